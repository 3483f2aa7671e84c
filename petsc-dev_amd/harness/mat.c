/* Mat: public wrappers (size/state checks, dispatch through mat->ops -- src/mat/interface/matrix.c)
 * and the string-keyed type registry (src/mat/interface/matreg.c:42-82,137-180). */
#include "petscimpl.h"

#define MAXTYPES 16
typedef PetscErrorCode (*MatCreateFn)(Mat);
static struct { char name[32]; MatCreateFn fn; } mat_types[MAXTYPES];
static int n_mat_types = 0;

PetscErrorCode MatRegister(const char name[], const char path[], const char fname[], MatCreateFn fn) {
  (void)path; (void)fname;
  for (int i = 0; i < n_mat_types; i++) if (!strcmp(mat_types[i].name, name)) { mat_types[i].fn = fn; return 0; }
  if (n_mat_types >= MAXTYPES) SETERRQ(0, PETSC_ERR_PLIB, "Mat type table full");
  snprintf(mat_types[n_mat_types].name, 32, "%s", name);
  mat_types[n_mat_types++].fn = fn;
  return 0;
}

#define MatValid(A, arg) do { if (!(A)) SETERRQ(0, PETSC_ERR_ARG_NULL, "Null Object: Parameter # %d", arg); } while (0)
#define MatTypeSet(A, arg) do { MatValid(A, arg); if (!(A)->data) SETERRQ((A)->comm, PETSC_ERR_ARG_TYPENOTSET, "Mat type not set: Parameter # %d", arg); } while (0)
#define MatAssembled(A) do { if (!(A)->assembled) SETERRQ((A)->comm, PETSC_ERR_ARG_WRONGSTATE, "Not for unassembled matrix"); } while (0)

PetscErrorCode MatCreate(PetscComm comm, Mat *A) {
  Mat B;
  PetscErrorCode ierr = PetscMalloc(sizeof(*B), &B);CHKERRQ(ierr);
  memset(B, 0, sizeof(*B));
  B->comm = comm;
  B->m_req = B->n_req = B->M_req = B->N_req = -1;
  *A = B;
  return 0;
}
PetscErrorCode MatSetSizes(Mat A, PetscInt m, PetscInt n, PetscInt M, PetscInt N) {
  MatValid(A, 1);
  if (M > 0 && m > M) SETERRQ(A->comm, PETSC_ERR_ARG_INCOMP, "Local row size %d cannot be larger than global row size %d", m, M);
  if (N > 0 && n > N) SETERRQ(A->comm, PETSC_ERR_ARG_INCOMP, "Local column size %d cannot be larger than global column size %d", n, N);
  A->m_req = m; A->n_req = n; A->M_req = M; A->N_req = N;
  if (A->pending_type[0] && !A->type_name[0]) {   /* the type was chosen first: its constructor can run now */
    char t[32];
    snprintf(t, sizeof(t), "%s", A->pending_type);
    A->pending_type[0] = 0;
    return MatSetType(A, t);
  }
  return 0;
}
PetscErrorCode MatSetType(Mat A, MatType type) {
  PetscErrorCode ierr;
  MatValid(A, 1);
  if (!strcmp(A->type_name, type)) return 0;
  for (int i = 0; i < n_mat_types; i++) {
    if (!strcmp(mat_types[i].name, type)) {
      if (A->ops->destroy) { ierr = (*A->ops->destroy)(A);CHKERRQ(ierr); }   /* matreg.c:68-71 */
      memset(A->ops, 0, sizeof(A->ops));
      ierr = PetscObjectListDestroy_Private((PetscObject)A);CHKERRQ(ierr);
      A->data = NULL; A->spptr = NULL;
      if (!A->rmap) {
        if (A->m_req == -1 && A->M_req == -1) {   /* sizes not known yet (e.g. MatSetType, then MatLoad): remember the choice */
          snprintf(A->pending_type, sizeof(A->pending_type), "%s", type);
          return 0;
        }
        ierr = PetscLayoutCreateSetUp(A->comm, A->m_req, A->M_req, &A->rmap);CHKERRQ(ierr);
        ierr = PetscLayoutCreateSetUp(A->comm, A->n_req, A->N_req, &A->cmap);CHKERRQ(ierr);
      }
      ierr = (*mat_types[i].fn)(A);CHKERRQ(ierr);
      return 0;
    }
  }
  SETERRQ(A->comm, PETSC_ERR_ARG_UNKNOWN_TYPE, "Unknown Mat type given: %s", type);
}
PetscErrorCode MatSetFromOptions(Mat A) {
  char t[64];
  PetscBool set;
  PetscErrorCode ierr = PetscOptionsGetString(NULL, "-mat_type", t, sizeof(t), &set);CHKERRQ(ierr);
  if (set || (!A->type_name[0] && !A->pending_type[0])) {                    /* gcreate.c:188-193: -mat_type, else the default type (aij) for a matrix without one */
    if (!set) snprintf(t, sizeof(t), "%s", MATAIJ);
    ierr = MatSetType(A, t);CHKERRQ(ierr);
  }
  if (A->ops->setfromoptions) { ierr = (*A->ops->setfromoptions)(A);CHKERRQ(ierr); }   /* gcreate.c:201-203 */
  return 0;
}
/* MatDuplicate, matrix.c:4023-4055 */
PetscErrorCode MatDuplicate(Mat A, MatDuplicateOption op, Mat *M) {
  PetscErrorCode ierr;
  MatTypeSet(A, 1);
  if (!A->assembled) SETERRQ(A->comm, PETSC_ERR_ARG_WRONGSTATE, "Not for unassembled matrix");
  if (A->factortype) SETERRQ(A->comm, PETSC_ERR_ARG_WRONGSTATE, "Not for factored matrix");
  *M = NULL;
  if (!A->ops->duplicate) SETERRQ(A->comm, PETSC_ERR_SUP, "Not written for this matrix type");
  ierr = (*A->ops->duplicate)(A, op, M);CHKERRQ(ierr);
  PetscObjectStateIncrease(*M);
  return 0;
}
PetscErrorCode MatSetOptionsPrefix(Mat A, const char prefix[]) { MatValid(A, 1); snprintf(A->prefix, sizeof(A->prefix), "%s", prefix ? prefix : ""); return 0; }
PetscErrorCode MatGetType(Mat A, MatType *type) { MatValid(A, 1); *type = A->type_name; return 0; }
PetscErrorCode MatSetUp(Mat A) {
  MatTypeSet(A, 1);
  if (!A->preallocated && A->ops->setup) { PetscErrorCode ierr = (*A->ops->setup)(A);CHKERRQ(ierr); }
  return 0;
}
PetscErrorCode MatSetValues(Mat A, PetscInt m, const PetscInt idxm[], PetscInt n, const PetscInt idxn[], const PetscScalar v[], InsertMode addv) {
  PetscErrorCode ierr;
  MatTypeSet(A, 1);
  if (!m || !n) return 0;
  if (!A->preallocated) { ierr = MatSetUp(A);CHKERRQ(ierr); }
  ierr = (*A->ops->setvalues)(A, m, idxm, n, idxn, v, addv);CHKERRQ(ierr);
  A->assembled = PETSC_FALSE;   /* matrix.c:1083 */
  return 0;
}
/* MatSetValuesBatch, matrix.c:1698-1722: nb square blocks of bs x bs values (row-major), rows[b*bs .. b*bs+bs) their
 * row and column indices, ADD_VALUES; MatAssemblyBegin/End must follow as after MatSetValues */
PetscErrorCode MatSetValuesBatch(Mat A, PetscInt nb, PetscInt bs, PetscInt rows[], const PetscScalar v[]) {
  PetscErrorCode ierr;
  MatTypeSet(A, 1);
  if (nb <= 0 || bs <= 0) return 0;
  if (!rows || !v) SETERRQ(A->comm, PETSC_ERR_ARG_NULL, "Null array");
  if (!A->preallocated) { ierr = MatSetUp(A);CHKERRQ(ierr); }
  if (A->ops->setvaluesbatch) { ierr = (*A->ops->setvaluesbatch)(A, nb, bs, rows, v);CHKERRQ(ierr); }
  else {
    for (PetscInt b = 0; b < nb; b++) { ierr = MatSetValues(A, bs, &rows[(size_t)b * bs], bs, &rows[(size_t)b * bs], &v[(size_t)b * bs * bs], ADD_VALUES);CHKERRQ(ierr); }
  }
  A->assembled = PETSC_FALSE;
  return 0;
}
PetscErrorCode MatAssemblyBegin(Mat A, MatAssemblyType type) {
  MatTypeSet(A, 1);
  if (A->ops->assemblybegin) { PetscErrorCode ierr = (*A->ops->assemblybegin)(A, type);CHKERRQ(ierr); }
  return 0;
}
PetscErrorCode MatAssemblyEnd(Mat A, MatAssemblyType type) {   /* matrix.c:4881 */
  PetscErrorCode ierr;
  MatTypeSet(A, 1);
  if (!A->preallocated) { ierr = MatSetUp(A);CHKERRQ(ierr); }
  if (A->ops->assemblyend) { ierr = (*A->ops->assemblyend)(A, type);CHKERRQ(ierr); }
  if (type == MAT_FINAL_ASSEMBLY) { A->assembled = PETSC_TRUE; A->was_assembled = PETSC_TRUE; }
  A->state++;
  return 0;
}
PetscErrorCode MatDestroy(Mat *A) {
  PetscErrorCode ierr;
  if (!*A) return 0;
  if ((*A)->ops->destroy) { ierr = (*(*A)->ops->destroy)(*A);CHKERRQ(ierr); }
  ierr = PetscLayoutDestroy(&(*A)->rmap);CHKERRQ(ierr);
  ierr = PetscLayoutDestroy(&(*A)->cmap);CHKERRQ(ierr);
  ierr = PetscObjectListDestroy_Private((PetscObject)*A);CHKERRQ(ierr);
  free(*A); *A = NULL;
  return 0;
}
PetscErrorCode MatGetSize(Mat A, PetscInt *M, PetscInt *N) { MatValid(A, 1); if (M) *M = A->rmap->N; if (N) *N = A->cmap->N; return 0; }
PetscErrorCode MatGetLocalSize(Mat A, PetscInt *m, PetscInt *n) { MatValid(A, 1); if (m) *m = A->rmap->n; if (n) *n = A->cmap->n; return 0; }
PetscErrorCode MatGetOwnershipRange(Mat A, PetscInt *rs, PetscInt *re) { MatValid(A, 1); if (rs) *rs = A->rmap->rstart; if (re) *re = A->rmap->rend; return 0; }

PetscErrorCode MatGetVecs(Mat A, Vec *right, Vec *left) {   /* matrix.c:8009-8037 */
  MatTypeSet(A, 1);
  PetscErrorCode ierr = (*A->ops->getvecs)(A, right, left);CHKERRQ(ierr);
  return 0;
}

PetscErrorCode MatMult(Mat A, Vec x, Vec y) {   /* matrix.c:2132-2158 */
  PetscErrorCode ierr;
  MatTypeSet(A, 1); MatAssembled(A);
  if (x == y) SETERRQ(A->comm, PETSC_ERR_ARG_IDN, "x and y must be different vectors");
  if (A->cmap->N != x->map->N) SETERRQ(A->comm, PETSC_ERR_ARG_SIZ, "Mat mat,Vec x: global dim %d %d", A->cmap->N, x->map->N);
  if (A->rmap->N != y->map->N) SETERRQ(A->comm, PETSC_ERR_ARG_SIZ, "Mat mat,Vec y: global dim %d %d", A->rmap->N, y->map->N);
  if (A->rmap->n != y->map->n) SETERRQ(A->comm, PETSC_ERR_ARG_SIZ, "Mat mat,Vec y: local dim %d %d", A->rmap->n, y->map->n);
  if (!A->ops->mult) SETERRQ(A->comm, PETSC_ERR_SUP, "This matrix type does not have a multiply defined");
  ierr = (*A->ops->mult)(A, x, y);CHKERRQ(ierr);
  PetscObjectStateIncrease(y);
  return 0;
}
PetscErrorCode MatMultAdd(Mat A, Vec v1, Vec v2, Vec v3) {   /* matrix.c:2304 */
  PetscErrorCode ierr;
  MatTypeSet(A, 1); MatAssembled(A);
  if (A->cmap->N != v1->map->N) SETERRQ(A->comm, PETSC_ERR_ARG_SIZ, "Mat mat,Vec v1: global dim %d %d", A->cmap->N, v1->map->N);
  if (A->rmap->n != v3->map->n) SETERRQ(A->comm, PETSC_ERR_ARG_SIZ, "Mat mat,Vec v3: local dim %d %d", A->rmap->n, v3->map->n);
  if (A->rmap->n != v2->map->n) SETERRQ(A->comm, PETSC_ERR_ARG_SIZ, "Mat mat,Vec v2: local dim %d %d", A->rmap->n, v2->map->n);
  if (v1 == v3) SETERRQ(A->comm, PETSC_ERR_ARG_IDN, "v1 and v3 must be different vectors");
  ierr = (*A->ops->multadd)(A, v1, v2, v3);CHKERRQ(ierr);
  PetscObjectStateIncrease(v3);
  return 0;
}
PetscErrorCode MatMultTranspose(Mat A, Vec x, Vec y) {   /* matrix.c:2187 */
  PetscErrorCode ierr;
  MatTypeSet(A, 1); MatAssembled(A);
  if (x == y) SETERRQ(A->comm, PETSC_ERR_ARG_IDN, "x and y must be different vectors");
  if (A->rmap->N != x->map->N) SETERRQ(A->comm, PETSC_ERR_ARG_SIZ, "Mat mat,Vec x: global dim %d %d", A->rmap->N, x->map->N);
  if (A->cmap->N != y->map->N) SETERRQ(A->comm, PETSC_ERR_ARG_SIZ, "Mat mat,Vec y: global dim %d %d", A->cmap->N, y->map->N);
  if (!A->ops->multtranspose) SETERRQ(A->comm, PETSC_ERR_SUP, "This matrix type does not have a multiply tranpose defined");
  ierr = (*A->ops->multtranspose)(A, x, y);CHKERRQ(ierr);
  PetscObjectStateIncrease(y);
  return 0;
}
PetscErrorCode MatMultTransposeAdd(Mat A, Vec v1, Vec v2, Vec v3) {
  PetscErrorCode ierr;
  MatTypeSet(A, 1); MatAssembled(A);
  if (v1 == v3) SETERRQ(A->comm, PETSC_ERR_ARG_IDN, "v1 and v3 must be different vectors");
  if (A->rmap->N != v1->map->N) SETERRQ(A->comm, PETSC_ERR_ARG_SIZ, "Mat mat,Vec v1: global dim %d %d", A->rmap->N, v1->map->N);
  if (A->cmap->N != v2->map->N) SETERRQ(A->comm, PETSC_ERR_ARG_SIZ, "Mat mat,Vec v2: global dim %d %d", A->cmap->N, v2->map->N);
  if (A->cmap->N != v3->map->N) SETERRQ(A->comm, PETSC_ERR_ARG_SIZ, "Mat mat,Vec v3: global dim %d %d", A->cmap->N, v3->map->N);
  if (!A->ops->multtransposeadd) SETERRQ(A->comm, PETSC_ERR_SUP, "Mat type %s", A->type_name);
  ierr = (*A->ops->multtransposeadd)(A, v1, v2, v3);CHKERRQ(ierr);
  PetscObjectStateIncrease(v3);
  return 0;
}
PetscErrorCode MatGetDiagonal(Mat A, Vec d) {   /* matrix.c:4080 */
  PetscErrorCode ierr;
  MatTypeSet(A, 1); MatAssembled(A);
  if (!A->ops->getdiagonal) SETERRQ(A->comm, PETSC_ERR_SUP, "Mat type %s", A->type_name);
  ierr = (*A->ops->getdiagonal)(A, d);CHKERRQ(ierr);
  PetscObjectStateIncrease(d);
  return 0;
}
PetscErrorCode MatScale(Mat A, PetscScalar a) {
  PetscErrorCode ierr;
  MatTypeSet(A, 1); MatAssembled(A);
  if (!A->ops->scale) SETERRQ(A->comm, PETSC_ERR_SUP, "Mat type %s", A->type_name);
  if (a != 1.0) { ierr = (*A->ops->scale)(A, a);CHKERRQ(ierr); A->state++; }
  return 0;
}
PetscErrorCode MatDiagonalScale(Mat A, Vec l, Vec r) {   /* matrix.c MatDiagonalScale: A <- diag(l) A diag(r), either may be NULL */
  PetscErrorCode ierr;
  MatTypeSet(A, 1); MatAssembled(A);
  if (!A->ops->diagonalscale) SETERRQ(A->comm, PETSC_ERR_SUP, "Mat type %s", A->type_name);
  if (!l && !r) return 0;
  ierr = (*A->ops->diagonalscale)(A, l, r);CHKERRQ(ierr);
  A->state++;
  return 0;
}
PetscErrorCode MatZeroEntries(Mat A) {
  PetscErrorCode ierr;
  MatTypeSet(A, 1);
  if (!A->ops->zeroentries) SETERRQ(A->comm, PETSC_ERR_SUP, "Mat type %s", A->type_name);
  ierr = (*A->ops->zeroentries)(A);CHKERRQ(ierr);
  A->state++;
  return 0;
}

/* ---- type-specific methods reached through composed functions, as in the reference (PetscTryMethod / PetscUseMethod,
 * e.g. MatSeqAIJSetPreallocation aij.c:3435, MatMPIAIJSetPreallocation mpiaij.c:4274, MatGetDiagonalBlock matrix.c) ---- */
PetscErrorCode MatSeqAIJSetPreallocation(Mat A, PetscInt nz, const PetscInt nnz[]) {
  PetscVoidFunction f;
  MatValid(A, 1);
  PetscErrorCode ierr = PetscObjectQueryFunction((PetscObject)A, "MatSeqAIJSetPreallocation_C", &f);CHKERRQ(ierr);
  if (f) { ierr = ((PetscErrorCode (*)(Mat, PetscInt, const PetscInt[]))f)(A, nz, nnz);CHKERRQ(ierr); }
  return 0;
}
PetscErrorCode MatMPIAIJSetPreallocation(Mat A, PetscInt d_nz, const PetscInt d_nnz[], PetscInt o_nz, const PetscInt o_nnz[]) {
  PetscVoidFunction f;
  MatValid(A, 1);
  PetscErrorCode ierr = PetscObjectQueryFunction((PetscObject)A, "MatMPIAIJSetPreallocation_C", &f);CHKERRQ(ierr);
  if (f) { ierr = ((PetscErrorCode (*)(Mat, PetscInt, const PetscInt[], PetscInt, const PetscInt[]))f)(A, d_nz, d_nnz, o_nz, o_nnz);CHKERRQ(ierr); }
  return 0;
}
/* MatSeqAIJSetPreallocationCSR / MatMPIAIJSetPreallocationCSR (aij.c:3795, mpiaij.c:3960): copy a CSR description in */
PetscErrorCode MatSeqAIJSetPreallocationCSR(Mat A, const PetscInt i[], const PetscInt j[], const PetscScalar v[]) {
  PetscVoidFunction f;
  MatValid(A, 1);
  PetscErrorCode ierr = PetscObjectQueryFunction((PetscObject)A, "MatSeqAIJSetPreallocationCSR_C", &f);CHKERRQ(ierr);
  if (!f) SETERRQ(A->comm, PETSC_ERR_SUP, "Mat type %s cannot be filled from CSR arrays", A->type_name);
  ierr = ((PetscErrorCode (*)(Mat, const PetscInt[], const PetscInt[], const PetscScalar[]))f)(A, i, j, v);CHKERRQ(ierr);
  return 0;
}
PetscErrorCode MatMPIAIJSetPreallocationCSR(Mat A, const PetscInt i[], const PetscInt j[], const PetscScalar v[]) {
  PetscVoidFunction f;
  MatValid(A, 1);
  PetscErrorCode ierr = PetscObjectQueryFunction((PetscObject)A, "MatMPIAIJSetPreallocationCSR_C", &f);CHKERRQ(ierr);
  if (!f) SETERRQ(A->comm, PETSC_ERR_SUP, "Mat type %s cannot be filled from CSR arrays", A->type_name);
  ierr = ((PetscErrorCode (*)(Mat, const PetscInt[], const PetscInt[], const PetscScalar[]))f)(A, i, j, v);CHKERRQ(ierr);
  return 0;
}
PetscErrorCode MatGetDiagonalBlock(Mat A, Mat *a) {   /* matrix.c MatGetDiagonalBlock: "MatGetDiagonalBlock_C", or A itself on one process */
  PetscVoidFunction f;
  MatTypeSet(A, 1);
  PetscErrorCode ierr = PetscObjectQueryFunction((PetscObject)A, "MatGetDiagonalBlock_C", &f);CHKERRQ(ierr);
  if (f) { ierr = ((PetscErrorCode (*)(Mat, Mat *))f)(A, a);CHKERRQ(ierr); return 0; }
  if (A->comm->size == 1) { *a = A; return 0; }
  SETERRQ(A->comm, PETSC_ERR_SUP, "Cannot get diagonal part for this matrix");
}
/* MatCreateSeqAIJWithArrays (aij.c:3845), MatCreateMPIAIJWithArrays (mpiaij.c:4046), MatCreateSeqBAIJWithArrays
 * (baij.c): here the arrays are copied (the sequential reference routines alias them) */
PetscErrorCode MatCreateSeqAIJWithArrays(PetscComm comm, PetscInt m, PetscInt n, PetscInt i[], PetscInt j[], PetscScalar a[], Mat *mat) {
  PetscErrorCode ierr;
  ierr = MatCreate(comm, mat);CHKERRQ(ierr);
  ierr = MatSetSizes(*mat, m, n, m, n);CHKERRQ(ierr);
  ierr = MatSetType(*mat, MATSEQAIJ);CHKERRQ(ierr);
  ierr = MatSeqAIJSetPreallocationCSR(*mat, i, j, a);CHKERRQ(ierr);
  return 0;
}
PetscErrorCode MatCreateMPIAIJWithArrays(PetscComm comm, PetscInt m, PetscInt n, PetscInt M, PetscInt N, const PetscInt i[], const PetscInt j[], const PetscScalar a[], Mat *mat) {
  PetscErrorCode ierr;
  if (i[0]) SETERRQ(comm, PETSC_ERR_ARG_OUTOFRANGE, "i (row indices) must start with 0");
  if (m < 0) SETERRQ(comm, PETSC_ERR_ARG_OUTOFRANGE, "local number of rows (m) cannot be PETSC_DECIDE, or negative");
  ierr = MatCreate(comm, mat);CHKERRQ(ierr);
  ierr = MatSetSizes(*mat, m, n, M, N);CHKERRQ(ierr);
  ierr = MatSetType(*mat, MATMPIAIJ);CHKERRQ(ierr);
  ierr = MatMPIAIJSetPreallocationCSR(*mat, i, j, a);CHKERRQ(ierr);
  return 0;
}
PetscErrorCode MatCreateSeqBAIJWithArrays(PetscComm comm, PetscInt bs, PetscInt m, PetscInt n, PetscInt i[], PetscInt j[], PetscScalar a[], Mat *mat) {
  PetscErrorCode ierr; PetscVoidFunction f;
  if (bs < 1 || m % bs || n % bs) SETERRQ(comm, PETSC_ERR_ARG_SIZ, "block size %d must divide the local sizes %d, %d", bs, m, n);
  ierr = MatCreate(comm, mat);CHKERRQ(ierr);
  ierr = MatSetSizes(*mat, m, n, m, n);CHKERRQ(ierr);
  ierr = MatSetType(*mat, MATSEQBAIJ);CHKERRQ(ierr);
  ierr = PetscObjectQueryFunction((PetscObject)*mat, "MatSeqBAIJSetPreallocationCSR_C", &f);CHKERRQ(ierr);
  if (!f) SETERRQ(comm, PETSC_ERR_SUP, "Mat type %s cannot be filled from block CSR arrays", (*mat)->type_name);
  ierr = ((PetscErrorCode (*)(Mat, PetscInt, const PetscInt[], const PetscInt[], const PetscScalar[]))f)(*mat, bs, i, j, a);CHKERRQ(ierr);
  return 0;
}

/* ---- factorisation interface (matrix.c) ---- */
PetscErrorCode MatFactorInfoInitialize(MatFactorInfo *info) { memset(info, 0, sizeof(*info)); return 0; }   /* matrix.c:2633 */
/* MatGetFactor, matrix.c:3937-3975: the operator's TYPE says which factorisations a solver package offers for it, through a
 * method composed under "MatGetFactor_<package>_C" */
PetscErrorCode MatGetFactor(Mat mat, const MatSolverPackage type, MatFactorType ftype, Mat *f) {
  PetscErrorCode ierr;
  char name[96];
  PetscVoidFunction conv = NULL;
  MatTypeSet(mat, 1);
  if (mat->factortype) SETERRQ(mat->comm, PETSC_ERR_ARG_WRONGSTATE, "Not for factored matrix");
  snprintf(name, sizeof(name), "MatGetFactor_%s_C", type);
  ierr = PetscObjectQueryFunction((PetscObject)mat, name, &conv);CHKERRQ(ierr);
  if (!conv) SETERRQ(mat->comm, PETSC_ERR_SUP, "Matrix format %s does not have a solver package %s. Perhaps you must ./configure with --download-%s", mat->type_name, type, type);
  ierr = ((PetscErrorCode (*)(Mat, MatFactorType, Mat *))conv)(mat, ftype, f);CHKERRQ(ierr);
  return 0;
}
PetscErrorCode MatGetFactorAvailable(Mat mat, const MatSolverPackage type, MatFactorType ftype, PetscBool *flg) {   /* matrix.c:3996 */
  PetscErrorCode ierr;
  char name[96];
  PetscVoidFunction conv = NULL;
  MatValid(mat, 1);
  *flg = PETSC_FALSE;
  snprintf(name, sizeof(name), "MatGetFactorAvailable_%s_C", type);
  ierr = PetscObjectQueryFunction((PetscObject)mat, name, &conv);CHKERRQ(ierr);
  if (conv) { ierr = ((PetscErrorCode (*)(Mat, MatFactorType, PetscBool *))conv)(mat, ftype, flg);CHKERRQ(ierr); }
  return 0;
}
#define FactorPair(fact, mat) do { MatValid(fact, 1); MatTypeSet(mat, 2); MatAssembled(mat); \
  if ((mat)->rmap->N != (mat)->cmap->N) SETERRQ((mat)->comm, PETSC_ERR_ARG_WRONG, "matrix must be square"); } while (0)
PetscErrorCode MatILUFactorSymbolic(Mat fact, Mat mat, IS row, IS col, const MatFactorInfo *info) {   /* matrix.c:5905 */
  FactorPair(fact, mat);
  if (info->levels < 0) SETERRQ(mat->comm, PETSC_ERR_ARG_OUTOFRANGE, "Levels of fill negative %d", (int)info->levels);
  if (!fact->ops->ilufactorsymbolic) SETERRQ(mat->comm, PETSC_ERR_SUP, "Matrix type %s symbolic ILU", mat->type_name);
  PetscErrorCode ierr = (*fact->ops->ilufactorsymbolic)(fact, mat, row, col, info);CHKERRQ(ierr);
  return 0;
}
PetscErrorCode MatLUFactorNumeric(Mat fact, Mat mat, const MatFactorInfo *info) {   /* matrix.c:2815 */
  FactorPair(fact, mat);
  if (!fact->ops->lufactornumeric) SETERRQ(mat->comm, PETSC_ERR_SUP, "Mat type %s numeric LU", mat->type_name);
  PetscErrorCode ierr = (*fact->ops->lufactornumeric)(fact, mat, info);CHKERRQ(ierr);
  fact->state++;
  return 0;
}
PetscErrorCode MatICCFactorSymbolic(Mat fact, Mat mat, IS perm, const MatFactorInfo *info) {   /* matrix.c:5968 */
  FactorPair(fact, mat);
  if (info->levels < 0) SETERRQ(mat->comm, PETSC_ERR_ARG_OUTOFRANGE, "Levels negative %d", (int)info->levels);
  if (!fact->ops->iccfactorsymbolic) SETERRQ(mat->comm, PETSC_ERR_SUP, "Matrix type %s symbolic ICC", mat->type_name);
  PetscErrorCode ierr = (*fact->ops->iccfactorsymbolic)(fact, mat, perm, info);CHKERRQ(ierr);
  return 0;
}
PetscErrorCode MatCholeskyFactorNumeric(Mat fact, Mat mat, const MatFactorInfo *info) {   /* matrix.c:2957 */
  FactorPair(fact, mat);
  if (!fact->ops->choleskyfactornumeric) SETERRQ(mat->comm, PETSC_ERR_SUP, "Mat type %s numeric factor Cholesky", mat->type_name);
  PetscErrorCode ierr = (*fact->ops->choleskyfactornumeric)(fact, mat, info);CHKERRQ(ierr);
  fact->state++;
  return 0;
}
PetscErrorCode MatSolve(Mat mat, Vec b, Vec x) {   /* matrix.c:3196-3225 */
  PetscErrorCode ierr;
  MatValid(mat, 1);
  if (x == b) SETERRQ(mat->comm, PETSC_ERR_ARG_IDN, "x and b must be different vectors");
  if (!mat->factortype) SETERRQ(mat->comm, PETSC_ERR_ARG_WRONGSTATE, "Unfactored matrix");
  if (mat->cmap->N != x->map->N) SETERRQ(mat->comm, PETSC_ERR_ARG_SIZ, "Mat mat,Vec x: global dim %d %d", mat->cmap->N, x->map->N);
  if (mat->rmap->N != b->map->N) SETERRQ(mat->comm, PETSC_ERR_ARG_SIZ, "Mat mat,Vec b: global dim %d %d", mat->rmap->N, b->map->N);
  if (!mat->rmap->N && !mat->cmap->N) return 0;
  if (!mat->ops->solve) SETERRQ(mat->comm, PETSC_ERR_SUP, "Mat type %s", mat->type_name);
  ierr = (*mat->ops->solve)(mat, b, x);CHKERRQ(ierr);
  PetscObjectStateIncrease(x);
  return 0;
}
