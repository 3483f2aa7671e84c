/* MATMPIAIJHIPMI355X: row-block distributed CSR.  Mat_MPIAIJ layout and MatMult choreography of
 * src/mat/impls/aij/mpi/mpiaij.c:1102-1238 and mmaij.c:9-161; the GPU subclass role of
 * src/mat/impls/aij/mpi/mpicusp/mpiaijcusp.cu:85-112,204-235.  A (diagonal block, local columns)
 * and B (off-diagonal block, columns compacted through garray) are MATSEQAIJHIPMI355X. */
#include "hipmi355ximpl.h"

#define MA(A) HipMPIAIJGet(A)

#if !defined(PETSCHIPMI355X_WITH_PETSC)   /* container, assembly and MatSetUpMultiply are the parent MATMPIAIJ's inside a PETSc tree */
static PetscErrorCode make_block(Mat parent, PetscInt m, PetscInt n, Mat *blk) {
  PetscErrorCode ierr;
  (void)parent;
  ierr = MatCreate(PETSC_COMM_SELF, blk);CHKERRQ(ierr);
  ierr = MatSetSizes(*blk, m, n, m, n);CHKERRQ(ierr);
  ierr = MatSetType(*blk, MATSEQAIJHIPMI355X);CHKERRQ(ierr);
  return 0;
}

/* "MatMPIAIJSetPreallocation_C" (MatMPIAIJSetPreallocation_MPIAIJ, mpiaij.c; recomposed by the GPU subclass as in
 * mpiaijcusp.cu:36-46,213-215) */
static PetscErrorCode MatMPIAIJSetPreallocation_MPIAIJHIP(Mat A, PetscInt d_nz, const PetscInt d_nnz[], PetscInt o_nz, const PetscInt o_nnz[]) {
  PetscErrorCode ierr;
  HipMPIAIJ *a = MA(A);
  if (d_nz == PETSC_DEFAULT || d_nz == PETSC_DECIDE) d_nz = 5;     /* mpiaij.c MatMPIAIJSetPreallocation_MPIAIJ */
  if (o_nz == PETSC_DEFAULT || o_nz == PETSC_DECIDE) o_nz = 2;
  if (!a->A) {
    ierr = make_block(A, A->rmap->n, A->cmap->n, &a->A);CHKERRQ(ierr);
    ierr = make_block(A, A->rmap->n, A->cmap->N, &a->B);CHKERRQ(ierr);
  }
  ierr = MatSeqAIJSetPreallocation(a->A, d_nz, d_nnz);CHKERRQ(ierr);
  ierr = MatSeqAIJSetPreallocation(a->B, o_nz, o_nnz);CHKERRQ(ierr);
  A->preallocated = PETSC_TRUE;
  return 0;
}
static PetscErrorCode MatSetUp_MPIAIJHIP(Mat A) { return MatMPIAIJSetPreallocation_MPIAIJHIP(A, PETSC_DEFAULT, NULL, PETSC_DEFAULT, NULL); }

static int cmp_int(const void *a, const void *b) { PetscInt x = *(const PetscInt *)a, y = *(const PetscInt *)b; return (x > y) - (x < y); }

/* MatDisAssemble_MPIAIJ, mmaij.c:170-231: an entry in an off-diagonal column the assembled matrix does not have yet.  The
 * off-diagonal block goes back to GLOBAL column numbers (the reference builds a full-width B and re-inserts every row; garray is
 * increasing, so mapping the stored indices through it keeps every row sorted), the local work vector, the scatter and garray
 * are discarded; the next final assembly builds them again (MatSetUpMultiply_MPIAIJ) -- on every process, see MatAssemblyEnd */
static PetscErrorCode MatDisAssemble_MPIAIJHIP(Mat A) {
  PetscErrorCode ierr;
  HipMPIAIJ *a = MA(A);
  if (!a->garray) return 0;
  HipAIJ *B = HipAIJGet(a->B);
  for (PetscInt r = 0; r < B->m; r++)
    for (PetscInt k = B->i[r]; k < B->i[r] + B->ilen[r]; k++) B->j[k] = a->garray[B->j[k]];
  B->n = A->cmap->N;
  ierr = PetscLayoutDestroy(&a->B->cmap);CHKERRQ(ierr);
  ierr = PetscLayoutCreateSetUp(PETSC_COMM_SELF, A->cmap->N, A->cmap->N, &a->B->cmap);CHKERRQ(ierr);
  ierr = VecDestroy(&a->lvec);CHKERRQ(ierr);
  ierr = HipScatterDestroy(&a->hscat);CHKERRQ(ierr);
  HipFree(a->garray); a->garray = NULL; a->ec = 0;
  ierr = MatSeqAIJHIPSetCompressedRow(a->B, PETSC_FALSE);CHKERRQ(ierr);      /* (and: the device copy of B belongs to the old columns) */
  HipStateIncrease(a->B);
  return 0;
}

/* MatSetValues_MPIAIJ, mpiaij.c:517-560: locally owned rows go into A / B; rows of other processes are stashed until the
 * assembly (mpiaij.c:552-558, matstash.c), unless MAT_NO_OFF_PROC_ENTRIES-like behaviour is asked with nothing */
static PetscErrorCode MatSetValues_MPIAIJHIP(Mat A, PetscInt m, const PetscInt im[], PetscInt n, const PetscInt in[], const PetscScalar v[], InsertMode addv) {
  PetscErrorCode ierr;
  HipMPIAIJ *a = MA(A);
  for (PetscInt i = 0; i < m; i++) {
    if (im[i] < 0) continue;
    if (im[i] >= A->rmap->N) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_OUTOFRANGE, "Row too large: row %d max %d", im[i], A->rmap->N - 1);
    if (im[i] < a->rstart || im[i] >= a->rend) {
      for (PetscInt j = 0; j < n; j++) {
        if (in[j] < 0) continue;
        if (in[j] >= A->cmap->N) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_OUTOFRANGE, "Column too large: col %d max %d", in[j], A->cmap->N - 1);
        ierr = HipStashAdd(&a->stash, im[i], in[j], v[i * n + j], (int)addv);CHKERRQ(ierr);
      }
      continue;
    }
    PetscInt row = im[i] - a->rstart;
    for (PetscInt j = 0; j < n; j++) {
      PetscInt col = in[j];
      if (col < 0) continue;
      if (col >= A->cmap->N) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_OUTOFRANGE, "Column too large: col %d max %d", col, A->cmap->N - 1);
      if (col >= a->cstart && col < a->cend) {
        PetscInt lc = col - a->cstart;
        ierr = MatSetValues(a->A, 1, &row, 1, &lc, &v[i * n + j], addv);CHKERRQ(ierr);
      } else {
        PetscInt bc = col;
        if (a->garray) {   /* assembled before: B's columns are compacted (colmap lookup, mpiaij.c:540-550) */
          PetscInt *p = (PetscInt *)bsearch(&col, a->garray, (size_t)a->ec, sizeof(PetscInt), cmp_int);
          if (p) bc = (PetscInt)(p - a->garray);
          else { ierr = MatDisAssemble_MPIAIJHIP(A);CHKERRQ(ierr); }            /* mpiaij.c:545-549: from here on B takes global columns */
        }
        ierr = MatSetValues(a->B, 1, &row, 1, &bc, &v[i * n + j], addv);CHKERRQ(ierr);
      }
    }
  }
  return 0;
}

/* MatSetUpMultiply_MPIAIJ, mmaij.c:9-161 */
PetscErrorCode MatSetUpMultiply_MPIAIJ(Mat mat) {
  PetscErrorCode ierr;
  HipMPIAIJ *aij = MA(mat);
  HipAIJ *B = HipAIJGet(aij->B);
  PetscInt nzB = B->nz, ec = 0, *garray, *tmp;
  /* garray = sorted distinct global columns of B (mmaij.c:27-50) */
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(nzB, 1), &tmp);CHKERRQ(ierr);
  memcpy(tmp, B->j, sizeof(PetscInt) * (size_t)nzB);
  qsort(tmp, (size_t)nzB, sizeof(PetscInt), cmp_int);
  for (PetscInt k = 0; k < nzB; k++) if (k == 0 || tmp[k] != tmp[k - 1]) tmp[ec++] = tmp[k];
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(ec + 1), &garray);CHKERRQ(ierr);
  memcpy(garray, tmp, sizeof(PetscInt) * (size_t)ec);
  HipFree(tmp);
  /* compact out the extra columns in B (mmaij.c:56-63) */
  for (PetscInt k = 0; k < nzB; k++) {
    PetscInt *p = (PetscInt *)bsearch(&B->j[k], garray, (size_t)ec, sizeof(PetscInt), cmp_int);
    B->j[k] = (PetscInt)(p - garray);
  }
  B->n = ec;
  ierr = PetscLayoutDestroy(&aij->B->cmap);CHKERRQ(ierr);
  ierr = PetscLayoutCreateSetUp(PETSC_COMM_SELF, ec, ec, &aij->B->cmap);CHKERRQ(ierr);
  HipStateIncrease(aij->B);
  /* local vector that is used to scatter into (mmaij.c:102) */
  ierr = VecCreateSeqHIPMI355X(PETSC_COMM_SELF, ec, &aij->lvec);CHKERRQ(ierr);
  /* generate the scatter context (mmaij.c:131-148) */
  ierr = HipScatterCreate_PtoS_MPIAIJ(HipObjComm(mat), mat->cmap, ec, garray, &aij->hscat);CHKERRQ(ierr);
  aij->garray = garray; aij->ec = ec;
  ierr = MatSeqAIJHIPSetCompressedRow(aij->B, PETSC_TRUE);CHKERRQ(ierr);
  return 0;
}

static PetscErrorCode MatAssemblyEnd_MPIAIJHIP(Mat A, MatAssemblyType mode) {   /* mpiaij.c:590-720 */
  PetscErrorCode ierr;
  HipMPIAIJ *a = MA(A);
  {   /* the stashed off-process entries reach their owners (MatAssemblyBegin/End_MPIAIJ: MatStashScatterBegin .. GetMesg),
       * rank after rank and in the order they were set */
    PetscInt nr, *ri, *rj; PetscScalar *rv; int smode;
    ierr = HipStashExchange(HipObjComm(A), &a->stash, &nr, &ri, &rj, &rv, &smode);CHKERRQ(ierr);
    for (PetscInt k = 0; k < nr; k++) {
      if (ri[k] < a->rstart || ri[k] >= a->rend) continue;
      ierr = MatSetValues_MPIAIJHIP(A, 1, &ri[k], 1, &rj[k], &rv[k], (InsertMode)smode);
      if (ierr) { HipFree(ri); HipFree(rj); HipFree(rv); CHKERRQ(ierr); }
    }
    HipFree(ri); HipFree(rj); HipFree(rv);
  }
  if (mode == MAT_FLUSH_ASSEMBLY) return 0;
  ierr = MatAssemblyBegin(a->A, mode);CHKERRQ(ierr);
  ierr = MatAssemblyEnd(a->A, mode);CHKERRQ(ierr);
  ierr = MatAssemblyBegin(a->B, mode);CHKERRQ(ierr);
  ierr = MatAssemblyEnd(a->B, mode);CHKERRQ(ierr);
  /* if one process has disassembled, all of them must (the scatter is rebuilt collectively): mpiaij.c:694-700 */
  if (A->was_assembled) {
    double dis = a->garray ? 0.0 : 1.0;
    if (HipCommAllreduce(HipObjComm(A), &dis, 1, 1, 0)) SETERRQ(HipObjComm(A), PETSC_ERR_LIB, "allreduce failed");
    if (dis > 0.0 && a->garray) { ierr = MatDisAssemble_MPIAIJHIP(A);CHKERRQ(ierr); }
  }
  if (!a->garray) { ierr = MatSetUpMultiply_MPIAIJ(A);CHKERRQ(ierr); }   /* mpiaij.c:701-703 */
  return 0;
}

#endif

static PetscErrorCode MatMult_MPIAIJHIP(Mat A, Vec xx, Vec yy) {   /* mpiaij.c:1102-1116 */
  PetscErrorCode ierr;
  HipMPIAIJ *a = MA(A);
  if (xx->map->n != A->cmap->n) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_SIZ, "Incompatible partition of A (%d) and xx (%d)", A->cmap->n, xx->map->n);
  /* Same dependences as the reference's Begin / mult / End sequence, but the diagonal-block SpMV is queued before
   * the host spends its ~tens of microseconds enqueueing the RCCL group: "x is final" is marked first. */
  ierr = HipScatterMarkReady(a->hscat, xx);CHKERRQ(ierr);
  ierr = (*a->A->ops->mult)(a->A, xx, yy);CHKERRQ(ierr);                                         /* compute stream */
  ierr = HipScatterBegin(a->hscat, xx, a->lvec, INSERT_VALUES, SCATTER_FORWARD);CHKERRQ(ierr);   /* halo stream, overlaps */
  ierr = HipScatterEnd(a->hscat, xx, a->lvec, INSERT_VALUES, SCATTER_FORWARD);CHKERRQ(ierr);
  ierr = (*a->B->ops->multadd)(a->B, a->lvec, yy, yy);CHKERRQ(ierr);
  return 0;
}
static PetscErrorCode MatMultAdd_MPIAIJHIP(Mat A, Vec xx, Vec yy, Vec zz) {   /* mpiaij.c:1132-1143 */
  PetscErrorCode ierr;
  HipMPIAIJ *a = MA(A);
  ierr = HipScatterBegin(a->hscat, xx, a->lvec, INSERT_VALUES, SCATTER_FORWARD);CHKERRQ(ierr);
  ierr = (*a->A->ops->multadd)(a->A, xx, yy, zz);CHKERRQ(ierr);
  ierr = HipScatterEnd(a->hscat, xx, a->lvec, INSERT_VALUES, SCATTER_FORWARD);CHKERRQ(ierr);
  ierr = (*a->B->ops->multadd)(a->B, a->lvec, zz, zz);CHKERRQ(ierr);
  return 0;
}
static PetscErrorCode MatMultTranspose_MPIAIJHIP(Mat A, Vec xx, Vec yy) {   /* mpiaij.c:1147-1174, !merged branch */
  PetscErrorCode ierr;
  HipMPIAIJ *a = MA(A);
  ierr = (*a->B->ops->multtranspose)(a->B, xx, a->lvec);CHKERRQ(ierr);
  ierr = HipScatterBegin(a->hscat, a->lvec, yy, ADD_VALUES, SCATTER_REVERSE);CHKERRQ(ierr);
  ierr = (*a->A->ops->multtranspose)(a->A, xx, yy);CHKERRQ(ierr);
  ierr = HipScatterEnd(a->hscat, a->lvec, yy, ADD_VALUES, SCATTER_REVERSE);CHKERRQ(ierr);
  return 0;
}
static PetscErrorCode MatMultTransposeAdd_MPIAIJHIP(Mat A, Vec xx, Vec yy, Vec zz) {   /* mpiaij.c:1223-1238 */
  PetscErrorCode ierr;
  HipMPIAIJ *a = MA(A);
  ierr = (*a->B->ops->multtranspose)(a->B, xx, a->lvec);CHKERRQ(ierr);
  ierr = HipScatterBegin(a->hscat, a->lvec, zz, ADD_VALUES, SCATTER_REVERSE);CHKERRQ(ierr);
  ierr = (*a->A->ops->multtransposeadd)(a->A, xx, yy, zz);CHKERRQ(ierr);
  ierr = HipScatterEnd(a->hscat, a->lvec, zz, ADD_VALUES, SCATTER_REVERSE);CHKERRQ(ierr);
  return 0;
}
static PetscErrorCode MatGetDiagonal_MPIAIJHIP(Mat A, Vec v) {   /* mpiaij.c:1246-1256 */
  if (A->rmap->N != A->cmap->N) SETERRQ(HipObjComm(A), PETSC_ERR_SUP, "Supports only square matrix where A->A is diag block");
  if (A->rmap->rstart != A->cmap->rstart || A->rmap->rend != A->cmap->rend) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_SIZ, "row partition must equal col partition");
  return (*MA(A)->A->ops->getdiagonal)(MA(A)->A, v);
}
static PetscErrorCode MatScale_MPIAIJHIP(Mat A, PetscScalar aa) {
  PetscErrorCode ierr;
  ierr = MatScale(MA(A)->A, aa);CHKERRQ(ierr);
  ierr = MatScale(MA(A)->B, aa);CHKERRQ(ierr);
  return 0;
}
/* MatDiagonalScale_MPIAIJ, mpiaij.c:2183-2213: the right vector's ghost values come through the MatMult scatter */
static PetscErrorCode MatDiagonalScale_MPIAIJHIP(Mat A, Vec ll, Vec rr) {
  PetscErrorCode ierr;
  HipMPIAIJ *aij = MA(A);
  if (rr) {
    if (rr->map->n != A->cmap->n) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_SIZ, "right vector non-conforming local size");
    ierr = HipScatterBegin(aij->hscat, rr, aij->lvec, INSERT_VALUES, SCATTER_FORWARD);CHKERRQ(ierr);
  }
  if (ll) {
    if (ll->map->n != A->rmap->n) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_SIZ, "left vector non-conforming local size");
    ierr = MatDiagonalScale(aij->B, ll, NULL);CHKERRQ(ierr);
  }
  ierr = MatDiagonalScale(aij->A, ll, rr);CHKERRQ(ierr);
  if (rr) {
    ierr = HipScatterEnd(aij->hscat, rr, aij->lvec, INSERT_VALUES, SCATTER_FORWARD);CHKERRQ(ierr);
    ierr = MatDiagonalScale(aij->B, NULL, aij->lvec);CHKERRQ(ierr);
  }
  return 0;
}
static PetscErrorCode MatZeroEntries_MPIAIJHIP(Mat A) {
  PetscErrorCode ierr;
  ierr = MatZeroEntries(MA(A)->A);CHKERRQ(ierr);
  ierr = MatZeroEntries(MA(A)->B);CHKERRQ(ierr);
  return 0;
}
#if defined(PETSCHIPMI355X_WITH_PETSC)
#include "mpiaijhipmi355x_ctor.h"   /* integration/petsc-3.3/: the constructor as a subclass of the reference's MATMPIAIJ */
#else
static PetscErrorCode MatDestroy_MPIAIJHIP(Mat A) {
  PetscErrorCode ierr;
  HipMPIAIJ *a = MA(A);
  if (!a) return 0;
  ierr = MatDestroy(&a->A);CHKERRQ(ierr);
  ierr = MatDestroy(&a->B);CHKERRQ(ierr);
  ierr = VecDestroy(&a->lvec);CHKERRQ(ierr);
  ierr = HipScatterDestroy(&a->hscat);CHKERRQ(ierr);
  HipFree(a->garray);
  HipStashFree(&a->stash);
  HipFree(a); A->data = NULL;
  return 0;
}

static PetscErrorCode MatMPIAIJSetPreallocationCSR_MPIAIJHIP(Mat A, const PetscInt i[], const PetscInt j[], const PetscScalar a[]);
static PetscErrorCode MatGetDiagonalBlock_MPIAIJHIP(Mat A, Mat *a) { *a = MA(A)->A; return 0; }   /* MatGetDiagonalBlock_MPIAIJ, mpiaij.c */

PetscErrorCode MatCreate_MPIAIJHIPMI355X(Mat B) {   /* MatCreate_MPIAIJCUSP, mpiaijcusp.cu:204-235 */
  PetscErrorCode ierr;
  HipMPIAIJ *a;
  ierr = PetscMalloc(sizeof(*a), &a);CHKERRQ(ierr);
  memset(a, 0, sizeof(*a));
  a->rstart = B->rmap->rstart; a->rend = B->rmap->rend; a->cstart = B->cmap->rstart; a->cend = B->cmap->rend;
  B->data = a;
  ierr = PetscObjectChangeTypeName((PetscObject)B, MATMPIAIJHIPMI355X);CHKERRQ(ierr);
  B->ops->setvalues = MatSetValues_MPIAIJHIP;
  B->ops->mult = MatMult_MPIAIJHIP;
  B->ops->multadd = MatMultAdd_MPIAIJHIP;
  B->ops->multtranspose = MatMultTranspose_MPIAIJHIP;
  B->ops->multtransposeadd = MatMultTransposeAdd_MPIAIJHIP;
  B->ops->getdiagonal = MatGetDiagonal_MPIAIJHIP;
  B->ops->assemblyend = MatAssemblyEnd_MPIAIJHIP;
  B->ops->zeroentries = MatZeroEntries_MPIAIJHIP;
  B->ops->setup = MatSetUp_MPIAIJHIP;
  B->ops->scale = MatScale_MPIAIJHIP;
  B->ops->diagonalscale = MatDiagonalScale_MPIAIJHIP;
  B->ops->destroy = MatDestroy_MPIAIJHIP;
  B->ops->getvecs = MatGetVecs_HIPMI355X;
  ierr = PetscObjectComposeFunction((PetscObject)B, "MatMPIAIJSetPreallocation_C", "MatMPIAIJSetPreallocation_MPIAIJHIP", (PetscVoidFunction)MatMPIAIJSetPreallocation_MPIAIJHIP);CHKERRQ(ierr);
  ierr = PetscObjectComposeFunction((PetscObject)B, "MatMPIAIJSetPreallocationCSR_C", "MatMPIAIJSetPreallocationCSR_MPIAIJHIP", (PetscVoidFunction)MatMPIAIJSetPreallocationCSR_MPIAIJHIP);CHKERRQ(ierr);
  ierr = PetscObjectComposeFunction((PetscObject)B, "MatGetDiagonalBlock_C", "MatGetDiagonalBlock_MPIAIJHIP", (PetscVoidFunction)MatGetDiagonalBlock_MPIAIJHIP);CHKERRQ(ierr);
  return 0;
}
#endif

/* base name "aijhipmi355x" -> seq or mpi by communicator size (MatRegisterBaseName, matreg.c:161-180) */
PetscErrorCode MatCreate_AIJHIPMI355X(Mat B) {
  return (HipCommSize(HipObjComm(B)) == 1) ? MatCreate_SeqAIJHIPMI355X(B) : MatCreate_MPIAIJHIPMI355X(B);
}

#if !defined(PETSCHIPMI355X_WITH_PETSC)
/* "MatMPIAIJSetPreallocationCSR_C" (MatMPIAIJSetPreallocationCSR_MPIAIJ, mpiaij.c:3900-3960): fills the matrix from this
 * rank's rows in CSR form, global ascending column indices.  The split is the column test of MatSetValues_MPIAIJ in bulk. */
static PetscErrorCode MatMPIAIJSetPreallocationCSR_MPIAIJHIP(Mat A, const PetscInt i[], const PetscInt j[], const PetscScalar a[]) {
  PetscErrorCode ierr;
  const PetscInt m = A->rmap->n;
  MPI_Comm comm = HipObjComm(A);
  HipMPIAIJ *aij = MA(A);
  (void)comm;
  if (i[0]) SETERRQ(comm, PETSC_ERR_ARG_OUTOFRANGE, "i (row indices) must start with 0");
  if (aij->A) { ierr = MatDestroy(&aij->A);CHKERRQ(ierr); ierr = MatDestroy(&aij->B);CHKERRQ(ierr); }
  PetscInt cs = aij->cstart, ce = aij->cend, nd = 0, no = 0;
  for (PetscInt k = 0; k < i[m]; k++) { if (j[k] >= cs && j[k] < ce) nd++; else no++; }
  PetscInt *di, *dj, *oi, *oj; PetscScalar *da, *oa;
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(m + 1), &di);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(m + 1), &oi);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(nd, 1), &dj);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(no, 1), &oj);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)PetscMax(nd, 1), &da);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)PetscMax(no, 1), &oa);CHKERRQ(ierr);
  nd = no = 0; di[0] = oi[0] = 0;
  for (PetscInt r = 0; r < m; r++) {
    for (PetscInt k = i[r]; k < i[r + 1]; k++) {
      if (j[k] < 0 || j[k] >= A->cmap->N) SETERRQ(comm, PETSC_ERR_ARG_OUTOFRANGE, "Column %d out of range [0,%d) in row %d", j[k], A->cmap->N, r);
      if (k > i[r] && j[k] <= j[k - 1]) SETERRQ(comm, PETSC_ERR_ARG_WRONG, "columns of row %d must be ascending and distinct", r);
      if (j[k] >= cs && j[k] < ce) { dj[nd] = j[k] - cs; da[nd++] = a[k]; }
      else { oj[no] = j[k]; oa[no++] = a[k]; }
    }
    di[r + 1] = nd; oi[r + 1] = no;
  }
  ierr = make_block(A, m, A->cmap->n, &aij->A);CHKERRQ(ierr);
  ierr = MatSeqAIJSetPreallocationCSR(aij->A, di, dj, da);CHKERRQ(ierr);
  ierr = make_block(A, m, A->cmap->N, &aij->B);CHKERRQ(ierr);
  ierr = MatSeqAIJSetPreallocationCSR(aij->B, oi, oj, oa);CHKERRQ(ierr);
  HipFree(di); HipFree(dj); HipFree(da); HipFree(oi); HipFree(oj); HipFree(oa);
  A->preallocated = PETSC_TRUE;
  ierr = MatSetUpMultiply_MPIAIJ(A);CHKERRQ(ierr);
  A->assembled = PETSC_TRUE; A->was_assembled = PETSC_TRUE; HipStateIncrease(A);
  return 0;
}

#endif

PetscErrorCode MatMPIAIJGetSeqAIJ(Mat A, Mat *Ad, Mat *Ao, const PetscInt **garray) {   /* mpiaij.c MatMPIAIJGetSeqAIJ */
  if (!A || strcmp(HipObjTypeName(A), MATMPIAIJHIPMI355X)) SETERRQ(0, PETSC_ERR_ARG_WRONG, "not an MPIAIJHIPMI355X matrix");
  if (Ad) *Ad = MA(A)->A;
  if (Ao) *Ao = MA(A)->B;
  if (garray) *garray = MA(A)->garray;
  return 0;
}
PetscErrorCode MatMPIAIJGetScatter(Mat A, VecScatter *ctx, Vec *lvec, PetscInt *ec) {
  if (!A || strcmp(HipObjTypeName(A), MATMPIAIJHIPMI355X)) SETERRQ(0, PETSC_ERR_ARG_WRONG, "not an MPIAIJHIPMI355X matrix");
  if (ctx) *ctx = (VecScatter)MA(A)->hscat;
  if (lvec) *lvec = MA(A)->lvec;
  if (ec) *ec = MA(A)->ec;
  return 0;
}

/* ---- bench.py: how the halo exchange of MatMult_MPIAIJ (mpiaij.c:1102-1116; VecScatterBegin_1/End_1, vpscat.h:14-233) sits beside the
 * diagonal-block product.  Event pairs: the diagonal product on the compute stream (MatHIPMI355XSetTiming of the diagonal block) and
 * the exchange on the halo stream (HipScatterSetTiming), one pair of each per MatMult.  Reported over the products timed so far:
 *   halo_ms     sum of the halo stream's busy spans (pack, grouped send/recv, unpack)
 *   overlap_ms  the part of those spans inside the diagonal product's span
 *   exposed_ms  what the compute stream waits for after its diagonal product has ended (halo end - product end, where positive)
 *   send_bytes  bytes this rank sends per product, neighbours: to how many ranks ---- */
PetscErrorCode MatMPIAIJHIPMI355XSetHaloTiming(Mat A, PetscBool on) {
  PetscErrorCode ierr;
  if (!A || strcmp(HipObjTypeName(A), MATMPIAIJHIPMI355X)) SETERRQ(0, PETSC_ERR_ARG_WRONG, "not an MPIAIJHIPMI355X matrix");
  ierr = MatHIPMI355XSetTiming(MA(A)->A, on);CHKERRQ(ierr);
  if (MA(A)->hscat) { ierr = HipScatterSetTiming(MA(A)->hscat, on);CHKERRQ(ierr); }
  return 0;
}
PetscErrorCode MatMPIAIJHIPMI355XGetHaloTiming(Mat A, PetscInt *nproducts, PetscLogDouble *halo_ms, PetscLogDouble *overlap_ms, PetscLogDouble *exposed_ms,
                                               PetscLogDouble *send_bytes, PetscInt *neighbours) {
  if (!A || strcmp(HipObjTypeName(A), MATMPIAIJHIPMI355X)) SETERRQ(0, PETSC_ERR_ARG_WRONG, "not an MPIAIJHIPMI355X matrix");
  HipScatter sc = MA(A)->hscat;
  Mat_SeqAIJHIP *d = (Mat_SeqAIJHIP *)MA(A)->A->spptr;
  double halo = 0.0, over = 0.0, expo = 0.0;
  PetscInt n = 0;
  if (sc && d && sc->time_ev && d->time_ev) {
    n = PetscMin(sc->time_n, d->time_n);
    for (PetscInt k = 0; k < n; k++) {
      mi355x_event_t s0 = d->time_ev[2 * k], s1 = d->time_ev[2 * k + 1], h0 = sc->time_ev[2 * k], h1 = sc->time_ev[2 * k + 1];
      float ts1 = 0.f, th0 = 0.f, th1 = 0.f;     /* all measured from the start of the diagonal product */
      CHKHIP(mi355x_event_synchronize(s1));
      CHKHIP(mi355x_event_synchronize(h1));
      CHKHIP(mi355x_event_elapsed_ms(s0, s1, &ts1));
      CHKHIP(mi355x_event_elapsed_ms(s0, h0, &th0));
      CHKHIP(mi355x_event_elapsed_ms(s0, h1, &th1));
      const double lo = th0 > 0.f ? th0 : 0.0, hi = th1 < ts1 ? th1 : ts1;
      halo += (double)th1 - (double)th0;
      if (hi > lo) over += hi - lo;
      if (th1 > ts1) expo += (double)th1 - (double)ts1;
    }
  }
  if (nproducts) *nproducts = n;
  if (halo_ms) *halo_ms = halo;
  if (overlap_ms) *overlap_ms = over;
  if (exposed_ms) *exposed_ms = expo;
  if (send_bytes) *send_bytes = sc ? (double)sizeof(PetscScalar) * (double)sc->to.starts[sc->to.n] : 0.0;
  if (neighbours) *neighbours = sc ? sc->to.n : 0;
  return 0;
}
