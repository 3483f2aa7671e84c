/* PC interface (src/ksp/pc/interface/precon.c) and the preconditioners of the north star:
 * PCNONE, PCJACOBI (src/ksp/pc/impls/jacobi/jacobi.c:125-198,266-277) and PCBJACOBI with one block
 * per rank (src/ksp/pc/impls/bjacobi/bjacobi.c:738-761,858-923). */
#include "petscimpl.h"

PetscErrorCode PCCreate(PetscComm comm, PC *newpc) {
  PC pc;
  PetscErrorCode ierr = PetscMiniInitialize();CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(*pc), &pc);CHKERRQ(ierr);
  memset(pc, 0, sizeof(*pc));
  pc->comm = comm;
  *newpc = pc;
  return 0;
}
#define MAXPCTYPES 16
static struct { char name[32]; PetscErrorCode (*fn)(PC); } pc_types[MAXPCTYPES + 1];
static int n_pc_types = 0;
PetscErrorCode PCRegister(const char name[], const char path[], const char fname[], PetscErrorCode (*fn)(PC)) {   /* src/ksp/pc/interface/pcregis.c */
  (void)path; (void)fname;
  for (int i = 0; i < n_pc_types; i++) if (!strcmp(pc_types[i].name, name)) { pc_types[i].fn = fn; return 0; }
  if (n_pc_types >= MAXPCTYPES) SETERRQ(0, PETSC_ERR_PLIB, "PC type table full");
  snprintf(pc_types[n_pc_types].name, 32, "%s", name);
  pc_types[n_pc_types++].fn = fn;
  return 0;
}
/* can this operator be ILU-factored by the default package?  (PCGetDefaultType_Private asks MatGetFactorAvailable, precon.c:24-33) */
static PetscBool mat_has_ilu(Mat A) {
  PetscBool flg = PETSC_FALSE;
  if (MatGetFactorAvailable(A, MATSOLVERPETSC, MAT_FACTOR_ILU, &flg)) return PETSC_FALSE;
  return flg;
}
/* the harness's own types: KSPRegisterAll / PCRegisterAll (itregis.c, pcregis.c), for the types restated here */
PetscErrorCode PetscMiniInitialize(void) {
  static int done = 0;
  PetscErrorCode ierr;
  if (done) return 0;
  done = 1;
  ierr = PCRegister(PCNONE, 0, "PCCreate_None", PCCreate_None);CHKERRQ(ierr);
  ierr = PCRegister(PCJACOBI, 0, "PCCreate_Jacobi", PCCreate_Jacobi);CHKERRQ(ierr);
  ierr = PCRegister(PCBJACOBI, 0, "PCCreate_BJacobi", PCCreate_BJacobi);CHKERRQ(ierr);
  ierr = PCRegister(PCILU, 0, "PCCreate_ILU", PCCreate_ILU);CHKERRQ(ierr);      /* pcfactor.c: control flow only; the factored matrix comes from the operator's type */
  ierr = PCRegister(PCICC, 0, "PCCreate_ICC", PCCreate_ICC);CHKERRQ(ierr);
  ierr = KSPRegister(KSPCG, 0, "KSPCreate_CG", KSPCreate_CG);CHKERRQ(ierr);
  ierr = KSPRegister(KSPGROPPCG, 0, "KSPCreate_GROPPCG", KSPCreate_GROPPCG);CHKERRQ(ierr);
  ierr = KSPRegister(KSPPIPECG, 0, "KSPCreate_PIPECG", KSPCreate_PIPECG);CHKERRQ(ierr);
  ierr = KSPRegister(KSPGMRES, 0, "KSPCreate_GMRES", KSPCreate_GMRES);CHKERRQ(ierr);
  ierr = KSPRegister(KSPBCGS, 0, "KSPCreate_BCGS", KSPCreate_BCGS);CHKERRQ(ierr);
  ierr = KSPRegister(KSPPREONLY, 0, "KSPCreate_PREONLY", KSPCreate_PREONLY);CHKERRQ(ierr);
  return 0;
}

PetscErrorCode PCSetType(PC pc, PCType type) {
  PetscErrorCode ierr;
  if (!pc) SETERRQ(0, PETSC_ERR_ARG_NULL, "Null PC");
  if (!strcmp(pc->type_name, type)) return 0;
  for (int i = 0; i < n_pc_types; i++) {
    if (!strcmp(pc_types[i].name, type)) {
      if (pc->ops->destroy) { ierr = (*pc->ops->destroy)(pc);CHKERRQ(ierr); }
      memset(pc->ops, 0, sizeof(pc->ops));
      pc->data = NULL; pc->setupcalled = 0;
      ierr = (*pc_types[i].fn)(pc);CHKERRQ(ierr);
      snprintf(pc->type_name, sizeof(pc->type_name), "%s", type);
      return 0;
    }
  }
  SETERRQ(pc->comm, PETSC_ERR_ARG_UNKNOWN_TYPE, "Unable to find requested PC type %s", type);
}
PetscErrorCode PCGetType(PC pc, PCType *type) { *type = pc->type_name; return 0; }
PetscErrorCode PCSetOperators(PC pc, Mat Amat, Mat Pmat, MatStructure flag) {
  (void)flag;
  if (!pc) SETERRQ(0, PETSC_ERR_ARG_NULL, "Null PC");
  pc->mat = Amat; pc->pmat = Pmat ? Pmat : Amat;
  if (pc->setupcalled == 2) pc->setupcalled = 1;
  return 0;
}
PetscErrorCode PCGetOperators(PC pc, Mat *Amat, Mat *Pmat, MatStructure *flag) {   /* precon.c PCGetOperators */
  if (!pc) SETERRQ(0, PETSC_ERR_ARG_NULL, "Null PC");
  if (Amat) *Amat = pc->mat;
  if (Pmat) *Pmat = pc->pmat;
  if (flag) *flag = SAME_NONZERO_PATTERN;
  return 0;
}
PetscErrorCode PCSetFromOptions(PC pc) {
  PetscErrorCode ierr; char t[64]; PetscBool set;
  ierr = PetscOptionsGetString(pc->prefix, "-pc_type", t, sizeof(t), &set);CHKERRQ(ierr);
  if (set) { ierr = PCSetType(pc, t);CHKERRQ(ierr); }
  if (pc->ops->setfromoptions) { ierr = (*pc->ops->setfromoptions)(pc);CHKERRQ(ierr); }
  return 0;
}
PetscErrorCode PCSetUp(PC pc) {   /* precon.c:~800 */
  PetscErrorCode ierr;
  if (!pc) SETERRQ(0, PETSC_ERR_ARG_NULL, "Null PC");
  if (pc->setupcalled > 1) return 0;
  if (!pc->mat) SETERRQ(pc->comm, PETSC_ERR_ARG_WRONGSTATE, "Matrix must be set first");
  if (!pc->type_name[0]) {
    /* PCGetDefaultType_Private (precon.c:14-53): ILU on one process when the matrix can be factored, block Jacobi on several */
    if (pc->comm->size > 1) { ierr = PCSetType(pc, PCBJACOBI);CHKERRQ(ierr); }
    else { ierr = PCSetType(pc, mat_has_ilu(pc->pmat) ? PCILU : PCJACOBI);CHKERRQ(ierr); }
  }
  if (pc->ops->setup) { ierr = (*pc->ops->setup)(pc);CHKERRQ(ierr); }
  pc->setupcalled = 2;
  return 0;
}
PetscErrorCode PCSetUpOnBlocks(PC pc) {   /* precon.c:857-868 */
  if (!pc) SETERRQ(0, PETSC_ERR_ARG_NULL, "Null PC");
  if (!pc->ops->setuponblocks) return 0;
  return (*pc->ops->setuponblocks)(pc);
}
PetscErrorCode PCApply(PC pc, Vec x, Vec y) {   /* precon.c:369-388 */
  PetscErrorCode ierr;
  if (!pc) SETERRQ(0, PETSC_ERR_ARG_NULL, "Null PC");
  if (x == y) SETERRQ(pc->comm, PETSC_ERR_ARG_IDN, "x and y must be different vectors");
  if (pc->setupcalled < 2) { ierr = PCSetUp(pc);CHKERRQ(ierr); }
  if (!pc->ops->apply) SETERRQ(pc->comm, PETSC_ERR_SUP, "PC does not have apply");
  ierr = (*pc->ops->apply)(pc, x, y);CHKERRQ(ierr);
  return 0;
}
PetscErrorCode PCDestroy(PC *ppc) {
  PetscErrorCode ierr;
  PC pc = *ppc;
  if (!pc) return 0;
  if (pc->ops->destroy) { ierr = (*pc->ops->destroy)(pc);CHKERRQ(ierr); }
  ierr = PetscObjectListDestroy_Private((PetscObject)pc);CHKERRQ(ierr);
  free(pc); *ppc = NULL;
  return 0;
}

/* ---------------------------------------------------------------- PCNONE (src/ksp/pc/impls/none/none.c) */
static PetscErrorCode PCApply_None(PC pc, Vec x, Vec y) { (void)pc; return VecCopy(x, y); }
PetscErrorCode PCCreate_None(PC pc) { pc->ops->apply = PCApply_None; return 0; }

/* ---------------------------------------------------------------- PCJACOBI */
typedef struct { Vec diag; } PC_Jacobi;

/* PCSetUp_Jacobi, jacobi.c:125-198: MatGetDiagonal -> VecReciprocal -> zeros replaced by 1 in a host loop over
 * VecGetArray (jacobi.c:182-190).  A vector type may offer the last two steps as one method of its own,
 * "VecJacobiInvert_C" (d = (d == 0) ? 1 : 1/d, same values), which spares an accelerator type the host round trip. */
static PetscErrorCode PCSetUp_Jacobi(PC pc) {
  PetscErrorCode ierr;
  PC_Jacobi *jac = (PC_Jacobi *)pc->data;
  PetscVoidFunction f;
  if (!jac->diag) { ierr = MatGetVecs(pc->pmat, &jac->diag, NULL);CHKERRQ(ierr); }
  ierr = MatGetDiagonal(pc->pmat, jac->diag);CHKERRQ(ierr);
  ierr = PetscObjectQueryFunction((PetscObject)jac->diag, "VecJacobiInvert_C", &f);CHKERRQ(ierr);
  if (f) { ierr = ((PetscErrorCode (*)(Vec))f)(jac->diag);CHKERRQ(ierr); PetscObjectStateIncrease(jac->diag); }
  else {
    PetscScalar *x; PetscInt n;
    ierr = VecReciprocal(jac->diag);CHKERRQ(ierr);
    ierr = VecGetLocalSize(jac->diag, &n);CHKERRQ(ierr);
    ierr = VecGetArray(jac->diag, &x);CHKERRQ(ierr);
    for (PetscInt i = 0; i < n; i++) if (x[i] == 0.0) x[i] = 1.0;
    ierr = VecRestoreArray(jac->diag, &x);CHKERRQ(ierr);
  }
  return 0;
}
static PetscErrorCode PCApply_Jacobi(PC pc, Vec x, Vec y) {   /* jacobi.c:266-277 */
  PC_Jacobi *jac = (PC_Jacobi *)pc->data;
  return VecPointwiseMult(y, x, jac->diag);
}
static PetscErrorCode PCDestroy_Jacobi(PC pc) {
  PC_Jacobi *jac = (PC_Jacobi *)pc->data;
  if (jac) { PetscErrorCode ierr = VecDestroy(&jac->diag);CHKERRQ(ierr); free(jac); pc->data = NULL; }
  return 0;
}
PetscErrorCode PCCreate_Jacobi(PC pc) {
  PC_Jacobi *jac;
  PetscErrorCode ierr = PetscMalloc(sizeof(*jac), &jac);CHKERRQ(ierr);
  jac->diag = NULL;
  pc->data = jac;
  pc->ops->setup = PCSetUp_Jacobi; pc->ops->apply = PCApply_Jacobi; pc->ops->destroy = PCDestroy_Jacobi;
  return 0;
}

/* ---------------------------------------------------------------- PCBJACOBI: one block per rank (PCSetUp_BJacobi_Singleblock,
 * bjacobi.c:858-923) or several local blocks (PCSetUp_BJacobi_Multiblock, bjacobi.c:1060-1230); blocks that span ranks
 * (-pc_bjacobi_blocks < size: the multiproc case) are outside the ported path */
typedef struct {
  PetscInt nloc;            /* local blocks */
  PetscInt *starts;         /* nloc + 1 local row offsets */
  KSP *ksp;                 /* one sub-KSP per block, options prefix "sub_" */
  Vec *x, *y;               /* sequential work vectors per block */
  Mat *block;               /* nloc == 1: the diagonal block itself (not owned); else extracted copies (owned) */
  PetscBool owns_blocks;
  PetscBool merged;         /* several blocks, every sub-solver KSPPREONLY + PCILU: ONE solver over the block-diagonal matrix */
} PC_BJacobi;
typedef PetscErrorCode (*MatSeqAIJGetArraysFn)(Mat, PetscInt *, const PetscInt **, const PetscInt **, const PetscScalar **);

/* the [s, e) x [s, e) diagonal block of a sequential AIJ matrix, as a matrix of the same type (MatGetSubMatrices with the
 * contiguous index sets of bjacobi.c:1115-1130) */
static PetscErrorCode extract_diagonal_block(Mat A, PetscInt s, PetscInt e, Mat *sub) {
  PetscErrorCode ierr;
  PetscVoidFunction f = NULL;
  PetscInt m; const PetscInt *ai, *aj; const PetscScalar *aa;
  ierr = PetscObjectQueryFunction((PetscObject)A, "MatSeqAIJGetArrays_C", &f);CHKERRQ(ierr);
  if (!f) SETERRQ(A->comm, PETSC_ERR_SUP, "several block-Jacobi blocks per process need a sequential AIJ diagonal block, got %s", A->type_name);
  ierr = ((MatSeqAIJGetArraysFn)f)(A, &m, &ai, &aj, &aa);CHKERRQ(ierr);
  if (A->rmap->n != m) SETERRQ(A->comm, PETSC_ERR_SUP, "several block-Jacobi blocks per process are ported for block size 1");
  PetscInt nz = 0, *si, *sj; PetscScalar *sa;
  for (PetscInt r = s; r < e; r++) for (PetscInt k = ai[r]; k < ai[r + 1]; k++) if (aj[k] >= s && aj[k] < e) nz++;
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(e - s + 1), &si);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(nz, 1), &sj);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)PetscMax(nz, 1), &sa);CHKERRQ(ierr);
  nz = 0; si[0] = 0;
  for (PetscInt r = s; r < e; r++) {
    for (PetscInt k = ai[r]; k < ai[r + 1]; k++) if (aj[k] >= s && aj[k] < e) { sj[nz] = aj[k] - s; sa[nz] = aa[k]; nz++; }
    si[r - s + 1] = nz;
  }
  ierr = MatCreateSeqAIJWithArrays(PETSC_COMM_SELF, e - s, e - s, si, sj, sa, sub);   /* the arrays are copied */
  free(si); free(sj); free(sa);
  CHKERRQ(ierr);
  return 0;
}

/* the matrix with every coupling between different blocks dropped.  Its ILU(0) factors are the blocks' ILU(0) factors side by
 * side (nothing couples them), and one triangular solve over it does all blocks' solves at once -- their dependency levels
 * overlap -- with each block's arithmetic unchanged: what the reference computes block after block (bjacobi.c:1140-1180) */
static PetscErrorCode extract_block_diagonal(Mat A, PetscInt nloc, const PetscInt *starts, Mat *sub) {
  PetscErrorCode ierr;
  PetscVoidFunction f = NULL;
  PetscInt m; const PetscInt *ai, *aj; const PetscScalar *aa;
  ierr = PetscObjectQueryFunction((PetscObject)A, "MatSeqAIJGetArrays_C", &f);CHKERRQ(ierr);
  if (!f) SETERRQ(A->comm, PETSC_ERR_SUP, "several block-Jacobi blocks per process need a sequential AIJ diagonal block, got %s", A->type_name);
  ierr = ((MatSeqAIJGetArraysFn)f)(A, &m, &ai, &aj, &aa);CHKERRQ(ierr);
  if (A->rmap->n != m) SETERRQ(A->comm, PETSC_ERR_SUP, "several block-Jacobi blocks per process are ported for block size 1");
  PetscInt nz = 0, *si, *sj; PetscScalar *sa;
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(m + 1), &si);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(ai[m], 1), &sj);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)PetscMax(ai[m], 1), &sa);CHKERRQ(ierr);
  si[0] = 0;
  for (PetscInt b = 0; b < nloc; b++)
    for (PetscInt r = starts[b]; r < starts[b + 1]; r++) {
      for (PetscInt k = ai[r]; k < ai[r + 1]; k++) if (aj[k] >= starts[b] && aj[k] < starts[b + 1]) { sj[nz] = aj[k]; sa[nz] = aa[k]; nz++; }
      si[r + 1] = nz;
    }
  ierr = MatCreateSeqAIJWithArrays(PETSC_COMM_SELF, m, m, si, sj, sa, sub);
  free(si); free(sj); free(sa);
  CHKERRQ(ierr);
  return 0;
}

static PetscErrorCode PCSetUp_BJacobi(PC pc) {
  PetscErrorCode ierr;
  PC_BJacobi *bj = (PC_BJacobi *)pc->data;
  PetscInt nb = pc->comm->size; PetscBool set; char prefix[80];
  Mat diag;
  snprintf(prefix, sizeof(prefix), "%s", pc->prefix);
  ierr = PetscOptionsGetInt(prefix, "-pc_bjacobi_blocks", &nb, &set);CHKERRQ(ierr);
  if (!set) nb = pc->comm->size;
  if (nb < pc->comm->size) SETERRQ(pc->comm, PETSC_ERR_SUP, "%d blocks on %d processes: blocks that span processes (PCSetUp_BJacobi_Multiproc) are outside the ported path", nb, pc->comm->size);
  /* bjacobi.c:118-150: the blocks are dealt evenly to the ranks, a rank's rows evenly to its blocks */
  const PetscInt nloc = nb / pc->comm->size + ((nb % pc->comm->size) > pc->comm->rank ? 1 : 0);
  ierr = MatGetDiagonalBlock(pc->pmat, &diag);CHKERRQ(ierr);
  const PetscInt M = diag->rmap->n;
  if (nloc > 1 && nloc > M && M > 0) SETERRQ(pc->comm, PETSC_ERR_ARG_OUTOFRANGE, "more block-Jacobi blocks (%d) than local rows (%d)", nloc, M);
  if (bj->ksp && bj->nloc != nloc) SETERRQ(pc->comm, PETSC_ERR_SUP, "the number of block-Jacobi blocks cannot change after set-up");
  if (!bj->ksp) {
    bj->nloc = nloc;
    ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(nloc + 1), &bj->starts);CHKERRQ(ierr);
    ierr = PetscMalloc(sizeof(KSP) * (size_t)nloc, &bj->ksp);CHKERRQ(ierr);
    ierr = PetscMalloc(sizeof(Vec) * (size_t)nloc, &bj->x);CHKERRQ(ierr);
    ierr = PetscMalloc(sizeof(Vec) * (size_t)nloc, &bj->y);CHKERRQ(ierr);
    ierr = PetscMalloc(sizeof(Mat) * (size_t)nloc, &bj->block);CHKERRQ(ierr);
    memset(bj->ksp, 0, sizeof(KSP) * (size_t)nloc); memset(bj->x, 0, sizeof(Vec) * (size_t)nloc);
    memset(bj->y, 0, sizeof(Vec) * (size_t)nloc); memset(bj->block, 0, sizeof(Mat) * (size_t)nloc);
    bj->starts[0] = 0;
    for (PetscInt i = 0; i < nloc; i++) bj->starts[i + 1] = bj->starts[i] + M / nloc + ((M % nloc) > i ? 1 : 0);
    bj->owns_blocks = (PetscBool)(nloc > 1);
    if (nloc > 1) {   /* all sub-solvers "apply ILU(0) once" (the defaults): one solver over the block-diagonal matrix */
      char kt[32] = "", pt[32] = "", mg[16] = ""; PetscBool ks, ps, ms;
      snprintf(prefix, sizeof(prefix), "%ssub_", pc->prefix);
      ierr = PetscOptionsGetString(prefix, "-ksp_type", kt, sizeof(kt), &ks);CHKERRQ(ierr);
      ierr = PetscOptionsGetString(prefix, "-pc_type", pt, sizeof(pt), &ps);CHKERRQ(ierr);
      ierr = PetscOptionsGetString(pc->prefix, "-pc_bjacobi_merge_blocks", mg, sizeof(mg), &ms);CHKERRQ(ierr);   /* 0: block after block */
      /* ILU(0), or ICC(0) when the type can factor the blocks of a block-diagonal matrix independently ("PCFactorSetIndependentBlocks_C", asked below) */
      bj->merged = (PetscBool)((!ks || !strcmp(kt, KSPPREONLY)) && (!ps || !strcmp(pt, PCILU) || !strcmp(pt, PCICC)) && !(ms && (!strcmp(mg, "0") || !strcmp(mg, "false"))));
    }
  }
  const PetscInt nsolvers = bj->merged ? 1 : nloc;
  for (PetscInt i = 0; i < nsolvers; i++) {
    if (nloc == 1) bj->block[0] = diag;
    else {
      if (bj->block[i]) { ierr = MatDestroy(&bj->block[i]);CHKERRQ(ierr); }     /* values may have changed: extract again */
      if (bj->merged) { ierr = extract_block_diagonal(diag, nloc, bj->starts, &bj->block[0]);CHKERRQ(ierr); }
      else { ierr = extract_diagonal_block(diag, bj->starts[i], bj->starts[i + 1], &bj->block[i]);CHKERRQ(ierr); }
    }
    if (!bj->ksp[i]) {
      ierr = KSPCreate(PETSC_COMM_SELF, &bj->ksp[i]);CHKERRQ(ierr);
      ierr = KSPSetType(bj->ksp[i], KSPPREONLY);CHKERRQ(ierr);
      snprintf(prefix, sizeof(prefix), "%ssub_", pc->prefix);
      ierr = KSPSetOptionsPrefix(bj->ksp[i], prefix);CHKERRQ(ierr);
      /* the sub-PC's type is left unset: PCSetUp picks the default (ILU(0) on a sequential AIJ block, precon.c:14-53) */
      ierr = MatGetVecs(bj->block[i], &bj->x[i], &bj->y[i]);CHKERRQ(ierr);
    }
    ierr = KSPSetOperators(bj->ksp[i], bj->block[i], bj->block[i], SAME_NONZERO_PATTERN);CHKERRQ(ierr);
    ierr = KSPSetFromOptions(bj->ksp[i]);CHKERRQ(ierr);
    if (bj->merged) {   /* a factorisation with a shift strategy must treat the blocks as the separate matrices they stand for */
      PetscVoidFunction fb = NULL;
      PC sub = bj->ksp[i]->pc;
      if (!sub->type_name[0] && mat_has_ilu(bj->block[i])) { ierr = PCSetType(sub, PCILU);CHKERRQ(ierr); }   /* the default PCSetUp would pick (precon.c:14-53), now, so that it can be asked */
      ierr = PetscObjectQueryFunction((PetscObject)bj->ksp[i]->pc, "PCFactorSetIndependentBlocks_C", &fb);CHKERRQ(ierr);
      if (fb) { ierr = ((PetscErrorCode (*)(PC, PetscInt, const PetscInt *))fb)(bj->ksp[i]->pc, nloc, bj->starts);CHKERRQ(ierr); }
      else if (!strcmp(bj->ksp[i]->pc->type_name, PCICC)) SETERRQ(pc->comm, PETSC_ERR_SUP, "this PCICC cannot factor independent blocks: use -pc_bjacobi_merge_blocks 0");
    }
    /* the block solvers themselves are set up by PCSetUpOnBlocks, i.e. at the start of KSPSolve (or at their first application): a program
     * may still change them through PCBJacobiGetSubKSP after KSPSetUp (tutorials/ex7.c:166-195) */
  }
  if (bj->merged) for (PetscInt i = 1; i < nloc; i++) bj->ksp[i] = bj->ksp[0];   /* PCBJacobiGetSubKSP: every block answers with the one solver */
  return 0;
}
static PetscErrorCode PCSetUpOnBlocks_BJacobi(PC pc) {   /* PCSetUpOnBlocks_BJacobi_Singleblock / _Multiblock, bjacobi.c:726,985 */
  PC_BJacobi *bj = (PC_BJacobi *)pc->data;
  const PetscInt nsolvers = bj->merged ? 1 : bj->nloc;
  for (PetscInt i = 0; i < nsolvers; i++) { PetscErrorCode ierr = KSPSetUp(bj->ksp[i]);CHKERRQ(ierr); }
  return 0;
}

/* PCApply_BJacobi_Singleblock / _Multiblock, bjacobi.c:738-761,1140-1180: the local arrays of the parallel vectors are placed
 * into the sequential work vectors of each block (VecGetArray + VecPlaceArray at the block's offset), the sub-KSP solves, the
 * arrays are reset.  A vector type whose data does not live on the host may offer the same aliasing as a method of its own
 * ("VecShareSubArrayBegin_C" / "VecShareSubArrayEnd_C": sub takes parent's storage from an offset on; write != 0 for the
 * output vector), which avoids a host round trip per application. */
typedef PetscErrorCode (*VecShareSubFn)(Vec sub, Vec parent, PetscInt offset, PetscBool write);
static PetscErrorCode PCApply_BJacobi(PC pc, Vec x, Vec y) {
  PetscErrorCode ierr, ierr2;
  PC_BJacobi *bj = (PC_BJacobi *)pc->data;
  PetscVoidFunction fb, fe;
  ierr = PetscObjectQueryFunction((PetscObject)bj->x[0], "VecShareSubArrayBegin_C", &fb);CHKERRQ(ierr);
  ierr = PetscObjectQueryFunction((PetscObject)bj->x[0], "VecShareSubArrayEnd_C", &fe);CHKERRQ(ierr);
  const PetscInt nsolvers = bj->merged ? 1 : bj->nloc;
  if (fb && fe) {
    for (PetscInt i = 0; i < nsolvers; i++) {
      const PetscInt off = bj->merged ? 0 : bj->starts[i];
      ierr = ((VecShareSubFn)fb)(bj->x[i], x, off, PETSC_FALSE);CHKERRQ(ierr);
      ierr = ((VecShareSubFn)fb)(bj->y[i], y, off, PETSC_TRUE);
      if (ierr) { ((VecShareSubFn)fe)(bj->x[i], x, off, PETSC_FALSE); CHKERRQ(ierr); }
      ierr = KSPSolve(bj->ksp[i], bj->x[i], bj->y[i]);
      /* the aliases come off first, also when the sub-solve failed: the work vectors own their storage again */
      ierr2 = ((VecShareSubFn)fe)(bj->x[i], x, off, PETSC_FALSE);
      if (!ierr2) ierr2 = ((VecShareSubFn)fe)(bj->y[i], y, off, PETSC_TRUE);
      CHKERRQ(ierr); CHKERRQ(ierr2);
    }
  } else {
    const PetscScalar *xa; PetscScalar *ya;
    ierr = VecGetArrayRead(x, &xa);CHKERRQ(ierr);
    ierr = VecGetArray(y, &ya);CHKERRQ(ierr);
    ierr = 0; ierr2 = 0;
    for (PetscInt i = 0; i < nsolvers && !ierr && !ierr2; i++) {
      const PetscInt off = bj->merged ? 0 : bj->starts[i];
      ierr = VecPlaceArray(bj->x[i], xa + off);
      if (!ierr) ierr = VecPlaceArray(bj->y[i], ya + off);
      if (!ierr) ierr = KSPSolve(bj->ksp[i], bj->x[i], bj->y[i]);
      ierr2 = VecResetArray(bj->x[i]);
      if (!ierr2) ierr2 = VecResetArray(bj->y[i]);
    }
    CHKERRQ(ierr); CHKERRQ(ierr2);
    ierr = VecRestoreArrayRead(x, &xa);CHKERRQ(ierr);
    ierr = VecRestoreArray(y, &ya);CHKERRQ(ierr);
  }
  PetscObjectStateIncrease(y);   /* y changed through the alias: cached norms are stale (VecRestoreArray does this in bjacobi.c:758) */
  return 0;
}
static PetscErrorCode PCDestroy_BJacobi(PC pc) {
  PetscErrorCode ierr;
  PC_BJacobi *bj = (PC_BJacobi *)pc->data;
  if (!bj) return 0;
  for (PetscInt i = 0; i < (bj->merged ? 1 : bj->nloc) && bj->ksp; i++) {
    ierr = KSPDestroy(&bj->ksp[i]);CHKERRQ(ierr);
    ierr = VecDestroy(&bj->x[i]);CHKERRQ(ierr);
    ierr = VecDestroy(&bj->y[i]);CHKERRQ(ierr);
    if (bj->owns_blocks && bj->block[i]) { ierr = MatDestroy(&bj->block[i]);CHKERRQ(ierr); }
  }
  free(bj->starts); free(bj->ksp); free(bj->x); free(bj->y); free(bj->block);
  free(bj); pc->data = NULL;
  return 0;
}
PetscErrorCode PCBJacobiGetSubKSP(PC pc, PetscInt *n_local, PetscInt *first_local, KSP **ksp) {
  if (strcmp(pc->type_name, PCBJACOBI)) SETERRQ(pc->comm, PETSC_ERR_ARG_WRONG, "Cannot get subsolvers for this preconditioner");
  PC_BJacobi *bj = (PC_BJacobi *)pc->data;
  if (!bj->ksp) SETERRQ(pc->comm, PETSC_ERR_ARG_WRONGSTATE, "Must call KSPSetUp() or PCSetUp() first");
  if (n_local) *n_local = bj->nloc;
  if (first_local) {   /* blocks are numbered rank after rank (bjacobi.c PCBJacobiGetSubKSP_BJacobi) */
    PetscInt nb = pc->comm->size; PetscBool set;
    PetscOptionsGetInt(pc->prefix, "-pc_bjacobi_blocks", &nb, &set);
    if (!set) nb = pc->comm->size;
    PetscInt first = 0;
    for (int r = 0; r < pc->comm->rank; r++) first += nb / pc->comm->size + ((nb % pc->comm->size) > r ? 1 : 0);
    *first_local = first;
  }
  if (ksp) *ksp = bj->ksp;
  return 0;
}
PetscErrorCode PCCreate_BJacobi(PC pc) {
  PC_BJacobi *bj;
  PetscErrorCode ierr = PetscMalloc(sizeof(*bj), &bj);CHKERRQ(ierr);
  memset(bj, 0, sizeof(*bj));
  pc->data = bj;
  pc->ops->setup = PCSetUp_BJacobi; pc->ops->apply = PCApply_BJacobi; pc->ops->destroy = PCDestroy_BJacobi; pc->ops->setuponblocks = PCSetUpOnBlocks_BJacobi;
  return 0;
}
