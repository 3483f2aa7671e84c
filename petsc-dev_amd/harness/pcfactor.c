/* HARNESS file: PCILU and PCICC as the reference has them (src/ksp/pc/impls/factor/ilu/ilu.c, icc/icc.c, factor.c) -- pure control
 * flow over the Mat factorisation interface: MatGetFactor(pmat, "petsc", MAT_FACTOR_ILU | ICC) -> symbolic -> numeric at set-up,
 * MatSolve at every application.  WHICH factorisation and WHICH triangular solve run is the factored matrix's business: the
 * operator's type answers MatGetFactor ("MatGetFactor_petsc_C"), exactly how the reference's GPU back end gets its solves under
 * an unchanged PCILU (src/mat/impls/aij/seq/seqcusparse/aijcusparse.cu:57-75,372-445).  Nothing here knows about devices.  Inside
 * a PETSc tree this file is not used: PETSc's own PCILU / PCICC / PCBJACOBI drive the same slots of the plug-in's factored matrix. */
#include "petscimpl.h"

typedef struct {
  Mat fact;
  MatFactorInfo info;
  MatFactorType factortype;
  int symbolic_done;
  PetscInt nblk, *blk;      /* "PCFactorSetIndependentBlocks_C" (block Jacobi solving its blocks as one block-diagonal system) */
} PC_Factor;

static PetscErrorCode PCFactorGetMatrix_Factor(PC pc, Mat *mat) { *mat = ((PC_Factor *)pc->data)->fact; return 0; }   /* factor.c */
PetscErrorCode PCFactorGetMatrix(PC pc, Mat *mat) {   /* precon.c:1017 */
  if (!pc) SETERRQ(0, PETSC_ERR_ARG_NULL, "Null PC");
  if (!pc->ops->getfactoredmatrix) SETERRQ(pc->comm, PETSC_ERR_SUP, "PC type does not support getting factor matrix");
  return (*pc->ops->getfactoredmatrix)(pc, mat);
}

/* the harness's block Jacobi hands several ILU(0) / ICC(0) blocks to ONE factorisation of the block-diagonal matrix; a
 * factorisation with a shift strategy has to know where the blocks are to treat each as the matrix of its own it stands for.
 * Passed on to the factored matrix ("MatFactorSetIndependentBlocks_C") if its type knows the method. */
static PetscErrorCode PCFactorSetIndependentBlocks_Factor(PC pc, PetscInt nblk, const PetscInt *starts) {
  PC_Factor *f = (PC_Factor *)pc->data;
  PetscErrorCode ierr;
  free(f->blk); f->blk = NULL; f->nblk = 0;
  if (nblk > 0) {
    ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(nblk + 1), &f->blk);CHKERRQ(ierr);
    memcpy(f->blk, starts, sizeof(PetscInt) * (size_t)(nblk + 1));
    f->nblk = nblk;
  }
  if (pc->setupcalled == 2) pc->setupcalled = 1;
  return 0;
}

static PetscErrorCode PCSetFromOptions_Factor(PC pc) {   /* factor.c PCSetFromOptions_Factor, ilu.c:110-150: the options the path uses */
  PC_Factor *f = (PC_Factor *)pc->data;
  PetscErrorCode ierr; PetscInt iv; PetscBool set; char t[64]; PetscReal r;
  ierr = PetscOptionsGetInt(pc->prefix, "-pc_factor_levels", &iv, &set);CHKERRQ(ierr);
  if (set) f->info.levels = iv;
  ierr = PetscOptionsGetString(pc->prefix, "-pc_factor_shift_type", t, sizeof(t), &set);CHKERRQ(ierr);
  if (set) {
    if (!strcmp(t, "none") || !strcmp(t, "NONE")) f->info.shifttype = (PetscReal)MAT_SHIFT_NONE;
    else if (!strcmp(t, "nonzero") || !strcmp(t, "NONZERO")) f->info.shifttype = (PetscReal)MAT_SHIFT_NONZERO;
    else if (!strcmp(t, "positive_definite") || !strcmp(t, "POSITIVE_DEFINITE")) f->info.shifttype = (PetscReal)MAT_SHIFT_POSITIVE_DEFINITE;
    else if (!strcmp(t, "inblocks") || !strcmp(t, "INBLOCKS")) f->info.shifttype = (PetscReal)MAT_SHIFT_INBLOCKS;
    else SETERRQ(pc->comm, PETSC_ERR_ARG_UNKNOWN_TYPE, "Unknown shift type %s", t);
  }
  ierr = PetscOptionsGetReal(pc->prefix, "-pc_factor_shift_amount", &r, &set);CHKERRQ(ierr);
  if (set) f->info.shiftamount = r;
  ierr = PetscOptionsGetReal(pc->prefix, "-pc_factor_zeropivot", &r, &set);CHKERRQ(ierr);
  if (set) f->info.zeropivot = r;
  return 0;
}

/* PCSetUp_ILU (ilu.c:152-240) / PCSetup_ICC (icc.c:26-56), natural ordering (row = col = perm = identity, passed as NULL), not in
 * place: the first set-up gets the factored matrix and does the symbolic phase, every set-up the numeric one */
static PetscErrorCode PCSetUp_Factor(PC pc) {
  PC_Factor *f = (PC_Factor *)pc->data;
  PetscErrorCode ierr;
  if (!f->fact) {
    ierr = MatGetFactor(pc->pmat, MATSOLVERPETSC, f->factortype, &f->fact);CHKERRQ(ierr);
    ierr = MatSetOptionsPrefix(f->fact, pc->prefix);CHKERRQ(ierr);
    f->symbolic_done = 0;
  }
  {
    PetscVoidFunction fb = NULL;
    ierr = PetscObjectQueryFunction((PetscObject)f->fact, "MatFactorSetIndependentBlocks_C", &fb);CHKERRQ(ierr);
    if (fb) { ierr = ((PetscErrorCode (*)(Mat, PetscInt, const PetscInt *))fb)(f->fact, f->nblk, f->blk);CHKERRQ(ierr); }
    else if (f->nblk > 0 && f->info.shifttype != (PetscReal)MAT_SHIFT_NONE) SETERRQ(pc->comm, PETSC_ERR_SUP, "this factorisation cannot treat the blocks of a block-diagonal matrix independently");
  }
  if (!f->symbolic_done) {
    if (f->factortype == MAT_FACTOR_ILU) { ierr = MatILUFactorSymbolic(f->fact, pc->pmat, NULL, NULL, &f->info);CHKERRQ(ierr); }
    else { ierr = MatICCFactorSymbolic(f->fact, pc->pmat, NULL, &f->info);CHKERRQ(ierr); }
    f->symbolic_done = 1;
  }
  if (f->factortype == MAT_FACTOR_ILU) { ierr = MatLUFactorNumeric(f->fact, pc->pmat, &f->info);CHKERRQ(ierr); }
  else { ierr = MatCholeskyFactorNumeric(f->fact, pc->pmat, &f->info);CHKERRQ(ierr); }
  return 0;
}
static PetscErrorCode PCApply_Factor(PC pc, Vec x, Vec y) { return MatSolve(((PC_Factor *)pc->data)->fact, x, y); }   /* PCApply_ILU ilu.c:262, PCApply_ICC icc.c:65 */
static PetscErrorCode PCDestroy_Factor(PC pc) {
  PC_Factor *f = (PC_Factor *)pc->data;
  if (f) { PetscErrorCode ierr = MatDestroy(&f->fact);CHKERRQ(ierr); free(f->blk); free(f); pc->data = NULL; }
  (void)PetscObjectComposeFunction((PetscObject)pc, "PCFactorSetIndependentBlocks_C", "", (PetscVoidFunction)NULL);
  return 0;
}
static PetscErrorCode create_factor(PC pc, MatFactorType ft, MatFactorShiftType shift) {
  PC_Factor *f;
  PetscErrorCode ierr = PetscMalloc(sizeof(*f), &f);CHKERRQ(ierr);
  memset(f, 0, sizeof(*f));
  ierr = MatFactorInfoInitialize(&f->info);CHKERRQ(ierr);
  f->factortype = ft;
  f->info.levels = 0.; f->info.fill = 1.0;
  f->info.dt = f->info.dtcount = f->info.dtcol = PETSC_DEFAULT;
  f->info.shifttype = (PetscReal)shift;
  f->info.shiftamount = 100.0 * 2.220446049250313e-16;    /* 100 PETSC_MACHINE_EPSILON, ilu.c:388-389 / icc.c:198-199 */
  f->info.zeropivot = 100.0 * 2.220446049250313e-16;
  f->info.pivotinblocks = 1.0;
  pc->data = f;
  pc->ops->setup = PCSetUp_Factor; pc->ops->apply = PCApply_Factor; pc->ops->destroy = PCDestroy_Factor;
  pc->ops->setfromoptions = PCSetFromOptions_Factor; pc->ops->getfactoredmatrix = PCFactorGetMatrix_Factor;
  ierr = PetscObjectComposeFunction((PetscObject)pc, "PCFactorSetIndependentBlocks_C", "PCFactorSetIndependentBlocks_Factor", (PetscVoidFunction)PCFactorSetIndependentBlocks_Factor);CHKERRQ(ierr);
  return 0;
}
PetscErrorCode PCCreate_ILU(PC pc) { return create_factor(pc, MAT_FACTOR_ILU, MAT_SHIFT_NONZERO); }             /* ilu.c:375-389 */
PetscErrorCode PCCreate_ICC(PC pc) { return create_factor(pc, MAT_FACTOR_ICC, MAT_SHIFT_POSITIVE_DEFINITE); }   /* icc.c:189-199 */
