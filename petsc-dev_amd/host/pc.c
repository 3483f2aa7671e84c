/* PC interface (src/ksp/pc/interface/precon.c) and the preconditioners of the north star:
 * PCNONE, PCJACOBI (src/ksp/pc/impls/jacobi/jacobi.c:125-198,266-277) and PCBJACOBI with one block
 * per rank (src/ksp/pc/impls/bjacobi/bjacobi.c:738-761,858-923). */
#include "petscimpl.h"

PetscErrorCode PCCreate(MPI_Comm comm, PC *newpc) {
  PC pc;
  PetscErrorCode ierr = PetscMalloc(sizeof(*pc), &pc);CHKERRQ(ierr);
  memset(pc, 0, sizeof(*pc));
  pc->comm = comm;
  *newpc = pc;
  return 0;
}
static struct { const char *name; PetscErrorCode (*fn)(PC); } pc_types[] = {
  {PCNONE, PCCreate_None}, {PCJACOBI, PCCreate_Jacobi}, {PCBJACOBI, PCCreate_BJacobi}, {PCILU, PCCreate_ILU}, {NULL, NULL}};

PetscErrorCode PCSetType(PC pc, PCType type) {
  PetscErrorCode ierr;
  if (!pc) SETERRQ(0, PETSC_ERR_ARG_NULL, "Null PC");
  if (!strcmp(pc->type_name, type)) return 0;
  for (int i = 0; pc_types[i].name; i++) {
    if (!strcmp(pc_types[i].name, type)) {
      if (pc->ops->destroy) { ierr = (*pc->ops->destroy)(pc);CHKERRQ(ierr); }
      memset(pc->ops, 0, sizeof(pc->ops));
      pc->data = NULL; pc->setupcalled = 0;
      ierr = (*pc_types[i].fn)(pc);CHKERRQ(ierr);
      snprintf(pc->type_name, sizeof(pc->type_name), "%s", type);
      return 0;
    }
  }
  SETERRQ(pc->comm, PETSC_ERR_ARG_UNKNOWN_TYPE, "Unable to find requested PC type %s (ported: none, jacobi, bjacobi, ilu)", type);
}
PetscErrorCode PCGetType(PC pc, PCType *type) { *type = pc->type_name; return 0; }
PetscErrorCode PCSetOperators(PC pc, Mat Amat, Mat Pmat, MatStructure flag) {
  (void)flag;
  if (!pc) SETERRQ(0, PETSC_ERR_ARG_NULL, "Null PC");
  pc->mat = Amat; pc->pmat = Pmat ? Pmat : Amat;
  if (pc->setupcalled == 2) pc->setupcalled = 1;
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
    /* PCGetDefaultType_Private (precon.c:14-53): ILU on one process (when the matrix can be factored, i.e.
     * sequential AIJ here), block Jacobi on several */
    if (pc->comm->size > 1) { ierr = PCSetType(pc, PCBJACOBI);CHKERRQ(ierr); }
    else { ierr = PCSetType(pc, !strcmp(pc->pmat->type_name, MATSEQAIJHIPMI355X) ? PCILU : PCJACOBI);CHKERRQ(ierr); }
  }
  if (pc->ops->setup) { ierr = (*pc->ops->setup)(pc);CHKERRQ(ierr); }
  pc->setupcalled = 2;
  return 0;
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
  free(pc); *ppc = NULL;
  return 0;
}

/* ---------------------------------------------------------------- PCNONE (src/ksp/pc/impls/none/none.c) */
static PetscErrorCode PCApply_None(PC pc, Vec x, Vec y) { (void)pc; return VecCopy(x, y); }
PetscErrorCode PCCreate_None(PC pc) { pc->ops->apply = PCApply_None; return 0; }

/* ---------------------------------------------------------------- PCJACOBI */
typedef struct { Vec diag; } PC_Jacobi;

/* PCSetUp_Jacobi, jacobi.c:125-198: MatGetDiagonal -> VecReciprocal -> zeros replaced by 1.  The
 * reference does the last step in a host loop over VecGetArray (forcing a D2H/H2D round trip for a
 * GPU vector); d = (d == 0) ? 1 : 1/d is one device kernel here (same values: VecReciprocal leaves
 * zeros as zeros, the loop then sets them to 1). */
static PetscErrorCode PCSetUp_Jacobi(PC pc) {
  PetscErrorCode ierr;
  PC_Jacobi *jac = (PC_Jacobi *)pc->data;
  PetscScalar *d; PetscDeviceCtx *dc;
  if (!jac->diag) { ierr = MatGetVecs(pc->pmat, &jac->diag, NULL);CHKERRQ(ierr); }
  ierr = MatGetDiagonal(pc->pmat, jac->diag);CHKERRQ(ierr);
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  ierr = VecHIPMI355XGetArray(jac->diag, &d);CHKERRQ(ierr);
  CHKHIP(mi355x_vec_jacobi_invert(dc->h, (size_t)jac->diag->map->n, d, NULL));
  ierr = VecHIPMI355XRestoreArray(jac->diag, &d);CHKERRQ(ierr);
  return 0;
}
static PetscErrorCode PCApply_Jacobi(PC pc, Vec x, Vec y) {   /* jacobi.c:266-277 */
  PC_Jacobi *jac = (PC_Jacobi *)pc->data;
  return VecPointwiseMult(y, x, jac->diag);
}
/* for the fused CG update (krylov.c): the vector PCApply_Jacobi multiplies by, or NULL when pc is another type */
PetscErrorCode PCJacobiGetInverseDiagonal_Private(PC pc, Vec *d) {
  PetscErrorCode ierr;
  *d = NULL;
  if (!pc || pc->ops->apply != PCApply_Jacobi) return 0;
  if (pc->setupcalled < 2) { ierr = PCSetUp(pc);CHKERRQ(ierr); }
  *d = ((PC_Jacobi *)pc->data)->diag;
  return 0;
}
PetscBool PCIsNone_Private(PC pc) { return (PetscBool)(pc && pc->ops->apply == PCApply_None); }
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

/* ---------------------------------------------------------------- PCBJACOBI, one block per rank */
typedef struct {
  KSP ksp;        /* sub-KSP on the local diagonal block, options prefix "sub_" */
  Vec x, y;       /* sequential work vectors */
  Mat block;
} PC_BJacobi;

extern PetscErrorCode MatMPIAIJGetSeqAIJ(Mat, Mat *, Mat *, const PetscInt **);

static PetscErrorCode PCSetUp_BJacobi(PC pc) {   /* PCSetUp_BJacobi_Singleblock, bjacobi.c:858-923 */
  PetscErrorCode ierr;
  PC_BJacobi *bj = (PC_BJacobi *)pc->data;
  PetscInt nb; PetscBool set; char prefix[80];
  snprintf(prefix, sizeof(prefix), "%s", pc->prefix);
  ierr = PetscOptionsGetInt(prefix, "-pc_bjacobi_blocks", &nb, &set);CHKERRQ(ierr);
  if (set && nb != pc->comm->size) SETERRQ(pc->comm, PETSC_ERR_SUP, "%d blocks on %d processes: only one block per process (PCSetUp_BJacobi_Singleblock) is on the ported path", nb, pc->comm->size);
  if (!strcmp(pc->pmat->type_name, MATMPIAIJHIPMI355X)) { ierr = MatMPIAIJGetSeqAIJ(pc->pmat, &bj->block, NULL, NULL);CHKERRQ(ierr); }   /* MatGetDiagonalBlock */
  else bj->block = pc->pmat;
  if (!bj->ksp) {
    PC subpc;
    ierr = KSPCreate(PETSC_COMM_SELF, &bj->ksp);CHKERRQ(ierr);
    ierr = KSPSetType(bj->ksp, KSPPREONLY);CHKERRQ(ierr);
    snprintf(prefix, sizeof(prefix), "%ssub_", pc->prefix);
    ierr = KSPSetOptionsPrefix(bj->ksp, prefix);CHKERRQ(ierr);
    ierr = KSPGetPC(bj->ksp, &subpc);CHKERRQ(ierr);
    /* the sub-PC defaults to ILU(0) like the reference's (precon.c:14-53, PCSetUp of the sub-KSP) */
    ierr = PCSetType(subpc, PCILU);CHKERRQ(ierr);
    ierr = VecCreateSeqHIPMI355X(PETSC_COMM_SELF, bj->block->rmap->n, &bj->x);CHKERRQ(ierr);
    ierr = VecCreateSeqHIPMI355X(PETSC_COMM_SELF, bj->block->rmap->n, &bj->y);CHKERRQ(ierr);
  }
  ierr = KSPSetOperators(bj->ksp, bj->block, bj->block, SAME_NONZERO_PATTERN);CHKERRQ(ierr);
  ierr = KSPSetFromOptions(bj->ksp);CHKERRQ(ierr);
  ierr = KSPSetUp(bj->ksp);CHKERRQ(ierr);
  return 0;
}

/* PCApply_BJacobi_Singleblock, bjacobi.c:738-761.  The reference aliases the parallel vectors' HOST
 * arrays into sequential work vectors (VecGetArray + VecPlaceArray), which drags a GPU vector through
 * the host on every application.  Here the aliasing is done on the DEVICE pointers. */
static PetscErrorCode PCApply_BJacobi(PC pc, Vec x, Vec y) {
  PetscErrorCode ierr;
  PC_BJacobi *bj = (PC_BJacobi *)pc->data;
  Vec_HIPMI355X *bx = (Vec_HIPMI355X *)bj->x->data, *by = (Vec_HIPMI355X *)bj->y->data;
  const PetscScalar *dx; PetscScalar *dy, *sx = bx->dev, *sy = by->dev;
  int vx = bx->valid, vy = by->valid;
  ierr = VecHIPGetRead(x, &dx);CHKERRQ(ierr);
  ierr = VecHIPGetWrite(y, &dy);CHKERRQ(ierr);
  bx->dev = (PetscScalar *)dx; bx->valid = VALID_DEVICE; PetscObjectStateIncrease(bj->x);
  by->dev = dy; by->valid = VALID_DEVICE; PetscObjectStateIncrease(bj->y);
  ierr = KSPSolve(bj->ksp, bj->x, bj->y);
  bx->dev = sx; bx->valid = vx; by->dev = sy; by->valid = vy;   /* the aliases come off first, also when the sub-solve failed: the work vectors own sx / sy */
  CHKERRQ(ierr);
  ierr = VecHIPRestoreWrite(y);CHKERRQ(ierr);
  PetscObjectStateIncrease(y);   /* y changed through the alias: cached norms are stale (VecRestoreArray does this in bjacobi.c:758) */
  return 0;
}
static PetscErrorCode PCDestroy_BJacobi(PC pc) {
  PetscErrorCode ierr;
  PC_BJacobi *bj = (PC_BJacobi *)pc->data;
  if (!bj) return 0;
  ierr = KSPDestroy(&bj->ksp);CHKERRQ(ierr);
  ierr = VecDestroy(&bj->x);CHKERRQ(ierr);
  ierr = VecDestroy(&bj->y);CHKERRQ(ierr);
  free(bj); pc->data = NULL;
  return 0;
}
PetscErrorCode PCBJacobiGetSubKSP(PC pc, PetscInt *n_local, PetscInt *first_local, KSP **ksp) {
  if (strcmp(pc->type_name, PCBJACOBI)) SETERRQ(pc->comm, PETSC_ERR_ARG_WRONG, "Cannot get subsolvers for this preconditioner");
  PC_BJacobi *bj = (PC_BJacobi *)pc->data;
  if (!bj->ksp) SETERRQ(pc->comm, PETSC_ERR_ARG_WRONGSTATE, "Must call KSPSetUp() or PCSetUp() first");
  if (n_local) *n_local = 1;
  if (first_local) *first_local = pc->comm->rank;
  if (ksp) *ksp = &bj->ksp;
  return 0;
}
PetscErrorCode PCCreate_BJacobi(PC pc) {
  PC_BJacobi *bj;
  PetscErrorCode ierr = PetscMalloc(sizeof(*bj), &bj);CHKERRQ(ierr);
  memset(bj, 0, sizeof(*bj));
  pc->data = bj;
  pc->ops->setup = PCSetUp_BJacobi; pc->ops->apply = PCApply_BJacobi; pc->ops->destroy = PCDestroy_BJacobi;
  return 0;
}
