/* PCPBJACOBI on the device for the BAIJ type: src/ksp/pc/impls/pbjacobi/pbjacobi.c.
 *   set-up : MatInvertBlockDiagonal_SeqBAIJ (src/mat/impls/baij/seq/baij.c:13-160) on the HOST copy of the matrix -- every
 *            diagonal block inverted in place by LINPACK dgefa + dgedi, the algorithm of PetscKernel_A_gets_inverse_A_N
 *            (dgefa.c, dgefa2.c .. dgefa7.c, dgedi.c) -- then one upload of the inverted blocks;
 *   apply  : PCApply_PBJacobi_N (pbjacobi.c:20-200), one kernel (mi355x_pbjacobi_apply), the reference's row sums. */
#include "hipmi355ximpl.h"

typedef struct {
  PetscInt mbs, bs;
  PetscScalar *d_idiag;     /* mbs * bs * bs, HBM */
  int setup_state;
} PC_PBJacobi_HIP;

/* in-place inverse of a column-major n x n block (n <= 16); returns the 1-based zero-pivot row or 0 */
static int block_inverse(int n, PetscScalar *a) {
  int ipvt[16]; PetscScalar work[16];
#define E(i, j) a[(i) + (j) * n]
  for (int k = 0; k + 1 < n; k++) {                       /* dgefa: elimination with partial pivoting, multipliers negated */
    int l = k; PetscReal big = PetscAbsScalar(E(k, k));
    for (int i = k + 1; i < n; i++) if (PetscAbsScalar(E(i, k)) > big) { big = PetscAbsScalar(E(i, k)); l = i; }
    ipvt[k] = l;
    if (E(l, k) == 0.0) return k + 1;
    if (l != k) { PetscScalar t = E(l, k); E(l, k) = E(k, k); E(k, k) = t; }
    const PetscScalar m = -1. / E(k, k);
    for (int i = k + 1; i < n; i++) E(i, k) *= m;
    for (int j = k + 1; j < n; j++) {
      const PetscScalar t = E(l, j);
      if (l != k) { E(l, j) = E(k, j); E(k, j) = t; }
      for (int i = k + 1; i < n; i++) E(i, j) += t * E(i, k);
    }
  }
  ipvt[n - 1] = n - 1;
  if (E(n - 1, n - 1) == 0.0) return n;
  for (int k = 0; k < n; k++) {                           /* dgedi: inverse(U) ... */
    E(k, k) = 1.0 / E(k, k);
    const PetscScalar t0 = -E(k, k);
    for (int i = 0; i < k; i++) E(i, k) *= t0;
    for (int j = k + 1; j < n; j++) {
      const PetscScalar t = E(k, j);
      E(k, j) = 0.0;
      for (int i = 0; i <= k; i++) E(i, j) += t * E(i, k);
    }
  }
  for (int k = n - 2; k >= 0; k--) {                      /* ... times inverse(L), interchanges undone */
    for (int i = k + 1; i < n; i++) { work[i] = E(i, k); E(i, k) = 0.0; }
    for (int j = k + 1; j < n; j++) { const PetscScalar t = work[j]; for (int i = 0; i < n; i++) E(i, k) += t * E(i, j); }
    const int l = ipvt[k];
    if (l != k) for (int i = 0; i < n; i++) { const PetscScalar t = E(i, k); E(i, k) = E(i, l); E(i, l) = t; }
  }
#undef E
  return 0;
}

static PetscErrorCode PCSetUp_PBJacobi_HIP(PC pc) {
  PetscErrorCode ierr;
  PC_PBJacobi_HIP *jac = (PC_PBJacobi_HIP *)pc->data;
  Mat A = pc->pmat;
  PetscInt mbs; const PetscInt *bi, *bj; const PetscScalar *ba;
  PetscDeviceCtx *dc;
  if (A->rmap->n != A->cmap->n) SETERRQ(HipObjComm(pc), PETSC_ERR_SUP, "Supported only for square matrices and square storage");
  if (strcmp(HipObjTypeName(A), MATSEQBAIJHIPMI355X) && strcmp(HipObjTypeName(A), MATSEQAIJHIPMI355X)) SETERRQ(HipObjComm(pc), PETSC_ERR_SUP, "PCPBJACOBI on the device needs a sequential (B)AIJHIPMI355X matrix, got %s", HipObjTypeName(A));
  if (jac->d_idiag && jac->setup_state == HipObjState(A)) return 0;
  ierr = MatSeqAIJGetArrays(A, &mbs, &bi, &bj, &ba);CHKERRQ(ierr);
  const PetscInt bs = mbs ? A->rmap->n / mbs : 1, bs2 = bs * bs;
  if (bs > 16) SETERRQ(HipObjComm(pc), PETSC_ERR_SUP, "not supported for block size %d", bs);
  PetscScalar *idiag;
  ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)PetscMax(mbs * bs2, 1), &idiag);CHKERRQ(ierr);
  for (PetscInt i = 0; i < mbs; i++) {
    PetscInt k = bi[i];
    while (k < bi[i + 1] && bj[k] != i) k++;
    if (k == bi[i + 1]) { HipFree(idiag); SETERRQ(HipObjComm(pc), PETSC_ERR_ARG_WRONGSTATE, "Matrix is missing diagonal block %d", i); }
    memcpy(idiag + (size_t)i * bs2, ba + (size_t)k * bs2, sizeof(PetscScalar) * (size_t)bs2);
    const int z = block_inverse((int)bs, idiag + (size_t)i * bs2);
    if (z) { HipFree(idiag); SETERRQ(HipObjComm(pc), 71 /* PETSC_ERR_MAT_LU_ZRPVT */, "Zero pivot, row %d", i * bs + z - 1); }
  }
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  if (jac->d_idiag && (jac->mbs != mbs || jac->bs != bs)) { mi355x_free(jac->d_idiag); jac->d_idiag = NULL; }
  if (!jac->d_idiag) CHKHIP(mi355x_malloc((void **)&jac->d_idiag, sizeof(PetscScalar) * (size_t)PetscMax(mbs * bs2, 1)));
  CHKHIP(mi355x_memcpy_h2d(dc->h, jac->d_idiag, idiag, sizeof(PetscScalar) * (size_t)(mbs * bs2)));
  CHKHIP(mi355x_handle_synchronize(dc->h));
  HipFree(idiag);
  jac->mbs = mbs; jac->bs = bs; jac->setup_state = HipObjState(A);
  return 0;
}

static PetscErrorCode PCApply_PBJacobi_HIP(PC pc, Vec x, Vec y) {
  PetscErrorCode ierr;
  PC_PBJacobi_HIP *jac = (PC_PBJacobi_HIP *)pc->data;
  const PetscScalar *dx; PetscScalar *dy; PetscDeviceCtx *dc;
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  ierr = VecHIPGetRead(x, &dx);CHKERRQ(ierr);
  ierr = VecHIPGetWrite(y, &dy);CHKERRQ(ierr);
  CHKHIP(mi355x_pbjacobi_apply(dc->h, jac->mbs, jac->bs, jac->d_idiag, dx, dy));
  ierr = VecHIPRestoreWrite(y);CHKERRQ(ierr);
  HipStateIncrease(y);
  ierr = PetscLogFlops((2.0 * jac->bs * jac->bs - jac->bs) * jac->mbs);CHKERRQ(ierr);   /* pbjacobi.c:88 (15 m for bs = 3) */
  return 0;
}

static PetscErrorCode PCDestroy_PBJacobi_HIP(PC pc) {
  PC_PBJacobi_HIP *jac = (PC_PBJacobi_HIP *)pc->data;
  if (jac) { if (jac->d_idiag) mi355x_free(jac->d_idiag); HipFree(jac); pc->data = NULL; }
  return 0;
}

PetscErrorCode PCCreate_PBJacobi_HIPMI355X(PC pc) {
  PC_PBJacobi_HIP *jac;
  PetscErrorCode ierr = PetscMalloc(sizeof(*jac), &jac);CHKERRQ(ierr);
  memset(jac, 0, sizeof(*jac));
  jac->setup_state = -1;
  pc->data = jac;
  pc->ops->setup = PCSetUp_PBJacobi_HIP;
  pc->ops->apply = PCApply_PBJacobi_HIP;
  pc->ops->destroy = PCDestroy_PBJacobi_HIP;
  return 0;
}
