/* The plug-in's own Krylov solvers, shipped the way the reference ships a solver: as KSP types registered with KSPRegister
 * (include/petscksp.h:83-123) and selected with -ksp_type:
 *
 *     KSPCGHIPMI355X    "cghipmi355x"      CG        (the recurrence of KSPCG,    src/ksp/ksp/impls/cg/cg.c:92-286)
 *     KSPGMRESHIPMI355X "gmreshipmi355x"   GMRES(m)  (the cycle of KSPGMRES,      src/ksp/ksp/impls/gmres/gmres.c:118-409, borthog2.c:35-119)
 *     KSPBCGSHIPMI355X  "bcgshipmi355x"    BiCGStab  (the recurrence of KSPBCGS,  src/ksp/ksp/impls/bcgs/bcgs.c:43-160)
 *
 * They compute what the reference's solvers compute -- same iterates, same residual history, same exits, bit for bit against
 * the op-by-op sequences (tests/test_host_gpu.py) -- but group the vector work between two reductions into the fused sweeps a
 * Vec / Mat type may offer by name ("VecKrylovFusedOps_C", "MatMultTDotBegin_C", "MatMultDiagonalScale_C";
 * include/petsckrylovfused.h), keep scalars that only the device needs on the device, and queue the front half of the next
 * iteration before the host has looked at the residual norm.  On vectors of a type without those methods every step falls
 * back to the public Vec / Mat / PC calls, so the types work on any PETSc objects.
 *
 * Written against the public API and the KSP implementation header only (petsc-private/kspimpl.h inside a PETSc tree, the
 * harness's petscimpl.h on a box without PETSc): nothing here reaches into a Vec, a Mat or a PC.  An unchanged PETSc program
 * gets these solvers with -ksp_type cghipmi355x etc.; with its own -ksp_type cg it gets PETSc's KSPSolve_CG over the same
 * Vec/Mat ops, one kernel per call. */
#include "hipmi355ximpl.h"

/* ------------------------------------------------------------------------------------------------ what the types offer */
static const VecKrylovFusedOps *vec_fused_ops(Vec x) {
  PetscVoidFunction f = NULL;
  if (PetscObjectQueryFunction((PetscObject)x, "VecKrylovFusedOps_C", &f) || !f) return NULL;
  return ((VecKrylovFusedOpsGetFn)f)();
}
static PetscErrorCode mat_mult_scaled(Mat A, Vec d, Vec x, Vec y, PetscBool *ok) {        /* y = d .* (A x) in one kernel */
  PetscVoidFunction f = NULL;
  PetscErrorCode ierr;
  *ok = PETSC_FALSE;
  ierr = PetscObjectQueryFunction((PetscObject)A, "MatMultDiagonalScale_C", &f);CHKERRQ(ierr);
  if (f) { ierr = ((MatMultDiagonalScaleFn)f)(A, d, x, y, ok);CHKERRQ(ierr); }
  return 0;
}
static PetscErrorCode mat_mult_with_dot(Mat A, Vec x, Vec y, PetscBool *ok) {             /* y = A x, x'y left on the device */
  PetscVoidFunction f = NULL;
  PetscErrorCode ierr;
  *ok = PETSC_FALSE;
  ierr = PetscObjectQueryFunction((PetscObject)A, "MatMultTDotBegin_C", &f);CHKERRQ(ierr);
  if (f) { ierr = ((MatMultTDotBeginFn)f)(A, x, y, ok);CHKERRQ(ierr); }
  return 0;
}
static PetscErrorCode option_level(KSP ksp, const char *name, PetscInt lo, PetscInt hi, PetscInt *level) {   /* "-name <int|true|false>" */
  char t[16]; PetscBool set;
  PetscErrorCode ierr = PetscOptionsGetString(HipObjPrefix(ksp), name, t, sizeof(t), &set);CHKERRQ(ierr);
  if (!set) return 0;
  if (!strcmp(t, "false")) *level = lo;
  else if (!strcmp(t, "true") || !t[0]) *level = hi;
  else { PetscInt v = (PetscInt)atoi(t); *level = v < lo ? lo : (v > hi ? hi : v); }
  return 0;
}

/* A preconditioner that is a diagonal scaling can ride along in a fused sweep.  Which diagonal: asked of the PC itself through
 * its public face -- PCApply to a vector of ones returns exactly the numbers PCApply_Jacobi multiplies by (1.0 * d is d), for
 * every variant of PCJACOBI (-pc_jacobi_rowmax, ...) and without knowing the PC's data structure. */
typedef enum { PCKIND_GENERAL = 0, PCKIND_IDENTITY, PCKIND_DIAGONAL } HipPCKind;
static PetscErrorCode pc_diagonal_form(KSP ksp, Vec like, Vec *dinv, HipPCKind *kind) {
  PetscErrorCode ierr;
  PCType t = NULL;
  *kind = PCKIND_GENERAL;
  ierr = PCGetType(ksp->pc, &t);CHKERRQ(ierr);
  if (t && !strcmp(t, PCNONE)) *kind = PCKIND_IDENTITY;
  else if (t && !strcmp(t, PCJACOBI)) {
    Vec ones;
    if (!*dinv) { ierr = VecDuplicate(like, dinv);CHKERRQ(ierr); }
    ierr = VecDuplicate(like, &ones);CHKERRQ(ierr);
    ierr = VecSet(ones, 1.0);CHKERRQ(ierr);
    ierr = PCApply(ksp->pc, ones, *dinv);CHKERRQ(ierr);
    ierr = VecDestroy(&ones);CHKERRQ(ierr);
    *kind = PCKIND_DIAGONAL;
  }
  return 0;
}
#define KSPCheckDot(ksp, v) do { if (PetscIsInfOrNanScalar(v)) SETERRQ(HipObjComm(ksp), PETSC_ERR_FP, "Infinite or not-a-number generated in dot product"); } while (0)
#define KSPTestConvergence(ksp, it, rn) (*(ksp)->converged)((ksp), (it), (rn), &(ksp)->reason, (ksp)->cnvP)
static PetscErrorCode report(KSP ksp, PetscInt it, PetscReal rn) {   /* what every solver does with a new residual norm */
  PetscErrorCode ierr;
  ksp->rnorm = rn;
  KSPLogResidualHistory(ksp, rn);
  ierr = KSPMonitor(ksp, it, rn);CHKERRQ(ierr);
  return 0;
}

/* ================================================================================================ CG
 * -ksp_cg_fused <0..4> (default 4); iterates and history carry the same bits at levels 0..3:
 *  0  every step through the public Vec / Mat / PC calls (what KSPCG does);
 *  1  with PCJACOBI or PCNONE the five calls between the two reductions of an iteration -- x += a p, r -= a w, z = B r, |z|,
 *     z'r (cg.c:206-232) -- are ONE sweep returning z'z, z'r and r'r; any other PC: norm and dot share one VecDotNorm2;
 *  2  as 1, and p'w never leaves the device: the sweep forms a = beta / (p'w) itself and applies the reference's break-down
 *     tests before touching anything; p'w comes back with the sums, so the host takes the same exits one kernel later;
 *  3  as 2, and while the host waits for the sums of iteration i the device already runs p = z + b p (b formed on the device),
 *     w = A p and p'w of iteration i+1.  If the convergence test ends the solve only work vectors have been touched.  Not done
 *     for the last permitted iteration nor within 10x of the target;
 *  4  as 3 with p'w produced by the SpMV pass itself (another summation tree: agrees with the other levels to rounding, not bit for
 *     bit; deterministic).  Sequential AIJ matrices whose product kernel leaves per-block sums ("MatMultTDotBegin_C"); level 3 otherwise. */
typedef struct { PetscInt level; Vec dinv; } KSP_CGHIP;

static PetscErrorCode KSPSetUp_CGHIP(KSP ksp) { return KSPDefaultGetWork(ksp, 3); }
static PetscErrorCode KSPSetFromOptions_CGHIP(KSP ksp) { return option_level(ksp, "-ksp_cg_fused", 0, 4, &((KSP_CGHIP *)ksp->data)->level); }
static PetscErrorCode KSPDestroy_CGHIP(KSP ksp) {
  KSP_CGHIP *cg = (KSP_CGHIP *)ksp->data;
  if (cg) { PetscErrorCode ierr = VecDestroy(&cg->dinv);CHKERRQ(ierr); HipFree(cg); ksp->data = NULL; }
  return 0;
}

static PetscReal cg_norm_from_sums(KSPNormType nt, PetscScalar zz, PetscScalar zr, PetscScalar rr) {
  switch (nt) {                                          /* the square is reduced, then rooted (pvec2.c:62-64) */
  case KSP_NORM_PRECONDITIONED:   return PetscSqrtReal(zz);
  case KSP_NORM_UNPRECONDITIONED: return PetscSqrtReal(rr);
  case KSP_NORM_NATURAL:          return PetscSqrtReal(PetscAbsScalar(zr));
  default:                        return 0.0;
  }
}

static PetscErrorCode KSPSolve_CGHIP(KSP ksp) {
  PetscErrorCode ierr;
  KSP_CGHIP *cg = (KSP_CGHIP *)ksp->data;
  Vec x = ksp->vec_sol, rhs = ksp->vec_rhs, r = ksp->work[0], z = ksp->work[1], p = ksp->work[2], w = z;   /* w = A p shares z's storage, as cg.c:122 */
  Mat A;
  const KSPNormType nt = ksp->normtype;
  const VecKrylovFusedOps *F = vec_fused_ops(x);
  const PetscInt level = F ? cg->level : 0;
  HipPCKind pck = PCKIND_GENERAL;
  Vec d = NULL;
  PetscBool sweep = PETSC_FALSE;          /* the one-sweep update is available for these vectors and this PC */
  PetscBool ahead = PETSC_FALSE;          /* p, w = A p and p'w of the coming iteration are already queued */
  PetscScalar pw = 0.0, pw_prev, rz, rz_prev = 1.0, step;
  PetscReal rn = 0.0;
  PetscInt it;

  ierr = PCGetOperators(ksp->pc, &A, NULL, NULL);CHKERRQ(ierr);
  if (level > 0) {
    ierr = pc_diagonal_form(ksp, x, &cg->dinv, &pck);CHKERRQ(ierr);
    if (pck != PCKIND_GENERAL) { d = pck == PCKIND_DIAGONAL ? cg->dinv : NULL; ierr = F->cg_update_check(x, r, z, p, w, d, &sweep);CHKERRQ(ierr); }
  }
  const PetscBool on_device = (PetscBool)(sweep && level > 1);

  /* ---- residual of the initial guess, its norm in the flavour the test wants, the first z'r ---- */
  ksp->its = 0;
  if (ksp->guess_zero) { ierr = VecCopy(rhs, r);CHKERRQ(ierr); }
  else { ierr = KSP_MatMult(ksp, A, x, r);CHKERRQ(ierr); ierr = VecAYPX(r, -1.0, rhs);CHKERRQ(ierr); }
  if (nt == KSP_NORM_PRECONDITIONED) { ierr = KSP_PCApply(ksp, r, z);CHKERRQ(ierr); ierr = VecNorm(z, NORM_2, &rn);CHKERRQ(ierr); }
  else if (nt == KSP_NORM_UNPRECONDITIONED) { ierr = VecNorm(r, NORM_2, &rn);CHKERRQ(ierr); }
  else if (nt == KSP_NORM_NATURAL) {
    ierr = KSP_PCApply(ksp, r, z);CHKERRQ(ierr);
    ierr = VecTDot(z, r, &rz);CHKERRQ(ierr);
    KSPCheckDot(ksp, rz);
    rn = PetscSqrtReal(PetscAbsScalar(rz));
  } else if (nt != KSP_NORM_NONE) SETERRQ(HipObjComm(ksp), PETSC_ERR_SUP, "norm type %d", (int)nt);
  ierr = report(ksp, 0, rn);CHKERRQ(ierr);
  ierr = KSPTestConvergence(ksp, 0, rn);CHKERRQ(ierr);
  if (ksp->reason) return 0;
  if (nt == KSP_NORM_UNPRECONDITIONED || nt == KSP_NORM_NONE) { ierr = KSP_PCApply(ksp, r, z);CHKERRQ(ierr); }
  if (nt != KSP_NORM_NATURAL) { ierr = VecTDot(z, r, &rz);CHKERRQ(ierr); KSPCheckDot(ksp, rz); }

  it = 0;
  do {
    PetscBool pw_on_device = PETSC_FALSE, have_sums = PETSC_FALSE;
    PetscScalar zz = 0.0, zr = 0.0, rr = 0.0;
    ksp->its = it + 1;
    if (rz == 0.0) { ksp->reason = KSP_CONVERGED_ATOL; break; }
    if (it > 0 && rz * rz_prev < 0.0) { ksp->reason = KSP_DIVERGED_INDEFINITE_PC; break; }
    pw_prev = pw;
    if (ahead) { pw_on_device = PETSC_TRUE; ahead = PETSC_FALSE; }       /* queued behind the previous sweep */
    else {
      if (it == 0) { ierr = VecCopy(z, p);CHKERRQ(ierr); }                /* p <- z */
      else { ierr = VecAYPX(p, rz / rz_prev, z);CHKERRQ(ierr); }          /* p <- z + b p */
      if (on_device && level > 3) { ierr = mat_mult_with_dot(A, p, w, &pw_on_device);CHKERRQ(ierr); }
      if (!pw_on_device) {
        ierr = KSP_MatMult(ksp, A, p, w);CHKERRQ(ierr);                   /* w <- A p */
        if (on_device) { ierr = F->tdot_begin(p, w, &pw_on_device);CHKERRQ(ierr); }
        if (!pw_on_device) { ierr = VecTDot(p, w, &pw);CHKERRQ(ierr); }
      }
    }
    rz_prev = rz;
    if (pw_on_device) {
      ierr = F->cg_update_dev_begin(x, r, z, p, w, d, rz, pw_prev, (PetscBool)(it > 0));CHKERRQ(ierr);
      if (level > 2 && it + 1 < ksp->max_it && (nt == KSP_NORM_NONE || rn > 10.0 * ksp->ttol)) {
        PetscBool ok = PETSC_FALSE;                                        /* front half of iteration it+1 */
        ierr = F->aypx_dev(p, rz, z);CHKERRQ(ierr);
        if (level > 3) { ierr = mat_mult_with_dot(A, p, w, &ok);CHKERRQ(ierr); }
        if (!ok) {
          ierr = KSP_MatMult(ksp, A, p, w);CHKERRQ(ierr);
          ierr = F->tdot_begin(p, w, &ok);CHKERRQ(ierr);
          if (!ok) SETERRQ(HipObjComm(ksp), PETSC_ERR_PLIB, "split dot refused after it had been accepted");
        }
        ahead = PETSC_TRUE;
      }
      ierr = F->cg_update_dev_end(x, &zz, &zr, &rr, &pw);CHKERRQ(ierr);
      have_sums = PETSC_TRUE;
    }
    KSPCheckDot(ksp, pw);
    if (pw == 0.0 || (it > 0 && pw * pw_prev <= 0.0)) { ksp->reason = KSP_DIVERGED_INDEFINITE_MAT; break; }
    step = rz / pw;
    if (!have_sums && sweep) { ierr = F->cg_update(x, r, z, p, w, d, step, &zz, &zr, &rr, &have_sums);CHKERRQ(ierr); }
    if (have_sums) {
      if (nt == KSP_NORM_NATURAL) KSPCheckDot(ksp, zr);
      rn = cg_norm_from_sums(nt, zz, zr, rr);
    } else {
      ierr = VecAXPY(x, step, p);CHKERRQ(ierr);
      ierr = VecAXPY(r, -step, w);CHKERRQ(ierr);
      if (nt == KSP_NORM_PRECONDITIONED) {
        ierr = KSP_PCApply(ksp, r, z);CHKERRQ(ierr);
        if (level > 0) { PetscReal n2; ierr = VecDotNorm2(r, z, &zr, &n2);CHKERRQ(ierr); rn = PetscSqrtReal(n2); have_sums = PETSC_TRUE; }
        else { ierr = VecNorm(z, NORM_2, &rn);CHKERRQ(ierr); }
      } else if (nt == KSP_NORM_UNPRECONDITIONED) { ierr = VecNorm(r, NORM_2, &rn);CHKERRQ(ierr); }
      else if (nt == KSP_NORM_NATURAL) {
        ierr = KSP_PCApply(ksp, r, z);CHKERRQ(ierr);
        ierr = VecTDot(z, r, &rz);CHKERRQ(ierr);
        KSPCheckDot(ksp, rz);
        rn = PetscSqrtReal(PetscAbsScalar(rz));
      } else rn = 0.0;
    }
    ierr = report(ksp, it + 1, rn);CHKERRQ(ierr);
    ierr = KSPTestConvergence(ksp, it + 1, rn);CHKERRQ(ierr);
    if (ksp->reason) break;
    if (have_sums) rz = zr;
    else {
      if (nt == KSP_NORM_UNPRECONDITIONED || nt == KSP_NORM_NONE) { ierr = KSP_PCApply(ksp, r, z);CHKERRQ(ierr); }
      if (nt != KSP_NORM_NATURAL) { ierr = VecTDot(z, r, &rz);CHKERRQ(ierr); }
    }
    KSPCheckDot(ksp, rz);
    it++;
  } while (it < ksp->max_it);
  if (it >= ksp->max_it) ksp->reason = KSP_DIVERGED_ITS;
  return 0;
}

PetscErrorCode KSPCreate_CGHIPMI355X(KSP ksp) {
  KSP_CGHIP *cg;
  PetscErrorCode ierr = PetscMalloc(sizeof(*cg), &cg);CHKERRQ(ierr);
  cg->level = 4; cg->dinv = NULL;
  ksp->data = cg;
  ierr = KSPSetSupportedNorm(ksp, KSP_NORM_PRECONDITIONED, PC_LEFT, 2);CHKERRQ(ierr);    /* the four of KSPCG, cg.c:439-442 */
  ierr = KSPSetSupportedNorm(ksp, KSP_NORM_UNPRECONDITIONED, PC_LEFT, 1);CHKERRQ(ierr);
  ierr = KSPSetSupportedNorm(ksp, KSP_NORM_NATURAL, PC_LEFT, 1);CHKERRQ(ierr);
  ierr = KSPSetSupportedNorm(ksp, KSP_NORM_NONE, PC_LEFT, 1);CHKERRQ(ierr);
  ksp->ops->setup = KSPSetUp_CGHIP; ksp->ops->solve = KSPSolve_CGHIP; ksp->ops->setfromoptions = KSPSetFromOptions_CGHIP; ksp->ops->destroy = KSPDestroy_CGHIP;
  return 0;
}

/* ================================================================================================ GMRES(m)
 * -ksp_gmres_fused <bool> (default true).  Left PCJACOBI: the product writes d .* (A v) itself.  Without refinement the
 * Gram-Schmidt step is MDot (coefficients stay on the device) -> ONE sweep w -= V h with |w|^2 -> w *= 1/|w| off the
 * device-resident norm, one host wait instead of three ("gmres_orthog_normalize").  Same bits as the separate calls.
 * Options of KSPGMRES kept: -ksp_gmres_restart, -ksp_gmres_cgs_refinement_type, KSPGMRESSetRestart / SetCGSRefinementType. */
typedef struct {
  PetscInt m;                       /* restart length */
  PetscReal haptol;
  KSPGMRESCGSRefinementType refine;
  PetscBool fused;
  PetscScalar *H;                   /* Hessenberg matrix after the rotations, column j at H + j (m + 2) */
  PetscScalar *g, *cs, *sn, *coef, *y;   /* rotated right-hand side, rotations, Gram-Schmidt coefficients, triangular solve */
  Vec *V; PetscInt nV;              /* V[0], V[1]: scratch; V[2 + k]: k-th basis vector */
  Vec dinv;
} KSP_GMRESHIP;
#define GH(ksp) ((KSP_GMRESHIP *)(ksp)->data)
#define Hcol(gm, j) ((gm)->H + (size_t)(j) * (size_t)((gm)->m + 2))
#define BASIS(gm, k) ((gm)->V[2 + (k)])

static PetscErrorCode KSPGMRESSetRestart_GMRESHIP(KSP ksp, PetscInt restart) {
  if (restart < 1) SETERRQ(HipObjComm(ksp), PETSC_ERR_ARG_OUTOFRANGE, "Restart must be positive");
  if (GH(ksp)->V) SETERRQ(HipObjComm(ksp), PETSC_ERR_ORDER, "Must call KSPGMRESSetRestart() before KSPSetUp()");
  GH(ksp)->m = restart;
  return 0;
}
static PetscErrorCode KSPGMRESSetCGSRefinementType_GMRESHIP(KSP ksp, KSPGMRESCGSRefinementType type) { GH(ksp)->refine = type; return 0; }

static PetscErrorCode KSPSetFromOptions_GMRESHIP(KSP ksp) {
  PetscErrorCode ierr; PetscInt iv; PetscBool set; char t[64];
  ierr = PetscOptionsGetInt(HipObjPrefix(ksp), "-ksp_gmres_restart", &iv, &set);CHKERRQ(ierr);
  if (set) { ierr = KSPGMRESSetRestart_GMRESHIP(ksp, iv);CHKERRQ(ierr); }
  ierr = PetscOptionsGetString(HipObjPrefix(ksp), "-ksp_gmres_cgs_refinement_type", t, sizeof(t), &set);CHKERRQ(ierr);
  if (set) {
    if (!strcmp(t, "refine_never")) GH(ksp)->refine = KSP_GMRES_CGS_REFINE_NEVER;
    else if (!strcmp(t, "refine_ifneeded")) GH(ksp)->refine = KSP_GMRES_CGS_REFINE_IFNEEDED;
    else if (!strcmp(t, "refine_always")) GH(ksp)->refine = KSP_GMRES_CGS_REFINE_ALWAYS;
    else SETERRQ(HipObjComm(ksp), PETSC_ERR_ARG_UNKNOWN_TYPE, "Unknown refinement type %s", t);
  }
  iv = GH(ksp)->fused;
  ierr = option_level(ksp, "-ksp_gmres_fused", 0, 1, &iv);CHKERRQ(ierr);
  GH(ksp)->fused = (PetscBool)iv;
  return 0;
}

static PetscErrorCode KSPSetUp_GMRESHIP(KSP ksp) {
  PetscErrorCode ierr;
  KSP_GMRESHIP *gm = GH(ksp);
  const size_t m = (size_t)gm->m;
  Vec like = ksp->vec_sol, made = NULL;
  ierr = PetscMalloc(sizeof(PetscScalar) * (m + 2) * (m + 1), &gm->H);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscScalar) * (m + 2), &gm->g);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscScalar) * (m + 2), &gm->cs);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscScalar) * (m + 2), &gm->sn);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscScalar) * (m + 2), &gm->coef);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscScalar) * (m + 2), &gm->y);CHKERRQ(ierr);
  memset(gm->H, 0, sizeof(PetscScalar) * (m + 2) * (m + 1));
  memset(gm->g, 0, sizeof(PetscScalar) * (m + 2)); memset(gm->cs, 0, sizeof(PetscScalar) * (m + 2)); memset(gm->sn, 0, sizeof(PetscScalar) * (m + 2));
  if (!like) { Mat A; ierr = PCGetOperators(ksp->pc, &A, NULL, NULL);CHKERRQ(ierr); ierr = MatGetVecs(A, &made, NULL);CHKERRQ(ierr); like = made; }
  gm->nV = gm->m + 4;                                   /* all basis vectors up front (gmres.c:36-96 allocates them in chunks) */
  ierr = VecDuplicateVecs(like, gm->nV, &gm->V);CHKERRQ(ierr);
  if (made) { ierr = VecDestroy(&made);CHKERRQ(ierr); }
  return 0;
}

/* classical Gram-Schmidt against BASIS(0..k) through the public calls (borthog2.c:35-119), optionally twice */
static PetscErrorCode gmres_orthogonalize(KSP ksp, PetscInt k, PetscScalar *hcol) {
  PetscErrorCode ierr;
  KSP_GMRESHIP *gm = GH(ksp);
  PetscScalar *c = gm->coef;
  Vec w = BASIS(gm, k + 1);
  PetscBool again = (PetscBool)(gm->refine == KSP_GMRES_CGS_REFINE_ALWAYS);
  for (PetscInt j = 0; j <= k; j++) hcol[j] = 0.0;
  for (int pass = 0; pass < 2; pass++) {
    ierr = VecMDot(w, k + 1, &BASIS(gm, 0), c);CHKERRQ(ierr);
    for (PetscInt j = 0; j <= k; j++) c[j] = -c[j];
    ierr = VecMAXPY(w, k + 1, c, &BASIS(gm, 0));CHKERRQ(ierr);
    for (PetscInt j = 0; j <= k; j++) hcol[j] -= c[j];
    if (pass == 0 && gm->refine == KSP_GMRES_CGS_REFINE_IFNEEDED) {          /* borthog2.c:77-98 */
      PetscReal hn = 0.0, wn;
      for (PetscInt j = 0; j <= k; j++) hn += c[j] * c[j];
      hn = PetscSqrtReal(hn);
      ierr = VecNorm(w, NORM_2, &wn);CHKERRQ(ierr);
      if (wn < 1.0286 * hn) again = PETSC_TRUE;
    }
    if (!again) break;
  }
  return 0;
}

/* apply the k earlier rotations to column k, then the new one that annihilates H(k+1,k) (gmres.c:360-409) */
static void gmres_rotate(KSP ksp, PetscInt k, PetscBool happy, PetscReal *res) {
  KSP_GMRESHIP *gm = GH(ksp);
  PetscScalar *h = Hcol(gm, k);
  for (PetscInt j = 0; j < k; j++) {
    const PetscScalar t = h[j];
    h[j] = gm->cs[j] * t + gm->sn[j] * h[j + 1];
    h[j + 1] = gm->cs[j] * h[j + 1] - gm->sn[j] * t;
  }
  if (happy) { *res = 0.0; return; }
  const PetscScalar t = PetscSqrtReal(h[k] * h[k] + h[k + 1] * h[k + 1]);
  if (t == 0.0) { ksp->reason = KSP_DIVERGED_NULL; return; }
  gm->cs[k] = h[k] / t;
  gm->sn[k] = h[k + 1] / t;
  gm->g[k + 1] = -(gm->sn[k] * gm->g[k]);
  gm->g[k] = gm->cs[k] * gm->g[k];
  h[k] = gm->cs[k] * h[k] + gm->sn[k] * h[k + 1];
  *res = PetscAbsScalar(gm->g[k + 1]);
}

/* x += [M^-1] V y with R y = g over the first n columns (gmres.c:309-354) */
static PetscErrorCode gmres_update_solution(KSP ksp, PetscInt n) {
  PetscErrorCode ierr;
  KSP_GMRESHIP *gm = GH(ksp);
  PetscScalar *y = gm->y;
  if (n < 1) return 0;
  for (PetscInt k = n - 1; k >= 0; k--) {
    PetscScalar t = gm->g[k];
    for (PetscInt j = k + 1; j < n; j++) t = t - Hcol(gm, j)[k] * y[j];
    if (Hcol(gm, k)[k] == 0.0) { ksp->reason = KSP_DIVERGED_BREAKDOWN; return 0; }
    y[k] = t / Hcol(gm, k)[k];
  }
  ierr = VecSet(gm->V[0], 0.0);CHKERRQ(ierr);
  ierr = VecMAXPY(gm->V[0], n, y, &BASIS(gm, 0));CHKERRQ(ierr);
  if (ksp->pc_side == PC_RIGHT) {                                   /* KSPUnwindPreconditioner */
    ierr = KSP_PCApply(ksp, gm->V[0], gm->V[1]);CHKERRQ(ierr);
    ierr = VecCopy(gm->V[1], gm->V[0]);CHKERRQ(ierr);
  }
  ierr = VecAXPY(ksp->vec_sol, 1.0, gm->V[0]);CHKERRQ(ierr);
  return 0;
}

/* one restart cycle starting from the residual in BASIS(0) */
static PetscErrorCode gmres_cycle(KSP ksp, PetscInt *steps) {
  PetscErrorCode ierr;
  KSP_GMRESHIP *gm = GH(ksp);
  const VecKrylovFusedOps *F = gm->fused ? vec_fused_ops(BASIS(gm, 0)) : NULL;
  Vec d = NULL;
  Mat A;
  PetscReal res, wnorm;
  PetscInt k = 0;
  PetscBool happy = PETSC_FALSE, done;

  if (F && !F->gmres_orthog_normalize) F = NULL;
  ierr = PCGetOperators(ksp->pc, &A, NULL, NULL);CHKERRQ(ierr);
  if (gm->fused && ksp->pc_side == PC_LEFT) {
    HipPCKind kind;
    ierr = pc_diagonal_form(ksp, BASIS(gm, 0), &gm->dinv, &kind);CHKERRQ(ierr);
    if (kind == PCKIND_DIAGONAL) d = gm->dinv;
  }
  ierr = VecNormalize(BASIS(gm, 0), &res);CHKERRQ(ierr);
  gm->g[0] = res;
  ierr = report(ksp, ksp->its, res);CHKERRQ(ierr);
  *steps = 0;
  if (!res) { ksp->reason = KSP_CONVERGED_ATOL; return 0; }
  ierr = KSPTestConvergence(ksp, ksp->its, res);CHKERRQ(ierr);
  while (!ksp->reason && k < gm->m && ksp->its < ksp->max_it) {
    PetscScalar *hcol = Hcol(gm, k);
    if (k) { ierr = report(ksp, ksp->its, res);CHKERRQ(ierr); }
    done = PETSC_FALSE;                                                        /* BASIS(k+1) = [M^-1] A [M^-1] BASIS(k) */
    if (d) { ierr = mat_mult_scaled(A, d, BASIS(gm, k), BASIS(gm, k + 1), &done);CHKERRQ(ierr); }
    if (!done) { ierr = KSP_PCApplyBAorAB(ksp, BASIS(gm, k), BASIS(gm, k + 1), gm->V[1]);CHKERRQ(ierr); }
    done = PETSC_FALSE;
    if (F && gm->refine == KSP_GMRES_CGS_REFINE_NEVER) {
      ierr = F->gmres_orthog_normalize(BASIS(gm, k + 1), k + 1, &BASIS(gm, 0), gm->coef, &wnorm, &done);CHKERRQ(ierr);
      if (done) for (PetscInt j = 0; j <= k; j++) { const PetscScalar c = -gm->coef[j]; hcol[j] = 0.0; hcol[j] -= c; }   /* borthog2.c:52-66 */
    }
    if (!done) {
      ierr = gmres_orthogonalize(ksp, k, hcol);CHKERRQ(ierr);
      ierr = VecNormalize(BASIS(gm, k + 1), &wnorm);CHKERRQ(ierr);
    }
    hcol[k + 1] = wnorm;
    {                                                                          /* happy breakdown test (gmres.c:171-178) */
      PetscReal bound = PetscAbsScalar(wnorm / gm->g[k]);
      if (bound > gm->haptol) bound = gm->haptol;
      if (wnorm < bound) happy = PETSC_TRUE;
    }
    gmres_rotate(ksp, k, happy, &res);
    k++;
    ksp->its++;
    ksp->rnorm = res;
    if (ksp->reason) break;
    ierr = KSPTestConvergence(ksp, ksp->its, res);CHKERRQ(ierr);
    if (happy) {
      if (!ksp->reason) SETERRQ(HipObjComm(ksp), PETSC_ERR_PLIB, "You reached the happy break down, but convergence was not indicated. Residual norm = %g", (double)res);
      break;
    }
  }
  if (k && (ksp->reason || ksp->its >= ksp->max_it)) { ierr = report(ksp, ksp->its, res);CHKERRQ(ierr); }
  *steps = k;
  ierr = gmres_update_solution(ksp, k);CHKERRQ(ierr);
  return 0;
}

static PetscErrorCode KSPSolve_GMRESHIP(KSP ksp) {
  PetscErrorCode ierr;
  KSP_GMRESHIP *gm = GH(ksp);
  const PetscBool guess_zero = ksp->guess_zero;
  PetscInt total = 0, steps = 0;
  ksp->its = 0;
  ksp->reason = KSP_CONVERGED_ITERATING;
  while (!ksp->reason) {
    ierr = KSPInitialResidual(ksp, ksp->vec_sol, gm->V[0], gm->V[1], BASIS(gm, 0), ksp->vec_rhs);CHKERRQ(ierr);
    ierr = gmres_cycle(ksp, &steps);CHKERRQ(ierr);
    total += steps;
    if (total >= ksp->max_it) { if (!ksp->reason) ksp->reason = KSP_DIVERGED_ITS; break; }
    ksp->guess_zero = PETSC_FALSE;                    /* the next cycle starts from the iterate (gmres.c:236) */
  }
  ksp->guess_zero = guess_zero;
  return 0;
}

static PetscErrorCode KSPDestroy_GMRESHIP(KSP ksp) {
  PetscErrorCode ierr;
  KSP_GMRESHIP *gm = GH(ksp);
  if (!gm) return 0;
  HipFree(gm->H); HipFree(gm->g); HipFree(gm->cs); HipFree(gm->sn); HipFree(gm->coef); HipFree(gm->y);
  if (gm->V) { ierr = VecDestroyVecs(gm->nV, &gm->V);CHKERRQ(ierr); }
  ierr = VecDestroy(&gm->dinv);CHKERRQ(ierr);
  HipFree(gm); ksp->data = NULL;
  ierr = PetscObjectComposeFunction((PetscObject)ksp, "KSPGMRESSetRestart_C", "", (PetscVoidFunction)NULL);CHKERRQ(ierr);
  ierr = PetscObjectComposeFunction((PetscObject)ksp, "KSPGMRESSetCGSRefinementType_C", "", (PetscVoidFunction)NULL);CHKERRQ(ierr);
  return 0;
}

PetscErrorCode KSPCreate_GMRESHIPMI355X(KSP ksp) {
  KSP_GMRESHIP *gm;
  PetscErrorCode ierr = PetscMalloc(sizeof(*gm), &gm);CHKERRQ(ierr);
  memset(gm, 0, sizeof(*gm));
  gm->m = 30; gm->haptol = 1.0e-30; gm->refine = KSP_GMRES_CGS_REFINE_NEVER; gm->fused = PETSC_TRUE;   /* KSPGMRES's defaults, gmres.c:944-958 */
  ksp->data = gm;
  ierr = KSPSetSupportedNorm(ksp, KSP_NORM_PRECONDITIONED, PC_LEFT, 2);CHKERRQ(ierr);       /* gmres.c:909-910 */
  ierr = KSPSetSupportedNorm(ksp, KSP_NORM_UNPRECONDITIONED, PC_RIGHT, 1);CHKERRQ(ierr);
  ksp->ops->setup = KSPSetUp_GMRESHIP; ksp->ops->solve = KSPSolve_GMRESHIP; ksp->ops->setfromoptions = KSPSetFromOptions_GMRESHIP; ksp->ops->destroy = KSPDestroy_GMRESHIP;
  ierr = PetscObjectComposeFunction((PetscObject)ksp, "KSPGMRESSetRestart_C", "KSPGMRESSetRestart_GMRESHIP", (PetscVoidFunction)KSPGMRESSetRestart_GMRESHIP);CHKERRQ(ierr);
  ierr = PetscObjectComposeFunction((PetscObject)ksp, "KSPGMRESSetCGSRefinementType_C", "KSPGMRESSetCGSRefinementType_GMRESHIP", (PetscVoidFunction)KSPGMRESSetCGSRefinementType_GMRESHIP);CHKERRQ(ierr);
  return 0;
}

/* ================================================================================================ BiCGStab
 * -ksp_bcgs_fused <bool> (default true), left preconditioning.  PCJACOBI: both products write d .* (A x) themselves; PCJACOBI /
 * PCNONE on types without that method: PCApply fused with the dot(s) that follow it; the x / r update, |r| and the NEXT
 * iteration's rho are one sweep: 22 vector passes and 3 reductions per iteration instead of 27 and 4, identical bits. */
typedef struct { PetscBool fused; Vec dinv; } KSP_BCGSHIP;

static PetscErrorCode KSPSetUp_BCGSHIP(KSP ksp) { return KSPDefaultGetWork(ksp, 6); }
static PetscErrorCode KSPSetFromOptions_BCGSHIP(KSP ksp) {
  PetscInt iv = ((KSP_BCGSHIP *)ksp->data)->fused;
  PetscErrorCode ierr = option_level(ksp, "-ksp_bcgs_fused", 0, 1, &iv);CHKERRQ(ierr);
  ((KSP_BCGSHIP *)ksp->data)->fused = (PetscBool)iv;
  return 0;
}
static PetscErrorCode KSPDestroy_BCGSHIP(KSP ksp) {
  KSP_BCGSHIP *bc = (KSP_BCGSHIP *)ksp->data;
  if (bc) { PetscErrorCode ierr = VecDestroy(&bc->dinv);CHKERRQ(ierr); HipFree(bc); ksp->data = NULL; }
  return 0;
}

/* out = K in (K = B A, left preconditioning) and, when the types can, the reduction(s) that follow in the same pass.
 * which = 1: *s1 = (out, other); which = 2: *s1 = (other, out), *s2 = (out, out).  scratch: a vector for A in. */
static PetscErrorCode bcgs_apply(KSP ksp, Mat A, const VecKrylovFusedOps *F, HipPCKind pck, Vec d, Vec in, Vec out, Vec scratch, int which, Vec other,
                                 PetscScalar *s1, PetscReal *s2) {
  PetscErrorCode ierr;
  PetscBool done = PETSC_FALSE, scaled = PETSC_FALSE;
  if (F && pck != PCKIND_GENERAL) {
    if (d) { ierr = mat_mult_scaled(A, d, in, out, &scaled);CHKERRQ(ierr); }
    if (!scaled) {
      ierr = KSP_MatMult(ksp, A, in, scratch);CHKERRQ(ierr);
      if (which == 1) { ierr = F->pmult_dot(out, scratch, d, other, s1, &done);CHKERRQ(ierr); }
      else { ierr = F->pmult_dotnorm2(out, scratch, d, other, s1, s2, &done);CHKERRQ(ierr); }
      if (!done) { ierr = KSP_PCApply(ksp, scratch, out);CHKERRQ(ierr); }
    }
  } else { ierr = KSP_PCApplyBAorAB(ksp, in, out, scratch);CHKERRQ(ierr); }
  if (!done) {
    if (which == 1) { ierr = VecDot(out, other, s1);CHKERRQ(ierr); }
    else { ierr = VecDotNorm2(other, out, s1, s2);CHKERRQ(ierr); }
  }
  return 0;
}

static PetscErrorCode KSPSolve_BCGSHIP(KSP ksp) {
  PetscErrorCode ierr;
  KSP_BCGSHIP *bc = (KSP_BCGSHIP *)ksp->data;
  Vec x = ksp->vec_sol, rhs = ksp->vec_rhs, r = ksp->work[0], rshadow = ksp->work[1], v = ksp->work[2], t = ksp->work[3], s = ksp->work[4], p = ksp->work[5];
  Mat A;
  const VecKrylovFusedOps *F = bc->fused ? vec_fused_ops(x) : NULL;
  HipPCKind pck = PCKIND_GENERAL;
  Vec d = NULL;
  const PetscBool nonorm = (PetscBool)(ksp->normtype == KSP_NORM_NONE);
  PetscScalar rho = 0.0, rho_used, rho_prev = 1.0, alpha = 1.0, omega, omega_prev = 1.0, beta, sv;
  PetscReal rn = 0.0, tt;
  PetscBool have_rho = PETSC_FALSE;
  PetscInt it;

  if (ksp->pc_side == PC_RIGHT) SETERRQ(HipObjComm(ksp), PETSC_ERR_SUP, "right-preconditioned BiCGStab is outside the ported path");
  ierr = PCGetOperators(ksp->pc, &A, NULL, NULL);CHKERRQ(ierr);
  if (F) { ierr = pc_diagonal_form(ksp, x, &bc->dinv, &pck);CHKERRQ(ierr); if (pck == PCKIND_DIAGONAL) d = bc->dinv; }
  ierr = KSPInitialResidual(ksp, x, v, t, r, rhs);CHKERRQ(ierr);
  if (!nonorm) { ierr = VecNorm(r, NORM_2, &rn);CHKERRQ(ierr); }
  ksp->its = 0;
  ierr = report(ksp, 0, rn);CHKERRQ(ierr);
  ierr = KSPTestConvergence(ksp, 0, rn);CHKERRQ(ierr);
  if (ksp->reason) return 0;
  ierr = VecCopy(r, rshadow);CHKERRQ(ierr);
  ierr = VecSet(p, 0.0);CHKERRQ(ierr);
  ierr = VecSet(v, 0.0);CHKERRQ(ierr);
  it = 0;
  do {
    PetscBool done = PETSC_FALSE;
    if (!have_rho) { ierr = VecDot(r, rshadow, &rho);CHKERRQ(ierr); }
    have_rho = PETSC_FALSE;
    rho_used = rho;
    beta = (rho / rho_prev) * (alpha / omega_prev);
    ierr = VecAXPBYPCZ(p, 1.0, -omega_prev * beta, beta, r, v);CHKERRQ(ierr);          /* p <- r - omega beta v + beta p */
    ierr = bcgs_apply(ksp, A, F, pck, d, p, v, t, 1, rshadow, &sv, NULL);CHKERRQ(ierr); /* v <- K p, (v, r~) */
    if (sv == 0.0) SETERRQ(HipObjComm(ksp), PETSC_ERR_PLIB, "Divide by zero");
    alpha = rho / sv;
    ierr = VecWAXPY(s, -alpha, v, r);CHKERRQ(ierr);                                    /* s <- r - alpha v */
    ierr = bcgs_apply(ksp, A, F, pck, d, s, t, r, 2, s, &sv, &tt);CHKERRQ(ierr);        /* t <- K s, (s, t), (t, t) */
    if (tt == 0.0) {                                                                    /* t = 0: if s = 0 too, alpha p may be the solution */
      ierr = VecDot(s, s, &sv);CHKERRQ(ierr);
      if (sv != 0.0) { ksp->reason = KSP_DIVERGED_BREAKDOWN; break; }
      ierr = VecAXPY(x, alpha, p);CHKERRQ(ierr);
      ksp->its++;
      ksp->rnorm = 0.0;
      ksp->reason = KSP_CONVERGED_RTOL;
      KSPLogResidualHistory(ksp, rn);
      ierr = KSPMonitor(ksp, it + 1, 0.0);CHKERRQ(ierr);
      break;
    }
    omega = sv / tt;
    if (F && pck != PCKIND_GENERAL) {                                                   /* x, r, (r, r) and the next (r, r~) in one sweep */
      PetscScalar rr, rho_next;
      ierr = F->bcgs_update(x, r, p, s, t, rshadow, alpha, omega, &rr, &rho_next, &done);CHKERRQ(ierr);
      if (done) { rn = nonorm ? 0.0 : PetscSqrtReal(rr); rho = rho_next; have_rho = PETSC_TRUE; }
    }
    if (!done) {
      ierr = VecAXPBYPCZ(x, alpha, omega, 1.0, p, s);CHKERRQ(ierr);                    /* x <- x + alpha p + omega s */
      ierr = VecWAXPY(r, -omega, t, s);CHKERRQ(ierr);                                  /* r <- s - omega t */
      if (!nonorm) { ierr = VecNorm(r, NORM_2, &rn);CHKERRQ(ierr); }
    }
    rho_prev = rho_used;
    omega_prev = omega;
    ksp->its++;
    ierr = report(ksp, it + 1, rn);CHKERRQ(ierr);
    ierr = KSPTestConvergence(ksp, it + 1, rn);CHKERRQ(ierr);
    if (ksp->reason) break;
    if (rho_used == 0.0) { ksp->reason = KSP_DIVERGED_BREAKDOWN; break; }              /* bcgs.c:146: the rho this iteration used */
    it++;
  } while (it < ksp->max_it);
  if (it >= ksp->max_it) ksp->reason = KSP_DIVERGED_ITS;
  return 0;
}

PetscErrorCode KSPCreate_BCGSHIPMI355X(KSP ksp) {
  KSP_BCGSHIP *bc;
  PetscErrorCode ierr = PetscMalloc(sizeof(*bc), &bc);CHKERRQ(ierr);
  bc->fused = PETSC_TRUE; bc->dinv = NULL;
  ksp->data = bc;
  ierr = KSPSetSupportedNorm(ksp, KSP_NORM_PRECONDITIONED, PC_LEFT, 2);CHKERRQ(ierr);
  ksp->ops->setup = KSPSetUp_BCGSHIP; ksp->ops->solve = KSPSolve_BCGSHIP; ksp->ops->setfromoptions = KSPSetFromOptions_BCGSHIP; ksp->ops->destroy = KSPDestroy_BCGSHIP;
  return 0;
}
