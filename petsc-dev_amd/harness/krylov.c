/* HARNESS file: a plain restatement of the reference's cg.c / gmres.c / borthog2.c / bcgs.c / groppcg.c / pipecg.c for boxes
 * without PETSc (the GPU test box) -- the op-by-op sequences an UNCHANGED PETSc program drives over the plug-in's Vec/Mat types
 * (SURVEY 8a25).  In a PETSc tree this file is NOT used: the reference's own KSPSolve_CG / _GMRES / _BCGS take its place.  It is
 * not part of the product library (libpetschipmi355x.so), links against nothing device-specific and knows no fused kernel:
 * the plug-in's own solvers are separate KSP types (KSPCGHIPMI355X, KSPGMRESHIPMI355X, KSPBCGSHIPMI355X; host/kspfused.c),
 * registered through KSPRegister like any third-party KSP.
 *
 * Operation sequences follow the reference exactly (so iteration counts and residual histories are comparable):
 * KSPSolve_CG src/ksp/ksp/impls/cg/cg.c:92-286, KSPSolve_GMRES / KSPGMRESCycle src/ksp/ksp/impls/gmres/gmres.c:118-409 with
 * classical Gram-Schmidt src/ksp/ksp/impls/gmres/borthog2.c:35-119, KSPSolve_BCGS src/ksp/ksp/impls/bcgs/bcgs.c:43-160. */
#include "petscimpl.h"

/* ================================================================== CG */
typedef struct { PetscBool singlereduction; } KSP_CG;   /* cgimpl.h */
static PetscErrorCode KSPSetUp_CG(KSP ksp) {   /* cg.c:50-80 (no eigenvalue work): 3 work vectors, 5 with -ksp_cg_single_reduction */
  return KSPDefaultGetWork(ksp, ((KSP_CG *)ksp->data)->singlereduction ? 5 : 3);
}
static PetscErrorCode KSPSetFromOptions_CG(KSP ksp) {   /* cg.c:330-345 */
  char t[16]; PetscBool set;
  PetscErrorCode ierr = PetscOptionsGetString(ksp->prefix, "-ksp_cg_single_reduction", t, sizeof(t), &set);CHKERRQ(ierr);
  if (set) ((KSP_CG *)ksp->data)->singlereduction = (PetscBool)(strcmp(t, "0") && strcmp(t, "false"));
  return 0;
}
static PetscErrorCode KSPDestroy_CG(KSP ksp) { free(ksp->data); ksp->data = NULL; return 0; }

static PetscErrorCode KSPSolve_CG(KSP ksp) {
  PetscErrorCode ierr;
  PetscInt i;
  PetscScalar dpi = 0.0, a = 1.0, beta, betaold = 1.0, b = 0, dpiold, delta = 0.0;
  PetscReal dp = 0.0;
  const PetscBool single = ((KSP_CG *)ksp->data)->singlereduction;
  Vec X = ksp->vec_sol, B = ksp->vec_rhs, R = ksp->work[0], Z = ksp->work[1], P = ksp->work[2];
  Vec S = single ? ksp->work[3] : NULL, W = single ? ksp->work[4] : Z;   /* cg.c:116-122 */
  Mat Amat = ksp->pc->mat;

  const KSPNormType nt = ksp->normtype;   /* cg.c:136-161,233-260: which norm the convergence test sees */
  ksp->its = 0;
  if (!ksp->guess_zero) {
    ierr = KSP_MatMult(ksp, Amat, X, R);CHKERRQ(ierr);            /* r <- b - Ax */
    ierr = VecAYPX(R, -1.0, B);CHKERRQ(ierr);
  } else { ierr = VecCopy(B, R);CHKERRQ(ierr); }                 /* r <- b (x is 0) */
  switch (nt) {
  case KSP_NORM_PRECONDITIONED:
    ierr = KSP_PCApply(ksp, R, Z);CHKERRQ(ierr);                 /* z <- Br */
    ierr = VecNorm(Z, NORM_2, &dp);CHKERRQ(ierr);
    break;
  case KSP_NORM_UNPRECONDITIONED:
    ierr = VecNorm(R, NORM_2, &dp);CHKERRQ(ierr);
    break;
  case KSP_NORM_NATURAL:
    ierr = KSP_PCApply(ksp, R, Z);CHKERRQ(ierr);
    if (single) { ierr = KSP_MatMult(ksp, Amat, Z, S);CHKERRQ(ierr); ierr = VecTDot(Z, S, &delta);CHKERRQ(ierr); }
    ierr = VecTDot(Z, R, &beta);CHKERRQ(ierr);
    if (PetscIsInfOrNanScalar(beta)) SETERRQ(ksp->comm, PETSC_ERR_FP, "Infinite or not-a-number generated in dot product");
    dp = PetscSqrtReal(PetscAbsScalar(beta));
    break;
  case KSP_NORM_NONE: dp = 0.0; break;
  default: SETERRQ(ksp->comm, PETSC_ERR_SUP, "norm type %d", (int)nt);
  }
  KSPLogResidualHistory(ksp, dp);
  ierr = KSPMonitor(ksp, 0, dp);CHKERRQ(ierr);
  ksp->rnorm = dp;
  ierr = (*ksp->converged)(ksp, 0, dp, &ksp->reason, ksp->cnvP);CHKERRQ(ierr);
  if (ksp->reason) return 0;
  if (nt != KSP_NORM_PRECONDITIONED && nt != KSP_NORM_NATURAL) { ierr = KSP_PCApply(ksp, R, Z);CHKERRQ(ierr); }
  if (nt != KSP_NORM_NATURAL) {
    if (single) {                                                 /* cg.c:166-169 */
      ierr = KSP_MatMult(ksp, Amat, Z, S);CHKERRQ(ierr);
      ierr = VecTDot(Z, S, &delta);CHKERRQ(ierr);
    }
    ierr = VecTDot(Z, R, &beta);CHKERRQ(ierr);                   /* beta <- z'*r */
    if (PetscIsInfOrNanScalar(beta)) SETERRQ(ksp->comm, PETSC_ERR_FP, "Infinite or not-a-number generated in dot product");
  }

  i = 0;
  do {
    ksp->its = i + 1;
    if (beta == 0.0) { ksp->reason = KSP_CONVERGED_ATOL; break; }
    else if ((i > 0) && (beta * betaold < 0.0)) { ksp->reason = KSP_DIVERGED_INDEFINITE_PC; break; }
    if (!i) { ierr = VecCopy(Z, P);CHKERRQ(ierr); b = 0.0; }       /* p <- z */
    else { b = beta / betaold; ierr = VecAYPX(P, b, Z);CHKERRQ(ierr); }   /* p <- z + b*p */
    dpiold = dpi;
    if (!single || !i) {
      ierr = KSP_MatMult(ksp, Amat, P, W);CHKERRQ(ierr);           /* w <- Ap */
      ierr = VecTDot(P, W, &dpi);CHKERRQ(ierr);                    /* dpi <- p'w */
    } else {                                                        /* cg.c:200-203: recurrences instead of a product and a dot */
      ierr = VecAYPX(W, beta / betaold, S);CHKERRQ(ierr);          /* w <- Ap */
      dpi = delta - beta * beta * dpiold / (betaold * betaold);    /* dpi <- p'w */
    }
    betaold = beta;
    if (PetscIsInfOrNanScalar(dpi)) SETERRQ(ksp->comm, PETSC_ERR_FP, "Infinite or not-a-number generated in dot product");
    if ((dpi == 0.0) || ((i > 0) && (dpi * dpiold <= 0.0))) { ksp->reason = KSP_DIVERGED_INDEFINITE_MAT; break; }
    a = beta / dpi;
    ierr = VecAXPY(X, a, P);CHKERRQ(ierr);                         /* x <- x + ap */
    ierr = VecAXPY(R, -a, W);CHKERRQ(ierr);                        /* r <- r - aw */
    if (nt == KSP_NORM_PRECONDITIONED) {
      ierr = KSP_PCApply(ksp, R, Z);CHKERRQ(ierr);                 /* z <- Br */
      if (single) { ierr = KSP_MatMult(ksp, Amat, Z, S);CHKERRQ(ierr); }   /* cg.c:217-219 */
      ierr = VecNorm(Z, NORM_2, &dp);CHKERRQ(ierr);
    } else if (nt == KSP_NORM_UNPRECONDITIONED) {
      ierr = VecNorm(R, NORM_2, &dp);CHKERRQ(ierr);
    } else if (nt == KSP_NORM_NATURAL) {
      ierr = KSP_PCApply(ksp, R, Z);CHKERRQ(ierr);
      if (single) {
        PetscScalar tmp[2]; Vec vecs[2];
        vecs[0] = S; vecs[1] = R;
        ierr = KSP_MatMult(ksp, Amat, Z, S);CHKERRQ(ierr);
        ierr = VecMDot(Z, 2, vecs, tmp);CHKERRQ(ierr);
        delta = tmp[0]; beta = tmp[1];
      } else { ierr = VecTDot(Z, R, &beta);CHKERRQ(ierr); }
      if (PetscIsInfOrNanScalar(beta)) SETERRQ(ksp->comm, PETSC_ERR_FP, "Infinite or not-a-number generated in dot product");
      dp = PetscSqrtReal(PetscAbsScalar(beta));
    } else dp = 0.0;
    ksp->rnorm = dp;
    KSPLogResidualHistory(ksp, dp);
    ierr = KSPMonitor(ksp, i + 1, dp);CHKERRQ(ierr);
    ierr = (*ksp->converged)(ksp, i + 1, dp, &ksp->reason, ksp->cnvP);CHKERRQ(ierr);
    if (ksp->reason) break;
    if (nt != KSP_NORM_PRECONDITIONED && nt != KSP_NORM_NATURAL) {
      ierr = KSP_PCApply(ksp, R, Z);CHKERRQ(ierr);                 /* z <- Br */
      if (single) { ierr = KSP_MatMult(ksp, Amat, Z, S);CHKERRQ(ierr); }
    }
    if (nt != KSP_NORM_NATURAL) {
      if (single) {                                                 /* cg.c:263-270: one VecMDot(2) = one reduction for delta and beta */
        PetscScalar tmp[2]; Vec vecs[2];
        vecs[0] = S; vecs[1] = R;
        ierr = VecMDot(Z, 2, vecs, tmp);CHKERRQ(ierr);
        delta = tmp[0]; beta = tmp[1];
      } else { ierr = VecTDot(Z, R, &beta);CHKERRQ(ierr); }        /* beta <- z'*r */
      if (PetscIsInfOrNanScalar(beta)) SETERRQ(ksp->comm, PETSC_ERR_FP, "Infinite or not-a-number generated in dot product");
    }
    i++;
  } while (i < ksp->max_it);
  if (i >= ksp->max_it) ksp->reason = KSP_DIVERGED_ITS;
  return 0;
}
static void cg_family_norms(KSP ksp, PetscInt pref) {   /* cg.c:439-442, groppcg.c:170-173, pipecg.c:199-202 */
  ksp->normsupporttable[KSP_NORM_PRECONDITIONED][PC_LEFT] = pref;
  ksp->normsupporttable[KSP_NORM_UNPRECONDITIONED][PC_LEFT] = 1;
  ksp->normsupporttable[KSP_NORM_NATURAL][PC_LEFT] = 1;
  ksp->normsupporttable[KSP_NORM_NONE][PC_LEFT] = 1;
}
PetscErrorCode KSPCreate_CG(KSP ksp) {
  KSP_CG *cg;
  PetscErrorCode ierr = PetscMalloc(sizeof(*cg), &cg);CHKERRQ(ierr);
  cg->singlereduction = PETSC_FALSE;
  ksp->data = cg;
  cg_family_norms(ksp, 2);
  ksp->ops->setup = KSPSetUp_CG; ksp->ops->solve = KSPSolve_CG; ksp->ops->setfromoptions = KSPSetFromOptions_CG; ksp->ops->destroy = KSPDestroy_CG;
  return 0;
}

/* ================================================================== GROPPCG
 * Gropp's variant of CG (src/ksp/ksp/impls/cg/groppcg/groppcg.c:40-175): the same Krylov space, two
 * reductions per iteration, each overlapped with work that does not need its result -- (p,s) with the preconditioner
 * application, {norm, (r,z)} with the matrix product.  The reductions are split-phase (VecDotBegin/End,
 * PetscCommSplitReductionBegin): on several GPUs the all-reduce travels on the halo stream while the compute stream
 * runs the overlapped kernel.  Six work vectors, all four norm types. */
static PetscErrorCode KSPSetUp_GROPPCG(KSP ksp) { return KSPDefaultGetWork(ksp, 6); }
static PetscErrorCode KSPSolve_GROPPCG(KSP ksp) {
  PetscErrorCode ierr;
  PetscInt i;
  PetscScalar alpha, beta = 0.0, gamma, gammaNew, t;
  PetscReal dp = 0.0;
  Vec x = ksp->vec_sol, b = ksp->vec_rhs, r = ksp->work[0], p = ksp->work[1], s = ksp->work[2], S = ksp->work[3], z = ksp->work[4], Z = ksp->work[5];
  Mat Amat = ksp->pc->mat;
  const KSPNormType nt = ksp->normtype;

  ksp->its = 0;
  if (!ksp->guess_zero) {
    ierr = KSP_MatMult(ksp, Amat, x, r);CHKERRQ(ierr);           /* r <- b - Ax */
    ierr = VecAYPX(r, -1.0, b);CHKERRQ(ierr);
  } else { ierr = VecCopy(b, r);CHKERRQ(ierr); }
  ierr = KSP_PCApply(ksp, r, z);CHKERRQ(ierr);                   /* z <- Br */
  ierr = VecCopy(z, p);CHKERRQ(ierr);                            /* p <- z */
  ierr = VecDotBegin(r, z, &gamma);CHKERRQ(ierr);                /* gamma <- z'r, overlapped with s <- Ap */
  ierr = PetscCommSplitReductionBegin(r->comm);CHKERRQ(ierr);
  ierr = KSP_MatMult(ksp, Amat, p, s);CHKERRQ(ierr);
  ierr = VecDotEnd(r, z, &gamma);CHKERRQ(ierr);
  switch (nt) {
  case KSP_NORM_PRECONDITIONED: ierr = VecNorm(z, NORM_2, &dp);CHKERRQ(ierr); break;
  case KSP_NORM_UNPRECONDITIONED: ierr = VecNorm(r, NORM_2, &dp);CHKERRQ(ierr); break;
  case KSP_NORM_NATURAL:
    if (PetscIsInfOrNanScalar(gamma)) SETERRQ(ksp->comm, PETSC_ERR_FP, "Infinite or not-a-number generated in dot product");
    dp = PetscSqrtReal(PetscAbsScalar(gamma));
    break;
  case KSP_NORM_NONE: dp = 0.0; break;
  default: SETERRQ(ksp->comm, PETSC_ERR_SUP, "norm type %d", (int)nt);
  }
  KSPLogResidualHistory(ksp, dp);
  ierr = KSPMonitor(ksp, 0, dp);CHKERRQ(ierr);
  ksp->rnorm = dp;
  ierr = (*ksp->converged)(ksp, 0, dp, &ksp->reason, ksp->cnvP);CHKERRQ(ierr);
  if (ksp->reason) return 0;

  i = 0;
  do {
    ksp->its = i + 1;
    i++;
    ierr = VecDotBegin(p, s, &t);CHKERRQ(ierr);                  /* t <- p's, overlapped with S <- Bs */
    ierr = PetscCommSplitReductionBegin(p->comm);CHKERRQ(ierr);
    ierr = KSP_PCApply(ksp, s, S);CHKERRQ(ierr);
    ierr = VecDotEnd(p, s, &t);CHKERRQ(ierr);
    alpha = gamma / t;
    ierr = VecAXPY(x, alpha, p);CHKERRQ(ierr);                   /* x <- x + alpha p */
    ierr = VecAXPY(r, -alpha, s);CHKERRQ(ierr);                  /* r <- r - alpha s */
    ierr = VecAXPY(z, -alpha, S);CHKERRQ(ierr);                  /* z <- z - alpha S */
    if (nt == KSP_NORM_UNPRECONDITIONED) { ierr = VecNormBegin(r, NORM_2, &dp);CHKERRQ(ierr); }
    else if (nt == KSP_NORM_PRECONDITIONED) { ierr = VecNormBegin(z, NORM_2, &dp);CHKERRQ(ierr); }
    ierr = VecDotBegin(r, z, &gammaNew);CHKERRQ(ierr);           /* overlapped with Z <- Az */
    ierr = PetscCommSplitReductionBegin(r->comm);CHKERRQ(ierr);
    ierr = KSP_MatMult(ksp, Amat, z, Z);CHKERRQ(ierr);
    if (nt == KSP_NORM_UNPRECONDITIONED) { ierr = VecNormEnd(r, NORM_2, &dp);CHKERRQ(ierr); }
    else if (nt == KSP_NORM_PRECONDITIONED) { ierr = VecNormEnd(z, NORM_2, &dp);CHKERRQ(ierr); }
    ierr = VecDotEnd(r, z, &gammaNew);CHKERRQ(ierr);
    if (nt == KSP_NORM_NATURAL) {
      if (PetscIsInfOrNanScalar(gammaNew)) SETERRQ(ksp->comm, PETSC_ERR_FP, "Infinite or not-a-number generated in dot product");
      dp = PetscSqrtReal(PetscAbsScalar(gammaNew));
    } else if (nt == KSP_NORM_NONE) dp = 0.0;
    ksp->rnorm = dp;
    KSPLogResidualHistory(ksp, dp);
    ierr = KSPMonitor(ksp, i, dp);CHKERRQ(ierr);
    ierr = (*ksp->converged)(ksp, i, dp, &ksp->reason, ksp->cnvP);CHKERRQ(ierr);
    if (ksp->reason) break;
    beta = gammaNew / gamma;
    gamma = gammaNew;
    ierr = VecAYPX(p, beta, z);CHKERRQ(ierr);                    /* p <- z + beta p */
    ierr = VecAYPX(s, beta, Z);CHKERRQ(ierr);                    /* s <- Z + beta s */
  } while (i < ksp->max_it);
  if (i >= ksp->max_it && !ksp->reason) ksp->reason = KSP_DIVERGED_ITS;
  return 0;
}
PetscErrorCode KSPCreate_GROPPCG(KSP ksp) { cg_family_norms(ksp, 1); ksp->ops->setup = KSPSetUp_GROPPCG; ksp->ops->solve = KSPSolve_GROPPCG; return 0; }

/* ================================================================== PIPECG
 * Pipelined CG of Ghysels & Vanroose (src/ksp/ksp/impls/cg/pipecg/pipecg.c:49-205): ONE split-phase reduction per
 * iteration -- {norm or (r,u), (w,u)} -- overlapped with m = B w and n = A m; nine work vectors, four extra recurrences.
 * Restated as this snapshot has it, including the branch structure of its reductions (pipecg.c:124-131,138-145): with the
 * preconditioned or the unpreconditioned norm gamma = (r,u) is reduced in iteration 0 only and the recurrence then runs with
 * beta = 1; the natural norm and KSP_NORM_NONE refresh gamma every iteration and are the forms that converge like KSPCG. */
static PetscErrorCode KSPSetUp_PIPECG(KSP ksp) { return KSPDefaultGetWork(ksp, 9); }
static PetscErrorCode KSPSolve_PIPECG(KSP ksp) {
  PetscErrorCode ierr;
  PetscInt i;
  PetscScalar alpha = 0.0, beta = 0.0, gamma = 0.0, gammaold = 0.0, delta = 0.0;
  PetscReal dp = 0.0;
  Vec X = ksp->vec_sol, B = ksp->vec_rhs, M = ksp->work[0], Z = ksp->work[1], P = ksp->work[2], N = ksp->work[3], W = ksp->work[4],
      Q = ksp->work[5], U = ksp->work[6], R = ksp->work[7], S = ksp->work[8];
  Mat Amat = ksp->pc->mat;
  const KSPNormType nt = ksp->normtype;

  ksp->its = 0;
  if (!ksp->guess_zero) {
    ierr = KSP_MatMult(ksp, Amat, X, R);CHKERRQ(ierr);
    ierr = VecAYPX(R, -1.0, B);CHKERRQ(ierr);
  } else { ierr = VecCopy(B, R);CHKERRQ(ierr); }
  ierr = KSP_PCApply(ksp, R, U);CHKERRQ(ierr);
  switch (nt) {
  case KSP_NORM_PRECONDITIONED:
    ierr = VecNormBegin(U, NORM_2, &dp);CHKERRQ(ierr);
    ierr = PetscCommSplitReductionBegin(U->comm);CHKERRQ(ierr);
    ierr = KSP_MatMult(ksp, Amat, U, W);CHKERRQ(ierr);
    ierr = VecNormEnd(U, NORM_2, &dp);CHKERRQ(ierr);
    break;
  case KSP_NORM_UNPRECONDITIONED:
    ierr = VecNormBegin(R, NORM_2, &dp);CHKERRQ(ierr);
    ierr = PetscCommSplitReductionBegin(R->comm);CHKERRQ(ierr);
    ierr = KSP_MatMult(ksp, Amat, U, W);CHKERRQ(ierr);
    ierr = VecNormEnd(R, NORM_2, &dp);CHKERRQ(ierr);
    break;
  case KSP_NORM_NATURAL:
    ierr = VecDotBegin(R, U, &gamma);CHKERRQ(ierr);
    ierr = PetscCommSplitReductionBegin(R->comm);CHKERRQ(ierr);
    ierr = KSP_MatMult(ksp, Amat, U, W);CHKERRQ(ierr);
    ierr = VecDotEnd(R, U, &gamma);CHKERRQ(ierr);
    if (PetscIsInfOrNanScalar(gamma)) SETERRQ(ksp->comm, PETSC_ERR_FP, "Infinite or not-a-number generated in dot product");
    dp = PetscSqrtReal(PetscAbsScalar(gamma));
    break;
  case KSP_NORM_NONE:
    ierr = KSP_MatMult(ksp, Amat, U, W);CHKERRQ(ierr);
    dp = 0.0;
    break;
  default: SETERRQ(ksp->comm, PETSC_ERR_SUP, "norm type %d", (int)nt);
  }
  KSPLogResidualHistory(ksp, dp);
  ierr = KSPMonitor(ksp, 0, dp);CHKERRQ(ierr);
  ksp->rnorm = dp;
  ierr = (*ksp->converged)(ksp, 0, dp, &ksp->reason, ksp->cnvP);CHKERRQ(ierr);
  if (ksp->reason) return 0;

  i = 0;
  do {
    const int red = (i > 0 && nt == KSP_NORM_UNPRECONDITIONED) ? 1 : (i > 0 && nt == KSP_NORM_PRECONDITIONED) ? 2 : !(i == 0 && nt == KSP_NORM_NATURAL) ? 3 : 0;
    if (red == 1) { ierr = VecNormBegin(R, NORM_2, &dp);CHKERRQ(ierr); }
    else if (red == 2) { ierr = VecNormBegin(U, NORM_2, &dp);CHKERRQ(ierr); }
    else if (red == 3) { ierr = VecDotBegin(R, U, &gamma);CHKERRQ(ierr); }
    ierr = VecDotBegin(W, U, &delta);CHKERRQ(ierr);
    ierr = PetscCommSplitReductionBegin(R->comm);CHKERRQ(ierr);
    ierr = KSP_PCApply(ksp, W, M);CHKERRQ(ierr);                 /* the overlapped work: m = B w, n = A m */
    ierr = KSP_MatMult(ksp, Amat, M, N);CHKERRQ(ierr);
    if (red == 1) { ierr = VecNormEnd(R, NORM_2, &dp);CHKERRQ(ierr); }
    else if (red == 2) { ierr = VecNormEnd(U, NORM_2, &dp);CHKERRQ(ierr); }
    else if (red == 3) { ierr = VecDotEnd(R, U, &gamma);CHKERRQ(ierr); }
    ierr = VecDotEnd(W, U, &delta);CHKERRQ(ierr);
    if (i > 0) {
      if (nt == KSP_NORM_NATURAL) dp = PetscSqrtReal(PetscAbsScalar(gamma));
      else if (nt == KSP_NORM_NONE) dp = 0.0;
      ksp->rnorm = dp;
      KSPLogResidualHistory(ksp, dp);
      ierr = KSPMonitor(ksp, i, dp);CHKERRQ(ierr);
      ierr = (*ksp->converged)(ksp, i, dp, &ksp->reason, ksp->cnvP);CHKERRQ(ierr);
      if (ksp->reason) break;
    }
    if (i == 0) {
      alpha = gamma / delta;
      ierr = VecCopy(N, Z);CHKERRQ(ierr);
      ierr = VecCopy(M, Q);CHKERRQ(ierr);
      ierr = VecCopy(U, P);CHKERRQ(ierr);
      ierr = VecCopy(W, S);CHKERRQ(ierr);
    } else {
      beta = gamma / gammaold;
      alpha = gamma / (delta - beta / alpha * gamma);
      ierr = VecAYPX(Z, beta, N);CHKERRQ(ierr);
      ierr = VecAYPX(Q, beta, M);CHKERRQ(ierr);
      ierr = VecAYPX(P, beta, U);CHKERRQ(ierr);
      ierr = VecAYPX(S, beta, W);CHKERRQ(ierr);
    }
    ierr = VecAXPY(X, alpha, P);CHKERRQ(ierr);
    ierr = VecAXPY(U, -alpha, Q);CHKERRQ(ierr);
    ierr = VecAXPY(W, -alpha, Z);CHKERRQ(ierr);
    ierr = VecAXPY(R, -alpha, S);CHKERRQ(ierr);
    gammaold = gamma;
    i++;
    ksp->its = i;
  } while (i < ksp->max_it);
  if (i >= ksp->max_it) ksp->reason = KSP_DIVERGED_ITS;
  return 0;
}
PetscErrorCode KSPCreate_PIPECG(KSP ksp) { cg_family_norms(ksp, 1); ksp->ops->setup = KSPSetUp_PIPECG; ksp->ops->solve = KSPSolve_PIPECG; return 0; }

/* ================================================================== GMRES(m) */
typedef struct {
  PetscInt max_k;
  PetscReal haptol;
  KSPGMRESCGSRefinementType cgstype;
  PetscScalar *hh, *hes, *grs, *cc, *ss, *lhh, *nrs;
  Vec *vecs;      /* [0]=TEMP, [1]=TEMP_MATOP, [2+k]=VV(k) */
  PetscInt nvecs;
  PetscInt it;
} KSP_GMRES;
#define GM ((KSP_GMRES *)ksp->data)
#define HH(a, b) (g->hh + (size_t)(b) * (size_t)(g->max_k + 2) + (a))      /* gmresimpl.h */
#define HES(a, b) (g->hes + (size_t)(b) * (size_t)(g->max_k + 1) + (a))
#define VEC_TEMP g->vecs[0]
#define VEC_TEMP_MATOP g->vecs[1]
#define VEC_VV(i) g->vecs[2 + (i)]

static PetscErrorCode KSPGMRESSetRestart_GMRES(KSP ksp, PetscInt restart) {   /* gmres.c:752-770 */
  if (restart < 1) SETERRQ(ksp->comm, PETSC_ERR_ARG_OUTOFRANGE, "Restart must be positive");
  if (ksp->setupcalled) SETERRQ(ksp->comm, PETSC_ERR_ORDER, "Must call KSPGMRESSetRestart() before KSPSetUp()");
  GM->max_k = restart;
  return 0;
}
static PetscErrorCode KSPGMRESSetCGSRefinementType_GMRES(KSP ksp, KSPGMRESCGSRefinementType type) {
  GM->cgstype = type;
  return 0;
}
static PetscErrorCode KSPSetFromOptions_GMRES(KSP ksp) {
  PetscErrorCode ierr; PetscInt iv; PetscBool set; char t[64];
  ierr = PetscOptionsGetInt(ksp->prefix, "-ksp_gmres_restart", &iv, &set);CHKERRQ(ierr);
  if (set) { ierr = KSPGMRESSetRestart(ksp, iv);CHKERRQ(ierr); }
  ierr = PetscOptionsGetString(ksp->prefix, "-ksp_gmres_cgs_refinement_type", t, sizeof(t), &set);CHKERRQ(ierr);
  if (set) {
    if (!strcmp(t, "refine_always")) GM->cgstype = KSP_GMRES_CGS_REFINE_ALWAYS;
    else if (!strcmp(t, "refine_ifneeded")) GM->cgstype = KSP_GMRES_CGS_REFINE_IFNEEDED;
    else if (!strcmp(t, "refine_never")) GM->cgstype = KSP_GMRES_CGS_REFINE_NEVER;
    else SETERRQ(ksp->comm, PETSC_ERR_ARG_UNKNOWN_TYPE, "Unknown refinement type %s", t);
  }
  return 0;
}

static PetscErrorCode KSPSetUp_GMRES(KSP ksp) {   /* gmres.c:36-96; all max_k+2 basis vectors are allocated up front */
  PetscErrorCode ierr;
  KSP_GMRES *g = GM;
  PetscInt k = g->max_k;
  g->hh = (PetscScalar *)calloc((size_t)(k + 2) * (size_t)(k + 1), sizeof(PetscScalar));
  g->hes = (PetscScalar *)calloc((size_t)(k + 1) * (size_t)(k + 1), sizeof(PetscScalar));
  g->grs = (PetscScalar *)calloc((size_t)k + 2, sizeof(PetscScalar));
  g->cc = (PetscScalar *)calloc((size_t)k + 1, sizeof(PetscScalar));
  g->ss = (PetscScalar *)calloc((size_t)k + 1, sizeof(PetscScalar));
  g->lhh = (PetscScalar *)calloc((size_t)k + 2, sizeof(PetscScalar));
  g->nrs = (PetscScalar *)calloc((size_t)k + 2, sizeof(PetscScalar));
  if (!g->hh || !g->hes || !g->grs || !g->cc || !g->ss || !g->lhh || !g->nrs) SETERRQ(ksp->comm, PETSC_ERR_MEM, "out of memory");
  g->nvecs = k + 4;
  if (ksp->vec_sol) { ierr = VecDuplicateVecs(ksp->vec_sol, g->nvecs, &g->vecs);CHKERRQ(ierr); }
  else {
    Vec r;
    ierr = MatGetVecs(ksp->pc->mat, &r, NULL);CHKERRQ(ierr);
    ierr = VecDuplicateVecs(r, g->nvecs, &g->vecs);CHKERRQ(ierr);
    ierr = VecDestroy(&r);CHKERRQ(ierr);
  }
  return 0;
}

/* KSPGMRESClassicalGramSchmidtOrthogonalization, borthog2.c:35-119 */
static PetscErrorCode gmres_orthog(KSP ksp, PetscInt it) {
  PetscErrorCode ierr;
  KSP_GMRES *g = GM;
  PetscScalar *hh = HH(0, it), *hes = HES(0, it), *lhh = g->lhh;
  PetscBool refine = (PetscBool)(g->cgstype == KSP_GMRES_CGS_REFINE_ALWAYS);
  for (PetscInt j = 0; j <= it; j++) { hh[j] = 0.0; hes[j] = 0.0; }
  ierr = VecMDot(VEC_VV(it + 1), it + 1, &VEC_VV(0), lhh);CHKERRQ(ierr);        /* <v,vnew> */
  for (PetscInt j = 0; j <= it; j++) lhh[j] = -lhh[j];
  ierr = VecMAXPY(VEC_VV(it + 1), it + 1, lhh, &VEC_VV(0));CHKERRQ(ierr);
  for (PetscInt j = 0; j <= it; j++) { hh[j] -= lhh[j]; hes[j] -= lhh[j]; }
  if (g->cgstype == KSP_GMRES_CGS_REFINE_IFNEEDED) {
    PetscReal hnrm = 0.0, wnrm;
    for (PetscInt j = 0; j <= it; j++) hnrm += lhh[j] * lhh[j];
    hnrm = PetscSqrtReal(hnrm);
    ierr = VecNorm(VEC_VV(it + 1), NORM_2, &wnrm);CHKERRQ(ierr);
    if (wnrm < 1.0286 * hnrm) refine = PETSC_TRUE;
  }
  if (refine) {
    ierr = VecMDot(VEC_VV(it + 1), it + 1, &VEC_VV(0), lhh);CHKERRQ(ierr);
    for (PetscInt j = 0; j <= it; j++) lhh[j] = -lhh[j];
    ierr = VecMAXPY(VEC_VV(it + 1), it + 1, lhh, &VEC_VV(0));CHKERRQ(ierr);
    for (PetscInt j = 0; j <= it; j++) { hh[j] -= lhh[j]; hes[j] -= lhh[j]; }
  }
  return 0;
}

/* KSPGMRESUpdateHessenberg, gmres.c:360-409: host Givens rotations */
static PetscErrorCode gmres_update_hessenberg(KSP ksp, PetscInt it, PetscBool hapend, PetscReal *res) {
  KSP_GMRES *g = GM;
  PetscScalar *hh = HH(0, it), *cc = g->cc, *ss = g->ss, tt;
  for (PetscInt j = 1; j <= it; j++) {
    tt = *hh;
    *hh = *cc * tt + *ss * *(hh + 1);
    hh++;
    *hh = *cc++ * *hh - (*ss++ * tt);
  }
  if (!hapend) {
    tt = sqrt(*hh * *hh + *(hh + 1) * *(hh + 1));
    if (tt == 0.0) { ksp->reason = KSP_DIVERGED_NULL; return 0; }
    *cc = *hh / tt;
    *ss = *(hh + 1) / tt;
    g->grs[it + 1] = -(*ss * g->grs[it]);
    g->grs[it] = *cc * g->grs[it];
    *hh = *cc * *hh + *ss * *(hh + 1);
    *res = PetscAbsScalar(g->grs[it + 1]);
  } else *res = 0.0;
  return 0;
}

/* KSPGMRESBuildSoln, gmres.c:309-354 (left preconditioning: no unwinding) */
static PetscErrorCode gmres_build_soln(KSP ksp, PetscInt it) {
  PetscErrorCode ierr;
  KSP_GMRES *g = GM;
  PetscScalar tt, *nrs = g->nrs;
  if (it < 0) return 0;
  if (*HH(it, it) != 0.0) nrs[it] = g->grs[it] / *HH(it, it);
  else { ksp->reason = KSP_DIVERGED_BREAKDOWN; return 0; }
  for (PetscInt ii = 1; ii <= it; ii++) {
    PetscInt k = it - ii;
    tt = g->grs[k];
    for (PetscInt j = k + 1; j <= it; j++) tt = tt - *HH(k, j) * nrs[j];
    if (*HH(k, k) == 0.0) { ksp->reason = KSP_DIVERGED_BREAKDOWN; return 0; }
    nrs[k] = tt / *HH(k, k);
  }
  ierr = VecSet(VEC_TEMP, 0.0);CHKERRQ(ierr);
  ierr = VecMAXPY(VEC_TEMP, it + 1, nrs, &VEC_VV(0));CHKERRQ(ierr);
  if (ksp->pc_side == PC_RIGHT) {   /* KSPUnwindPreconditioner */
    ierr = KSP_PCApply(ksp, VEC_TEMP, VEC_TEMP_MATOP);CHKERRQ(ierr);
    ierr = VecCopy(VEC_TEMP_MATOP, VEC_TEMP);CHKERRQ(ierr);
  }
  ierr = VecAXPY(ksp->vec_sol, 1.0, VEC_TEMP);CHKERRQ(ierr);
  return 0;
}

/* KSPGMRESCycle, gmres.c:118-209 */
static PetscErrorCode gmres_cycle(PetscInt *itcount, KSP ksp) {
  PetscErrorCode ierr;
  KSP_GMRES *g = GM;
  PetscReal res_norm, res, hapbnd, tt;
  PetscInt it = 0, max_k = g->max_k;
  PetscBool hapend = PETSC_FALSE;

  ierr = VecNormalize(VEC_VV(0), &res_norm);CHKERRQ(ierr);
  res = res_norm;
  g->grs[0] = res_norm;
  ksp->rnorm = res;
  g->it = it - 1;
  KSPLogResidualHistory(ksp, res);
  ierr = KSPMonitor(ksp, ksp->its, res);CHKERRQ(ierr);
  if (!res) { if (itcount) *itcount = 0; ksp->reason = KSP_CONVERGED_ATOL; return 0; }
  ierr = (*ksp->converged)(ksp, ksp->its, res, &ksp->reason, ksp->cnvP);CHKERRQ(ierr);
  while (!ksp->reason && it < max_k && ksp->its < ksp->max_it) {
    if (it) { KSPLogResidualHistory(ksp, res); ierr = KSPMonitor(ksp, ksp->its, res);CHKERRQ(ierr); }
    g->it = it - 1;
    ierr = KSP_PCApplyBAorAB(ksp, VEC_VV(it), VEC_VV(1 + it), VEC_TEMP_MATOP);CHKERRQ(ierr);
    ierr = gmres_orthog(ksp, it);CHKERRQ(ierr);                    /* update hessenberg matrix and do Gram-Schmidt */
    ierr = VecNormalize(VEC_VV(it + 1), &tt);CHKERRQ(ierr);       /* vv(i+1) . vv(i+1) */
    *HH(it + 1, it) = tt;
    *HES(it + 1, it) = tt;
    hapbnd = PetscAbsScalar(tt / g->grs[it]);                      /* happy breakdown test */
    if (hapbnd > g->haptol) hapbnd = g->haptol;
    if (tt < hapbnd) hapend = PETSC_TRUE;
    ierr = gmres_update_hessenberg(ksp, it, hapend, &res);CHKERRQ(ierr);
    it++;
    g->it = it - 1;
    ksp->its++;
    ksp->rnorm = res;
    if (ksp->reason) break;
    ierr = (*ksp->converged)(ksp, ksp->its, res, &ksp->reason, ksp->cnvP);CHKERRQ(ierr);
    if (hapend) {
      if (!ksp->reason) SETERRQ(ksp->comm, PETSC_ERR_PLIB, "You reached the happy break down, but convergence was not indicated. Residual norm = %g", res);
      break;
    }
  }
  if (it && (ksp->reason || ksp->its >= ksp->max_it)) { KSPLogResidualHistory(ksp, res); ierr = KSPMonitor(ksp, ksp->its, res);CHKERRQ(ierr); }
  if (itcount) *itcount = it;
  ierr = gmres_build_soln(ksp, it - 1);CHKERRQ(ierr);
  return 0;
}

static PetscErrorCode KSPSolve_GMRES(KSP ksp) {   /* gmres.c:213-243 */
  PetscErrorCode ierr;
  KSP_GMRES *g = GM;
  PetscInt its = 0, itcount = 0;
  PetscBool guess_zero = ksp->guess_zero;
  ksp->its = 0;
  ksp->reason = KSP_CONVERGED_ITERATING;
  while (!ksp->reason) {
    ierr = KSPInitialResidual(ksp, ksp->vec_sol, VEC_TEMP, VEC_TEMP_MATOP, VEC_VV(0), ksp->vec_rhs);CHKERRQ(ierr);
    ierr = gmres_cycle(&its, ksp);CHKERRQ(ierr);
    itcount += its;
    if (itcount >= ksp->max_it) { if (!ksp->reason) ksp->reason = KSP_DIVERGED_ITS; break; }
    ksp->guess_zero = PETSC_FALSE;   /* every future call to KSPInitialResidual() will have nonzero guess */
  }
  ksp->guess_zero = guess_zero;
  return 0;
}
static PetscErrorCode KSPDestroy_GMRES(KSP ksp) {
  KSP_GMRES *g = GM;
  if (!g) return 0;
  free(g->hh); free(g->hes); free(g->grs); free(g->cc); free(g->ss); free(g->lhh); free(g->nrs);
  if (g->vecs) { PetscErrorCode ierr = VecDestroyVecs(g->nvecs, &g->vecs);CHKERRQ(ierr); }
  free(g); ksp->data = NULL;
  (void)PetscObjectComposeFunction((PetscObject)ksp, "KSPGMRESSetRestart_C", "", (PetscVoidFunction)NULL);   /* gmres.c:288-291 */
  (void)PetscObjectComposeFunction((PetscObject)ksp, "KSPGMRESSetCGSRefinementType_C", "", (PetscVoidFunction)NULL);
  return 0;
}
PetscErrorCode KSPCreate_GMRES(KSP ksp) {   /* gmres.c KSPCreate_GMRES: restart 30, haptol 1e-30, refine never */
  KSP_GMRES *g;
  PetscErrorCode ierr = PetscMalloc(sizeof(*g), &g);CHKERRQ(ierr);
  memset(g, 0, sizeof(*g));
  g->max_k = 30; g->haptol = 1.0e-30; g->cgstype = KSP_GMRES_CGS_REFINE_NEVER;
  ksp->data = g;
  ksp->normsupporttable[KSP_NORM_PRECONDITIONED][PC_LEFT] = 2;      /* gmres.c:909-910 */
  ksp->normsupporttable[KSP_NORM_UNPRECONDITIONED][PC_RIGHT] = 1;
  ksp->ops->setup = KSPSetUp_GMRES; ksp->ops->solve = KSPSolve_GMRES; ksp->ops->destroy = KSPDestroy_GMRES;
  ksp->ops->setfromoptions = KSPSetFromOptions_GMRES;
  ierr = PetscObjectComposeFunction((PetscObject)ksp, "KSPGMRESSetRestart_C", "KSPGMRESSetRestart_GMRES", (PetscVoidFunction)KSPGMRESSetRestart_GMRES);CHKERRQ(ierr);   /* gmres.c:931-942 */
  ierr = PetscObjectComposeFunction((PetscObject)ksp, "KSPGMRESSetCGSRefinementType_C", "KSPGMRESSetCGSRefinementType_GMRES", (PetscVoidFunction)KSPGMRESSetCGSRefinementType_GMRES);CHKERRQ(ierr);
  return 0;
}

/* ================================================================== BiCGStab */
static PetscErrorCode KSPSetUp_BCGS(KSP ksp) { return KSPDefaultGetWork(ksp, 6); }   /* bcgs.c:13 */

static PetscErrorCode KSPSolve_BCGS(KSP ksp) {
  PetscErrorCode ierr;
  PetscInt i;
  PetscScalar rho = 0.0, rhoold, alpha, beta, omega, omegaold, d1;
  PetscReal dp = 0.0, d2;
  Vec X = ksp->vec_sol, B = ksp->vec_rhs, R = ksp->work[0], RP = ksp->work[1], V = ksp->work[2], T = ksp->work[3], S = ksp->work[4], P = ksp->work[5];

  const PetscBool nonorm = (PetscBool)(ksp->normtype == KSP_NORM_NONE);   /* bcgs.c:76,131: smoother use, no norms, KSPSkipConverged */
  ierr = KSPInitialResidual(ksp, X, V, T, R, B);CHKERRQ(ierr);   /* initial preconditioned residual */
  if (!nonorm) { ierr = VecNorm(R, NORM_2, &dp);CHKERRQ(ierr); }
  ksp->its = 0;
  ksp->rnorm = dp;
  KSPLogResidualHistory(ksp, dp);
  ierr = KSPMonitor(ksp, 0, dp);CHKERRQ(ierr);
  ierr = (*ksp->converged)(ksp, 0, dp, &ksp->reason, ksp->cnvP);CHKERRQ(ierr);
  if (ksp->reason) return 0;
  ierr = VecCopy(R, RP);CHKERRQ(ierr);                           /* rp == r */
  rhoold = 1.0; alpha = 1.0; omegaold = 1.0;
  ierr = VecSet(P, 0.0);CHKERRQ(ierr);
  ierr = VecSet(V, 0.0);CHKERRQ(ierr);
  i = 0;
  do {
    ierr = VecDot(R, RP, &rho);CHKERRQ(ierr);                    /* rho <- (r,rp) */
    beta = (rho / rhoold) * (alpha / omegaold);
    ierr = VecAXPBYPCZ(P, 1.0, -omegaold * beta, beta, R, V);CHKERRQ(ierr);   /* p <- r - omega*beta*v + beta*p */
    ierr = KSP_PCApplyBAorAB(ksp, P, V, T);CHKERRQ(ierr);        /* v <- K p */
    ierr = VecDot(V, RP, &d1);CHKERRQ(ierr);
    if (d1 == 0.0) SETERRQ(ksp->comm, PETSC_ERR_PLIB, "Divide by zero");
    alpha = rho / d1;
    ierr = VecWAXPY(S, -alpha, V, R);CHKERRQ(ierr);              /* s <- r - a v */
    ierr = KSP_PCApplyBAorAB(ksp, S, T, R);CHKERRQ(ierr);        /* t <- K s */
    ierr = VecDotNorm2(S, T, &d1, &d2);CHKERRQ(ierr);
    if (d2 == 0.0) {
      /* t is 0: if s is 0 too, alpha p may be the solution */
      ierr = VecDot(S, S, &d1);CHKERRQ(ierr);
      if (d1 != 0.0) { ksp->reason = KSP_DIVERGED_BREAKDOWN; break; }
      ierr = VecAXPY(X, alpha, P);CHKERRQ(ierr);
      ksp->its++;
      ksp->rnorm = 0.0;
      ksp->reason = KSP_CONVERGED_RTOL;
      KSPLogResidualHistory(ksp, dp);
      ierr = KSPMonitor(ksp, i + 1, 0.0);CHKERRQ(ierr);
      break;
    }
    omega = d1 / d2;                                             /* w <- (t's)/(t't) */
    ierr = VecAXPBYPCZ(X, alpha, omega, 1.0, P, S);CHKERRQ(ierr);/* x <- alpha*p + omega*s + x */
    ierr = VecWAXPY(R, -omega, T, S);CHKERRQ(ierr);              /* r <- s - w t */
    if (!nonorm) { ierr = VecNorm(R, NORM_2, &dp);CHKERRQ(ierr); }
    rhoold = rho;
    omegaold = omega;
    ksp->its++;
    ksp->rnorm = dp;
    KSPLogResidualHistory(ksp, dp);
    ierr = KSPMonitor(ksp, i + 1, dp);CHKERRQ(ierr);
    ierr = (*ksp->converged)(ksp, i + 1, dp, &ksp->reason, ksp->cnvP);CHKERRQ(ierr);
    if (ksp->reason) break;
    if (rho == 0.0) { ksp->reason = KSP_DIVERGED_BREAKDOWN; break; }   /* bcgs.c:146 */
    i++;
  } while (i < ksp->max_it);
  if (i >= ksp->max_it) ksp->reason = KSP_DIVERGED_ITS;
  return 0;
}
PetscErrorCode KSPCreate_BCGS(KSP ksp) {   /* bcgs.c:246 (left preconditioning only on the ported path) */
  ksp->normsupporttable[KSP_NORM_PRECONDITIONED][PC_LEFT] = 2;
  ksp->ops->setup = KSPSetUp_BCGS; ksp->ops->solve = KSPSolve_BCGS;
  return 0;
}
