/* HARNESS file: the Krylov drivers of the test harness (libpetscharness.so), for boxes without PETSc -- what an UNCHANGED PETSc program
 * asks of the plug-in's Vec / Mat types, call by call.  In a PETSc tree this file is NOT used: the reference's own
 * KSPSolve_CG / _GMRES / _BCGS take its place.  It is not part of the product library (libpetschipmi355x.so), links against nothing
 * device-specific and knows no fused kernel; the plug-in's own solvers are separate KSP types (host/kspfused.c).
 *
 * The CONTRACT of every solver below is the reference's sequence of Vec / Mat / PC calls and of convergence checkpoints, nothing else:
 * with the same sequence the iteration counts, the residual histories and (on one rank) the bits are the reference's.  The sequences
 * were read from
 *   CG            src/ksp/ksp/impls/cg/cg.c:92-286        (-ksp_cg_single_reduction: cg.c:116-122,166-169,200-203,263-270)
 *   Gropp CG      src/ksp/ksp/impls/cg/groppcg/groppcg.c:40-175
 *   pipelined CG  src/ksp/ksp/impls/cg/pipecg/pipecg.c:49-205
 *   GMRES(m)      src/ksp/ksp/impls/gmres/gmres.c:118-409, classical Gram-Schmidt src/ksp/ksp/impls/gmres/borthog2.c:35-119
 *   BiCGStab      src/ksp/ksp/impls/bcgs/bcgs.c:43-160
 * and are written here over a few shared pieces: one checkpoint routine (history, monitor, convergence test), one start-residual
 * routine for the CG family, index-addressed Hessenberg columns for GMRES. */
#include "petscimpl.h"

#define OK(call) do { PetscErrorCode e_ = (call); CHKERRQ(e_); } while (0)
#define FINITE(ksp, v) do { if (PetscIsInfOrNanScalar(v)) SETERRQ((ksp)->comm, PETSC_ERR_FP, "Infinite or not-a-number generated in dot product"); } while (0)

/* ------------------------------------------------------------------ shared pieces */

/* a convergence checkpoint: the norm goes to the history and the monitor, then the test decides ksp->reason */
static PetscErrorCode checkpoint(KSP ksp, PetscInt it, PetscReal rn) {
  ksp->rnorm = rn;
  KSPLogResidualHistory(ksp, rn);
  OK(KSPMonitor(ksp, it, rn));
  return (*ksp->converged)(ksp, it, rn, &ksp->reason, ksp->cnvP);
}

/* the CG family starts from res = rhs - A x, or from a copy of rhs when x is known to be zero */
static PetscErrorCode cg_family_start(KSP ksp, Mat A, Vec x, Vec rhs, Vec res) {
  if (ksp->guess_zero) return VecCopy(rhs, res);
  OK(KSP_MatMult(ksp, A, x, res));
  return VecAYPX(res, -1.0, rhs);
}

/* norm types the CG family accepts, with the priority of the preconditioned one (cg.c:439-442, groppcg.c:170-173, pipecg.c:199-202) */
static void cg_family_norm_table(KSP ksp, PetscInt preconditioned_priority) {
  static const KSPNormType all[] = {KSP_NORM_UNPRECONDITIONED, KSP_NORM_NATURAL, KSP_NORM_NONE};
  for (size_t k = 0; k < sizeof(all) / sizeof(all[0]); k++) ksp->normsupporttable[all[k]][PC_LEFT] = 1;
  ksp->normsupporttable[KSP_NORM_PRECONDITIONED][PC_LEFT] = preconditioned_priority;
}

static PetscBool option_is_on(const char *text) { return (PetscBool)(strcmp(text, "0") != 0 && strcmp(text, "false") != 0); }

/* ================================================================== CG
 * Work vectors: res, z = B res, dir; with -ksp_cg_single_reduction two more (Az = A z, Adir), and then A dir and dir' A dir come from
 * recurrences after the first iteration, and z'Az, z'res from ONE VecMDot. */
typedef struct { PetscBool one_reduction; } CgData;
#define CGD(ksp) ((CgData *)(ksp)->data)

static PetscErrorCode KSPSetUp_CG(KSP ksp) { return KSPDefaultGetWork(ksp, CGD(ksp)->one_reduction ? 5 : 3); }   /* cg.c:50-80, no eigenvalue work */
static PetscErrorCode KSPDestroy_CG(KSP ksp) { free(ksp->data); ksp->data = NULL; return 0; }
static PetscErrorCode KSPSetFromOptions_CG(KSP ksp) {   /* cg.c:330-345 */
  char text[16]; PetscBool given;
  OK(PetscOptionsGetString(ksp->prefix, "-ksp_cg_single_reduction", text, sizeof(text), &given));
  if (given) CGD(ksp)->one_reduction = option_is_on(text);
  return 0;
}

/* z <- B res and, in the one-reduction form, Az <- A z right behind it */
static PetscErrorCode cg_precondition(KSP ksp, PetscBool one_red, Mat A, Vec res, Vec z, Vec Az) {
  OK(KSP_PCApply(ksp, res, z));
  if (one_red) OK(KSP_MatMult(ksp, A, z, Az));
  return 0;
}
/* the products the next direction needs: z'res, and z'Az with it in the same reduction when the one-reduction form runs */
static PetscErrorCode cg_products(KSP ksp, PetscBool one_red, Vec z, Vec res, Vec Az, PetscScalar *zAz, PetscScalar *rz) {
  if (one_red) {
    Vec against[2] = {Az, res}; PetscScalar both[2];
    OK(VecMDot(z, 2, against, both));
    *zAz = both[0]; *rz = both[1];
  } else OK(VecTDot(z, res, rz));
  FINITE(ksp, *rz);
  return 0;
}
/* before the first iteration the same two products are two separate dots, the product A z between the norm and them */
static PetscErrorCode cg_first_products(KSP ksp, PetscBool one_red, Mat A, Vec z, Vec res, Vec Az, PetscScalar *zAz, PetscScalar *rz) {
  if (one_red) { OK(KSP_MatMult(ksp, A, z, Az)); OK(VecTDot(z, Az, zAz)); }
  OK(VecTDot(z, res, rz));
  FINITE(ksp, *rz);
  return 0;
}

static PetscErrorCode KSPSolve_CG(KSP ksp) {
  const PetscBool one_red = CGD(ksp)->one_reduction;
  const KSPNormType norm = ksp->normtype;
  const PetscBool norm_has_z = (PetscBool)(norm == KSP_NORM_PRECONDITIONED || norm == KSP_NORM_NATURAL);   /* the norm's own work leaves z = B res behind */
  Vec x = ksp->vec_sol, rhs = ksp->vec_rhs, res = ksp->work[0], z = ksp->work[1], dir = ksp->work[2];
  Vec Az = one_red ? ksp->work[3] : NULL, Adir = one_red ? ksp->work[4] : z;   /* without the extra vectors A dir borrows z (dead until the next B res) */
  Mat A = ksp->pc->mat;
  PetscScalar rz = 0.0, rz_last = 1.0, pAp = 0.0, pAp_last, zAz = 0.0, step;
  PetscReal rn = 0.0;
  PetscInt k = 0;

  ksp->its = 0;
  OK(cg_family_start(ksp, A, x, rhs, res));
  switch (norm) {                                                   /* iteration 0's checkpoint */
  case KSP_NORM_PRECONDITIONED: OK(KSP_PCApply(ksp, res, z)); OK(VecNorm(z, NORM_2, &rn)); break;
  case KSP_NORM_UNPRECONDITIONED: OK(VecNorm(res, NORM_2, &rn)); break;
  case KSP_NORM_NATURAL:                                             /* sqrt |z'res|: the recurrence's first product serves */
    OK(KSP_PCApply(ksp, res, z));
    OK(cg_first_products(ksp, one_red, A, z, res, Az, &zAz, &rz));
    rn = PetscSqrtReal(PetscAbsScalar(rz));
    break;
  case KSP_NORM_NONE: break;
  default: SETERRQ(ksp->comm, PETSC_ERR_SUP, "norm type %d", (int)norm);
  }
  OK(checkpoint(ksp, 0, rn));
  if (ksp->reason) return 0;
  if (!norm_has_z) OK(KSP_PCApply(ksp, res, z));
  if (norm != KSP_NORM_NATURAL) OK(cg_first_products(ksp, one_red, A, z, res, Az, &zAz, &rz));

  for (;;) {
    ksp->its = k + 1;
    if (rz == 0.0) { ksp->reason = KSP_CONVERGED_ATOL; break; }
    if (k > 0 && rz * rz_last < 0.0) { ksp->reason = KSP_DIVERGED_INDEFINITE_PC; break; }
    if (k == 0) OK(VecCopy(z, dir));
    else OK(VecAYPX(dir, rz / rz_last, z));                         /* dir <- z + (rz / rz_last) dir */
    pAp_last = pAp;
    if (!one_red || k == 0) {
      OK(KSP_MatMult(ksp, A, dir, Adir));
      OK(VecTDot(dir, Adir, &pAp));
    } else {                                                         /* the same two by recurrence */
      OK(VecAYPX(Adir, rz / rz_last, Az));
      pAp = zAz - rz * rz * pAp_last / (rz_last * rz_last);
    }
    rz_last = rz;
    FINITE(ksp, pAp);
    if (pAp == 0.0 || (k > 0 && pAp * pAp_last <= 0.0)) { ksp->reason = KSP_DIVERGED_INDEFINITE_MAT; break; }
    step = rz / pAp;
    OK(VecAXPY(x, step, dir));
    OK(VecAXPY(res, -step, Adir));
    /* this iteration's checkpoint; a norm that needs z produces it (and A z) now, the others after the test */
    if (norm_has_z) OK(cg_precondition(ksp, one_red, A, res, z, Az));
    if (norm == KSP_NORM_PRECONDITIONED) OK(VecNorm(z, NORM_2, &rn));
    else if (norm == KSP_NORM_UNPRECONDITIONED) OK(VecNorm(res, NORM_2, &rn));
    else if (norm == KSP_NORM_NATURAL) { OK(cg_products(ksp, one_red, z, res, Az, &zAz, &rz)); rn = PetscSqrtReal(PetscAbsScalar(rz)); }
    else rn = 0.0;
    OK(checkpoint(ksp, k + 1, rn));
    if (ksp->reason) break;
    if (!norm_has_z) OK(cg_precondition(ksp, one_red, A, res, z, Az));
    if (norm != KSP_NORM_NATURAL) OK(cg_products(ksp, one_red, z, res, Az, &zAz, &rz));
    if (++k >= ksp->max_it) break;
  }
  if (k >= ksp->max_it) ksp->reason = KSP_DIVERGED_ITS;
  return 0;
}

PetscErrorCode KSPCreate_CG(KSP ksp) {
  CgData *data;
  OK(PetscMalloc(sizeof(*data), &data));
  data->one_reduction = PETSC_FALSE;
  ksp->data = data;
  cg_family_norm_table(ksp, 2);
  ksp->ops->setup = KSPSetUp_CG;
  ksp->ops->solve = KSPSolve_CG;
  ksp->ops->setfromoptions = KSPSetFromOptions_CG;
  ksp->ops->destroy = KSPDestroy_CG;
  return 0;
}

/* ================================================================== Gropp's CG
 * Two reductions per iteration, both split-phase (VecDotBegin / PetscCommSplitReductionBegin / VecDotEnd) with independent work between
 * the halves: dir'Adir travels while B Adir is applied, {the norm, res'z} while A z is formed -- on several GPUs the all-reduce is
 * on the halo stream while the compute stream runs that work.  The preconditioned residual and A z are updated by recurrence, which
 * costs three more vectors than CG (six in all). */
static PetscErrorCode KSPSetUp_GROPPCG(KSP ksp) { return KSPDefaultGetWork(ksp, 6); }

static PetscErrorCode KSPSolve_GROPPCG(KSP ksp) {
  const KSPNormType norm = ksp->normtype;
  Vec x = ksp->vec_sol, rhs = ksp->vec_rhs;
  Vec res = ksp->work[0], dir = ksp->work[1], Adir = ksp->work[2], BAdir = ksp->work[3], z = ksp->work[4], Az = ksp->work[5];
  Vec normed = norm == KSP_NORM_UNPRECONDITIONED ? res : norm == KSP_NORM_PRECONDITIONED ? z : NULL;   /* whose 2-norm the test sees, if any vector's */
  Mat A = ksp->pc->mat;
  PetscScalar rz, rz_new, pAp, step;
  PetscReal rn = 0.0;
  PetscInt k = 0;

  ksp->its = 0;
  OK(cg_family_start(ksp, A, x, rhs, res));
  OK(KSP_PCApply(ksp, res, z));
  OK(VecCopy(z, dir));
  OK(VecDotBegin(res, z, &rz));                                     /* res'z in flight over Adir <- A dir */
  OK(PetscCommSplitReductionBegin(res->comm));
  OK(KSP_MatMult(ksp, A, dir, Adir));
  OK(VecDotEnd(res, z, &rz));
  if (norm != KSP_NORM_PRECONDITIONED && norm != KSP_NORM_UNPRECONDITIONED && norm != KSP_NORM_NATURAL && norm != KSP_NORM_NONE) SETERRQ(ksp->comm, PETSC_ERR_SUP, "norm type %d", (int)norm);
  if (normed) OK(VecNorm(normed, NORM_2, &rn));
  else if (norm == KSP_NORM_NATURAL) { FINITE(ksp, rz); rn = PetscSqrtReal(PetscAbsScalar(rz)); }
  OK(checkpoint(ksp, 0, rn));
  if (ksp->reason) return 0;

  do {
    ksp->its = ++k;
    OK(VecDotBegin(dir, Adir, &pAp));                               /* dir'Adir in flight over BAdir <- B Adir */
    OK(PetscCommSplitReductionBegin(dir->comm));
    OK(KSP_PCApply(ksp, Adir, BAdir));
    OK(VecDotEnd(dir, Adir, &pAp));
    step = rz / pAp;
    OK(VecAXPY(x, step, dir));
    OK(VecAXPY(res, -step, Adir));
    OK(VecAXPY(z, -step, BAdir));                                   /* z stays B res without another application */
    if (normed) OK(VecNormBegin(normed, NORM_2, &rn));
    OK(VecDotBegin(res, z, &rz_new));                               /* {norm, res'z} in flight over Az <- A z */
    OK(PetscCommSplitReductionBegin(res->comm));
    OK(KSP_MatMult(ksp, A, z, Az));
    if (normed) OK(VecNormEnd(normed, NORM_2, &rn));
    OK(VecDotEnd(res, z, &rz_new));
    if (norm == KSP_NORM_NATURAL) { FINITE(ksp, rz_new); rn = PetscSqrtReal(PetscAbsScalar(rz_new)); }
    else if (norm == KSP_NORM_NONE) rn = 0.0;
    OK(checkpoint(ksp, k, rn));
    if (ksp->reason) break;
    OK(VecAYPX(dir, rz_new / rz, z));
    OK(VecAYPX(Adir, rz_new / rz, Az));                             /* A dir by the same recurrence */
    rz = rz_new;
  } while (k < ksp->max_it);
  if (k >= ksp->max_it && !ksp->reason) ksp->reason = KSP_DIVERGED_ITS;
  return 0;
}
PetscErrorCode KSPCreate_GROPPCG(KSP ksp) {
  cg_family_norm_table(ksp, 1);
  ksp->ops->setup = KSPSetUp_GROPPCG;
  ksp->ops->solve = KSPSolve_GROPPCG;
  return 0;
}

/* ================================================================== pipelined CG (Ghysels & Vanroose)
 * ONE split-phase reduction per iteration -- {a norm or res'u, w'u} -- in flight over m <- B w and n <- A m; nine work vectors, four
 * of them recurrences of products.  Restated as this snapshot of the reference has it, including which quantity each iteration
 * reduces (pipecg.c:124-131,138-145): with the preconditioned or the unpreconditioned norm res'u is reduced in iteration 0 only and
 * the direction recurrence then runs with ratio 1; the natural norm and KSP_NORM_NONE refresh it every iteration and are the forms
 * that converge like KSPCG. */
static PetscErrorCode KSPSetUp_PIPECG(KSP ksp) { return KSPDefaultGetWork(ksp, 9); }

enum { PIPE_NOTHING, PIPE_NORM, PIPE_RU };                          /* what rides with w'u in an iteration's reduction */

static PetscErrorCode KSPSolve_PIPECG(KSP ksp) {
  const KSPNormType norm = ksp->normtype;
  Vec x = ksp->vec_sol, rhs = ksp->vec_rhs;
  Vec m = ksp->work[0], ABs_rec = ksp->work[1], dir = ksp->work[2], n = ksp->work[3], w = ksp->work[4];
  Vec Bs_rec = ksp->work[5], u = ksp->work[6], res = ksp->work[7], s = ksp->work[8];
  Vec normed = norm == KSP_NORM_UNPRECONDITIONED ? res : norm == KSP_NORM_PRECONDITIONED ? u : NULL;
  Mat A = ksp->pc->mat;
  PetscScalar step = 0.0, ratio, ru = 0.0, ru_last = 0.0, wu = 0.0;
  PetscReal rn = 0.0;
  PetscInt k = 0;

  ksp->its = 0;
  OK(cg_family_start(ksp, A, x, rhs, res));
  OK(KSP_PCApply(ksp, res, u));
  /* iteration 0's norm, its reduction in flight over w <- A u */
  if (norm != KSP_NORM_PRECONDITIONED && norm != KSP_NORM_UNPRECONDITIONED && norm != KSP_NORM_NATURAL && norm != KSP_NORM_NONE) SETERRQ(ksp->comm, PETSC_ERR_SUP, "norm type %d", (int)norm);
  if (normed) { OK(VecNormBegin(normed, NORM_2, &rn)); OK(PetscCommSplitReductionBegin(normed->comm)); }
  else if (norm == KSP_NORM_NATURAL) { OK(VecDotBegin(res, u, &ru)); OK(PetscCommSplitReductionBegin(res->comm)); }
  OK(KSP_MatMult(ksp, A, u, w));
  if (normed) OK(VecNormEnd(normed, NORM_2, &rn));
  else if (norm == KSP_NORM_NATURAL) { OK(VecDotEnd(res, u, &ru)); FINITE(ksp, ru); rn = PetscSqrtReal(PetscAbsScalar(ru)); }
  OK(checkpoint(ksp, 0, rn));
  if (ksp->reason) return 0;

  do {
    int rides;
    if (k > 0 && normed) rides = PIPE_NORM;
    else if (k == 0 && norm == KSP_NORM_NATURAL) rides = PIPE_NOTHING;   /* res'u is iteration 0's norm already */
    else rides = PIPE_RU;
    if (rides == PIPE_NORM) OK(VecNormBegin(normed, NORM_2, &rn));
    else if (rides == PIPE_RU) OK(VecDotBegin(res, u, &ru));
    OK(VecDotBegin(w, u, &wu));
    OK(PetscCommSplitReductionBegin(res->comm));
    OK(KSP_PCApply(ksp, w, m));                                      /* the work the reduction hides behind */
    OK(KSP_MatMult(ksp, A, m, n));
    if (rides == PIPE_NORM) OK(VecNormEnd(normed, NORM_2, &rn));
    else if (rides == PIPE_RU) OK(VecDotEnd(res, u, &ru));
    OK(VecDotEnd(w, u, &wu));
    if (k > 0) {
      if (norm == KSP_NORM_NATURAL) rn = PetscSqrtReal(PetscAbsScalar(ru));
      else if (norm == KSP_NORM_NONE) rn = 0.0;
      OK(checkpoint(ksp, k, rn));
      if (ksp->reason) break;
      ratio = ru / ru_last;
      step = ru / (wu - ratio / step * ru);
      OK(VecAYPX(ABs_rec, ratio, n));
      OK(VecAYPX(Bs_rec, ratio, m));
      OK(VecAYPX(dir, ratio, u));
      OK(VecAYPX(s, ratio, w));
    } else {
      step = ru / wu;
      OK(VecCopy(n, ABs_rec));
      OK(VecCopy(m, Bs_rec));
      OK(VecCopy(u, dir));
      OK(VecCopy(w, s));
    }
    OK(VecAXPY(x, step, dir));
    OK(VecAXPY(u, -step, Bs_rec));
    OK(VecAXPY(w, -step, ABs_rec));
    OK(VecAXPY(res, -step, s));
    ru_last = ru;
    ksp->its = ++k;
  } while (k < ksp->max_it);
  if (k >= ksp->max_it) ksp->reason = KSP_DIVERGED_ITS;
  return 0;
}
PetscErrorCode KSPCreate_PIPECG(KSP ksp) {
  cg_family_norm_table(ksp, 1);
  ksp->ops->setup = KSPSetUp_PIPECG;
  ksp->ops->solve = KSPSolve_PIPECG;
  return 0;
}

/* ================================================================== GMRES(m), left or right preconditioning, classical Gram-Schmidt
 * Host state: the rotated Hessenberg matrix, one column per Arnoldi step, column k holding k + 2 numbers at H[k * ld ...]; the Givens
 * pairs; the rotated right-hand side.  Device state: a pool of restart + 4 vectors -- the update, a scratch for the two-step operator,
 * and the basis. */
typedef struct {
  PetscInt restart, ld;
  PetscReal happy_tol;
  KSPGMRESCGSRefinementType refinement;
  PetscScalar *H, *giv_c, *giv_s, *rot_rhs, *coef, *y;
  Vec *pool;
  PetscInt npool;
} GmresData;
#define GMD(ksp) ((GmresData *)(ksp)->data)
#define UPDATE(g) ((g)->pool[0])
#define SCRATCH(g) ((g)->pool[1])
#define BASIS(g) ((g)->pool + 2)

static PetscErrorCode KSPGMRESSetRestart_GMRES(KSP ksp, PetscInt restart) {   /* gmres.c:752-770 */
  if (restart < 1) SETERRQ(ksp->comm, PETSC_ERR_ARG_OUTOFRANGE, "Restart must be positive");
  if (ksp->setupcalled) SETERRQ(ksp->comm, PETSC_ERR_ORDER, "Must call KSPGMRESSetRestart() before KSPSetUp()");
  GMD(ksp)->restart = restart;
  return 0;
}
static PetscErrorCode KSPGMRESSetCGSRefinementType_GMRES(KSP ksp, KSPGMRESCGSRefinementType type) { GMD(ksp)->refinement = type; return 0; }

static PetscErrorCode KSPSetFromOptions_GMRES(KSP ksp) {
  static const struct { const char *name; KSPGMRESCGSRefinementType type; } known[] = {
    {"refine_never", KSP_GMRES_CGS_REFINE_NEVER}, {"refine_ifneeded", KSP_GMRES_CGS_REFINE_IFNEEDED}, {"refine_always", KSP_GMRES_CGS_REFINE_ALWAYS}};
  PetscInt m; PetscBool given; char text[64];
  OK(PetscOptionsGetInt(ksp->prefix, "-ksp_gmres_restart", &m, &given));
  if (given) OK(KSPGMRESSetRestart(ksp, m));
  OK(PetscOptionsGetString(ksp->prefix, "-ksp_gmres_cgs_refinement_type", text, sizeof(text), &given));
  if (given) {
    size_t k = 0;
    while (k < 3 && strcmp(text, known[k].name)) k++;
    if (k == 3) SETERRQ(ksp->comm, PETSC_ERR_ARG_UNKNOWN_TYPE, "Unknown refinement type %s", text);
    GMD(ksp)->refinement = known[k].type;
  }
  return 0;
}

static PetscErrorCode KSPSetUp_GMRES(KSP ksp) {   /* gmres.c:36-96; the whole basis is allocated up front */
  GmresData *g = GMD(ksp);
  const size_t m = (size_t)g->restart;
  PetscScalar *host = (PetscScalar *)calloc((m + 2) * (m + 1) + 5 * (m + 2), sizeof(PetscScalar));   /* one block: H, then five vectors of m + 2 */
  if (!host) SETERRQ(ksp->comm, PETSC_ERR_MEM, "out of memory");
  g->ld = g->restart + 2;
  g->H = host; host += (m + 2) * (m + 1);
  g->giv_c = host; host += m + 2;
  g->giv_s = host; host += m + 2;
  g->rot_rhs = host; host += m + 2;
  g->coef = host; host += m + 2;
  g->y = host;
  g->npool = g->restart + 4;
  if (ksp->vec_sol) return VecDuplicateVecs(ksp->vec_sol, g->npool, &g->pool);
  Vec like;
  OK(MatGetVecs(ksp->pc->mat, &like, NULL));
  OK(VecDuplicateVecs(like, g->npool, &g->pool));
  return VecDestroy(&like);
}

/* classical Gram-Schmidt of basis vector k + 1 against 0..k (borthog2.c:35-119): all k + 1 dots in one VecMDot, all subtractions in one
 * VecMAXPY; a second pass always, never, or when the first one cancelled too much (Daniel et al.'s 1 / sqrt 2 criterion in the
 * reference's constant).  Column k of H accumulates the coefficients. */
static PetscErrorCode gmres_orthogonalize(KSP ksp, PetscInt k) {
  GmresData *g = GMD(ksp);
  PetscScalar *h = g->H + (size_t)k * (size_t)g->ld, *coef = g->coef;
  Vec *V = BASIS(g), fresh = V[k + 1];
  PetscBool again = (PetscBool)(g->refinement == KSP_GMRES_CGS_REFINE_ALWAYS);
  for (PetscInt j = 0; j <= k; j++) h[j] = 0.0;
  for (int pass = 0; pass < 2; pass++) {
    OK(VecMDot(fresh, k + 1, V, coef));
    for (PetscInt j = 0; j <= k; j++) coef[j] = -coef[j];
    OK(VecMAXPY(fresh, k + 1, coef, V));
    for (PetscInt j = 0; j <= k; j++) h[j] -= coef[j];
    if (pass == 0 && g->refinement == KSP_GMRES_CGS_REFINE_IFNEEDED) {
      PetscReal removed = 0.0, left;
      for (PetscInt j = 0; j <= k; j++) removed += coef[j] * coef[j];
      removed = PetscSqrtReal(removed);
      OK(VecNorm(fresh, NORM_2, &left));
      if (left < 1.0286 * removed) again = PETSC_TRUE;
    }
    if (!again) break;
  }
  return 0;
}

/* the earlier rotations applied to column k of H, then (unless the happy breakdown ended the cycle) the new one that clears H(k+1, k)
 * (KSPGMRESUpdateHessenberg, gmres.c:360-409); *rn: the residual norm the rotated right-hand side now implies */
static PetscErrorCode gmres_rotate(KSP ksp, PetscInt k, PetscBool happy, PetscReal *rn) {
  GmresData *g = GMD(ksp);
  PetscScalar *h = g->H + (size_t)k * (size_t)g->ld, *c = g->giv_c, *s = g->giv_s, *rhs = g->rot_rhs;
  for (PetscInt j = 0; j < k; j++) {
    const PetscScalar upper = h[j];
    h[j] = c[j] * upper + s[j] * h[j + 1];
    h[j + 1] = c[j] * h[j + 1] - (s[j] * upper);
  }
  if (happy) { *rn = 0.0; return 0; }
  const PetscScalar len = sqrt(h[k] * h[k] + h[k + 1] * h[k + 1]);
  if (len == 0.0) { ksp->reason = KSP_DIVERGED_NULL; return 0; }
  c[k] = h[k] / len;
  s[k] = h[k + 1] / len;
  rhs[k + 1] = -(s[k] * rhs[k]);
  rhs[k] = c[k] * rhs[k];
  h[k] = c[k] * h[k] + s[k] * h[k + 1];
  *rn = PetscAbsScalar(rhs[k + 1]);
  return 0;
}

/* x <- x + [B] V y with R y = rotated right-hand side, R the leading (last + 1)^2 triangle of H (KSPGMRESBuildSoln, gmres.c:309-354) */
static PetscErrorCode gmres_update_solution(KSP ksp, PetscInt last) {
  GmresData *g = GMD(ksp);
  const size_t ld = (size_t)g->ld;
  PetscScalar *y = g->y;
  if (last < 0) return 0;
  for (PetscInt r = last; r >= 0; r--) {
    PetscScalar acc = g->rot_rhs[r];
    for (PetscInt j = r + 1; j <= last; j++) acc = acc - g->H[(size_t)j * ld + r] * y[j];
    if (g->H[(size_t)r * ld + r] == 0.0) { ksp->reason = KSP_DIVERGED_BREAKDOWN; return 0; }
    y[r] = acc / g->H[(size_t)r * ld + r];
  }
  OK(VecSet(UPDATE(g), 0.0));
  OK(VecMAXPY(UPDATE(g), last + 1, y, BASIS(g)));
  if (ksp->pc_side == PC_RIGHT) {                                   /* KSPUnwindPreconditioner */
    OK(KSP_PCApply(ksp, UPDATE(g), SCRATCH(g)));
    OK(VecCopy(SCRATCH(g), UPDATE(g)));
  }
  return VecAXPY(ksp->vec_sol, 1.0, UPDATE(g));
}

/* one restart cycle from the residual in basis vector 0 (KSPGMRESCycle, gmres.c:118-209).  The history and the monitor see a step's
 * norm when the NEXT step starts (or when the cycle ends for good), the convergence test sees it at once. */
static PetscErrorCode gmres_cycle(KSP ksp, PetscInt *steps) {
  GmresData *g = GMD(ksp);
  Vec *V = BASIS(g);
  PetscReal rn, beta, hnext, happy_bound;
  PetscBool happy = PETSC_FALSE;
  PetscInt k = 0;

  *steps = 0;
  OK(VecNormalize(V[0], &beta));
  g->rot_rhs[0] = beta;
  ksp->rnorm = rn = beta;
  KSPLogResidualHistory(ksp, rn);
  OK(KSPMonitor(ksp, ksp->its, rn));
  if (!rn) { ksp->reason = KSP_CONVERGED_ATOL; return 0; }
  OK((*ksp->converged)(ksp, ksp->its, rn, &ksp->reason, ksp->cnvP));
  while (!ksp->reason && k < g->restart && ksp->its < ksp->max_it) {
    if (k) { KSPLogResidualHistory(ksp, rn); OK(KSPMonitor(ksp, ksp->its, rn)); }
    OK(KSP_PCApplyBAorAB(ksp, V[k], V[k + 1], SCRATCH(g)));
    OK(gmres_orthogonalize(ksp, k));
    OK(VecNormalize(V[k + 1], &hnext));
    g->H[(size_t)k * (size_t)g->ld + k + 1] = hnext;
    happy_bound = PetscAbsScalar(hnext / g->rot_rhs[k]);          /* the happy breakdown: the new vector vanished relative to the residual */
    if (happy_bound > g->happy_tol) happy_bound = g->happy_tol;
    if (hnext < happy_bound) happy = PETSC_TRUE;
    OK(gmres_rotate(ksp, k, happy, &rn));
    k++;
    ksp->its++;
    ksp->rnorm = rn;
    if (ksp->reason) break;
    OK((*ksp->converged)(ksp, ksp->its, rn, &ksp->reason, ksp->cnvP));
    if (happy) {
      if (!ksp->reason) SETERRQ(ksp->comm, PETSC_ERR_PLIB, "You reached the happy break down, but convergence was not indicated. Residual norm = %g", rn);
      break;
    }
  }
  if (k && (ksp->reason || ksp->its >= ksp->max_it)) { KSPLogResidualHistory(ksp, rn); OK(KSPMonitor(ksp, ksp->its, rn)); }
  *steps = k;
  return gmres_update_solution(ksp, k - 1);
}

static PetscErrorCode KSPSolve_GMRES(KSP ksp) {   /* gmres.c:213-243: cycles until a reason turns up or the iteration budget is spent */
  GmresData *g = GMD(ksp);
  const PetscBool caller_guess_zero = ksp->guess_zero;
  PetscInt total = 0, steps = 0;
  ksp->its = 0;
  ksp->reason = KSP_CONVERGED_ITERATING;
  while (!ksp->reason) {
    OK(KSPInitialResidual(ksp, ksp->vec_sol, UPDATE(g), SCRATCH(g), BASIS(g)[0], ksp->vec_rhs));
    OK(gmres_cycle(ksp, &steps));
    total += steps;
    if (total >= ksp->max_it) { if (!ksp->reason) ksp->reason = KSP_DIVERGED_ITS; break; }
    ksp->guess_zero = PETSC_FALSE;                                  /* x holds a cycle's update from here on */
  }
  ksp->guess_zero = caller_guess_zero;
  return 0;
}

static PetscErrorCode KSPDestroy_GMRES(KSP ksp) {
  GmresData *g = GMD(ksp);
  if (!g) return 0;
  free(g->H);                                                        /* the one host block */
  if (g->pool) OK(VecDestroyVecs(g->npool, &g->pool));
  free(g);
  ksp->data = NULL;
  (void)PetscObjectComposeFunction((PetscObject)ksp, "KSPGMRESSetRestart_C", "", (PetscVoidFunction)NULL);   /* gmres.c:288-291 */
  (void)PetscObjectComposeFunction((PetscObject)ksp, "KSPGMRESSetCGSRefinementType_C", "", (PetscVoidFunction)NULL);
  return 0;
}

PetscErrorCode KSPCreate_GMRES(KSP ksp) {   /* defaults of the reference's constructor: restart 30, happy-breakdown tolerance 1e-30, no refinement */
  GmresData *g;
  OK(PetscMalloc(sizeof(*g), &g));
  memset(g, 0, sizeof(*g));
  g->restart = 30;
  g->happy_tol = 1.0e-30;
  g->refinement = KSP_GMRES_CGS_REFINE_NEVER;
  ksp->data = g;
  ksp->normsupporttable[KSP_NORM_PRECONDITIONED][PC_LEFT] = 2;      /* gmres.c:909-910 */
  ksp->normsupporttable[KSP_NORM_UNPRECONDITIONED][PC_RIGHT] = 1;
  ksp->ops->setup = KSPSetUp_GMRES;
  ksp->ops->solve = KSPSolve_GMRES;
  ksp->ops->destroy = KSPDestroy_GMRES;
  ksp->ops->setfromoptions = KSPSetFromOptions_GMRES;
  OK(PetscObjectComposeFunction((PetscObject)ksp, "KSPGMRESSetRestart_C", "KSPGMRESSetRestart_GMRES", (PetscVoidFunction)KSPGMRESSetRestart_GMRES));   /* gmres.c:931-942 */
  OK(PetscObjectComposeFunction((PetscObject)ksp, "KSPGMRESSetCGSRefinementType_C", "KSPGMRESSetCGSRefinementType_GMRES", (PetscVoidFunction)KSPGMRESSetCGSRefinementType_GMRES));
  return 0;
}

/* ================================================================== BiCGStab (left preconditioning)
 * Six work vectors: res, the shadow residual, v = K dir, t = K s, s, dir, with K the preconditioned operator applied in one
 * KSP_PCApplyBAorAB.  KSP_NORM_NONE (smoother use, bcgs.c:76,131) skips the two norms and nothing else. */
static PetscErrorCode KSPSetUp_BCGS(KSP ksp) { return KSPDefaultGetWork(ksp, 6); }   /* bcgs.c:13 */

static PetscErrorCode KSPSolve_BCGS(KSP ksp) {
  const PetscBool want_norm = (PetscBool)(ksp->normtype != KSP_NORM_NONE);
  Vec x = ksp->vec_sol, rhs = ksp->vec_rhs;
  Vec res = ksp->work[0], shadow = ksp->work[1], v = ksp->work[2], t = ksp->work[3], s = ksp->work[4], dir = ksp->work[5];
  PetscScalar rho = 0.0, rho_last = 1.0, alpha = 1.0, omega, omega_last = 1.0, beta, dot;
  PetscReal rn = 0.0, tt;
  PetscInt k = 0;

  OK(KSPInitialResidual(ksp, x, v, t, res, rhs));
  if (want_norm) OK(VecNorm(res, NORM_2, &rn));
  ksp->its = 0;
  OK(checkpoint(ksp, 0, rn));
  if (ksp->reason) return 0;
  OK(VecCopy(res, shadow));
  OK(VecSet(dir, 0.0));
  OK(VecSet(v, 0.0));
  do {
    OK(VecDot(res, shadow, &rho));
    beta = (rho / rho_last) * (alpha / omega_last);
    OK(VecAXPBYPCZ(dir, 1.0, -omega_last * beta, beta, res, v));    /* dir <- res + beta (dir - omega v) */
    OK(KSP_PCApplyBAorAB(ksp, dir, v, t));
    OK(VecDot(v, shadow, &dot));
    if (dot == 0.0) SETERRQ(ksp->comm, PETSC_ERR_PLIB, "Divide by zero");
    alpha = rho / dot;
    OK(VecWAXPY(s, -alpha, v, res));
    OK(KSP_PCApplyBAorAB(ksp, s, t, res));                          /* (res is free: s replaced it) */
    OK(VecDotNorm2(s, t, &dot, &tt));                               /* s't and t't in one sweep */
    if (tt == 0.0) {                                                 /* K s vanished: with s = 0 as well, x + alpha dir is the solution */
      OK(VecDot(s, s, &dot));
      if (dot != 0.0) { ksp->reason = KSP_DIVERGED_BREAKDOWN; break; }
      OK(VecAXPY(x, alpha, dir));
      ksp->its++;
      ksp->rnorm = 0.0;
      ksp->reason = KSP_CONVERGED_RTOL;
      KSPLogResidualHistory(ksp, rn);                               /* (the reference logs the previous norm here, bcgs.c:118) */
      OK(KSPMonitor(ksp, k + 1, 0.0));
      break;
    }
    omega = dot / tt;
    OK(VecAXPBYPCZ(x, alpha, omega, 1.0, dir, s));
    OK(VecWAXPY(res, -omega, t, s));
    if (want_norm) OK(VecNorm(res, NORM_2, &rn));
    rho_last = rho;
    omega_last = omega;
    ksp->its++;
    OK(checkpoint(ksp, k + 1, rn));
    if (ksp->reason) break;
    if (rho == 0.0) { ksp->reason = KSP_DIVERGED_BREAKDOWN; break; }   /* bcgs.c:146 */
  } while (++k < ksp->max_it);
  if (k >= ksp->max_it) ksp->reason = KSP_DIVERGED_ITS;
  return 0;
}
PetscErrorCode KSPCreate_BCGS(KSP ksp) {   /* bcgs.c:246 (left preconditioning only on this path) */
  ksp->normsupporttable[KSP_NORM_PRECONDITIONED][PC_LEFT] = 2;
  ksp->ops->setup = KSPSetUp_BCGS;
  ksp->ops->solve = KSPSolve_BCGS;
  return 0;
}
