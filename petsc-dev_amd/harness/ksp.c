/* KSP driver: the host-side control flow the reference runs unchanged over any Vec/Mat type
 * (src/ksp/ksp/interface/itfunc.c:175,335; itcreate.c:640-700; iterativ.c:702,911-986; itres.c:39;
 * macros include/petsc-private/kspimpl.h:181-188).  Only host scalars live here; every vector
 * operation dispatches through the Vec/Mat function tables to the HIP kernels. */
#include "petscimpl.h"
#include <stdio.h>
static const char *ksp_reason_name(KSPConvergedReason r);

#define KSPValid(k) do { if (!(k)) SETERRQ(0, PETSC_ERR_ARG_NULL, "Null KSP"); } while (0)

PetscErrorCode KSPCreate(PetscComm comm, KSP *inksp) {   /* itcreate.c:640-700 */
  PetscErrorCode ierr;
  KSP ksp;
  ierr = PetscMiniInitialize();CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(*ksp), &ksp);CHKERRQ(ierr);
  memset(ksp, 0, sizeof(*ksp));
  ksp->comm = comm;
  ksp->max_it = 10000; ksp->pc_side = PC_SIDE_DEFAULT; ksp->rtol = 1.e-5; ksp->abstol = 1.e-50; ksp->divtol = 1.e4;
  ksp->chknorm = -1; ksp->normtype = KSP_NORM_DEFAULT; ksp->rnorm = 0.0; ksp->its = 0; ksp->guess_zero = PETSC_TRUE;
  ksp->reason = KSP_CONVERGED_ITERATING;
  ksp->converged = KSPDefaultConverged;
  *inksp = ksp;
  return 0;
}

/* KSPSetSupportedNorm / KSPNormSupportTableReset_Private, itcreate.c:309-372: every KSPCreate_XXX declares the (norm, side)
 * pairs it can run with a preference; KSP_NORM_NONE is supported by every type unless it says otherwise */
PetscErrorCode KSPSetSupportedNorm(KSP ksp, KSPNormType normtype, PCSide pcside, PetscInt priority) {
  KSPValid(ksp);
  ksp->normsupporttable[normtype][pcside] = priority;
  return 0;
}
static void ksp_norm_table_reset(KSP ksp) {
  memset(ksp->normsupporttable, 0, sizeof(ksp->normsupporttable));
  ksp->normsupporttable[KSP_NORM_NONE][PC_LEFT] = 1;
  ksp->normsupporttable[KSP_NORM_NONE][PC_RIGHT] = 1;
}
static const char *const norm_names[] = {"NONE", "PRECONDITIONED", "UNPRECONDITIONED", "NATURAL"}, *const side_names[] = {"LEFT", "RIGHT", "SYMMETRIC"};
/* KSPSetUpNorms_Private, itcreate.c:341-372: the best supported pair among those the user's choices leave open */
static PetscErrorCode ksp_setup_norms(KSP ksp) {
  PetscInt best = 0, ibest = 0, jbest = 0;
  for (PetscInt i = 0; i < KSP_NORM_MAX; i++)
    for (PetscInt j = 0; j < PC_SIDE_MAX; j++) {
      if (ksp->normtype != KSP_NORM_DEFAULT && (PetscInt)ksp->normtype != i) continue;
      if (ksp->pc_side != PC_SIDE_DEFAULT && (PetscInt)ksp->pc_side != j) continue;
      if (ksp->normsupporttable[i][j] <= best) continue;
      if (ksp->normtype == KSP_NORM_DEFAULT && i == KSP_NORM_NONE && ksp->normsupporttable[i][j] <= 1) continue;   /* no norm is never the silent default (preonly asks for it with 2) */
      best = ksp->normsupporttable[i][j]; ibest = i; jbest = j;
    }
  if (best < 1) {
    if (ksp->normtype == KSP_NORM_DEFAULT && ksp->pc_side == PC_SIDE_DEFAULT) SETERRQ(ksp->comm, PETSC_ERR_PLIB, "The %s KSP implementation did not call KSPSetSupportedNorm()", ksp->type_name);
    if (ksp->normtype == KSP_NORM_DEFAULT) SETERRQ(ksp->comm, PETSC_ERR_SUP, "KSP %s does not support %s", ksp->type_name, side_names[ksp->pc_side]);
    if (ksp->pc_side == PC_SIDE_DEFAULT) SETERRQ(ksp->comm, PETSC_ERR_SUP, "KSP %s does not support %s", ksp->type_name, norm_names[ksp->normtype]);
    SETERRQ(ksp->comm, PETSC_ERR_SUP, "KSP %s does not support %s with %s", ksp->type_name, norm_names[ksp->normtype], side_names[ksp->pc_side]);
  }
  ksp->normtype = (KSPNormType)ibest;
  ksp->pc_side = (PCSide)jbest;
  return 0;
}

#define MAXKSPTYPES 16
static struct { char name[32]; PetscErrorCode (*fn)(KSP); } ksp_types[MAXKSPTYPES + 1];
static int n_ksp_types = 0;
PetscErrorCode KSPRegister(const char name[], const char path[], const char fname[], PetscErrorCode (*fn)(KSP)) {   /* src/ksp/ksp/interface/itregis.c */
  (void)path; (void)fname;
  for (int i = 0; i < n_ksp_types; i++) if (!strcmp(ksp_types[i].name, name)) { ksp_types[i].fn = fn; return 0; }
  if (n_ksp_types >= MAXKSPTYPES) SETERRQ(0, PETSC_ERR_PLIB, "KSP type table full");
  snprintf(ksp_types[n_ksp_types].name, 32, "%s", name);
  ksp_types[n_ksp_types++].fn = fn;
  return 0;
}

PetscErrorCode KSPSetType(KSP ksp, KSPType type) {
  PetscErrorCode ierr;
  KSPValid(ksp);
  if (!strcmp(ksp->type_name, type)) return 0;
  for (int i = 0; i < n_ksp_types; i++) {
    if (!strcmp(ksp_types[i].name, type)) {
      if (ksp->ops->destroy) { ierr = (*ksp->ops->destroy)(ksp);CHKERRQ(ierr); }
      if (ksp->work) { ierr = VecDestroyVecs(ksp->nwork, &ksp->work);CHKERRQ(ierr); ksp->nwork = 0; }
      memset(ksp->ops, 0, sizeof(ksp->ops));
      ksp->data = NULL; ksp->setupcalled = 0;
      ksp_norm_table_reset(ksp);
      snprintf(ksp->type_name, sizeof(ksp->type_name), "%s", type);
      ierr = (*ksp_types[i].fn)(ksp);CHKERRQ(ierr);
      return 0;
    }
  }
  SETERRQ(ksp->comm, PETSC_ERR_ARG_UNKNOWN_TYPE, "Unable to find requested KSP type %s", type);
}
PetscErrorCode KSPGetType(KSP ksp, KSPType *type) { KSPValid(ksp); *type = ksp->type_name; return 0; }
PetscErrorCode KSPGetPC(KSP ksp, PC *pc) {
  KSPValid(ksp);
  if (!ksp->pc) { PetscErrorCode ierr = PCCreate(ksp->comm, &ksp->pc);CHKERRQ(ierr); snprintf(ksp->pc->prefix, sizeof(ksp->pc->prefix), "%s", ksp->prefix); }
  *pc = ksp->pc;
  return 0;
}
PetscErrorCode KSPSetOperators(KSP ksp, Mat Amat, Mat Pmat, MatStructure flag) {
  PetscErrorCode ierr; PC pc;
  ierr = KSPGetPC(ksp, &pc);CHKERRQ(ierr);
  ierr = PCSetOperators(pc, Amat, Pmat, flag);CHKERRQ(ierr);
  if (ksp->setupcalled > 1) ksp->setupcalled = 1;   /* so that next solve call will call PCSetUp() on new matrix */
  return 0;
}
PetscErrorCode KSPSetTolerances(KSP ksp, PetscReal rtol, PetscReal abstol, PetscReal dtol, PetscInt maxits) {   /* itfunc.c */
  KSPValid(ksp);
  if (rtol != PETSC_DEFAULT) { if (rtol < 0.0 || 1.0 <= rtol) SETERRQ(ksp->comm, PETSC_ERR_ARG_OUTOFRANGE, "Relative tolerance %g must be non-negative and less than 1.0", rtol); ksp->rtol = rtol; }
  if (abstol != PETSC_DEFAULT) { if (abstol < 0.0) SETERRQ(ksp->comm, PETSC_ERR_ARG_OUTOFRANGE, "Absolute tolerance %g must be non-negative", abstol); ksp->abstol = abstol; }
  if (dtol != PETSC_DEFAULT) { if (dtol < 0.0) SETERRQ(ksp->comm, PETSC_ERR_ARG_OUTOFRANGE, "Divergence tolerance %g must be larger than 1.0", dtol); ksp->divtol = dtol; }
  if (maxits != PETSC_DEFAULT) { if (maxits < 0) SETERRQ(ksp->comm, PETSC_ERR_ARG_OUTOFRANGE, "Maximum number of iterations %d must be non-negative", maxits); ksp->max_it = maxits; }
  return 0;
}
PetscErrorCode KSPSetInitialGuessNonzero(KSP ksp, PetscBool flg) { KSPValid(ksp); ksp->guess_zero = (PetscBool)!flg; return 0; }
PetscErrorCode KSPSetNormType(KSP ksp, KSPNormType t) {   /* itcreate.c:221-235: no norm -> the test that only counts iterations */
  KSPValid(ksp);
  ksp->normtype = t;
  if (t == KSP_NORM_NONE) ksp->converged = KSPSkipConverged;
  else if (ksp->converged == KSPSkipConverged) ksp->converged = KSPDefaultConverged;
  if (ksp->setupcalled == 2) ksp->setupcalled = 1;   /* have KSPSetUp look at the combination again */
  return 0;
}
PetscErrorCode KSPSetPCSide(KSP ksp, PCSide side) {   /* itcreate.c KSPSetPCSide */
  KSPValid(ksp);
  if (side != PC_LEFT && side != PC_RIGHT) SETERRQ(ksp->comm, PETSC_ERR_SUP, "only left and right preconditioning are on the ported path");
  ksp->pc_side = side;
  if (ksp->setupcalled == 2) ksp->setupcalled = 1;
  return 0;
}
/* KSPGMRESSetRestart / KSPGMRESSetCGSRefinementType, gmres.c:730,798: PetscTryMethod -- whatever KSP type composed the
 * method answers, any other type ignores the call */
PetscErrorCode KSPGMRESSetRestart(KSP ksp, PetscInt restart) {
  PetscVoidFunction f = NULL;
  KSPValid(ksp);
  PetscErrorCode ierr = PetscObjectQueryFunction((PetscObject)ksp, "KSPGMRESSetRestart_C", &f);CHKERRQ(ierr);
  if (f) { ierr = ((PetscErrorCode (*)(KSP, PetscInt))f)(ksp, restart);CHKERRQ(ierr); }
  return 0;
}
PetscErrorCode KSPGMRESSetCGSRefinementType(KSP ksp, KSPGMRESCGSRefinementType type) {
  PetscVoidFunction f = NULL;
  KSPValid(ksp);
  PetscErrorCode ierr = PetscObjectQueryFunction((PetscObject)ksp, "KSPGMRESSetCGSRefinementType_C", &f);CHKERRQ(ierr);
  if (f) { ierr = ((PetscErrorCode (*)(KSP, KSPGMRESCGSRefinementType))f)(ksp, type);CHKERRQ(ierr); }
  return 0;
}
PetscErrorCode KSPSetOptionsPrefix(KSP ksp, const char prefix[]) {
  KSPValid(ksp);
  snprintf(ksp->prefix, sizeof(ksp->prefix), "%s", prefix ? prefix : "");
  if (ksp->pc) snprintf(ksp->pc->prefix, sizeof(ksp->pc->prefix), "%s", ksp->prefix);
  return 0;
}
PetscErrorCode KSPSetFromOptions(KSP ksp) {   /* itcl.c KSPSetFromOptions, the options the golden tests use */
  PetscErrorCode ierr; char t[64]; PetscBool set; PetscReal r; PetscInt iv; PC pc;
  KSPValid(ksp);
  ierr = KSPGetPC(ksp, &pc);CHKERRQ(ierr);
  ierr = PCSetFromOptions(pc);CHKERRQ(ierr);
  ierr = PetscOptionsGetString(ksp->prefix, "-ksp_type", t, sizeof(t), &set);CHKERRQ(ierr);
  if (set) { ierr = KSPSetType(ksp, t);CHKERRQ(ierr); }
  else if (!ksp->type_name[0]) { ierr = KSPSetType(ksp, KSPGMRES);CHKERRQ(ierr); }   /* default, itcl.c */
  ierr = PetscOptionsGetInt(ksp->prefix, "-ksp_max_it", &iv, &set);CHKERRQ(ierr); if (set) ksp->max_it = iv;
  ierr = PetscOptionsGetReal(ksp->prefix, "-ksp_rtol", &r, &set);CHKERRQ(ierr); if (set) ksp->rtol = r;
  ierr = PetscOptionsGetReal(ksp->prefix, "-ksp_atol", &r, &set);CHKERRQ(ierr); if (set) ksp->abstol = r;
  ierr = PetscOptionsGetReal(ksp->prefix, "-ksp_divtol", &r, &set);CHKERRQ(ierr); if (set) ksp->divtol = r;
  ierr = PetscOptionsGetString(ksp->prefix, "-ksp_initial_guess_nonzero", t, sizeof(t), &set);CHKERRQ(ierr); if (set) ksp->guess_zero = PETSC_FALSE;
  ierr = PetscOptionsGetString(ksp->prefix, "-ksp_monitor", t, sizeof(t), &set);CHKERRQ(ierr);
  if (set) { ierr = KSPMonitorSet(ksp, KSPMonitorDefault, NULL, NULL);CHKERRQ(ierr); }
  ierr = PetscOptionsGetString(ksp->prefix, "-ksp_monitor_short", t, sizeof(t), &set);CHKERRQ(ierr);
  if (set) { ierr = KSPMonitorSet(ksp, KSPMonitorDefaultShort, NULL, NULL);CHKERRQ(ierr); }
  ierr = PetscOptionsGetString(ksp->prefix, "-ksp_converged_reason", t, sizeof(t), &set);CHKERRQ(ierr);
  if (set) ksp->printreason = PETSC_TRUE;
  ierr = PetscOptionsGetString(ksp->prefix, "-ksp_pc_side", t, sizeof(t), &set);CHKERRQ(ierr);
  if (set) {
    if (!strcmp(t, "left")) { ierr = KSPSetPCSide(ksp, PC_LEFT);CHKERRQ(ierr); }
    else if (!strcmp(t, "right")) { ierr = KSPSetPCSide(ksp, PC_RIGHT);CHKERRQ(ierr); }
    else SETERRQ(ksp->comm, PETSC_ERR_ARG_UNKNOWN_TYPE, "Unknown preconditioning side %s", t);
  }
  ierr = PetscOptionsGetString(ksp->prefix, "-ksp_norm_type", t, sizeof(t), &set);CHKERRQ(ierr);   /* itcl.c: KSPNormTypes[] */
  if (set) {
    KSPNormType nt;
    if (!strcmp(t, "none")) nt = KSP_NORM_NONE;
    else if (!strcmp(t, "preconditioned")) nt = KSP_NORM_PRECONDITIONED;
    else if (!strcmp(t, "unpreconditioned")) nt = KSP_NORM_UNPRECONDITIONED;
    else if (!strcmp(t, "natural")) nt = KSP_NORM_NATURAL;
    else SETERRQ(ksp->comm, PETSC_ERR_ARG_UNKNOWN_TYPE, "Unknown norm type %s", t);
    ierr = KSPSetNormType(ksp, nt);CHKERRQ(ierr);
  }
  if (ksp->ops->setfromoptions) { ierr = (*ksp->ops->setfromoptions)(ksp);CHKERRQ(ierr); }
  return 0;
}

/* KSPDefaultGetWork / KSPGetVecs, iterativ.c:911-986: work vectors duplicate vec_sol (inherit its type) */
PetscErrorCode KSPDefaultGetWork(KSP ksp, PetscInt nw) {
  PetscErrorCode ierr;
  if (ksp->work) { ierr = VecDestroyVecs(ksp->nwork, &ksp->work);CHKERRQ(ierr); }
  ksp->nwork = nw;
  if (ksp->vec_sol) { ierr = VecDuplicateVecs(ksp->vec_sol, nw, &ksp->work);CHKERRQ(ierr); }
  else {
    Vec r;
    ierr = MatGetVecs(ksp->pc->mat, &r, NULL);CHKERRQ(ierr);
    ierr = VecDuplicateVecs(r, nw, &ksp->work);CHKERRQ(ierr);
    ierr = VecDestroy(&r);CHKERRQ(ierr);
  }
  return 0;
}

PetscErrorCode KSPSetUp(KSP ksp) {   /* itfunc.c:175 */
  PetscErrorCode ierr;
  KSPValid(ksp);
  if (!ksp->type_name[0]) { ierr = KSPSetType(ksp, KSPGMRES);CHKERRQ(ierr); }
  if (ksp->setupcalled == 2) return 0;
  if (!ksp->pc || !ksp->pc->mat) SETERRQ(ksp->comm, PETSC_ERR_ARG_WRONGSTATE, "Matrix must be set first");
  if (!ksp->setupcalled) { ierr = (*ksp->ops->setup)(ksp);CHKERRQ(ierr); }
  ierr = ksp_setup_norms(ksp);CHKERRQ(ierr);   /* itfunc.c:225 */
  if (ksp->normtype == KSP_NORM_NONE) ksp->converged = KSPSkipConverged;
  ierr = PCSetUp(ksp->pc);CHKERRQ(ierr);
  ksp->setupcalled = 2;
  return 0;
}

PetscErrorCode KSPSetUpOnBlocks(KSP ksp) {   /* itfunc.c:147-156 */
  PetscErrorCode ierr;
  KSPValid(ksp);
  if (!ksp->pc) return 0;
  ierr = PCSetUpOnBlocks(ksp->pc);CHKERRQ(ierr);
  return 0;
}

PetscErrorCode KSPSolve(KSP ksp, Vec b, Vec x) {   /* itfunc.c:335 */
  PetscErrorCode ierr;
  KSPValid(ksp);
  if (b == x) SETERRQ(ksp->comm, PETSC_ERR_SUP, "in-place solve (b == x) is outside the ported path");
  ksp->vec_rhs = b; ksp->vec_sol = x;
  ierr = KSPSetUp(ksp);CHKERRQ(ierr);
  ierr = KSPSetUpOnBlocks(ksp);CHKERRQ(ierr);                          /* itfunc.c:377 */
  if (ksp->guess_zero) { ierr = VecSet(ksp->vec_sol, 0.0);CHKERRQ(ierr); }
  if (ksp->res_hist_reset) ksp->res_hist_len = 0;
  ksp->reason = KSP_CONVERGED_ITERATING;
  ierr = (*ksp->ops->solve)(ksp);CHKERRQ(ierr);
  if (!ksp->reason) SETERRQ(ksp->comm, PETSC_ERR_PLIB, "Internal error, solver returned without setting converged reason");
  if (ksp->printreason && !ksp->comm->rank) {   /* -ksp_converged_reason, itfunc.c:662-668 */
    if (ksp->reason > 0) printf("Linear solve converged due to %s iterations %d\n", ksp_reason_name(ksp->reason), (int)ksp->its);
    else printf("Linear solve did not converge due to %s iterations %d\n", ksp_reason_name(ksp->reason), (int)ksp->its);
    fflush(stdout);
  }
  return 0;
}
PetscErrorCode KSPGetIterationNumber(KSP ksp, PetscInt *its) { KSPValid(ksp); *its = ksp->its; return 0; }
PetscErrorCode KSPGetResidualNorm(KSP ksp, PetscReal *r) { KSPValid(ksp); *r = ksp->rnorm; return 0; }
PetscErrorCode KSPGetConvergedReason(KSP ksp, KSPConvergedReason *reason) { KSPValid(ksp); *reason = ksp->reason; return 0; }
PetscErrorCode KSPSetResidualHistory(KSP ksp, PetscReal a[], PetscInt na, PetscBool reset) {
  PetscErrorCode ierr;
  KSPValid(ksp);
  free(ksp->res_hist_alloc); ksp->res_hist_alloc = NULL;
  if (na != PETSC_DECIDE && na != PETSC_DEFAULT && a) { ksp->res_hist = a; ksp->res_hist_max = na; }
  else {
    if (na == PETSC_DECIDE || na == PETSC_DEFAULT) na = 10000;
    ierr = PetscMalloc(sizeof(PetscReal) * (size_t)na, &ksp->res_hist_alloc);CHKERRQ(ierr);
    ksp->res_hist = ksp->res_hist_alloc; ksp->res_hist_max = na;
  }
  ksp->res_hist_len = 0; ksp->res_hist_reset = reset;
  return 0;
}
PetscErrorCode KSPGetResidualHistory(KSP ksp, PetscReal *a[], PetscInt *na) { KSPValid(ksp); if (a) *a = ksp->res_hist; if (na) *na = ksp->res_hist_len; return 0; }
PetscErrorCode KSPLogResidualHistory(KSP ksp, PetscReal norm) {   /* kspimpl.h KSPLogResidualHistory */
  if (ksp->res_hist && ksp->res_hist_max > ksp->res_hist_len) ksp->res_hist[ksp->res_hist_len++] = norm;
  return 0;
}
PetscErrorCode KSPMonitorSet(KSP ksp, PetscErrorCode (*monitor)(KSP, PetscInt, PetscReal, void *), void *mctx, PetscErrorCode (*destroy)(void **)) {
  KSPValid(ksp); (void)destroy;
  ksp->monitor = monitor; ksp->mctx = mctx;
  return 0;
}
/* KSPMonitorDefault / KSPMonitorDefaultShort, iterativ.c:178-195,484-503: what -ksp_monitor / -ksp_monitor_short print
 * (rank 0 of the KSP's communicator; a prefixed, i.e. inner, solver announces itself at iteration 0) */
PetscErrorCode KSPMonitorDefault(KSP ksp, PetscInt n, PetscReal rnorm, void *dummy) {
  (void)dummy;
  if (ksp->comm->rank) return 0;
  if (n == 0 && ksp->prefix[0]) printf("  Residual norms for %s solve.\n", ksp->prefix);
  printf("%3d KSP Residual norm %14.12e \n", (int)n, (double)rnorm);
  fflush(stdout);
  return 0;
}
PetscErrorCode KSPMonitorDefaultShort(KSP ksp, PetscInt its, PetscReal fnorm, void *dummy) {
  (void)dummy;
  if (ksp->comm->rank) return 0;
  if (its == 0 && ksp->prefix[0]) printf("  Residual norms for %s solve.\n", ksp->prefix);
  if (fnorm > 1.e-9) printf("%3d KSP Residual norm %g \n", (int)its, (double)fnorm);
  else if (fnorm > 1.e-11) printf("%3d KSP Residual norm %5.3e \n", (int)its, (double)fnorm);
  else printf("%3d KSP Residual norm < 1.e-11\n", (int)its);
  fflush(stdout);
  return 0;
}
static const char *ksp_reason_name(KSPConvergedReason r) {   /* KSPConvergedReasons[], dlregisksp.c:98-104 */
  static const char *const shifted[] = {"DIVERGED_INDEFINITE_MAT", "DIVERGED_NAN", "DIVERGED_INDEFINITE_PC", "DIVERGED_NONSYMMETRIC",
    "DIVERGED_BREAKDOWN_BICG", "DIVERGED_BREAKDOWN", "DIVERGED_DTOL", "DIVERGED_ITS", "DIVERGED_NULL", "", "CONVERGED_ITERATING",
    "CONVERGED_RTOL_NORMAL", "CONVERGED_RTOL", "CONVERGED_ATOL", "CONVERGED_ITS", "CONVERGED_CG_NEG_CURVE", "CONVERGED_CG_CONSTRAINED",
    "CONVERGED_STEP_LENGTH", "CONVERGED_HAPPY_BREAKDOWN", "CONVERGED_ATOL_NORMAL"};
  int k = (int)r + 10;
  return (k >= 0 && k < (int)(sizeof(shifted) / sizeof(shifted[0]))) ? shifted[k] : "UNKNOWN";
}
PetscErrorCode KSPMonitor(KSP ksp, PetscInt it, PetscReal rnorm) {
  if (ksp->monitor) { PetscErrorCode ierr = (*ksp->monitor)(ksp, it, rnorm, ksp->mctx);CHKERRQ(ierr); }
  return 0;
}
PetscErrorCode KSPDestroy(KSP *pksp) {
  PetscErrorCode ierr;
  KSP ksp = *pksp;
  if (!ksp) return 0;
  if (ksp->ops->destroy) { ierr = (*ksp->ops->destroy)(ksp);CHKERRQ(ierr); }
  if (ksp->work) { ierr = VecDestroyVecs(ksp->nwork, &ksp->work);CHKERRQ(ierr); }
  ierr = PCDestroy(&ksp->pc);CHKERRQ(ierr);
  ierr = PetscObjectListDestroy_Private((PetscObject)ksp);CHKERRQ(ierr);
  free(ksp->res_hist_alloc);
  free(ksp); *pksp = NULL;
  return 0;
}

/* kspimpl.h:181-188 */
PetscErrorCode KSP_MatMult(KSP ksp, Mat A, Vec x, Vec y) { (void)ksp; return MatMult(A, x, y); }
PetscErrorCode KSP_PCApply(KSP ksp, Vec x, Vec y) { return PCApply(ksp->pc, x, y); }
/* PCApplyBAorAB, src/ksp/pc/interface/precon.c:553-640 (no diagonal scaling) */
PetscErrorCode KSP_PCApplyBAorAB(KSP ksp, Vec x, Vec y, Vec w) {
  PetscErrorCode ierr;
  if (x == y) SETERRQ(ksp->comm, PETSC_ERR_ARG_IDN, "x and y must be different vectors");
  if (ksp->pc_side == PC_RIGHT) { ierr = PCApply(ksp->pc, x, w);CHKERRQ(ierr); ierr = MatMult(ksp->pc->mat, w, y);CHKERRQ(ierr); }
  else if (ksp->pc_side == PC_LEFT) { ierr = MatMult(ksp->pc->mat, x, w);CHKERRQ(ierr); ierr = PCApply(ksp->pc, w, y);CHKERRQ(ierr); }
  else SETERRQ(ksp->comm, PETSC_ERR_SUP, "symmetric preconditioning is outside the ported path");
  return 0;
}

/* KSPInitialResidual, itres.c:39-73 */
PetscErrorCode KSPInitialResidual(KSP ksp, Vec vsoln, Vec vt1, Vec vt2, Vec vres, Vec vb) {
  PetscErrorCode ierr;
  Mat Amat = ksp->pc->mat;
  if (!ksp->guess_zero) {
    ierr = KSP_MatMult(ksp, Amat, vsoln, vt1);CHKERRQ(ierr);
    ierr = VecCopy(vb, vt2);CHKERRQ(ierr);
    ierr = VecAXPY(vt2, -1.0, vt1);CHKERRQ(ierr);
    if (ksp->pc_side == PC_RIGHT) { ierr = VecCopy(vt2, vres);CHKERRQ(ierr); }
    else { ierr = KSP_PCApply(ksp, vt2, vres);CHKERRQ(ierr); }
  } else {
    ierr = VecCopy(vb, vt2);CHKERRQ(ierr);
    if (ksp->pc_side == PC_RIGHT) { ierr = VecCopy(vb, vres);CHKERRQ(ierr); }
    else if (ksp->pc_side == PC_LEFT) { ierr = KSP_PCApply(ksp, vb, vres);CHKERRQ(ierr); }
    else SETERRQ(ksp->comm, PETSC_ERR_SUP, "Invalid preconditioning side %d", (int)ksp->pc_side);
  }
  return 0;
}

/* KSPSkipConverged, iterativ.c:536-544: what KSPSetNormType(KSP_NORM_NONE) installs -- iterate until max_it */
PetscErrorCode KSPSkipConverged(KSP ksp, PetscInt n, PetscReal rnorm, KSPConvergedReason *reason, void *ctx) {
  (void)rnorm; (void)ctx;
  *reason = (n >= ksp->max_it) ? KSP_CONVERGED_ITS : KSP_CONVERGED_ITERATING;
  return 0;
}
/* KSPDefaultConverged, iterativ.c:702-783 */
PetscErrorCode KSPDefaultConverged(KSP ksp, PetscInt n, PetscReal rnorm, KSPConvergedReason *reason, void *ctx) {
  PetscErrorCode ierr;
  (void)ctx;
  *reason = KSP_CONVERGED_ITERATING;
  if (!n) {
    if (!ksp->guess_zero) {
      PetscReal snorm = 0.0;
      if (ksp->normtype == KSP_NORM_UNPRECONDITIONED || ksp->pc_side == PC_RIGHT) { ierr = VecNorm(ksp->vec_rhs, NORM_2, &snorm);CHKERRQ(ierr); }
      else {
        Vec z;
        ierr = VecDuplicate(ksp->vec_rhs, &z);CHKERRQ(ierr);
        ierr = KSP_PCApply(ksp, ksp->vec_rhs, z);CHKERRQ(ierr);
        if (ksp->normtype == KSP_NORM_NATURAL) { PetscScalar nrm; ierr = VecDot(ksp->vec_rhs, z, &nrm);CHKERRQ(ierr); snorm = PetscSqrtReal(PetscAbsScalar(nrm)); }
        else { ierr = VecNorm(z, NORM_2, &snorm);CHKERRQ(ierr); }
        ierr = VecDestroy(&z);CHKERRQ(ierr);
      }
      if (!snorm) snorm = rnorm;   /* zero RHS and nonzero guess */
      ksp->rnorm0 = snorm;
    } else ksp->rnorm0 = rnorm;
    ksp->ttol = PetscMax(ksp->rtol * ksp->rnorm0, ksp->abstol);
  }
  if (n <= ksp->chknorm) return 0;
  if (PetscIsInfOrNanScalar(rnorm)) *reason = KSP_DIVERGED_NAN;
  else if (rnorm <= ksp->ttol) *reason = (rnorm < ksp->abstol) ? KSP_CONVERGED_ATOL : KSP_CONVERGED_RTOL;
  else if (rnorm >= ksp->divtol * ksp->rnorm0) *reason = KSP_DIVERGED_DTOL;
  return 0;
}

/* KSPPREONLY, src/ksp/ksp/impls/preonly/preonly.c: x = PC(b) */
static PetscErrorCode KSPSetUp_PREONLY(KSP ksp) { (void)ksp; return 0; }
static PetscErrorCode KSPSolve_PREONLY(KSP ksp) {
  PetscErrorCode ierr;
  if (!ksp->guess_zero) SETERRQ(ksp->comm, PETSC_ERR_USER, "Running KSP of preonly doesn't make sense with nonzero initial guess\nyou probably want a KSP type of Richardson");
  ksp->its = 0;
  ierr = KSP_PCApply(ksp, ksp->vec_rhs, ksp->vec_sol);CHKERRQ(ierr);
  ksp->its = 1;
  ksp->reason = KSP_CONVERGED_ITS;
  return 0;
}
PetscErrorCode KSPCreate_PREONLY(KSP ksp) {   /* preonly.c:55-56 */
  ksp->ops->setup = KSPSetUp_PREONLY; ksp->ops->solve = KSPSolve_PREONLY;
  ksp->normsupporttable[KSP_NORM_NONE][PC_LEFT] = 2; ksp->normsupporttable[KSP_NORM_NONE][PC_RIGHT] = 1;
  return 0;
}
