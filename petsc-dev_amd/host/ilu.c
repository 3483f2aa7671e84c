/* Factored matrices of MATSEQAIJHIPMI355X, ILU(0) part, behind the reference's own factorisation interface:
 *
 *   MatGetFactor(A, "petsc", MAT_FACTOR_ILU | MAT_FACTOR_ICC, &F)      "MatGetFactor_petsc_C" composed on every matrix of the type
 *   MatILUFactorSymbolic(F, A, ...) / MatICCFactorSymbolic(F, A, ...)   F->ops->ilufactorsymbolic / iccfactorsymbolic
 *   MatLUFactorNumeric(F, A, info) / MatCholeskyFactorNumeric(F, A, info)
 *   MatSolve(F, b, x)                                                   F->ops->solve  = the device triangular solves
 *
 * so that an UNCHANGED PCILU / PCICC / PCBJACOBI (PETSc's own inside a PETSc tree, the harness's pcfactor.c on a box without
 * PETSc) reaches the device solves -- how the reference's GPU back end does it (MatGetFactor_seqaij_cusparse,
 * MatLUFactorNumeric_SeqAIJCUSPARSE installing MatSolve_SeqAIJCUSPARSE, src/mat/impls/aij/seq/seqcusparse/aijcusparse.cu:57-75,
 * 175-200,358-445), and like it the factorisation itself runs on the host copy of the matrix.
 *
 *   numeric : MatILUFactorSymbolic_SeqAIJ_ilu0 + MatLUFactorNumeric_SeqAIJ (src/mat/impls/aij/seq/aijfact.c:1628, :461), natural
 *             ordering.  Inside a PETSc tree those routines themselves (the parent class's); on the harness their restatement
 *             below.  Then a dependency-level analysis of L and U and one upload.
 *   solve   : MatSolve_SeqAIJ_NaturalOrdering (aijfact.c:3126) on the device, one lane per row in column order (same bits): by
 *             default two launches, one per triangular solve, with point-to-point hand-off of the solution values between
 *             wavefronts (mi355x_trisolve_*, csrc/trisolve.hip); -pc_factor_hipmi355x_trisolve level selects the level-scheduled
 *             kernels, one launch per dependency level (replayed from a hipGraph), which also serve systems with few levels. */
#include "hipmi355ximpl.h"
#if defined(PETSCHIPMI355X_WITH_PETSC)
#include <../src/mat/impls/aij/seq/aij.h>
EXTERN_C_BEGIN
extern PetscErrorCode MatGetFactor_seqaij_petsc(Mat, MatFactorType, Mat *);
EXTERN_C_END
#endif

/* ---------------------------------------------------------------- sync-free solves that gave up: noticed at the next host wait
 * A dependency wait of the sync-free kernels is bounded; a lane that gives up raises a flag in pinned host memory and the
 * application's result is unusable.  The flag cannot be read before the kernels have run, so every factored matrix with
 * sync-free plans is on a watch list, and every host wait the solvers already perform (reduction results, VecGetArray:
 * HipTriWatchCheck, called from vechip.c) looks at the flags: the first wait after an abort returns PETSC_ERR_LIB instead of
 * numbers computed from a poisoned vector, and the factor is switched to the level-by-level form of the same plans for every
 * later application (MatSolve below), ILU(0) and ICC(0) alike. */
static HipTriFactors **tri_watch = NULL;
static int tri_watch_cap = 0;
/* the list grows with the number of live factors (multigrid levels, fieldsplit blocks, many KSPs); a factor that cannot be put on
 * it (out of memory) never runs the sync-free kernels unwatched: it takes the level launches */
void HipTriWatchAdd(HipTriFactors *f) {
  for (int i = 0; i < tri_watch_cap; i++) if (tri_watch[i] == f) return;
  for (int i = 0; i < tri_watch_cap; i++) if (!tri_watch[i]) { tri_watch[i] = f; return; }
  const int ncap = tri_watch_cap ? 2 * tri_watch_cap : 64;
  HipTriFactors **nw = (HipTriFactors **)realloc(tri_watch, sizeof(*nw) * (size_t)ncap);
  if (!nw) { f->use_levels = 1; return; }
  memset(nw + tri_watch_cap, 0, sizeof(*nw) * (size_t)(ncap - tri_watch_cap));
  nw[tri_watch_cap] = f;
  tri_watch = nw; tri_watch_cap = ncap;
}
static void tri_watch_remove(HipTriFactors *f) { for (int i = 0; i < tri_watch_cap; i++) if (tri_watch[i] == f) tri_watch[i] = NULL; }
PetscErrorCode HipTriWatchCheck(void) {
  for (int i = 0; i < tri_watch_cap; i++) {
    HipTriFactors *f = tri_watch[i];
    int a = 0, b = 0;
    if (!f || !f->tri_lo || f->use_levels) continue;
    mi355x_trisolve_aborted(f->tri_lo, &a); mi355x_trisolve_aborted(f->tri_up, &b);
    if (a || b) {
      f->use_levels = 1; f->aborted = 1;
      SETERRQ(PETSC_COMM_SELF, PETSC_ERR_LIB, "a sync-free triangular solve gave up waiting for a dependency: results computed since that MatSolve are invalid; "
                                              "later applications of this factor use one launch per dependency level");
    }
  }
  return 0;
}

PetscErrorCode HipTriFactorsDestroy(HipTriFactors **pf) {
  HipTriFactors *f = *pf;
  if (!f) return 0;
  tri_watch_remove(f);
  if (f->owns_host) { HipFree(f->bi); HipFree(f->bj); HipFree(f->bdiag); HipFree(f->ba); }
  HipFree(f->levptrL); HipFree(f->levptrU); HipFree(f->blk); HipFree(f->rlevL); HipFree(f->rlevU);
  if (f->d_bi) mi355x_free(f->d_bi);
  if (f->d_bj) mi355x_free(f->d_bj);
  if (f->d_bdiag) mi355x_free(f->d_bdiag);
  if (f->d_ba) mi355x_free(f->d_ba);
  if (f->d_rowsL) mi355x_free(f->d_rowsL);
  if (f->d_rowsU) mi355x_free(f->d_rowsU);
  if (f->d_work) mi355x_free(f->d_work);
  if (f->graph) mi355x_graph_destroy(f->graph);
  if (f->tri_lo) mi355x_trisolve_plan_destroy(f->tri_lo);
  if (f->tri_up) mi355x_trisolve_plan_destroy(f->tri_up);
  HipFree(f);
  *pf = NULL;
  return 0;
}
/* forget one numeric factorisation (the symbolic choices, the block list and the watch slot stay) */
static void tri_reset_numeric(HipTriFactors *f) {
  if (f->owns_host) { HipFree(f->bi); HipFree(f->bj); HipFree(f->bdiag); HipFree(f->ba); }
  f->bi = f->bj = f->bdiag = NULL; f->ba = NULL;
  HipFree(f->levptrL); HipFree(f->levptrU); f->levptrL = f->levptrU = NULL;
  HipFree(f->rlevL); HipFree(f->rlevU); f->rlevL = f->rlevU = NULL;
  if (f->d_bi) mi355x_free(f->d_bi);
  if (f->d_bj) mi355x_free(f->d_bj);
  if (f->d_bdiag) mi355x_free(f->d_bdiag);
  if (f->d_ba) mi355x_free(f->d_ba);
  if (f->d_rowsL) mi355x_free(f->d_rowsL);
  if (f->d_rowsU) mi355x_free(f->d_rowsU);
  if (f->d_work) mi355x_free(f->d_work);
  if (f->graph) mi355x_graph_destroy(f->graph);
  if (f->tri_lo) mi355x_trisolve_plan_destroy(f->tri_lo);
  if (f->tri_up) mi355x_trisolve_plan_destroy(f->tri_up);
  f->d_bi = f->d_bj = f->d_bdiag = f->d_rowsL = f->d_rowsU = NULL; f->d_ba = f->d_work = NULL;
  f->graph = NULL; f->graph_tried = 0; f->tri_lo = f->tri_up = NULL;
  f->use_levels = 0; f->nshift = 0; f->nlevL = f->nlevU = 0;
  f->factored_state = -1;
}

/* block Jacobi solving all its ILU(0) / ICC(0) blocks as one block-diagonal system ("MatFactorSetIndependentBlocks_C", asked for
 * by the harness's PCILU / PCICC): every block is factored as the reference factors a matrix of its own (its own shift loop), so
 * the result is the blocks' factors side by side also when one block needs a shift */
static PetscErrorCode MatFactorSetIndependentBlocks_SeqAIJHIP(Mat F, PetscInt nblk, const PetscInt *starts) {
  HipTriFactors *f = HipTriGet(F);
  PetscErrorCode ierr;
  if (f->nblk == nblk && (!nblk || !memcmp(f->blk, starts, sizeof(PetscInt) * (size_t)(nblk + 1)))) return 0;
  HipFree(f->blk); f->blk = NULL; f->nblk = 0;
  if (nblk > 0) {
    ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(nblk + 1), &f->blk);CHKERRQ(ierr);
    memcpy(f->blk, starts, sizeof(PetscInt) * (size_t)(nblk + 1));
    f->nblk = nblk;
  }
  f->factored_state = -1;
  return 0;
}

/* rows sorted by dependency level (stable: ascending row inside a level) */
static PetscErrorCode level_order(PetscInt n, const PetscInt *lev, PetscInt nlev, PetscInt **ptr_out, PetscInt **rows_out) {
  PetscErrorCode ierr;
  PetscInt *ptr, *rows, *next;
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(nlev + 1), &ptr);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(n, 1), &rows);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(nlev + 1), &next);CHKERRQ(ierr);
  memset(ptr, 0, sizeof(PetscInt) * (size_t)(nlev + 1));
  for (PetscInt i = 0; i < n; i++) ptr[lev[i] + 1]++;
  for (PetscInt l = 0; l < nlev; l++) ptr[l + 1] += ptr[l];
  memcpy(next, ptr, sizeof(PetscInt) * (size_t)(nlev + 1));
  for (PetscInt i = 0; i < n; i++) rows[next[lev[i]]++] = i;
  HipFree(next);
  *ptr_out = ptr; *rows_out = rows;
  return 0;
}

static PetscErrorCode natural_ordering_only(Mat A, IS row, IS col, const MatFactorInfo *info, const char *what) {
  if (info->levels != 0.0) SETERRQ(HipObjComm(A), PETSC_ERR_SUP, "%s(%d): only zero fill is on the ported path", what, (int)info->levels);
#if defined(PETSCHIPMI355X_WITH_PETSC)
  { PetscErrorCode ierr; PetscBool id = PETSC_TRUE;
    if (row) { ierr = ISIdentity(row, &id);CHKERRQ(ierr); if (!id) SETERRQ(HipObjComm(A), PETSC_ERR_SUP, "%s on the device: natural ordering only (-pc_factor_mat_ordering_type natural)", what); }
    if (col) { ierr = ISIdentity(col, &id);CHKERRQ(ierr); if (!id) SETERRQ(HipObjComm(A), PETSC_ERR_SUP, "%s on the device: natural ordering only (-pc_factor_mat_ordering_type natural)", what); } }
#else
  if (row || col) SETERRQ(HipObjComm(A), PETSC_ERR_SUP, "%s: natural ordering only", what);
#endif
  return 0;
}

#include <time.h>
static double wall_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
#define SETUP_TICK(what) do { if (getenv("PETSC_HIPMI355X_SETUP_TIMING")) { const double t__ = wall_s(); fprintf(stderr, "[hipmi355x]   %-34s %.3f s\n", what, t__ - tick0); tick0 = t__; } } while (0)

#include <pthread.h>
#include <unistd.h>
/* the host factorisation hands its work arrays to a thread that returns them to the system; at most one such thread: it is joined before the next one starts and by PetscHIPMI355XFinalize (never left running behind the library) */
static pthread_t release_th; static int release_running = 0;
void HipFactorJoinHelpers(void) { if (release_running) { pthread_join(release_th, NULL); release_running = 0; } }
/* dependency levels of the rows of L (a row may start once the rows its L part names are done) and of U (backwards), from the
 * factor's pattern in the reference's layout; each is a sequential recurrence, the two run side by side */
typedef struct { PetscInt n; const PetscInt *bi, *bj, *bdiag; PetscInt *lev, nlev; } RowLevArg;
static void *row_levels_L(void *a_) {
  RowLevArg *a = (RowLevArg *)a_; const PetscInt *bi = a->bi, *bj = a->bj; PetscInt *lev = a->lev, nlev = 0;
  for (PetscInt i = 0; i < a->n; i++) {
    PetscInt l = 0;
    for (PetscInt q = bi[i]; q < bi[i + 1]; q++) l = PetscMax(l, lev[bj[q]] + 1);
    lev[i] = l; nlev = PetscMax(nlev, l + 1);
  }
  a->nlev = nlev;
  return NULL;
}
static void *row_levels_U(void *a_) {
  RowLevArg *a = (RowLevArg *)a_; const PetscInt *bj = a->bj, *bdiag = a->bdiag; PetscInt *lev = a->lev, nlev = 0;
  for (PetscInt i = a->n - 1; i >= 0; i--) {
    PetscInt l = 0; const PetscInt s0 = bdiag[i + 1] + 1, nz = bdiag[i] - bdiag[i + 1] - 1;
    for (PetscInt q = 0; q < nz; q++) l = PetscMax(l, lev[bj[s0 + q]] + 1);
    lev[i] = l; nlev = PetscMax(nlev, l + 1);
  }
  a->nlev = nlev;
  return NULL;
}
static PetscErrorCode ilu0_row_levels(PetscInt n, const PetscInt *bi, const PetscInt *bj, const PetscInt *bdiag, PetscInt **levL, PetscInt *nlevL, PetscInt **levU, PetscInt *nlevU) {
  PetscErrorCode ierr;
  RowLevArg aL = {n, bi, bj, bdiag, NULL, 0}, aU = {n, bi, bj, bdiag, NULL, 0};
  pthread_t th; int side = 0;
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(n, 1), &aL.lev);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(n, 1), &aU.lev);CHKERRQ(ierr);
  if (n >= 200000) side = !pthread_create(&th, NULL, row_levels_L, &aL);
  if (!side) row_levels_L(&aL);
  row_levels_U(&aU);
  if (side) pthread_join(th, NULL);
  *levL = aL.lev; *nlevL = aL.nlev; *levU = aU.lev; *nlevU = aU.nlev;
  return 0;
}

#if !defined(PETSCHIPMI355X_WITH_PETSC)
/* one pass of the numeric ILU(0) over the rows [r0, r1) of an independent block with a given diagonal shift, rows taken level by
 * level (dependency levels of L) and a level's rows dealt to the threads */
typedef struct {
  const PetscInt *ai, *aj; const PetscScalar *aa;
  const PetscInt *bi, *bj, *bdiag; PetscScalar *ba;
  PetscInt n, r0, r1, nlev, *levptr, *rows;
  PetscReal zeropivot, shift_amount;
  int nth;
  PetscScalar **rtmp;                 /* a dense work row per thread */
  volatile PetscInt fail_row, fail_level; volatile PetscReal fail_value;
  volatile int go;                    /* the gate the threads start at: 1 go, -1 leave (not all of them could be created) */
  int oversubscribed;                 /* more threads than cores of this rank */
  volatile int bar_count, bar_sense;  /* sense-reversing barrier: the levels are short (tens of microseconds), a futex sleep per level costs more */
  pthread_mutex_t mtx;
} IluPass;
typedef struct { IluPass *p; int tid; } IluArg;

/* row i of MatLUFactorNumeric_SeqAIJ (aijfact.c:505-570): dense work row, multipliers in column order, the pivot stored inverted;
 * returns 1 when the pivot fails MatPivotCheck_nz (matimpl.h:512-528) */
static int ilu0_factor_row(const IluPass *p, PetscScalar *rtmp, PetscInt i, PetscReal *badval) {
  const PetscInt *ai = p->ai, *aj = p->aj, *bi = p->bi, *bj = p->bj, *bdiag = p->bdiag; const PetscScalar *aa = p->aa; PetscScalar *ba = p->ba;
  PetscInt nzl = bi[i + 1] - bi[i], nzu = bdiag[i] - bdiag[i + 1];
  PetscReal rs = 0.0;
  for (PetscInt j = 0; j < nzl; j++) rtmp[bj[bi[i] + j]] = 0.0;
  for (PetscInt j = 0; j < nzu; j++) rtmp[bj[bdiag[i + 1] + 1 + j]] = 0.0;
  for (PetscInt q = ai[i]; q < ai[i + 1]; q++) rtmp[aj[q]] = aa[q];
  rtmp[i] += p->shift_amount;
  for (PetscInt kk = 0; kk < nzl; kk++) {
    const PetscInt row = bj[bi[i] + kk];
    PetscScalar *pc_ = rtmp + row;
    if (*pc_ != 0.0) {
      const PetscScalar multiplier = *pc_ * ba[bdiag[row]];
      *pc_ = multiplier;
      const PetscInt *pj = bj + bdiag[row + 1] + 1;
      const PetscScalar *pv = ba + bdiag[row + 1] + 1;
      const PetscInt nz = bdiag[row] - bdiag[row + 1] - 1;
      for (PetscInt j = 0; j < nz; j++) rtmp[pj[j]] -= multiplier * pv[j];
    }
  }
  for (PetscInt j = 0; j < nzl; j++) { ba[bi[i] + j] = rtmp[bj[bi[i] + j]]; rs += PetscAbsScalar(ba[bi[i] + j]); }
  for (PetscInt j = 0; j < nzu - 1; j++) { ba[bdiag[i + 1] + 1 + j] = rtmp[bj[bdiag[i + 1] + 1 + j]]; rs += PetscAbsScalar(ba[bdiag[i + 1] + 1 + j]); }
  if (PetscAbsScalar(rtmp[i]) <= p->zeropivot * rs) { *badval = PetscAbsScalar(rtmp[i]); return 1; }
  ba[bdiag[i]] = 1.0 / rtmp[i];
  return 0;
}
#include <sched.h>
static void ilu0_barrier(IluPass *p, int *sense) {
  *sense = !*sense;
  if (__sync_add_and_fetch(&p->bar_count, 1) == p->nth) { p->bar_count = 0; __sync_synchronize(); p->bar_sense = *sense; }
  else {   /* more threads than this rank's cores (an explicit -mat_factor_hipmi355x_threads): give the core away at once instead of spinning */
    const int limit = p->oversubscribed ? 1 : 4000;
    int spins = 0; while (p->bar_sense != *sense) { if (++spins > limit) { sched_yield(); spins = 0; } } }
  __sync_synchronize();
}
static void *ilu0_worker(void *arg_) {
  IluArg *arg = (IluArg *)arg_;
  IluPass *p = arg->p;
  PetscScalar *rtmp = p->rtmp[arg->tid];
  int sense = 0;
  if (p->nth > 1 && arg->tid > 0) { while (p->go == 0) sched_yield(); if (p->go < 0) return NULL; }
  for (PetscInt l = 0; l < p->nlev; l++) {
    const PetscInt a = p->levptr[l], b = p->levptr[l + 1], cnt = b - a;
    const PetscInt lo = a + (PetscInt)((long)cnt * arg->tid / p->nth), hi = a + (PetscInt)((long)cnt * (arg->tid + 1) / p->nth);
    for (PetscInt t = lo; t < hi; t++) {
      const PetscInt i = p->rows[t];
      PetscReal bad;
      if (i < p->r0 || i >= p->r1) continue;
      if (ilu0_factor_row(p, rtmp, i, &bad)) {
        pthread_mutex_lock(&p->mtx);
        if (p->fail_row < 0 || i < p->fail_row) { p->fail_row = i; p->fail_value = bad; }
        if (p->fail_level < 0 || l < p->fail_level) p->fail_level = l;
        pthread_mutex_unlock(&p->mtx);
        break;
      }
    }
    if (p->nth > 1) ilu0_barrier(p, &sense);
    /* one barrier per level: a thread that is already in level l + 1 may record a failure there before a slower thread has looked
     * at level l's outcome; that thread goes on to level l + 1 like everybody else and they all leave after ITS barrier */
    if (p->fail_level >= 0 && p->fail_level <= l) break;
  }
  return NULL;
}
static PetscErrorCode ilu0_run_pass(IluPass *p) {
  IluArg args[64]; pthread_t th[64];
  for (int t = 0; t < p->nth; t++) { args[t].p = p; args[t].tid = t; }
  pthread_mutex_init(&p->mtx, NULL);
  if (p->nth == 1) { ilu0_worker(&args[0]); pthread_mutex_destroy(&p->mtx); return 0; }
  p->bar_count = 0; p->bar_sense = 0; p->go = 0;
  int started = 0;
  for (int t = 1; t < p->nth; t++) { if (pthread_create(&th[t], NULL, ilu0_worker, &args[t])) break; started = t; }
  if (started != p->nth - 1) {            /* could not start them all: the ones that did are still at the gate and leave from there */
    p->go = -1;
    for (int t = 1; t <= started; t++) pthread_join(th[t], NULL);
    pthread_mutex_destroy(&p->mtx);
    return PETSC_ERR_LIB;
  }
  p->go = 1;
  ilu0_worker(&args[0]);
  for (int t = 1; t < p->nth; t++) pthread_join(th[t], NULL);
  pthread_mutex_destroy(&p->mtx);
  return 0;
}
#endif

#if !defined(PETSCHIPMI355X_WITH_PETSC)
static void *ilu0_release_thread(void *a_) { void **a = (void **)a_; for (int i = 0; a[i]; i++) free(a[i]); free(a); return NULL; }
typedef struct { const PetscInt *ai, *aj; PetscInt *adiag, *bi, *bj, *bdiag; volatile PetscInt missing; } IluSym;
static void ilu0_sym_diag(void *c_, PetscInt lo, PetscInt hi) {
  IluSym *c = (IluSym *)c_;
  for (PetscInt i = lo; i < hi; i++) {
    PetscInt d = -1;
    for (PetscInt q = c->ai[i]; q < c->ai[i + 1]; q++) if (c->aj[q] == i) { d = q; break; }
    c->adiag[i] = d;
    if (d < 0) c->missing = i;             /* (the caller looks for the first such row) */
  }
}
/* the pattern of A, L part forward, U part from the last row backwards with the diagonal at each row's end (aijfact.c:1660-1685) */
static void ilu0_sym_pattern(void *c_, PetscInt lo, PetscInt hi) {
  IluSym *c = (IluSym *)c_;
  for (PetscInt i = lo; i < hi; i++) {
    const PetscInt nzl = c->adiag[i] - c->ai[i], nzu = c->ai[i + 1] - c->adiag[i] - 1;
    PetscInt *l = c->bj + c->bi[i], *u = c->bj + c->bdiag[i + 1] + 1;
    for (PetscInt j = 0; j < nzl; j++) l[j] = c->aj[c->ai[i] + j];
    for (PetscInt j = 0; j < nzu; j++) u[j] = c->aj[c->adiag[i] + 1 + j];
    u[nzu] = i;
  }
}
/* MatILUFactorSymbolic_SeqAIJ_ilu0 + MatLUFactorNumeric_SeqAIJ restated for the harness (inside a PETSc tree the parent's
 * routines run instead): the pattern of A, L part forward, U part from the last row backwards (aijfact.c:1660-1685); row by row
 * with a dense work row, pivots stored inverted (aijfact.c:505-570); MatPivotCheck_nz's restarts (matimpl.h:512-528) */
static PetscErrorCode ilu0_factor_host(Mat F, Mat A, const MatFactorInfo *info) {
  PetscErrorCode ierr;
  HipTriFactors *f = HipTriGet(F);
  PetscInt n; const PetscInt *ai, *aj; const PetscScalar *aa;
  double tick0 = wall_s();
  ierr = MatSeqAIJGetArrays(A, &n, &ai, &aj, &aa);CHKERRQ(ierr);
  f->n = n; f->nz = ai[n]; f->owns_host = PETSC_TRUE;
  PetscInt *adiag;
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(n, 1), &adiag);CHKERRQ(ierr);
  { IluSym sy = {ai, aj, adiag, NULL, NULL, NULL, -1};
    HipParallelRanges(n, ilu0_sym_diag, &sy);
    if (sy.missing >= 0) {
      PetscInt first = 0;
      while (first < n && adiag[first] >= 0) first++;
      HipFree(adiag);
      SETERRQ(HipObjComm(A), PETSC_ERR_ARG_WRONGSTATE, "Matrix is missing diagonal entry %d", first);
    } }
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(n + 1), &f->bi);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(f->nz + 1), &f->bj);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(n + 1), &f->bdiag);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)(f->nz + 1), &f->ba);CHKERRQ(ierr);
  SETUP_TICK("factor: diagonal positions");
  f->ba[f->nz] = 0.0;                      /* (every other entry is written by the numeric pass) */
  PetscInt *bi = f->bi, *bj = f->bj, *bdiag = f->bdiag; PetscScalar *ba = f->ba;
  /* the two pointer arrays are running sums (one light sequential pass each); the column copies are per row, on host threads */
  bi[0] = 0;
  for (PetscInt i = 0; i < n; i++) bi[i + 1] = bi[i] + (adiag[i] - ai[i]);
  bdiag[n] = bi[n] - 1;
  for (PetscInt i = n - 1; i >= 0; i--) bdiag[i] = bdiag[i + 1] + (ai[i + 1] - adiag[i] - 1) + 1;
  { IluSym sy = {ai, aj, adiag, bi, bj, bdiag, -1};
    HipParallelRanges(n, ilu0_sym_pattern, &sy); }
  SETUP_TICK("factor: pattern of L and U");
  const PetscReal zeropivot = info->zeropivot, shiftamount = info->shiftamount;
  const PetscBool shift_nz = (PetscBool)(info->shifttype == (PetscReal)MAT_SHIFT_NONZERO);
  f->nshift = 0;
  const PetscInt whole[2] = {0, n};
  const PetscInt nblk = (f->nblk > 0 && f->blk[f->nblk] == n) ? f->nblk : 1, *blk = (f->nblk > 0 && f->blk[f->nblk] == n) ? f->blk : whole;
  for (PetscInt bb = 0; bb < nblk; bb++)
    for (PetscInt i = blk[bb]; i < blk[bb + 1]; i++)
      if (ai[i] < ai[i + 1] && (aj[ai[i]] < blk[bb] || aj[ai[i + 1] - 1] >= blk[bb + 1])) { HipFree(adiag); SETERRQ(HipObjComm(A), PETSC_ERR_ARG_WRONG, "row %d couples to a column outside its independent block", i); }
  /* Rows of one dependency level of L do not read each other: a level's rows are factored by several host threads, one barrier per
   * level.  Every row's own arithmetic is the sequential loop's (ilu0_factor_row), so the factor carries the same bits whatever
   * the thread count; a pivot that fails MatPivotCheck_nz anywhere ends the pass for everybody and the block restarts shifted. */
  IluPass ps;
  memset(&ps, 0, sizeof(ps));
  ps.ai = ai; ps.aj = aj; ps.aa = aa; ps.bi = bi; ps.bj = bj; ps.bdiag = bdiag; ps.ba = ba; ps.zeropivot = zeropivot; ps.n = n;
  { PetscInt nth = 1; PetscBool set;
    if (n >= 200000) nth = (PetscInt)HipHostThreads(16);
    ierr = PetscOptionsGetInt(NULL, "-mat_factor_hipmi355x_threads", &nth, &set);CHKERRQ(ierr);
    if (nth < 1) nth = 1;
    if (nth > 64) nth = 64;
    ps.nth = (int)nth; ps.oversubscribed = nth > (PetscInt)HipHostThreads(64); }
  /* levels of L (the threaded passes below take a level's rows together) and of U, kept for the solves' analysis */
  HipFree(f->rlevL); HipFree(f->rlevU); f->rlevL = f->rlevU = NULL;
  ierr = ilu0_row_levels(n, bi, bj, bdiag, &f->rlevL, &f->nlevL, &f->rlevU, &f->nlevU);
  if (!ierr) { ps.nlev = f->nlevL; ierr = level_order(n, f->rlevL, f->nlevL, &ps.levptr, &ps.rows); }
  if (ierr) { HipFree(adiag); CHKERRQ(ierr); }
  SETUP_TICK("factor: levels of L and U");
  ierr = PetscMalloc(sizeof(PetscScalar *) * (size_t)ps.nth, &ps.rtmp);CHKERRQ(ierr);
  for (int t = 0; t < ps.nth; t++) { ps.rtmp[t] = (PetscScalar *)calloc((size_t)n + 1, sizeof(PetscScalar)); if (!ps.rtmp[t]) SETERRQ(HipObjComm(A), PETSC_ERR_MEM, "out of memory"); }
  for (PetscInt bb = 0; bb < nblk; bb++) {
    PetscInt nshift = 0;
    ps.r0 = blk[bb]; ps.r1 = blk[bb + 1]; ps.shift_amount = 0.0;
    for (;;) {   /* MAT_SHIFT_NONZERO, PCILU's default on a SeqAIJ matrix (ilu.c:387): a pivot that fails MatPivotCheck_nz restarts the
                  * factorisation with the diagonal shifted by shiftamount, then by twice that, ... (aijfact.c:507-592) */
      ps.fail_row = -1; ps.fail_level = -1;
      ierr = ilu0_run_pass(&ps);
      if (ierr) break;
      if (ps.fail_row < 0) break;
      if (!shift_nz) { ierr = PETSC_ERR_ARG_WRONG; break; }
      ps.shift_amount = nshift ? ps.shift_amount * 2.0 : shiftamount;
      nshift++;
      if (nshift > 80) { ierr = PETSC_ERR_ARG_WRONG; break; }
    }
    if (ierr) break;
    f->nshift = PetscMax(f->nshift, nshift);
  }
  SETUP_TICK("factor: numeric passes");
  /* returning 16 dense work rows (2 GB of touched pages at 16.7 M rows) to the system takes 0.12 s: off the caller's path */
  { void **junk = (void **)malloc(sizeof(void *) * (size_t)(ps.nth + 4)); int k = 0;
    if (junk) {
      for (int t = 0; t < ps.nth; t++) junk[k++] = ps.rtmp[t];
      junk[k++] = ps.levptr; junk[k++] = ps.rows; junk[k++] = adiag; junk[k] = NULL;
      HipFactorJoinHelpers();
      if (n < 200000 || pthread_create(&release_th, NULL, ilu0_release_thread, junk)) ilu0_release_thread(junk);
      else release_running = 1;
    } else { for (int t = 0; t < ps.nth; t++) free(ps.rtmp[t]); HipFree(ps.levptr); HipFree(ps.rows); HipFree(adiag); }
    HipFree(ps.rtmp); }
  SETUP_TICK("factor: work arrays released");
  if (ierr) SETERRQ(HipObjComm(A), 71 /* PETSC_ERR_MAT_LU_ZRPVT */, "Zero pivot row %d value %g%s", ps.fail_row, ps.fail_value, shift_nz ? ": still there after 80 diagonal shifts" : "");
  return 0;
}
#endif

typedef struct { const PetscInt *bi, *bdiag; const PetscScalar *ba; PetscInt *rlL, *rpU, *rlU; PetscScalar *dinv; } RowArr;
static void ilu0_row_arrays(void *c_, PetscInt lo, PetscInt hi) {
  RowArr *c = (RowArr *)c_;
  for (PetscInt i = lo; i < hi; i++) {
    c->rlL[i] = c->bi[i + 1] - c->bi[i];
    c->rpU[i] = c->bdiag[i + 1] + 1; c->rlU[i] = c->bdiag[i] - c->bdiag[i + 1] - 1; c->dinv[i] = c->ba[c->bdiag[i]];
  }
}
/* dependency levels of the two triangular factors, the sync-free plans, the level lists: everything MatSolve needs, from the
 * host factor in f->bi / bj / bdiag / ba (the reference's layout, whoever computed it) */
static PetscErrorCode ilu0_analyse_and_upload(Mat F, Mat A) {
  PetscErrorCode ierr;
  HipTriFactors *f = HipTriGet(F);
  const PetscInt n = f->n, *bi = f->bi, *bj = f->bj, *bdiag = f->bdiag; const PetscScalar *ba = f->ba;
  PetscDeviceCtx *dc;
  PetscInt *lev, *levU, *rowsL = NULL, *rowsU = NULL;
  const double ta0 = wall_s();
  if (f->rlevL && f->rlevU) { lev = f->rlevL; levU = f->rlevU; f->rlevL = f->rlevU = NULL; }   /* the host factorisation's own analysis */
  else { ierr = ilu0_row_levels(n, bi, bj, bdiag, &lev, &f->nlevL, &levU, &f->nlevU);CHKERRQ(ierr); }
  if (getenv("PETSC_HIPMI355X_SETUP_TIMING")) fprintf(stderr, "[hipmi355x] ILU(0): row levels %.3f s\n", wall_s() - ta0);
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  {   /* sync-free solves: worth it as soon as the level launches would be a launch-bound chain */
    char mode[32] = "syncfree"; PetscBool set;
    ierr = PetscOptionsGetString(HipObjPrefix(F), "-pc_factor_hipmi355x_trisolve", mode, sizeof(mode), &set);CHKERRQ(ierr);
    if (strcmp(mode, "syncfree") && strcmp(mode, "level")) SETERRQ(HipObjComm(F), PETSC_ERR_ARG_WRONG, "-pc_factor_hipmi355x_trisolve <syncfree|level>, got %s", mode);
    if (!strcmp(mode, "syncfree") && n > 0 && (f->nlevL + f->nlevU > 16 || set)) {
      PetscInt *rpU, *rlU, *rlL; PetscScalar *dinv;
      ierr = PetscMalloc(sizeof(PetscInt) * (size_t)n, &rpU);CHKERRQ(ierr);
      ierr = PetscMalloc(sizeof(PetscInt) * (size_t)n, &rlU);CHKERRQ(ierr);
      ierr = PetscMalloc(sizeof(PetscInt) * (size_t)n, &rlL);CHKERRQ(ierr);
      ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)n, &dinv);CHKERRQ(ierr);
      { RowArr ra = {bi, bdiag, ba, rlL, rpU, rlU, dinv};
        HipParallelRanges(n, ilu0_row_arrays, &ra); }
      /* -pc_factor_hipmi355x_trisolve_order <column|level>.  column: every row is summed in column order, the bits of
       * MatSolve_SeqAIJ_NaturalOrdering.  level: in the order of its dependencies' levels (a row then waits on its last
       * entries only) -- the default where the reference does not run the natural-ordering routine either: a matrix with
       * inodes, whose factor it solves with MatSolve_SeqAIJ_Inode (inode.c; MatLUFactorNumeric_SeqAIJ_Inode installs it),
       * in yet another order.  There the two agree to rounding. */
      char ord[32] = "", nodeopt[16] = ""; PetscInt nodes = 0; const PetscInt *nsizes = NULL; int by_level, rc = 1; PetscBool nset;
      ierr = PetscOptionsGetString(HipObjPrefix(F), "-pc_factor_hipmi355x_trisolve_order", ord, sizeof(ord), &set);CHKERRQ(ierr);
      if (set && strcmp(ord, "column") && strcmp(ord, "level")) SETERRQ(HipObjComm(F), PETSC_ERR_ARG_WRONG, "-pc_factor_hipmi355x_trisolve_order <column|level>, got %s", ord);
      ierr = MatSeqAIJHIPGetInodes(A, &nodes, &nsizes);CHKERRQ(ierr);
      if (getenv("PETSC_HIPMI355X_SETUP_TIMING")) fprintf(stderr, "[hipmi355x] ILU(0): ... row arrays + inode query at %.3f s\n", wall_s() - ta0);
      ierr = PetscOptionsGetString(HipObjPrefix(F), "-pc_factor_hipmi355x_trisolve_nodes", nodeopt, sizeof(nodeopt), &nset);CHKERRQ(ierr);
      f->nodes = 0;
      if (nodes > 0 && !(nset && (!strcmp(nodeopt, "0") || !strcmp(nodeopt, "false")))) {
        /* The factor of a matrix with inodes: the reference solves it node by node (MatSolve_SeqAIJ_Inode, inode.c:2327-2760;
         * MatLUFactorNumeric_SeqAIJ_Inode installs it), and so does the device: one lane per NODE, dependency levels over nodes
         * (a node's rows were consecutive levels of the row-granular analysis), the shared column list walked once per node, two
         * columns at a time as the reference routine does.  -pc_factor_hipmi355x_trisolve_order column: in column order -- the
         * reference routine's bits; level (the default for such factors, as for the row-granular plans): oldest dependency
         * first -- agreement to rounding, 1.6x faster on the FEM stand-in (profiles/r03_ilu_fem_nodes.log).
         * -pc_factor_hipmi355x_trisolve_nodes 0 keeps the row-granular plans. */
        PetscInt *nstart, *nodeof, *nlevL, *nlevU, nlL = 0, nlU = 0;
        ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(nodes + 1), &nstart);CHKERRQ(ierr);
        ierr = PetscMalloc(sizeof(PetscInt) * (size_t)n, &nodeof);CHKERRQ(ierr);
        ierr = PetscMalloc(sizeof(PetscInt) * (size_t)nodes, &nlevL);CHKERRQ(ierr);
        ierr = PetscMalloc(sizeof(PetscInt) * (size_t)nodes, &nlevU);CHKERRQ(ierr);
        nstart[0] = 0;
        for (PetscInt u = 0; u < nodes; u++) { nstart[u + 1] = nstart[u] + nsizes[u]; for (PetscInt r = nstart[u]; r < nstart[u + 1] && r < n; r++) nodeof[r] = u; }
        if (nstart[nodes] == n) {
          for (PetscInt u = 0; u < nodes; u++) {                     /* a node may start once the nodes its FIRST row references are done */
            PetscInt l = 0; const PetscInt r0 = nstart[u];
            for (PetscInt q = bi[r0]; q < bi[r0 + 1]; q++) l = PetscMax(l, nlevL[nodeof[bj[q]]] + 1);
            nlevL[u] = l; nlL = PetscMax(nlL, l + 1);
          }
          for (PetscInt u = nodes - 1; u >= 0; u--) {                /* upper: the columns of the node's LAST row */
            PetscInt l = 0; const PetscInt rL = nstart[u + 1] - 1;
            for (PetscInt q = 0; q < rlU[rL]; q++) l = PetscMax(l, nlevU[nodeof[bj[rpU[rL] + q]]] + 1);
            nlevU[u] = l; nlU = PetscMax(nlU, l + 1);
          }
          by_level = set ? !strcmp(ord, "level") : 1;
          if (getenv("PETSC_HIPMI355X_SETUP_TIMING")) fprintf(stderr, "[hipmi355x] ILU(0): ... node levels at %.3f s\n", wall_s() - ta0);
          /* on request first with whole dependency nodes as columns (a fixed number of dofs per node: one gather per dependency
           * node; measured no faster, so not the default), then the general node plans, then (below) row by row */
          PetscInt bc = 0; PetscBool bset;
          ierr = PetscOptionsGetInt(HipObjPrefix(F), "-pc_factor_hipmi355x_trisolve_block_columns", &bc, &bset);CHKERRQ(ierr);
          for (int blk = bc ? 1 : 0; blk >= 0; blk--) {
            rc = mi355x_trisolve_plan_create_nodes_pair(dc->h, n, nodes, nstart, by_level, blk, nlL, nlevL, bi, rlL, nlU, nlevU, rpU, rlU, bj, ba, dinv, &f->tri_lo, &f->tri_up);
            if (!rc) { f->nodes = nodes; f->nlevL_nodes = nlL; f->nlevU_nodes = nlU; f->by_level = by_level; f->block_columns = blk; break; }
            if (f->tri_lo) mi355x_trisolve_plan_destroy(f->tri_lo);      /* not of that shape: the next, more general form */
            if (f->tri_up) mi355x_trisolve_plan_destroy(f->tri_up);
            f->tri_lo = f->tri_up = NULL;
          }
        }
        HipFree(nstart); HipFree(nodeof); HipFree(nlevL); HipFree(nlevU);
      }
      if (rc) {
        by_level = set ? !strcmp(ord, "level") : (nodes > 0);
        f->by_level = by_level;
        rc = mi355x_trisolve_plan_create_pair(dc->h, n, by_level, f->nlevL, lev, bi, rlL, bj, ba, f->nlevU, levU, rpU, rlU, bj, ba, dinv, NULL, &f->tri_lo, &f->tri_up);
      }
      HipFree(rpU); HipFree(rlU); HipFree(rlL); HipFree(dinv);
      if (getenv("PETSC_HIPMI355X_SETUP_TIMING")) fprintf(stderr, "[hipmi355x] ILU(0): ... plans made at %.3f s\n", wall_s() - ta0);
      if (rc) {   /* e.g. a factor too large for 32-bit sliced-ELL offsets: the level kernels serve */
        if (f->tri_lo) mi355x_trisolve_plan_destroy(f->tri_lo);
        if (f->tri_up) mi355x_trisolve_plan_destroy(f->tri_up);
        f->tri_lo = f->tri_up = NULL;
      } else HipTriWatchAdd(f);
    }
  }
  if (!f->tri_lo) {   /* the level-scheduled kernels work on the reference's layout itself, rows listed level by level */
    ierr = level_order(n, lev, f->nlevL, &f->levptrL, &rowsL);CHKERRQ(ierr);
    ierr = level_order(n, levU, f->nlevU, &f->levptrU, &rowsU);CHKERRQ(ierr);
    CHKHIP(mi355x_malloc((void **)&f->d_bi, sizeof(PetscInt) * (size_t)(n + 1)));
    CHKHIP(mi355x_malloc((void **)&f->d_bj, sizeof(PetscInt) * (size_t)(f->nz + 1)));
    CHKHIP(mi355x_malloc((void **)&f->d_bdiag, sizeof(PetscInt) * (size_t)(n + 1)));
    CHKHIP(mi355x_malloc((void **)&f->d_ba, sizeof(PetscScalar) * (size_t)(f->nz + 1)));
    CHKHIP(mi355x_malloc((void **)&f->d_rowsL, sizeof(PetscInt) * (size_t)PetscMax(n, 1)));
    CHKHIP(mi355x_malloc((void **)&f->d_rowsU, sizeof(PetscInt) * (size_t)PetscMax(n, 1)));
    CHKHIP(mi355x_memcpy_h2d(dc->h, f->d_bi, bi, sizeof(PetscInt) * (size_t)(n + 1)));
    CHKHIP(mi355x_memcpy_h2d(dc->h, f->d_bj, bj, sizeof(PetscInt) * (size_t)(f->nz + 1)));
    CHKHIP(mi355x_memcpy_h2d(dc->h, f->d_bdiag, bdiag, sizeof(PetscInt) * (size_t)(n + 1)));
    CHKHIP(mi355x_memcpy_h2d(dc->h, f->d_ba, ba, sizeof(PetscScalar) * (size_t)(f->nz + 1)));
    CHKHIP(mi355x_memcpy_h2d(dc->h, f->d_rowsL, rowsL, sizeof(PetscInt) * (size_t)n));
    CHKHIP(mi355x_memcpy_h2d(dc->h, f->d_rowsU, rowsU, sizeof(PetscInt) * (size_t)n));
    CHKHIP(mi355x_handle_synchronize(dc->h));
  }
  HipFree(rowsL); HipFree(rowsU); HipFree(lev); HipFree(levU);
  return 0;
}

static PetscErrorCode MatSolve_SeqAIJHIP_ILU(Mat F, Vec b, Vec x);

static PetscErrorCode MatLUFactorNumeric_SeqAIJHIP(Mat F, Mat A, const MatFactorInfo *info) {   /* MatLUFactorNumeric_SeqAIJCUSPARSE, aijcusparse.cu:358-376 */
  PetscErrorCode ierr;
  HipTriFactors *f = HipTriGet(F);
  const double t0 = wall_s();
  if (A->rmap->n != A->cmap->n) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_WRONG, "Must be square matrix, rows %d columns %d", A->rmap->n, A->cmap->n);
  if (f->factored_state == HipObjState(A) && f->factored_of == (void *)A && (f->tri_lo || f->d_ba)) return 0;   /* same operator, same values: nothing to redo */
  tri_reset_numeric(f);
#if defined(PETSCHIPMI355X_WITH_PETSC)
  ierr = MatLUFactorNumeric_SeqAIJ(F, A, info);CHKERRQ(ierr);            /* the parent's factorisation into F's own Mat_SeqAIJ */
  { Mat_SeqAIJ *b = (Mat_SeqAIJ *)F->data;
    f->n = A->rmap->n; f->nz = b->nz; f->bi = b->i; f->bj = b->j; f->bdiag = b->diag; f->ba = b->a; f->owns_host = PETSC_FALSE; }
#else
  ierr = ilu0_factor_host(F, A, info);CHKERRQ(ierr);
#endif
  const double t1 = wall_s();
  ierr = ilu0_analyse_and_upload(F, A);CHKERRQ(ierr);
  if (getenv("PETSC_HIPMI355X_SETUP_TIMING")) fprintf(stderr, "[hipmi355x] ILU(0) n=%d: host factorisation %.3f s, level analysis + plans + upload %.3f s\n", (int)f->n, t1 - t0, wall_s() - t1);
  F->ops->solve = MatSolve_SeqAIJHIP_ILU;                                 /* aijcusparse.cu:372-373 */
  f->factored_state = HipObjState(A); f->factored_of = (void *)A;
  return 0;
}

static PetscErrorCode MatILUFactorSymbolic_SeqAIJHIP(Mat F, Mat A, IS row, IS col, const MatFactorInfo *info) {   /* aijcusparse.cu:175-186 */
  PetscErrorCode ierr = natural_ordering_only(A, row, col, info, "ILU");CHKERRQ(ierr);
  if (strcmp(HipObjTypeName(A), MATSEQAIJHIPMI355X)) SETERRQ(HipObjComm(A), PETSC_ERR_SUP, "ILU on the device needs a sequential AIJ matrix of this type (use -pc_type bjacobi -sub_pc_type ilu in parallel); got %s", HipObjTypeName(A));
#if defined(PETSCHIPMI355X_WITH_PETSC)
  ierr = MatILUFactorSymbolic_SeqAIJ(F, A, row, col, info);CHKERRQ(ierr);
#endif
  HipTriGet(F)->factored_state = -1;
  F->ops->lufactornumeric = MatLUFactorNumeric_SeqAIJHIP;
  return 0;
}

/* y = U^-1 L^-1 b through whatever the analysis prepared.  After a sync-free application gave up (HipTriWatchCheck), the same
 * plans run level by level. */
PetscErrorCode HipTriFactorsApply(Mat F, HipTriFactors *f, Vec b, Vec x, PetscLogDouble flops) {
  PetscErrorCode ierr;
  const PetscScalar *db; PetscScalar *dx; PetscDeviceCtx *dc;
  int rc;
  if (!f->n) return 0;
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  ierr = VecHIPGetRead(b, &db);CHKERRQ(ierr);
  ierr = VecHIPGetWrite(x, &dx);CHKERRQ(ierr);
  if (f->use_levels) rc = mi355x_trisolve_apply_levels(dc->h, f->tri_lo, f->tri_up, db, dx);
  else {
    rc = mi355x_trisolve_apply(dc->h, f->tri_lo, f->tri_up, db, dx);
    if (rc == 719) {   /* hipErrorLaunchFailure: an earlier application gave up and no host wait has noticed yet: this one runs level by level */
      f->use_levels = 1; f->aborted = 1;
      rc = mi355x_trisolve_apply_levels(dc->h, f->tri_lo, f->tri_up, db, dx);
    } else if (rc == 1 || rc == 701 || rc == 720) {   /* hipErrorInvalidValue / LaunchOutOfResources / cooperative too large: the sync-free kernels could not be launched on this
                                                        * device (LDS, registers): nothing ran, so the same plans serve one launch per level from now on */
      f->use_levels = 1;
      rc = mi355x_trisolve_apply_levels(dc->h, f->tri_lo, f->tri_up, db, dx);
    }
  }
  ierr = VecHIPRestoreWrite(x);CHKERRQ(ierr);      /* also on the error path: x is not left in write state */
  HipStateIncrease(x);
  CHKHIP(rc);
  ierr = PetscLogFlops(flops);CHKERRQ(ierr);
  (void)F;
  return 0;
}

static PetscErrorCode MatSolve_SeqAIJHIP_ILU(Mat F, Vec b, Vec x) {   /* MatSolve_SeqAIJCUSPARSE_NaturalOrdering, aijcusparse.cu:419-445 */
  PetscErrorCode ierr;
  HipTriFactors *f = HipTriGet(F);
  const PetscScalar *db; PetscScalar *dx; PetscDeviceCtx *dc;
  if (f->tri_lo) return HipTriFactorsApply(F, f, b, x, 2.0 * f->nz - f->n);
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  ierr = VecHIPGetRead(b, &db);CHKERRQ(ierr);
  ierr = VecHIPGetWrite(x, &dx);CHKERRQ(ierr);
  /* The level launches are a launch-bound inner loop (766 + 766 kernels for P7(256)): with more than a handful of levels they
   * are captured once into a hipGraph that works in place on a fixed buffer and replayed per application (copy in, one graph
   * launch, copy out).  Same kernels, same order, same bits. */
  if (!f->graph_tried && f->nlevL + f->nlevU > 16) {
    f->graph_tried = 1;
    if (!mi355x_malloc((void **)&f->d_work, sizeof(PetscScalar) * (size_t)PetscMax(f->n, 1)) && !mi355x_graph_capture_begin(dc->h)) {
      int bad = 0;
      for (PetscInt l = 0; l < f->nlevL && !bad; l++)
        bad = mi355x_ilu0_lower_level(dc->h, f->levptrL[l + 1] - f->levptrL[l], f->d_rowsL + f->levptrL[l], f->d_bi, f->d_bj, f->d_ba, f->d_work, f->d_work);
      for (PetscInt l = 0; l < f->nlevU && !bad; l++)
        bad = mi355x_ilu0_upper_level(dc->h, f->levptrU[l + 1] - f->levptrU[l], f->d_rowsU + f->levptrU[l], f->d_bj, f->d_ba, f->d_bdiag, f->d_work);
      void *g = NULL;
      if (mi355x_graph_capture_end(dc->h, &g) || bad) g = NULL;   /* capture failed: stay with plain launches */
      f->graph = g;
    }
  }
  int rc = 0;
  if (f->graph) {
    rc = mi355x_vec_copy(dc->h, (size_t)f->n, db, f->d_work);
    if (!rc) rc = mi355x_graph_launch(dc->h, f->graph);
    if (!rc) rc = mi355x_vec_copy(dc->h, (size_t)f->n, f->d_work, dx);
  } else {
    for (PetscInt l = 0; l < f->nlevL && !rc; l++)
      rc = mi355x_ilu0_lower_level(dc->h, f->levptrL[l + 1] - f->levptrL[l], f->d_rowsL + f->levptrL[l], f->d_bi, f->d_bj, f->d_ba, db, dx);
    for (PetscInt l = 0; l < f->nlevU && !rc; l++)
      rc = mi355x_ilu0_upper_level(dc->h, f->levptrU[l + 1] - f->levptrU[l], f->d_rowsU + f->levptrU[l], f->d_bj, f->d_ba, f->d_bdiag, dx);
  }
  ierr = VecHIPRestoreWrite(x);CHKERRQ(ierr);
  HipStateIncrease(x);
  CHKHIP(rc);
  ierr = PetscLogFlops(2.0 * f->nz - f->n);CHKERRQ(ierr);
  return 0;
}

/* ---------------------------------------------------------------- MatGetFactor */
#if defined(PETSCHIPMI355X_WITH_PETSC)
static PetscErrorCode MatDestroy_Factor_SeqAIJHIP(Mat F) {   /* the factor's device state first, spptr zeroed, then the parent's destroy (aijcusp.cu:584-586) */
  HipTriFactors *f = HipTriGet(F);
  PetscErrorCode (*parent)(Mat) = f ? f->parent_destroy : NULL;
  PetscErrorCode ierr = HipTriFactorsDestroy(&f);CHKERRQ(ierr);
  F->spptr = 0;
  if (parent) { ierr = (*parent)(F);CHKERRQ(ierr); }
  return 0;
}
#endif

PetscErrorCode MatGetFactor_seqaijhipmi355x_petsc(Mat A, MatFactorType ftype, Mat *B) {   /* MatGetFactor_seqaij_cusparse, aijcusparse.cu:57-75 */
  PetscErrorCode ierr;
  HipTriFactors *f;
  if (ftype != MAT_FACTOR_ILU && ftype != MAT_FACTOR_ICC) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_SUP, "Factor type not supported for HIPMI355X matrix types (ILU(0) and ICC(0) are)");
  ierr = PetscMalloc(sizeof(*f), &f);CHKERRQ(ierr);
  memset(f, 0, sizeof(*f));
  f->kind = ftype; f->factored_state = -1;
#if defined(PETSCHIPMI355X_WITH_PETSC)
  ierr = MatGetFactor_seqaij_petsc(A, ftype, B);CHKERRQ(ierr);           /* the parent's factor matrix (MATSEQAIJ / MATSEQSBAIJ) with its host routines */
  f->parent_destroy = (*B)->ops->destroy;
  (*B)->spptr = f;
  (*B)->ops->destroy = MatDestroy_Factor_SeqAIJHIP;
#else
  { const PetscInt n = A->rmap->n;
    ierr = MatCreate(HipObjComm(A), B);CHKERRQ(ierr);
    ierr = MatSetSizes(*B, n, n, n, n);CHKERRQ(ierr);
    ierr = MatSetType(*B, MATSEQAIJHIPMI355X);CHKERRQ(ierr);
    ((Mat_SeqAIJHIP *)(*B)->spptr)->tri = f; }
#endif
  (*B)->ops->ilufactorsymbolic = MatILUFactorSymbolic_SeqAIJHIP;
  (*B)->ops->iccfactorsymbolic = MatICCFactorSymbolic_SeqAIJHIP;
  (*B)->factortype = ftype;
  ierr = PetscObjectComposeFunction((PetscObject)*B, "MatFactorSetIndependentBlocks_C", "MatFactorSetIndependentBlocks_SeqAIJHIP", (PetscVoidFunction)MatFactorSetIndependentBlocks_SeqAIJHIP);CHKERRQ(ierr);
  return 0;
}
PetscErrorCode MatGetFactorAvailable_seqaijhipmi355x_petsc(Mat A, MatFactorType ftype, PetscBool *flg) {   /* MatGetFactorAvailable_seqaij_petsc, aijfact.c:97 */
  (void)A;
  *flg = (PetscBool)(ftype == MAT_FACTOR_ILU || ftype == MAT_FACTOR_ICC);
  return 0;
}

/* ---------------------------------------------------------------- introspection (tests / DESIGN.md), through the PC's public face */
static PetscErrorCode pc_factors(PC pc, MatFactorType kind, HipTriFactors **f) {
  PetscErrorCode ierr;
  Mat F = NULL;
  ierr = PCFactorGetMatrix(pc, &F);CHKERRQ(ierr);
  if (!F || !F->factortype || !HipTriGet(F) || HipTriGet(F)->kind != kind) SETERRQ(HipObjComm(pc), PETSC_ERR_ARG_WRONG, "not a set-up %s over a HIPMI355X factored matrix", kind == MAT_FACTOR_ILU ? "PCILU" : "PCICC");
  *f = HipTriGet(F);
  return 0;
}
/* 1 when MatSolve runs the sync-free solves (two launches), 0 for the level-scheduled kernels; *aborted: a dependency wait gave up */
PetscErrorCode PCILUGetSolver_HIPMI355X(PC pc, PetscInt *syncfree, PetscInt *aborted) {
  HipTriFactors *f;
  PetscErrorCode ierr = pc_factors(pc, MAT_FACTOR_ILU, &f);CHKERRQ(ierr);
  int a = 0, b = 0;
  if (syncfree) *syncfree = (f->tri_lo && !f->use_levels) ? 1 : 0;
  if (f->tri_lo) {
    PetscDeviceCtx *dc;
    ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
    CHKHIP(mi355x_handle_synchronize(dc->h));      /* the flag of everything queued so far */
    mi355x_trisolve_aborted(f->tri_lo, &a); mi355x_trisolve_aborted(f->tri_up, &b);
  }
  if (aborted) *aborted = a || b || f->aborted;
  return 0;
}
/* restarts of the factorisation with a larger diagonal shift (MAT_SHIFT_NONZERO); 0 for every matrix whose pivots pass */
PetscErrorCode PCILUGetShiftCount_HIPMI355X(PC pc, PetscInt *nshift) {
  HipTriFactors *f;
  PetscErrorCode ierr = pc_factors(pc, MAT_FACTOR_ILU, &f);CHKERRQ(ierr);
  *nshift = f->nshift;
  return 0;
}
/* node-blocked solves: the number of nodes the plans hold (0: row-granular) and the dependency levels over nodes */
PetscErrorCode PCILUGetNodeInfo_HIPMI355X(PC pc, PetscInt *nodes, PetscInt *nlevL, PetscInt *nlevU) {
  HipTriFactors *f;
  PetscErrorCode ierr = pc_factors(pc, MAT_FACTOR_ILU, &f);CHKERRQ(ierr);
  if (nodes) *nodes = f->nodes ? (f->block_columns ? -f->nodes : f->nodes) : 0;   /* negative: block-column plans */
  if (nlevL) *nlevL = f->nlevL_nodes;
  if (nlevU) *nlevU = f->nlevU_nodes;
  return 0;
}
/* levels of the two triangular solves */
PetscErrorCode PCILUGetLevels_HIPMI355X(PC pc, PetscInt *nlevL, PetscInt *nlevU) {
  HipTriFactors *f;
  PetscErrorCode ierr = pc_factors(pc, MAT_FACTOR_ILU, &f);CHKERRQ(ierr);
  if (nlevL) *nlevL = f->nlevL;
  if (nlevU) *nlevU = f->nlevU;
  return 0;
}
/* dependency levels of the two sweeps and the number of positive-definite shifts the factorisation took */
PetscErrorCode PCICCGetInfo_HIPMI355X(PC pc, PetscInt *nlevL, PetscInt *nlevU, PetscInt *nshift) {
  HipTriFactors *f;
  PetscErrorCode ierr = pc_factors(pc, MAT_FACTOR_ICC, &f);CHKERRQ(ierr);
  if (nlevL) *nlevL = f->nlevL;
  if (nlevU) *nlevU = f->nlevU;
  if (nshift) *nshift = f->nshift;
  return 0;
}
/* tests: make the next host wait see an aborted sync-free solve on this PC's factor */
PetscErrorCode PCFactorDebugSetAborted_HIPMI355X(PC pc) {
  PetscErrorCode ierr;
  Mat F = NULL;
  ierr = PCFactorGetMatrix(pc, &F);CHKERRQ(ierr);
  if (!F || !HipTriGet(F) || !HipTriGet(F)->tri_lo) SETERRQ(HipObjComm(pc), PETSC_ERR_ARG_WRONGSTATE, "no sync-free plans");
  CHKHIP(mi355x_trisolve_debug_set_aborted(HipTriGet(F)->tri_lo, 1));
  return 0;
}
