/* Factored matrices of MATSEQAIJHIPMI355X, ICC(0) part (SURVEY 8f.1 names ILU(0)/ICC(0)): the incomplete Cholesky factorisation the
 * reference offers for symmetric positive definite systems, i.e. the natural partner of KSPCG under PCBJACOBI.  Reached through
 * MatGetFactor(A, "petsc", MAT_FACTOR_ICC, &F) -> MatICCFactorSymbolic -> MatCholeskyFactorNumeric -> MatSolve (ilu.c has the
 * MatGetFactor side), i.e. by an unchanged PCICC.
 *   numeric : MatICCFactorSymbolic_SeqAIJ with levels 0 and natural ordering (the pattern of A's upper triangle, the diagonal last in
 *             its row, src/mat/impls/aij/seq/aijfact.c:2405-2600) + MatCholeskyFactorNumeric_SeqAIJ (aijfact.c:2076-2230) with
 *             PCICC's defaults (src/ksp/pc/impls/factor/icc/icc.c:189-200: MAT_SHIFT_POSITIVE_DEFINITE, zeropivot 100 eps), on the
 *             HOST copy of the matrix: the parent class's routines inside a PETSc tree, their restatement below on the harness;
 *             then the two triangular systems in row form and one upload.
 *   solve   : MatSolve_SeqSBAIJ_1_NaturalOrdering (src/mat/impls/sbaij/seq/sbaijfact2.c:1977-2015) on the device with the sync-free
 *             solves of csrc/trisolve.hip.  The reference sweeps U^T by COLUMNS (x[col] += v * x_i for the entries of row i, rows in
 *             ascending order): entry (i, c) therefore reaches x[c] after every earlier row's -- which is the order a row-oriented
 *             solve with L = U^T adds them in.  x[c] += v t is the bits of x[c] -= (-v) t, so the plans hold the negated values; the
 *             1/D(i) between the two sweeps is the upper solve's right-hand-side factor; its rows are stored last entry first, as the
 *             reference's backward loop reads them.  Same bits as the host loop. */
#include "hipmi355ximpl.h"
#if defined(PETSCHIPMI355X_WITH_PETSC)
#include <../src/mat/impls/aij/seq/aij.h>
#include <../src/mat/impls/sbaij/seq/sbaij.h>
#endif

static PetscErrorCode MatSolve_SeqAIJHIP_ICC(Mat F, Vec b, Vec x) {   /* PCApply_ICC (icc.c:65) -> MatSolve(fact, x, y) */
  HipTriFactors *f = HipTriGet(F);
  return HipTriFactorsApply(F, f, b, x, 4.0 * f->nz - 3.0 * f->n);
}

static PetscErrorCode icc0_plans(Mat F, PetscInt n, const PetscInt *ui, const PetscInt *uj, const PetscScalar *ua);
#include <time.h>
static double icc_wall_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
#define ICC_TICK(what) do { if (getenv("PETSC_HIPMI355X_SETUP_TIMING")) { const double t__ = icc_wall_s(); fprintf(stderr, "[hipmi355x]   %-34s %.3f s\n", what, t__ - tick0); tick0 = t__; } } while (0)

#if !defined(PETSCHIPMI355X_WITH_PETSC)
typedef struct { const PetscInt *ai, *aj; PetscInt *ui, *uj; volatile PetscInt missing; } IccSym;
static void icc0_sym_count(void *c_, PetscInt lo, PetscInt hi) {   /* ui[k + 1] = strictly upper entries of row k + the diagonal slot; -1: no diagonal entry */
  IccSym *c = (IccSym *)c_;
  for (PetscInt k = lo; k < hi; k++) {
    PetscInt cnt = 1; PetscBool hasd = PETSC_FALSE;
    for (PetscInt q = c->ai[k]; q < c->ai[k + 1]; q++) { if (c->aj[q] > k) cnt++; else if (c->aj[q] == k) hasd = PETSC_TRUE; }
    c->ui[k + 1] = hasd ? cnt : -1;
    if (!hasd) c->missing = k;
  }
}
static void icc0_sym_fill(void *c_, PetscInt lo, PetscInt hi) {    /* row k = its strictly upper entries in column order, then the diagonal slot */
  IccSym *c = (IccSym *)c_;
  for (PetscInt k = lo; k < hi; k++) {
    PetscInt w = c->ui[k];
    for (PetscInt q = c->ai[k]; q < c->ai[k + 1]; q++) if (c->aj[q] > k) c->uj[w++] = c->aj[q];
    c->uj[w] = k;
  }
}
/* MatICCFactorSymbolic_SeqAIJ (levels 0, natural ordering) + MatCholeskyFactorNumeric_SeqAIJ restated for the harness; inside a
 * PETSc tree the parent's routines run instead and leave the same arrays in F's Mat_SeqSBAIJ */
static PetscErrorCode icc0_factor_host(Mat F, Mat A, const MatFactorInfo *info) {
  PetscErrorCode ierr;
  HipTriFactors *f = HipTriGet(F);
  PetscInt n; const PetscInt *ai, *aj; const PetscScalar *aa;
  double tick0 = icc_wall_s();
  ierr = MatSeqAIJGetArrays(A, &n, &ai, &aj, &aa);CHKERRQ(ierr);
  f->n = n;
  if (!n) return 0;

  /* ---- symbolic: row k = its strictly upper entries in column order, then the diagonal slot ---- */
  PetscInt *ui, *uj, nz = 0; PetscScalar *ua;
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(n + 1), &ui);CHKERRQ(ierr);
  { IccSym sy = {ai, aj, ui, NULL, -1};                    /* per-row counts and copies on host threads, the running sum in between */
    HipParallelRanges(n, icc0_sym_count, &sy);
    if (sy.missing >= 0) {
      PetscInt first = 0;
      while (first < n && ui[first + 1] >= 0) first++;
      HipFree(ui);
      SETERRQ(HipObjComm(A), PETSC_ERR_ARG_WRONGSTATE, "Matrix is missing diagonal entry %d", first);
    }
    ui[0] = 0;
    for (PetscInt k = 0; k < n; k++) ui[k + 1] += ui[k];
    nz = ui[n];
    ierr = PetscMalloc(sizeof(PetscInt) * (size_t)nz, &uj);CHKERRQ(ierr);
    ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)nz, &ua);CHKERRQ(ierr);
    f->nz = nz;
    sy.uj = uj;
    HipParallelRanges(n, icc0_sym_fill, &sy); }

  ICC_TICK("icc: pattern of U");
  /* ---- numeric, left-looking over the rows; the shift of MatPivotCheck_pd (matimpl.h:532-553) restarts it.  One pass per
   * independent block (one block = the whole matrix unless "PCFactorSetIndependentBlocks_C" said otherwise) ---- */
  const PetscReal zeropivot = info->zeropivot;
  const PetscBool shift_pd = (PetscBool)(info->shifttype == (PetscReal)MAT_SHIFT_POSITIVE_DEFINITE);
  const PetscInt nshift_max = 5;
  PetscScalar *work; PetscInt *first, *list;
  ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)n, &work);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)n, &first);CHKERRQ(ierr);   /* first[i]: first entry of row i not yet folded into a later row */
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)n, &list);CHKERRQ(ierr);    /* list[c]: chain of the earlier rows whose next entry is in column c */
  const PetscInt whole[2] = {0, n};
  const PetscInt nblk = (f->nblk > 0 && f->blk[f->nblk] == n) ? f->nblk : 1, *blk = (f->nblk > 0 && f->blk[f->nblk] == n) ? f->blk : whole;
  for (PetscInt bb = 0; bb < nblk; bb++) {
    const PetscInt r0 = blk[bb], r1 = blk[bb + 1];
    for (PetscInt k = r0; k < r1; k++)
      for (PetscInt q = ui[k]; q < ui[k + 1]; q++)
        if (uj[q] >= r1) { HipFree(work); HipFree(first); HipFree(list); HipFree(ui); HipFree(uj); HipFree(ua); SETERRQ(HipObjComm(A), PETSC_ERR_ARG_WRONG, "row %d couples to column %d outside its independent block", k, uj[q]); }
    PetscReal shift_top = zeropivot, shift_amount = 0.0, shift_fraction = 0.0, shift_lo = 0.0, shift_hi = 1.0;
    PetscInt nshift = 0;
    for (PetscInt i = r0; i < r1; i++) {
      PetscScalar d = 0.0; PetscReal rs;
      for (PetscInt q = ai[i]; q < ai[i + 1]; q++) if (aj[q] == i) d = aa[q];
      rs = -PetscAbsScalar(d) - d;
      for (PetscInt q = ai[i]; q < ai[i + 1]; q++) rs += PetscAbsScalar(aa[q]);
      if (rs > shift_top) shift_top = rs;
    }
    shift_top *= 1.1;
    PetscBool again;
    do {
      again = PETSC_FALSE;
      for (PetscInt i = r0; i < r1; i++) list[i] = n;
      if (r0 < r1) first[r0] = ui[r0];
      for (PetscInt k = r0; k < r1; k++) {
        const PetscInt dslot = ui[k + 1] - 1;
        for (PetscInt q = ui[k]; q <= dslot; q++) { work[uj[q]] = 0.0; ua[q] = 0.0; }
        for (PetscInt q = ai[k]; q < ai[k + 1]; q++) if (aj[q] >= k) work[aj[q]] = aa[q];
        work[k] += shift_amount;
        PetscScalar dk = work[k];
        for (PetscInt i = list[k]; i < k;) {
          const PetscInt nexti = list[i], at = first[i];
          const PetscScalar m = -ua[at] * ua[ui[i + 1] - 1];     /* -U(i,k)/D(i): what the solve multiplies with */
          dk += m * ua[at];
          ua[at] = m;
          if (at + 1 < ui[i + 1] - 1) {
            for (PetscInt q = at + 1; q < ui[i + 1] - 1; q++) work[uj[q]] += m * ua[q];
            first[i] = at + 1;
            const PetscInt c = uj[at + 1]; list[i] = list[c]; list[c] = i;
          }
          i = nexti;
        }
        PetscReal rs = 0.0;
        if (ui[k] < dslot) {
          for (PetscInt q = ui[k]; q < dslot; q++) { ua[q] = work[uj[q]]; rs += PetscAbsScalar(ua[q]); }
          first[k] = ui[k];
          const PetscInt c = uj[ui[k]]; list[k] = list[c]; list[c] = k;
        }
        if (dk <= zeropivot * rs) {
          if (!shift_pd) { HipFree(work); HipFree(first); HipFree(list); HipFree(ui); HipFree(uj); HipFree(ua); SETERRQ(HipObjComm(A), 72 /* PETSC_ERR_MAT_CH_ZRPVT */, "Zero pivot row %d value %g tolerance %g", k, (double)dk, (double)(zeropivot * rs)); }
          if (nshift == nshift_max) shift_fraction = shift_hi;
          else { shift_lo = shift_fraction; shift_fraction = (shift_hi + shift_lo) / 2.; }
          shift_amount = shift_fraction * shift_top;
          nshift++;
          if (nshift > nshift_max + 1) { HipFree(work); HipFree(first); HipFree(list); HipFree(ui); HipFree(uj); HipFree(ua); SETERRQ(HipObjComm(A), 71 /* PETSC_ERR_MAT_LU_ZRPVT */, "ICC(0): no positive pivot in row %d even with the full diagonal shift", k); }
          again = PETSC_TRUE;
          break;
        }
        ua[dslot] = 1.0 / dk;
      }
    } while (again);
    f->nshift = PetscMax(f->nshift, nshift);
  }
  HipFree(work); HipFree(first); HipFree(list);
  ICC_TICK("icc: numeric");
  ierr = icc0_plans(F, n, ui, uj, ua);
  ICC_TICK("icc: plans");
  HipFree(ui); HipFree(uj); HipFree(ua);
  CHKERRQ(ierr);
  return 0;
}
#endif

#include <pthread.h>
typedef struct { PetscInt n; const PetscInt *ui, *uj; const PetscScalar *ua; PetscInt *lp, *ll, *lj; PetscScalar *lv; PetscInt *up, *ul, *uc; PetscScalar *uv, *ones, *dinv; PetscInt *levU, nlevU; } IccT;
static void icc0_count_columns(void *c_, PetscInt lo, PetscInt hi) {          /* lp[c + 1] = entries in column c, for the columns [lo, hi) */
  IccT *t = (IccT *)c_;
  for (PetscInt i = 0; i < hi && i < t->n; i++) {
    const PetscInt q0 = t->ui[i], q1 = t->ui[i + 1] - 1;                      /* strictly upper entries: columns > i (in any order) */
    for (PetscInt q = q0; q < q1; q++) { const PetscInt c = t->uj[q]; if (c >= lo && c < hi) t->lp[c + 1]++; }
  }
}
static void icc0_fill_columns(void *c_, PetscInt lo, PetscInt hi) {
  IccT *t = (IccT *)c_;
  for (PetscInt i = 0; i < hi && i < t->n; i++) {
    const PetscInt q0 = t->ui[i], q1 = t->ui[i + 1] - 1;
    for (PetscInt q = q0; q < q1; q++) {
      const PetscInt c = t->uj[q];
      if (c >= lo && c < hi) { const PetscInt at = t->lp[c] + t->ll[c]++; t->lj[at] = i; t->lv[at] = -t->ua[q]; }
    }
  }
}
static void icc0_upper_rows(void *c_, PetscInt lo, PetscInt hi) {
  IccT *t = (IccT *)c_;
  for (PetscInt i = lo; i < hi; i++) {
    PetscInt w = t->ui[i] - i;                                                /* every earlier row left its diagonal slot behind */
    t->up[i] = w; t->ul[i] = t->ui[i + 1] - 1 - t->ui[i];
    for (PetscInt q = t->ui[i + 1] - 2; q >= t->ui[i]; q--) { t->uc[w] = t->uj[q]; t->uv[w] = -t->ua[q]; w++; }
    t->ones[i] = 1.0; t->dinv[i] = t->ua[t->ui[i + 1] - 1];
  }
}
static void *icc0_levels_U(void *c_) {
  IccT *t = (IccT *)c_; PetscInt nlev = 0;
  for (PetscInt i = t->n - 1; i >= 0; i--) {
    PetscInt l = 0;
    for (PetscInt q = t->ui[i]; q < t->ui[i + 1] - 1; q++) l = PetscMax(l, t->levU[t->uj[q]] + 1);
    t->levU[i] = l; nlev = PetscMax(nlev, l + 1);
  }
  t->nlevU = nlev;
  return NULL;
}

/* the two triangular systems in row form, negated values, from the factor U (rows: strictly upper entries, then the inverted
 * diagonal) in the reference's layout; dependency levels; the sync-free plans */
static PetscErrorCode icc0_plans(Mat F, PetscInt n, const PetscInt *ui, const PetscInt *uj, const PetscScalar *ua) {
  PetscErrorCode ierr;
  HipTriFactors *f = HipTriGet(F);
  PetscDeviceCtx *dc;
  f->nz = ui[n];
  const PetscInt noff = f->nz - n;
  PetscInt *lp, *ll, *lj, *up, *ul, *uc, *levL, *levU; PetscScalar *lv, *uv, *ones, *dinv;
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(n + 1), &lp);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)n, &ll);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(noff, 1), &lj);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)PetscMax(noff, 1), &lv);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)n, &up);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)n, &ul);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(noff, 1), &uc);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)PetscMax(noff, 1), &uv);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)n, &levL);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)n, &levU);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)n, &ones);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)n, &dinv);CHKERRQ(ierr);
  /* U^T by COLUMN RANGES on host threads: a thread owns the columns [lo, hi), walks the rows in ascending order (so that row c of
   * U^T lists its entries in the order the column sweep adds them) and takes the entries whose column is its own.  The levels of U (a backward recurrence over U's own rows) are
   * computed beside it, those of U^T (forward, over the transposed rows) after it. */
  IccT tr = {n, ui, uj, ua, lp, ll, lj, lv, up, ul, uc, uv, ones, dinv, levU, 0};
  pthread_t thU; int sideU = 0;
  if (n >= 200000) sideU = !pthread_create(&thU, NULL, icc0_levels_U, &tr);
  memset(lp, 0, sizeof(PetscInt) * (size_t)(n + 1));
  HipParallelRanges(n, icc0_count_columns, &tr);
  for (PetscInt c = 0; c < n; c++) { ll[c] = 0; lp[c + 1] += lp[c]; }
  HipParallelRanges(n, icc0_fill_columns, &tr);
  f->nlevL = 0;
  for (PetscInt c = 0; c < n; c++) {
    PetscInt l = 0;
    for (PetscInt q = lp[c]; q < lp[c] + ll[c]; q++) l = PetscMax(l, levL[lj[q]] + 1);
    levL[c] = l; f->nlevL = PetscMax(f->nlevL, l + 1);
  }
  HipParallelRanges(n, icc0_upper_rows, &tr);             /* the backward loop reads a row from its last off-diagonal entry to its first */
  if (sideU) pthread_join(thU, NULL); else icc0_levels_U(&tr);
  f->nlevU = tr.nlevU;
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  int rc = mi355x_trisolve_plan_create_pair(dc->h, n, 0, f->nlevL, levL, lp, ll, lj, lv, f->nlevU, levU, up, ul, uc, uv, ones, dinv, &f->tri_lo, &f->tri_up);
  HipFree(lp); HipFree(ll); HipFree(lj); HipFree(lv); HipFree(up); HipFree(ul); HipFree(uc); HipFree(uv);
  HipFree(levL); HipFree(levU); HipFree(ones); HipFree(dinv);
  CHKHIP(rc);
  return 0;
}

static PetscErrorCode MatCholeskyFactorNumeric_SeqAIJHIP(Mat F, Mat A, const MatFactorInfo *info) {
  PetscErrorCode ierr;
  HipTriFactors *f = HipTriGet(F);
  if (A->rmap->n != A->cmap->n) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_WRONG, "Must be square matrix, rows %d columns %d", A->rmap->n, A->cmap->n);
  if (f->factored_state == HipObjState(A) && f->factored_of == (void *)A && (f->tri_lo || !A->rmap->n)) return 0;
  if (f->tri_lo) mi355x_trisolve_plan_destroy(f->tri_lo);
  if (f->tri_up) mi355x_trisolve_plan_destroy(f->tri_up);
  f->tri_lo = f->tri_up = NULL; f->use_levels = 0; f->nshift = 0; f->nlevL = f->nlevU = 0; f->factored_state = -1;
#if defined(PETSCHIPMI355X_WITH_PETSC)
  ierr = MatCholeskyFactorNumeric_SeqAIJ(F, A, info);CHKERRQ(ierr);     /* the parent's factorisation into F's own Mat_SeqSBAIJ (aijfact.c:2076) */
  { Mat_SeqSBAIJ *b = (Mat_SeqSBAIJ *)F->data;
    f->n = A->rmap->n;
    if (f->n) { ierr = icc0_plans(F, f->n, b->i, b->j, b->a);CHKERRQ(ierr); } }
#else
  ierr = icc0_factor_host(F, A, info);CHKERRQ(ierr);
#endif
  if (f->tri_lo) HipTriWatchAdd(f);
  F->ops->solve = MatSolve_SeqAIJHIP_ICC;
  f->factored_state = HipObjState(A); f->factored_of = (void *)A;
  return 0;
}

PetscErrorCode MatICCFactorSymbolic_SeqAIJHIP(Mat F, Mat A, IS perm, const MatFactorInfo *info) {
  PetscErrorCode ierr;
  if (info->levels != 0.0) SETERRQ(HipObjComm(A), PETSC_ERR_SUP, "ICC(%d): only zero fill is on the ported path", (int)info->levels);
  if (strcmp(HipObjTypeName(A), MATSEQAIJHIPMI355X)) SETERRQ(HipObjComm(A), PETSC_ERR_SUP, "ICC on the device needs a sequential AIJ matrix of this type (use -pc_type bjacobi -sub_pc_type icc in parallel); got %s", HipObjTypeName(A));
#if defined(PETSCHIPMI355X_WITH_PETSC)
  { PetscBool id = PETSC_TRUE;
    if (perm) { ierr = ISIdentity(perm, &id);CHKERRQ(ierr); }
    if (!id) SETERRQ(HipObjComm(A), PETSC_ERR_SUP, "ICC on the device: natural ordering only (-pc_factor_mat_ordering_type natural)");
    ierr = MatICCFactorSymbolic_SeqAIJ(F, A, perm, info);CHKERRQ(ierr); }
#else
  if (perm) SETERRQ(HipObjComm(A), PETSC_ERR_SUP, "ICC: natural ordering only");
  ierr = 0; (void)ierr;
#endif
  HipTriGet(F)->factored_state = -1;
  F->ops->choleskyfactornumeric = MatCholeskyFactorNumeric_SeqAIJHIP;
  return 0;
}

