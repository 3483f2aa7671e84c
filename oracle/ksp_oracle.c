/*
 * ksp_oracle.c -- TEST INFRASTRUCTURE (see oracle.h).  Sequential restatement of the reference's
 * KSPSolve for CG, GMRES(m) and BiCGStab with none / Jacobi / block-Jacobi preconditioning, built on
 * the orc_* Vec and SeqAIJ restatements (left preconditioning, preconditioned residual norm,
 * KSPDefaultConverged -- the defaults the reference's golden outputs were produced with).
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* KSPConvergedReason values, include/petscksp.h */
#define R_ITERATING 0
#define R_CONVERGED_RTOL 2
#define R_CONVERGED_ATOL 3
#define R_CONVERGED_ITS 4
#define R_DIVERGED_NULL (-2)
#define R_DIVERGED_ITS (-3)
#define R_DIVERGED_DTOL (-4)
#define R_DIVERGED_BREAKDOWN (-5)
#define R_DIVERGED_INDEFINITE_PC (-8)
#define R_DIVERGED_NAN (-9)
#define R_DIVERGED_INDEFINITE_MAT (-10)

typedef struct solver_s {
  int ksp_type, pc_type;
  double rtol, abstol, dtol;
  int max_it, restart, refine_always, guess_nonzero, cg_single;
  int norm_type;   /* KSPNormType: 0 none, 1 preconditioned (default), 2 unpreconditioned, 3 natural (petscksp.h) */
  int pc_right;    /* PC_RIGHT (GMRES only): A B y = b, x = B y; the norm is then the unpreconditioned one */
  int pb_bs;       /* PCPBJACOBI block size */
  int n;
  int inode;       /* -1: not checked yet; 1: Mat_CheckInode keeps the inode routines for this matrix; 0: plain */
  const int *ai, *aj;
  const double *aa;
  double *idiag;              /* Jacobi: 1/diag (0 -> 1), PCSetUp_Jacobi jacobi.c:170-190 */
  int *fi, *fj, *fdiag; double *fa;   /* ILU(0) factors in the reference's L / reversed-U layout */
  int nblocks;
  const int *blk;
  struct solver_s *sub;       /* block Jacobi sub-solvers */
  int **sbi, **sbj; double **sba;
  /* convergence state (KSPDefaultConverged, iterativ.c:702) */
  double rnorm0, ttol;
  int its, reason;
  double *hist; int hist_cap, nhist;
} solver;

void orc_ksp_default_opts(orc_ksp_opts *o) {
  memset(o, 0, sizeof(*o));
  o->ksp_type = ORC_KSP_GMRES; o->pc_type = ORC_PC_NONE;
  o->rtol = 1e-5; o->abstol = 1e-50; o->dtol = 1e4; o->max_it = 10000;   /* itcreate.c:662-666 */
  o->restart = 30;                                                        /* gmresimpl.h GMRES_DEFAULT_MAXK */
  o->norm_type = 1;                                                       /* KSP_NORM_PRECONDITIONED (left PC default) */
  o->sub_ksp_type = ORC_KSP_PREONLY; o->sub_pc_type = ORC_PC_JACOBI;
  o->sub_rtol = 1e-5; o->sub_abstol = 1e-50; o->sub_dtol = 1e4; o->sub_max_it = 10000;
}

static void monitor(solver *s, double r) { if (s->hist && s->nhist < s->hist_cap) s->hist[s->nhist] = r; s->nhist++; }

static int solve(solver *s, const double *b, double *x);

/* ---- ILU(0), natural ordering: src/mat/impls/aij/seq/aijfact.c ---- */
/* MatILUFactorSymbolic_SeqAIJ_ilu0 (:1628-1700): L rows forward (columns < i), then the U rows stored from the
 * last row backwards, each followed by its diagonal slot; bdiag[i] = position of the (inverted) diagonal. */
int orc_ilu0_factor(int n, const int *ai, const int *aj, const double *aa, int *bi, int *bj, int *bdiag, double *ba) {
  return orc_ilu0_factor_shift(n, ai, aj, aa, bi, bj, bdiag, ba, NULL);
}
/* ... with PCILU's default shift (ilu.c:387-389: MAT_SHIFT_NONZERO, shiftamount = zeropivot = 100 eps): a pivot with
 * |pivot| <= zeropivot * (sum of the row's other factor entries) restarts the factorisation with the diagonal shifted by
 * shiftamount, then twice that, ... (MatPivotCheck_nz, matimpl.h:512-528; aijfact.c:507-592).  *nshift: restarts taken. */
int orc_ilu0_factor_shift(int n, const int *ai, const int *aj, const double *aa, int *bi, int *bj, int *bdiag, double *ba, int *nshift_out) {
  int k = 0;
  int *adiag = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; i++) {
    adiag[i] = -1;
    for (int q = ai[i]; q < ai[i + 1]; q++) if (aj[q] == i) { adiag[i] = q; break; }
    if (adiag[i] < 0) { free(adiag); return 1; }   /* "Matrix is missing diagonal entry" */
  }
  bi[0] = 0;
  for (int i = 0; i < n; i++) {
    int nz = adiag[i] - ai[i];
    bi[i + 1] = bi[i] + nz;
    for (int j = 0; j < nz; j++) bj[k++] = aj[ai[i] + j];
  }
  bdiag[n] = bi[n] - 1;
  for (int i = n - 1; i >= 0; i--) {
    int nz = ai[i + 1] - adiag[i] - 1;
    for (int j = 0; j < nz; j++) bj[k++] = aj[adiag[i] + 1 + j];
    bj[k++] = i;
    bdiag[i] = bdiag[i + 1] + nz + 1;
  }
  /* MatLUFactorNumeric_SeqAIJ (:461-620), identity permutations */
  double *rtmp = (double *)calloc((size_t)n + 1, sizeof(double));
  const double zeropivot = 100.0 * 2.220446049250313e-16, shiftamount = 100.0 * 2.220446049250313e-16;
  double shift_amount = 0.0;
  int nshift = 0, again;
  do {
  again = 0;
  for (int i = 0; i < n; i++) {
    int nz = bi[i + 1] - bi[i];
    const int *bjtmp = bj + bi[i];
    for (int j = 0; j < nz; j++) rtmp[bjtmp[j]] = 0.0;
    nz = bdiag[i] - bdiag[i + 1];
    bjtmp = bj + bdiag[i + 1] + 1;
    for (int j = 0; j < nz; j++) rtmp[bjtmp[j]] = 0.0;
    for (int q = ai[i]; q < ai[i + 1]; q++) rtmp[aj[q]] = aa[q];
    rtmp[i] += shift_amount;
    double rs = 0.0;
    const int nzL = bi[i + 1] - bi[i];
    for (int kk = 0; kk < nzL; kk++) {
      const int row = bj[bi[i] + kk];
      double *pc = rtmp + row;
      if (*pc != 0.0) {
        const double multiplier = *pc * ba[bdiag[row]];
        *pc = multiplier;
        const int *pj = bj + bdiag[row + 1] + 1;
        const double *pv = ba + bdiag[row + 1] + 1;
        const int nzu = bdiag[row] - bdiag[row + 1] - 1;
        for (int j = 0; j < nzu; j++) rtmp[pj[j]] -= multiplier * pv[j];
      }
    }
    for (int j = 0; j < nzL; j++) { ba[bi[i] + j] = rtmp[bj[bi[i] + j]]; rs += fabs(ba[bi[i] + j]); }
    nz = bdiag[i] - bdiag[i + 1] - 1;
    for (int j = 0; j < nz; j++) { ba[bdiag[i + 1] + 1 + j] = rtmp[bj[bdiag[i + 1] + 1 + j]]; rs += fabs(ba[bdiag[i + 1] + 1 + j]); }
    if (fabs(rtmp[i]) <= zeropivot * rs) {     /* MatPivotCheck_nz */
      shift_amount = nshift ? shift_amount * 2.0 : shiftamount;
      nshift++;
      if (nshift > 80) { free(rtmp); free(adiag); return 2; }
      again = 1;
      break;
    }
    ba[bdiag[i]] = 1.0 / rtmp[i];   /* inverted diagonal */
  }
  } while (again);
  free(rtmp); free(adiag);
  if (nshift_out) *nshift_out = nshift;
  return 0;
}

/* MatSolve_SeqAIJ_NaturalOrdering (:3126-3172) with PetscSparseDenseMinusDot (aij.h:337-339) */
void orc_ilu0_solve(int n, const int *bi, const int *bj, const int *bdiag, const double *ba, const double *b, double *x) {
  if (!n) return;
  x[0] = b[0];
  for (int i = 1; i < n; i++) {
    double sum = b[i];
    for (int q = bi[i]; q < bi[i + 1]; q++) sum -= ba[q] * x[bj[q]];
    x[i] = sum;
  }
  for (int i = n - 1; i >= 0; i--) {
    const int s0 = bdiag[i + 1] + 1, nz = bdiag[i] - bdiag[i + 1] - 1;
    double sum = x[i];
    for (int q = 0; q < nz; q++) sum -= ba[s0 + q] * x[bj[s0 + q]];
    x[i] = sum * ba[s0 + nz];
  }
}

/* MatSolve_SeqAIJ_Inode (src/mat/impls/aij/seq/inode.c:2327-2760), natural ordering: what the reference runs on the ILU factor of
 * a matrix with inodes (MatLUFactorNumeric_SeqAIJ_Inode installs it, inode.c:1310-1320) -- e.g. every 3-dof FEM matrix.  ns[] = the
 * node sizes (1..5) of the FACTOR (the nodes of A: consecutive rows with identical column lists).  Per node: every row's sum runs over
 * the columns of the node's FIRST row (lower) / LAST row (upper) two products at a time -- sum -= v[j] t0 + v[j+1] t1, the two
 * products added to each other first -- an odd last column alone; THEN the couplings inside the node, row after row.  Written as
 * one loop over the node size: cases 1..5 of the reference are this loop unrolled. */
void orc_ilu0_solve_inode(int n, int nnodes, const int *ns, const int *bi, const int *bj, const int *bdiag, const double *ba, const double *b, double *x) {
  double sum[5];
  int row = 0;
  for (int i = 0; i < nnodes; i++) {                      /* forward solve the lower triangular */
    const int nsz = ns[i], nz = bi[row + 1] - bi[row];
    const int *vi = bj + bi[row];
    int j;
    for (int k = 0; k < nsz; k++) sum[k] = b[row + k];
    for (j = 0; j < nz - 1; j += 2) {
      const double t0 = x[vi[j]], t1 = x[vi[j + 1]];
      for (int k = 0; k < nsz; k++) { const double *v = ba + bi[row + k]; sum[k] -= v[j] * t0 + v[j + 1] * t1; }
    }
    if (j == nz - 1) {
      const double t0 = x[vi[j]];
      for (int k = 0; k < nsz; k++) { const double *v = ba + bi[row + k]; sum[k] -= v[j] * t0; }
    }
    for (int k = 1; k < nsz; k++) { const double *v = ba + bi[row + k]; for (int l = 0; l < k; l++) sum[k] -= v[nz + l] * sum[l]; }
    for (int k = 0; k < nsz; k++) x[row + k] = sum[k];
    row += nsz;
  }
  row = n - 1;
  for (int i = nnodes - 1; i >= 0; i--) {                 /* backward solve the upper triangular */
    const int nsz = ns[i], nz = bdiag[row] - bdiag[row + 1] - 1;
    const int *vi = bj + bdiag[row + 1] + 1;
    int j;
    /* sum[k] belongs to row - k; its stored row starts k entries earlier than the shared columns (the k later rows of the node) */
    for (int k = 0; k < nsz; k++) sum[k] = x[row - k];
    for (j = 0; j < nz - 1; j += 2) {
      const double t0 = x[vi[j]], t1 = x[vi[j + 1]];
      for (int k = 0; k < nsz; k++) { const double *v = ba + bdiag[row - k + 1] + 1; sum[k] -= v[j + k] * t0 + v[j + k + 1] * t1; }
    }
    if (j == nz - 1) {
      const double t0 = x[vi[j]];
      for (int k = 0; k < nsz; k++) { const double *v = ba + bdiag[row - k + 1] + 1; sum[k] -= v[j + k] * t0; }
    }
    for (int k = 0; k < nsz; k++) {
      const double *v = ba + bdiag[row - k + 1] + 1;
      /* couplings to the later rows of the node, nearest LAST: v[k-1] is row's column, v[0] is row-k+1's (inode.c:2580-2760) */
      for (int l = 0; l < k; l++) sum[k] -= v[k - 1 - l] * x[row - l];
      x[row - k] = sum[k] * v[nz + k];
    }
    row -= nsz;
  }
}

/* ---- ICC(0), natural ordering: MatICCFactorSymbolic_SeqAIJ (levels 0: the pattern of A's upper triangle, diagonal LAST in its
 * row, aijfact.c:2405-2600) and MatCholeskyFactorNumeric_SeqAIJ (aijfact.c:2076-2230) with PCICC's defaults (icc.c:189-200:
 * MAT_SHIFT_POSITIVE_DEFINITE, zeropivot 100 eps).  Row k of the factor ends up holding  -U(k,j)/D(k)-style multipliers: the
 * reference overwrites U(i,k) by uikdi = -U(i,k) * (1/D(i)) when row i is added into row k, and that is what the solve reads. ---- */
int orc_icc0_count(int n, const int *ai, const int *aj) {
  int nz = 0;
  for (int i = 0; i < n; i++) { nz++; for (int q = ai[i]; q < ai[i + 1]; q++) if (aj[q] > i) nz++; }
  return nz;
}
int orc_icc0_factor(int n, const int *ai, const int *aj, const double *aa, int *ui, int *uj, double *ua) {
  /* symbolic: strictly upper entries of row k in column order, then the diagonal slot */
  int nz = 0;
  ui[0] = 0;
  for (int k = 0; k < n; k++) {
    for (int q = ai[k]; q < ai[k + 1]; q++) if (aj[q] > k) uj[nz++] = aj[q];
    uj[nz++] = k;
    ui[k + 1] = nz;
  }
  const double zeropivot = 100.0 * 2.220446049250313e-16;
  double shift_top = zeropivot, shift_amount = 0.0, shift_fraction = 0.0, shift_lo = 0.0, shift_hi = 1.0;
  int nshift = 0;
  const int nshift_max = 5;
  for (int i = 0; i < n; i++) {            /* max over the rows of sum|a_ij| - |a_ii| - Re(a_ii): what makes the matrix diagonally dominant */
    double d = 0.0, rs;
    for (int q = ai[i]; q < ai[i + 1]; q++) if (aj[q] == i) d = aa[q];
    rs = -fabs(d) - d;
    for (int q = ai[i]; q < ai[i + 1]; q++) rs += fabs(aa[q]);
    if (rs > shift_top) shift_top = rs;
  }
  shift_top *= 1.1;
  double *rtmp = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  int *il = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1)), *c2r = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  int newshift;
  do {
    newshift = 0;
    for (int i = 0; i < n; i++) c2r[i] = n;      /* c2r[col]: head of the list of earlier rows with an entry in column col still to be applied */
    if (n) il[0] = 0;                            /* il[i]: first entry of row i not yet consumed */
    for (int k = 0; k < n; k++) {
      const int diagk = ui[k + 1] - 1;
      for (int q = ui[k]; q < ui[k + 1]; q++) rtmp[uj[q]] = 0.0;
      { int w = ui[k];                           /* the unfactored row's upper part; the factor's slots are zeroed as they are claimed */
        for (int q = ai[k]; q < ai[k + 1]; q++) if (aj[q] >= k) { rtmp[aj[q]] = aa[q]; ua[w++] = 0.0; } }
      rtmp[k] += shift_amount;
      double dk = rtmp[k];
      int i = c2r[k];
      while (i < k) {
        const int nexti = c2r[i], ili = il[i];
        const double uikdi = -ua[ili] * ua[ui[i + 1] - 1];
        dk += uikdi * ua[ili];
        ua[ili] = uikdi;
        const int jmin = ili + 1, jmax = ui[i + 1] - 1;      /* the rest of row i, without its diagonal slot */
        if (jmin < jmax) {
          for (int q = jmin; q < jmax; q++) rtmp[uj[q]] += uikdi * ua[q];
          il[i] = jmin;
          const int j = uj[jmin]; c2r[i] = c2r[j]; c2r[j] = i;
        }
        i = nexti;
      }
      double rs = 0.0;
      if (ui[k] < diagk) {
        for (int q = ui[k]; q < diagk; q++) { ua[q] = rtmp[uj[q]]; rs += fabs(ua[q]); }
        il[k] = ui[k];
        const int j = uj[ui[k]]; c2r[k] = c2r[j]; c2r[j] = k;
      }
      if (dk <= zeropivot * rs) {                /* MatPivotCheck_pd (matimpl.h:532-553): bisect the shift towards diagonal dominance */
        if (nshift == nshift_max) shift_fraction = shift_hi;
        else { shift_lo = shift_fraction; shift_fraction = (shift_hi + shift_lo) / 2.; }
        shift_amount = shift_fraction * shift_top;
        nshift++;
        newshift = 1;
        if (nshift > nshift_max + 1) { free(rtmp); free(il); free(c2r); return -1; }
        break;
      }
      ua[diagk] = 1.0 / dk;
    }
  } while (newshift);
  free(rtmp); free(il); free(c2r);
  return nshift;
}
void orc_icc0_solve(int n, const int *ui, const int *uj, const double *ua, const double *b, double *x) {
  for (int i = 0; i < n; i++) x[i] = b[i];
  for (int i = 0; i < n; i++) {                  /* U^T D y = b: column sweep, then the row's own entry times 1/D(i) */
    const int nz = ui[i + 1] - ui[i] - 1;
    const double xi = x[i];
    for (int q = 0; q < nz; q++) x[uj[ui[i] + q]] += ua[ui[i] + q] * xi;
    x[i] = xi * ua[ui[i] + nz];
  }
  for (int i = n - 2; i >= 0; i--) {             /* U x = y: the row's entries from its last off-diagonal one backwards */
    const int nz = ui[i + 1] - ui[i] - 1;
    double xi = x[i];
    for (int q = nz - 1; q >= 0; q--) xi += ua[ui[i] + q] * x[uj[ui[i] + q]];
    x[i] = xi;
  }
}

/* ---- PC ---- */
static void pc_setup(solver *s) {
  if (s->pc_type == ORC_PC_ICC) {
    int nz = orc_icc0_count(s->n, s->ai, s->aj);
    s->fi = (int *)malloc(sizeof(int) * (size_t)(s->n + 1));
    s->fj = (int *)malloc(sizeof(int) * (size_t)(nz + 1));
    s->fdiag = NULL;
    s->fa = (double *)calloc((size_t)nz + 1, sizeof(double));
    orc_icc0_factor(s->n, s->ai, s->aj, s->aa, s->fi, s->fj, s->fa);
  } else if (s->pc_type == ORC_PC_ILU) {
    int nz = s->ai[s->n];
    s->fi = (int *)malloc(sizeof(int) * (size_t)(s->n + 1));
    s->fj = (int *)malloc(sizeof(int) * (size_t)(nz + 1));
    s->fdiag = (int *)malloc(sizeof(int) * (size_t)(s->n + 1));
    s->fa = (double *)calloc((size_t)nz + 1, sizeof(double));
    orc_ilu0_factor(s->n, s->ai, s->aj, s->aa, s->fi, s->fj, s->fdiag, s->fa);
  } else if (s->pc_type == ORC_PC_JACOBI) {
    s->idiag = (double *)malloc(sizeof(double) * (size_t)s->n);
    orc_csr_get_diagonal(s->n, s->ai, s->aj, s->aa, s->idiag);   /* MatGetDiagonal */
    orc_vec_reciprocal((size_t)s->n, s->idiag);                   /* VecReciprocal */
    for (int i = 0; i < s->n; i++) if (s->idiag[i] == 0.0) s->idiag[i] = 1.0;
  } else if (s->pc_type == ORC_PC_PBJACOBI) {
    /* PCSetUp_PBJacobi (pbjacobi.c:228-275) -> MatInvertBlockDiagonal: the bs x bs diagonal blocks, column-major, inverted.
     * The blocks are read out of the point CSR (entries a BAIJ matrix stores as explicit zeros are zero here too). */
    const int bs = s->pb_bs > 0 ? s->pb_bs : 1, mbs = s->n / bs;
    s->idiag = (double *)calloc((size_t)mbs * bs * bs + 1, sizeof(double));
    for (int i = 0; i < s->n; i++)
      for (int k = s->ai[i]; k < s->ai[i + 1]; k++)
        if (s->aj[k] / bs == i / bs) s->idiag[(size_t)(i / bs) * bs * bs + (i % bs) + (size_t)(s->aj[k] % bs) * bs] = s->aa[k];
    for (int i = 0; i < mbs; i++) orc_block_inverse(bs, s->idiag + (size_t)i * bs * bs);
  } else if (s->pc_type == ORC_PC_BJACOBI) {
    /* PCSetUp_BJacobi_*: sub-KSP on each diagonal block (bjacobi.c:858-923; MatGetSubMatrices of
     * the contiguous diagonal blocks) */
    s->sub = (solver *)calloc((size_t)s->nblocks, sizeof(solver));
    s->sbi = (int **)calloc((size_t)s->nblocks, sizeof(int *));
    s->sbj = (int **)calloc((size_t)s->nblocks, sizeof(int *));
    s->sba = (double **)calloc((size_t)s->nblocks, sizeof(double *));
    for (int k = 0; k < s->nblocks; k++) {
      int lo = s->blk[k], hi = s->blk[k + 1], m = hi - lo, nz = 0;
      for (int r = lo; r < hi; r++)
        for (int q = s->ai[r]; q < s->ai[r + 1]; q++) if (s->aj[q] >= lo && s->aj[q] < hi) nz++;
      int *bi = (int *)malloc(sizeof(int) * (size_t)(m + 1));
      int *bj = (int *)malloc(sizeof(int) * (size_t)(nz > 0 ? nz : 1));
      double *ba = (double *)malloc(sizeof(double) * (size_t)(nz > 0 ? nz : 1));
      nz = 0; bi[0] = 0;
      for (int r = lo; r < hi; r++) {
        for (int q = s->ai[r]; q < s->ai[r + 1]; q++)
          if (s->aj[q] >= lo && s->aj[q] < hi) { bj[nz] = s->aj[q] - lo; ba[nz++] = s->aa[q]; }
        bi[r - lo + 1] = nz;
      }
      s->sbi[k] = bi; s->sbj[k] = bj; s->sba[k] = ba;
      solver *t = &s->sub[k];
      t->n = m; t->ai = bi; t->aj = bj; t->aa = ba; t->inode = -1;
      pc_setup(t);
    }
  }
}

static void pc_free(solver *s) {
  free(s->idiag); s->idiag = NULL;
  free(s->fi); free(s->fj); free(s->fdiag); free(s->fa); s->fi = s->fj = s->fdiag = NULL; s->fa = NULL;
  if (s->sub) {
    for (int k = 0; k < s->nblocks; k++) { pc_free(&s->sub[k]); free(s->sbi[k]); free(s->sbj[k]); free(s->sba[k]); }
    free(s->sub); free(s->sbi); free(s->sbj); free(s->sba);
    s->sub = NULL;
  }
}

/* PCApply: PCApply_Jacobi jacobi.c:266 (VecPointwiseMult(y,x,diag)); PCApply_None (VecCopy);
 * PCApply_BJacobi_Singleblock/Multiblock bjacobi.c:738,1140 (sub-KSPSolve per block, zero guess) */
static void pc_apply(solver *s, const double *x, double *y) {
  if (s->pc_type == ORC_PC_NONE) orc_vec_copy((size_t)s->n, x, y);
  else if (s->pc_type == ORC_PC_JACOBI) orc_vec_pointwise_mult((size_t)s->n, x, s->idiag, y);
  else if (s->pc_type == ORC_PC_ILU) orc_ilu0_solve(s->n, s->fi, s->fj, s->fdiag, s->fa, x, y);   /* PCApply_ILU -> MatSolve */
  else if (s->pc_type == ORC_PC_ICC) orc_icc0_solve(s->n, s->fi, s->fj, s->fa, x, y);               /* PCApply_ICC (icc.c:65) -> MatSolve */
  else if (s->pc_type == ORC_PC_PBJACOBI) orc_pbjacobi_apply(s->n / (s->pb_bs > 0 ? s->pb_bs : 1), s->pb_bs > 0 ? s->pb_bs : 1, s->idiag, x, y);
  else {
    for (int k = 0; k < s->nblocks; k++) {
      solver *t = &s->sub[k];
      t->hist = NULL; t->nhist = 0; t->hist_cap = 0;
      solve(t, x + s->blk[k], y + s->blk[k]);
    }
  }
}

/* MatMult as MATSEQAIJ dispatches it: MatMult_SeqAIJ_Inode once Mat_CheckInode has kept the inode routines (checked on
 * first use, as MatAssemblyEnd_SeqAIJ_Inode does once per pattern), MatMult_SeqAIJ otherwise */
static void mat_mult(solver *s, const double *x, double *y) {
  if (s->inode < 0) {
    int *ns = (int *)malloc(sizeof(int) * (size_t)(s->n + 1));
    s->inode = orc_check_inode(s->n, s->ai, s->aj, 5, ns) > 0;
    free(ns);
  }
  if (s->inode) orc_spmv_csr_inode(s->n, s->ai, s->aj, s->aa, x, y);
  else orc_spmv_csr(s->n, s->ai, s->aj, s->aa, x, y);
}
/* KSP_PCApplyBAorAB, PC_LEFT branch of PCApplyBAorAB precon.c:620-622 */
static void pc_apply_BA(solver *s, const double *x, double *y, double *w) {
  if (s->pc_right) { pc_apply(s, x, w); mat_mult(s, w, y); }   /* PC_RIGHT branch, precon.c:617-619 */
  else { mat_mult(s, x, w); pc_apply(s, w, y); }
}

/* KSPDefaultConverged, src/ksp/ksp/interface/iterativ.c:702-780 (left PC; the norm of the right-hand side for a
 * nonzero guess follows the norm type, :718-737); KSP_NORM_NONE installs KSPSkipConverged (itcreate.c:228-229,
 * iterativ.c:536-544) */
static void converged(solver *s, int n, double rnorm, const double *b) {
  s->reason = R_ITERATING;
  if (s->norm_type == 0) { if (n >= s->max_it) s->reason = R_CONVERGED_ITS; return; }
  if (!n) {
    if (s->guess_nonzero) {
      double snorm = 0.0;
      if (s->norm_type == 2 || s->pc_right) orc_vec_norm((size_t)s->n, 1, b, &snorm);
      else {
        double *z = (double *)malloc(sizeof(double) * (size_t)s->n);
        pc_apply(s, b, z);
        if (s->norm_type == 1) orc_vec_norm((size_t)s->n, 1, z, &snorm);
        else snorm = sqrt(fabs(orc_vec_dot((size_t)s->n, b, z)));
        free(z);
      }
      if (!snorm) snorm = rnorm;
      s->rnorm0 = snorm;
    } else s->rnorm0 = rnorm;
    s->ttol = fmax(s->rtol * s->rnorm0, s->abstol);
  }
  /* chknorm == -1 (itcreate.c:668) so the test is always made */
  if (isnan(rnorm) || isinf(rnorm)) s->reason = R_DIVERGED_NAN;
  else if (rnorm <= s->ttol) s->reason = (rnorm < s->abstol) ? R_CONVERGED_ATOL : R_CONVERGED_RTOL;
  else if (rnorm >= s->dtol * s->rnorm0) s->reason = R_DIVERGED_DTOL;
}

/* KSPInitialResidual, src/ksp/ksp/interface/itres.c:39-73 (PC_LEFT: B(b - A x); PC_RIGHT: b - A x) */
static void initial_residual(solver *s, const double *x, double *vt1, double *vt2, double *vres, const double *b) {
  size_t n = (size_t)s->n;
  if (s->guess_nonzero) {
    mat_mult(s, x, vt1);
    orc_vec_copy(n, b, vt2);
    orc_vec_axpy(n, -1.0, vt1, vt2);
    if (s->pc_right) orc_vec_copy(n, vt2, vres);
    else pc_apply(s, vt2, vres);
  } else {
    orc_vec_copy(n, b, vt2);
    if (s->pc_right) orc_vec_copy(n, b, vres);
    else pc_apply(s, b, vres);
  }
}

/* ---- KSPSolve_CG, src/ksp/ksp/impls/cg/cg.c:92-286 (all four norm types :136-161,233-260; -ksp_cg_single_reduction) ---- */
static void solve_cg(solver *s, const double *B, double *X) {
  size_t n = (size_t)s->n;
  const int single = s->cg_single, nt = s->norm_type;
  double *R = (double *)malloc(5 * n * sizeof(double)), *Z = R + n, *P = Z + n, *S = P + n, *W = single ? S + n : Z;
  double dpi = 0.0, a = 1.0, beta = 0.0, betaold = 1.0, b, dpiold, dp = 0.0, delta = 0.0;
  int i;
  s->its = 0;
  if (s->guess_nonzero) { mat_mult(s, X, R); orc_vec_aypx(n, -1.0, B, R); }
  else orc_vec_copy(n, B, R);
  if (nt == 1) { pc_apply(s, R, Z); orc_vec_norm(n, 1, Z, &dp); }
  else if (nt == 2) orc_vec_norm(n, 1, R, &dp);
  else if (nt == 3) {
    pc_apply(s, R, Z);
    if (single) { mat_mult(s, Z, S); delta = orc_vec_dot(n, Z, S); }
    beta = orc_vec_dot(n, Z, R);
    dp = sqrt(fabs(beta));
  } else dp = 0.0;
  monitor(s, dp);
  converged(s, 0, dp, B);
  if (s->reason) { free(R); return; }
  if (nt != 1 && nt != 3) pc_apply(s, R, Z);
  if (nt != 3) {
    if (single) { mat_mult(s, Z, S); delta = orc_vec_dot(n, Z, S); }
    beta = orc_vec_dot(n, Z, R);
  }
  i = 0;
  do {
    s->its = i + 1;
    if (beta == 0.0) { s->reason = R_CONVERGED_ATOL; break; }
    else if (i > 0 && beta * betaold < 0.0) { s->reason = R_DIVERGED_INDEFINITE_PC; break; }
    if (!i) { orc_vec_copy(n, Z, P); b = 0.0; }
    else { b = beta / betaold; orc_vec_aypx(n, b, Z, P); }
    dpiold = dpi;
    if (!single || !i) { mat_mult(s, P, W); dpi = orc_vec_dot(n, P, W); }
    else { orc_vec_aypx(n, beta / betaold, S, W); dpi = delta - beta * beta * dpiold / (betaold * betaold); }
    betaold = beta;
    if (dpi == 0.0 || (i > 0 && dpi * dpiold <= 0.0)) { s->reason = R_DIVERGED_INDEFINITE_MAT; break; }
    a = beta / dpi;
    orc_vec_axpy(n, a, P, X);
    orc_vec_axpy(n, -a, W, R);
    if (nt == 1) { pc_apply(s, R, Z); if (single) mat_mult(s, Z, S); orc_vec_norm(n, 1, Z, &dp); }
    else if (nt == 2) orc_vec_norm(n, 1, R, &dp);
    else if (nt == 3) {
      pc_apply(s, R, Z);
      if (single) { const double *vv[2] = {S, R}; double t2[2]; mat_mult(s, Z, S); orc_vec_mdot(n, 2, Z, vv, t2); delta = t2[0]; beta = t2[1]; }
      else beta = orc_vec_dot(n, Z, R);
      dp = sqrt(fabs(beta));
    } else dp = 0.0;
    monitor(s, dp);
    converged(s, i + 1, dp, B);
    if (s->reason) break;
    if (nt != 1 && nt != 3) { pc_apply(s, R, Z); if (single) mat_mult(s, Z, S); }
    if (nt != 3) {
      if (single) { const double *vv[2] = {S, R}; double t2[2]; orc_vec_mdot(n, 2, Z, vv, t2); delta = t2[0]; beta = t2[1]; }
      else beta = orc_vec_dot(n, Z, R);
    }
    i++;
  } while (i < s->max_it);
  if (i >= s->max_it && !(nt == 0 && s->reason)) s->reason = R_DIVERGED_ITS;
  free(R);
}

/* ---- KSPSolve_PIPECG, src/ksp/ksp/impls/cg/pipecg/pipecg.c:49-205 (split-phase reductions are immediate here); the branch
 * structure of its reductions is kept as it is: gamma is refreshed every iteration only with the natural norm or none ---- */
static void solve_pipecg(solver *sv, const double *B, double *X) {
  size_t n = (size_t)sv->n;
  const int nt = sv->norm_type;
  double *M = (double *)malloc(9 * n * sizeof(double)), *Z = M + n, *P = Z + n, *N = P + n, *W = N + n, *Q = W + n, *U = Q + n, *R = U + n, *S = R + n;
  double alpha = 0.0, beta = 0.0, gamma = 0.0, gammaold = 0.0, delta = 0.0, dp = 0.0;
  int i;
  sv->its = 0;
  if (sv->guess_nonzero) { mat_mult(sv, X, R); orc_vec_aypx(n, -1.0, B, R); }
  else orc_vec_copy(n, B, R);
  pc_apply(sv, R, U);
  if (nt == 1) { orc_vec_norm(n, 1, U, &dp); mat_mult(sv, U, W); }
  else if (nt == 2) { orc_vec_norm(n, 1, R, &dp); mat_mult(sv, U, W); }
  else if (nt == 3) { gamma = orc_vec_dot(n, R, U); mat_mult(sv, U, W); dp = sqrt(fabs(gamma)); }
  else { mat_mult(sv, U, W); dp = 0.0; }
  monitor(sv, dp);
  converged(sv, 0, dp, B);
  if (sv->reason) { free(M); return; }
  i = 0;
  do {
    if (i > 0 && nt == 2) orc_vec_norm(n, 1, R, &dp);
    else if (i > 0 && nt == 1) orc_vec_norm(n, 1, U, &dp);
    else if (!(i == 0 && nt == 3)) gamma = orc_vec_dot(n, R, U);
    delta = orc_vec_dot(n, W, U);
    pc_apply(sv, W, M);
    mat_mult(sv, M, N);
    if (i > 0) {
      if (nt == 3) dp = sqrt(fabs(gamma));
      else if (nt == 0) dp = 0.0;
      monitor(sv, dp);
      converged(sv, i, dp, B);
      if (sv->reason) break;
    }
    if (i == 0) {
      alpha = gamma / delta;
      orc_vec_copy(n, N, Z); orc_vec_copy(n, M, Q); orc_vec_copy(n, U, P); orc_vec_copy(n, W, S);
    } else {
      beta = gamma / gammaold;
      alpha = gamma / (delta - beta / alpha * gamma);
      orc_vec_aypx(n, beta, N, Z); orc_vec_aypx(n, beta, M, Q); orc_vec_aypx(n, beta, U, P); orc_vec_aypx(n, beta, W, S);
    }
    orc_vec_axpy(n, alpha, P, X);
    orc_vec_axpy(n, -alpha, Q, U);
    orc_vec_axpy(n, -alpha, Z, W);
    orc_vec_axpy(n, -alpha, S, R);
    gammaold = gamma;
    i++;
    sv->its = i;
  } while (i < sv->max_it);
  if (i >= sv->max_it) sv->reason = R_DIVERGED_ITS;
  free(M);
}

/* ---- KSPSolve_GROPPCG, src/ksp/ksp/impls/cg/groppcg/groppcg.c:40-175 (split-phase reductions are immediate here) ---- */
static void solve_groppcg(solver *sv, const double *B, double *X) {
  size_t n = (size_t)sv->n;
  const int nt = sv->norm_type;
  double *r = (double *)malloc(6 * n * sizeof(double)), *p = r + n, *s = p + n, *S = s + n, *z = S + n, *Z = z + n;
  double alpha, beta, gamma, gammaNew, t, dp = 0.0;
  int i;
  sv->its = 0;
  if (sv->guess_nonzero) { mat_mult(sv, X, r); orc_vec_aypx(n, -1.0, B, r); }
  else orc_vec_copy(n, B, r);
  pc_apply(sv, r, z);
  orc_vec_copy(n, z, p);
  gamma = orc_vec_dot(n, r, z);
  mat_mult(sv, p, s);
  if (nt == 1) orc_vec_norm(n, 1, z, &dp);
  else if (nt == 2) orc_vec_norm(n, 1, r, &dp);
  else if (nt == 3) dp = sqrt(fabs(gamma));
  else dp = 0.0;
  monitor(sv, dp);
  converged(sv, 0, dp, B);
  if (sv->reason) { free(r); return; }
  i = 0;
  do {
    sv->its = i + 1;
    i++;
    t = orc_vec_dot(n, p, s);
    pc_apply(sv, s, S);
    alpha = gamma / t;
    orc_vec_axpy(n, alpha, p, X);
    orc_vec_axpy(n, -alpha, s, r);
    orc_vec_axpy(n, -alpha, S, z);
    if (nt == 2) orc_vec_norm(n, 1, r, &dp);
    else if (nt == 1) orc_vec_norm(n, 1, z, &dp);
    gammaNew = orc_vec_dot(n, r, z);
    mat_mult(sv, z, Z);
    if (nt == 3) dp = sqrt(fabs(gammaNew));
    else if (nt == 0) dp = 0.0;
    monitor(sv, dp);
    converged(sv, i, dp, B);
    if (sv->reason) break;
    beta = gammaNew / gamma;
    gamma = gammaNew;
    orc_vec_aypx(n, beta, z, p);
    orc_vec_aypx(n, beta, Z, s);
  } while (i < sv->max_it);
  if (i >= sv->max_it && !sv->reason) sv->reason = R_DIVERGED_ITS;
  free(r);
}

/* ---- KSPSolve_GMRES, src/ksp/ksp/impls/gmres/gmres.c:118-409 + borthog2.c:35-119 ---- */
typedef struct {
  int max_k;
  double *hh, *hes, *grs, *cc, *ss, *lhh, *nrs;
  double *temp, *temp_matop, **vv;
} gm;
#define HH(a, b) (g->hh + (size_t)(b) * (g->max_k + 2) + (a))
#define HES(a, b) (g->hes + (size_t)(b) * (g->max_k + 1) + (a))

static double normalize(size_t n, double *x) {   /* VecNormalize rvector.c:299 */
  double nrm;
  orc_vec_norm(n, 1, x, &nrm);
  if (nrm != 0.0 && nrm != 1.0) orc_vec_scale(n, 1.0 / nrm, x);
  return nrm;
}

static void gmres_build_soln(solver *s, gm *g, double *X, int it) {   /* gmres.c:309-354 */
  size_t n = (size_t)s->n;
  if (it < 0) return;
  if (*HH(it, it) != 0.0) g->nrs[it] = g->grs[it] / *HH(it, it);
  else { s->reason = R_DIVERGED_BREAKDOWN; return; }
  for (int ii = 1; ii <= it; ii++) {
    int k = it - ii;
    double tt = g->grs[k];
    for (int j = k + 1; j <= it; j++) tt = tt - *HH(k, j) * g->nrs[j];
    if (*HH(k, k) == 0.0) { s->reason = R_DIVERGED_BREAKDOWN; return; }
    g->nrs[k] = tt / *HH(k, k);
  }
  orc_vec_set(n, 0.0, g->temp);
  orc_vec_maxpy(n, it + 1, g->nrs, (const double *const *)g->vv, g->temp);
  if (s->pc_right) { pc_apply(s, g->temp, g->temp_matop); orc_vec_copy(n, g->temp_matop, g->temp); }   /* KSPUnwindPreconditioner */
  orc_vec_axpy(n, 1.0, g->temp, X);
}

static void gmres_orthog(solver *s, gm *g, int it) {   /* borthog2.c:35-119 */
  size_t n = (size_t)s->n;
  double *hh = HH(0, it), *hes = HES(0, it), *lhh = g->lhh;
  int passes = s->refine_always ? 2 : 1;
  for (int j = 0; j <= it; j++) { hh[j] = 0.0; hes[j] = 0.0; }
  for (int p = 0; p < passes; p++) {
    orc_vec_mdot(n, it + 1, g->vv[it + 1], (const double *const *)g->vv, lhh);
    for (int j = 0; j <= it; j++) lhh[j] = -lhh[j];
    orc_vec_maxpy(n, it + 1, lhh, (const double *const *)g->vv, g->vv[it + 1]);
    for (int j = 0; j <= it; j++) { hh[j] -= lhh[j]; hes[j] -= lhh[j]; }
  }
}

static void gmres_update_hessenberg(solver *s, gm *g, int it, int hapend, double *res) {   /* gmres.c:360-409 */
  double *hh = HH(0, it), *cc = g->cc, *ss = g->ss, tt;
  for (int j = 1; j <= it; j++) {
    tt = *hh;
    *hh = *cc * tt + *ss * *(hh + 1);
    hh++;
    *hh = *cc++ * *hh - (*ss++ * tt);
  }
  if (!hapend) {
    tt = sqrt(*hh * *hh + *(hh + 1) * *(hh + 1));
    if (tt == 0.0) { s->reason = R_DIVERGED_NULL; return; }
    *cc = *hh / tt;
    *ss = *(hh + 1) / tt;
    g->grs[it + 1] = -(*ss * g->grs[it]);
    g->grs[it] = *cc * g->grs[it];
    *hh = *cc * *hh + *ss * *(hh + 1);
    *res = fabs(g->grs[it + 1]);
  } else *res = 0.0;
}

static int gmres_cycle(solver *s, gm *g, const double *B, double *X) {   /* gmres.c:118-209 */
  size_t n = (size_t)s->n;
  double res_norm, res, hapbnd, tt;
  int it = 0, hapend = 0;
  res_norm = normalize(n, g->vv[0]);
  res = res_norm;
  g->grs[0] = res_norm;
  monitor(s, res);
  if (!res) { s->reason = R_CONVERGED_ATOL; return 0; }
  converged(s, s->its, res, B);
  while (!s->reason && it < g->max_k && s->its < s->max_it) {
    if (it) monitor(s, res);
    pc_apply_BA(s, g->vv[it], g->vv[it + 1], g->temp_matop);
    gmres_orthog(s, g, it);
    tt = normalize(n, g->vv[it + 1]);
    *HH(it + 1, it) = tt;
    *HES(it + 1, it) = tt;
    hapbnd = fabs(tt / g->grs[it]);
    if (hapbnd > 1e-30) hapbnd = 1e-30;     /* haptol, gmres.c KSPCreate_GMRES */
    if (tt < hapbnd) hapend = 1;
    gmres_update_hessenberg(s, g, it, hapend, &res);
    it++;
    s->its++;
    if (s->reason) break;
    converged(s, s->its, res, B);
    if (hapend) break;
  }
  if (it && (s->reason || s->its >= s->max_it)) monitor(s, res);
  gmres_build_soln(s, g, X, it - 1);
  return it;
}

static void solve_gmres(solver *s, const double *B, double *X) {   /* gmres.c:213-243 */
  size_t n = (size_t)s->n;
  gm G, *g = &G;
  int k = s->restart, itcount = 0, guess = s->guess_nonzero;
  g->max_k = k;
  g->hh = (double *)calloc((size_t)(k + 2) * (k + 1), sizeof(double));
  g->hes = (double *)calloc((size_t)(k + 1) * (k + 1), sizeof(double));
  g->grs = (double *)calloc((size_t)k + 2, sizeof(double));
  g->cc = (double *)calloc((size_t)k + 1, sizeof(double));
  g->ss = (double *)calloc((size_t)k + 1, sizeof(double));
  g->lhh = (double *)calloc((size_t)k + 2, sizeof(double));
  g->nrs = (double *)calloc((size_t)k + 2, sizeof(double));
  g->vv = (double **)calloc((size_t)k + 2, sizeof(double *));
  double *store = (double *)calloc((size_t)(k + 4) * n + 1, sizeof(double));
  g->temp = store; g->temp_matop = store + n;
  for (int j = 0; j < k + 2; j++) g->vv[j] = store + (size_t)(j + 2) * n;
  s->its = 0;
  s->reason = R_ITERATING;
  while (!s->reason) {
    initial_residual(s, X, g->temp, g->temp_matop, g->vv[0], B);
    itcount += gmres_cycle(s, g, B, X);
    if (itcount >= s->max_it) { if (!s->reason) s->reason = R_DIVERGED_ITS; break; }
    s->guess_nonzero = 1;
  }
  s->guess_nonzero = guess;
  free(g->hh); free(g->hes); free(g->grs); free(g->cc); free(g->ss); free(g->lhh); free(g->nrs); free(g->vv); free(store);
}

/* ---- KSPSolve_BCGS, src/ksp/ksp/impls/bcgs/bcgs.c:43-160 (PC_LEFT) ---- */
static void solve_bcgs(solver *s, const double *B, double *X) {
  size_t n = (size_t)s->n;
  double *R = (double *)calloc(6 * n + 1, sizeof(double)), *RP = R + n, *V = RP + n, *T = V + n, *S = T + n, *P = S + n;
  double rho, rhoold, alpha, beta, omega, omegaold, d1, d2, dp = 0.0;
  int i;
  initial_residual(s, X, V, T, R, B);
  if (s->norm_type != 0) orc_vec_norm(n, 1, R, &dp);   /* bcgs.c:76: no norm with KSP_NORM_NONE */
  s->its = 0;
  monitor(s, dp);
  converged(s, 0, dp, B);
  if (s->reason) { free(R); return; }
  orc_vec_copy(n, R, RP);
  rhoold = 1.0; alpha = 1.0; omegaold = 1.0;
  orc_vec_set(n, 0.0, P);
  orc_vec_set(n, 0.0, V);
  i = 0;
  do {
    rho = orc_vec_dot(n, R, RP);
    beta = (rho / rhoold) * (alpha / omegaold);
    orc_vec_axpbypcz(n, 1.0, -omegaold * beta, beta, R, V, P);
    pc_apply_BA(s, P, V, T);
    d1 = orc_vec_dot(n, V, RP);
    if (d1 == 0.0) { s->reason = R_DIVERGED_BREAKDOWN; break; }   /* reference raises PETSC_ERR_PLIB */
    alpha = rho / d1;
    orc_vec_waxpy(n, -alpha, V, R, S);
    pc_apply_BA(s, S, T, R);
    orc_vec_dotnorm2(n, S, T, &d1, &d2);
    if (d2 == 0.0) {
      d1 = orc_vec_dot(n, S, S);
      if (d1 != 0.0) { s->reason = R_DIVERGED_BREAKDOWN; break; }
      orc_vec_axpy(n, alpha, P, X);
      s->its++;
      s->reason = R_CONVERGED_RTOL;
      monitor(s, 0.0);
      break;
    }
    omega = d1 / d2;
    orc_vec_axpbypcz(n, alpha, omega, 1.0, P, S, X);
    orc_vec_waxpy(n, -omega, T, S, R);
    if (s->norm_type != 0) orc_vec_norm(n, 1, R, &dp);   /* bcgs.c:131 */
    rhoold = rho; omegaold = omega;
    s->its++;
    monitor(s, dp);
    converged(s, i + 1, dp, B);
    if (s->reason) break;
    if (rho == 0.0) { s->reason = R_DIVERGED_BREAKDOWN; break; }
    i++;
  } while (i < s->max_it);
  if (i >= s->max_it) s->reason = R_DIVERGED_ITS;
  free(R);
}

/* KSPSolve, src/ksp/ksp/interface/itfunc.c:335: zero the guess unless told otherwise, then the type's solve;
 * KSPSolve_PREONLY (impls/preonly/preonly.c): x = PC(b), CONVERGED_ITS */
static int solve(solver *s, const double *b, double *x) {
  if (!s->guess_nonzero) orc_vec_set((size_t)s->n, 0.0, x);
  s->reason = R_ITERATING; s->its = 0;
  switch (s->ksp_type) {
  case ORC_KSP_CG: solve_cg(s, b, x); break;
  case ORC_KSP_GROPPCG: solve_groppcg(s, b, x); break;
  case ORC_KSP_PIPECG: solve_pipecg(s, b, x); break;
  case ORC_KSP_GMRES: solve_gmres(s, b, x); break;
  case ORC_KSP_BCGS: solve_bcgs(s, b, x); break;
  case ORC_KSP_PREONLY: pc_apply(s, b, x); s->its = 1; s->reason = R_CONVERGED_ITS; break;
  default: return 1;
  }
  return 0;
}

int orc_ksp_solve(const orc_ksp_opts *o, int n, const int *ai, const int *aj, const double *aa, const double *b,
                  double *x, double *hist, int hist_cap, int *nhist, int *its, int *reason) {
  solver S;
  memset(&S, 0, sizeof(S));
  S.ksp_type = o->ksp_type; S.pc_type = o->pc_type;
  S.rtol = o->rtol; S.abstol = o->abstol; S.dtol = o->dtol; S.max_it = o->max_it;
  S.restart = o->restart; S.refine_always = o->refine_always; S.guess_nonzero = o->guess_nonzero; S.cg_single = o->cg_single; S.norm_type = o->norm_type; S.pc_right = (o->pc_right && o->ksp_type == ORC_KSP_GMRES);
  S.pb_bs = o->pb_bs;
  S.n = n; S.ai = ai; S.aj = aj; S.aa = aa; S.inode = -1;
  S.nblocks = o->nblocks; S.blk = o->blk;
  S.hist = hist; S.hist_cap = hist_cap; S.nhist = 0;
  pc_setup(&S);
  if (S.pc_type == ORC_PC_BJACOBI) {
    for (int k = 0; k < S.nblocks; k++) {
      solver *t = &S.sub[k];
      pc_free(t);   /* sub PC type is set below, redo its set-up */
      t->ksp_type = o->blk_ksp_type ? o->blk_ksp_type[k] : o->sub_ksp_type; t->pc_type = o->blk_pc_type ? o->blk_pc_type[k] : o->sub_pc_type; t->norm_type = 1;
      t->rtol = o->blk_rtol ? o->blk_rtol[k] : o->sub_rtol; t->abstol = o->sub_abstol; t->dtol = o->sub_dtol; t->max_it = o->sub_max_it;
      t->restart = o->restart; t->refine_always = 0;
      pc_setup(t);
    }
  }
  int rc = solve(&S, b, x);
  if (nhist) *nhist = S.nhist;
  if (its) *its = S.its;
  if (reason) *reason = S.reason;
  pc_free(&S);
  return rc;
}
