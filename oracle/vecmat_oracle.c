/*
 * vecmat_oracle.c -- TEST INFRASTRUCTURE (see oracle.h).  Plain-C restatement of the
 * reference's sequential Vec and SeqAIJ/SeqBAIJ kernels.  Build: gcc -O2 -ffp-contract=off.
 */
#include "oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

/* ------------------------------------------------------------------ Vec */

/* VecSet_Seq, src/vec/vec/impls/seq/dvec2.c:722 */
void orc_vec_set(size_t n, double alpha, double *x) {
  for (size_t i = 0; i < n; i++) x[i] = alpha;
}

/* VecCopy_Seq, src/vec/vec/impls/seq/bvec2.c:464 (memcpy when x != y) */
void orc_vec_copy(size_t n, const double *x, double *y) {
  if (x != y) memmove(y, x, n * sizeof(double));
}

/* VecScale_Seq, src/vec/vec/impls/seq/bvec1.c:183: alpha==0 -> VecSet, alpha==1 -> nothing, else dscal */
void orc_vec_scale(size_t n, double alpha, double *x) {
  if (alpha == 0.0) orc_vec_set(n, 0.0, x);
  else if (alpha != 1.0)
    for (size_t i = 0; i < n; i++) x[i] = alpha * x[i];
}

/* VecSwap_Seq, src/vec/vec/impls/seq/bvec2.c:519 (dswap) */
void orc_vec_swap(size_t n, double *x, double *y) {
  if (x == y) return;
  for (size_t i = 0; i < n; i++) { double t = x[i]; x[i] = y[i]; y[i] = t; }
}

/* VecAXPY_Seq, src/vec/vec/impls/seq/bvec1.c:244: skipped for alpha==0, else daxpy (dy = dy + da*dx) */
void orc_vec_axpy(size_t n, double alpha, const double *x, double *y) {
  if (alpha == 0.0) return;
  for (size_t i = 0; i < n; i++) y[i] = y[i] + alpha * x[i];
}

/* VecAYPX_Seq, src/vec/vec/impls/seq/dvec2.c:971-1014 */
void orc_vec_aypx(size_t n, double alpha, const double *x, double *y) {
  if (alpha == 0.0) orc_vec_copy(n, x, y);
  else if (alpha == 1.0) orc_vec_axpy(n, alpha, x, y);
  else if (alpha == -1.0) { for (size_t i = 0; i < n; i++) y[i] = x[i] - y[i]; }
  else { for (size_t i = 0; i < n; i++) y[i] = x[i] + alpha * y[i]; }
}

/* VecAXPBY_Seq, src/vec/vec/impls/seq/bvec1.c:320-358 */
void orc_vec_axpby(size_t n, double alpha, double beta, const double *x, double *y) {
  if (alpha == 0.0) orc_vec_scale(n, beta, y);
  else if (beta == 1.0) orc_vec_axpy(n, alpha, x, y);
  else if (alpha == 1.0) orc_vec_aypx(n, beta, x, y);
  else if (beta == 0.0) { for (size_t i = 0; i < n; i++) y[i] = alpha * x[i]; }
  else { for (size_t i = 0; i < n; i++) y[i] = alpha * x[i] + beta * y[i]; }
}

/* VecWAXPY_Seq, src/vec/vec/impls/seq/dvec2.c:1082-1115 */
void orc_vec_waxpy(size_t n, double alpha, const double *x, const double *y, double *w) {
  if (alpha == 1.0) { for (size_t i = 0; i < n; i++) w[i] = y[i] + x[i]; }
  else if (alpha == -1.0) { for (size_t i = 0; i < n; i++) w[i] = y[i] - x[i]; }
  else if (alpha == 0.0) { memmove(w, y, n * sizeof(double)); }
  else { for (size_t i = 0; i < n; i++) w[i] = y[i] + alpha * x[i]; }
}

/* VecAXPBYPCZ_Seq, src/vec/vec/impls/seq/bvec1.c:418-455 (the non-threadcomm variant) */
void orc_vec_axpbypcz(size_t n, double alpha, double beta, double gamma, const double *x, const double *y, double *z) {
  if (alpha == 1.0) { for (size_t i = 0; i < n; i++) z[i] = x[i] + beta * y[i] + gamma * z[i]; }
  else if (gamma == 1.0) { for (size_t i = 0; i < n; i++) z[i] = alpha * x[i] + beta * y[i] + z[i]; }
  else if (gamma == 0.0) { for (size_t i = 0; i < n; i++) z[i] = alpha * x[i] + beta * y[i]; }
  else { for (size_t i = 0; i < n; i++) z[i] = alpha * x[i] + beta * y[i] + gamma * z[i]; }
}

/* VecPointwiseMult_Seq, src/vec/vec/impls/seq/bvec2.c:234-260 (aliasing cases give the same values) */
void orc_vec_pointwise_mult(size_t n, const double *x, const double *y, double *w) {
  for (size_t i = 0; i < n; i++) w[i] = x[i] * y[i];
}

/* VecPointwiseDivide_Seq, src/vec/vec/impls/seq/bvec2.c:298 */
void orc_vec_pointwise_divide(size_t n, const double *x, const double *y, double *w) {
  for (size_t i = 0; i < n; i++) w[i] = x[i] / y[i];
}

/* VecReciprocal_Default, src/vec/vec/utils/vinv.c: x[i] = 1/x[i] where x[i] != 0 */
void orc_vec_reciprocal(size_t n, double *x) {
  for (size_t i = 0; i < n; i++) if (x[i] != 0.0) x[i] = 1.0 / x[i];
}

/* VecMAXPY_Seq, src/vec/vec/impls/seq/dvec2.c:836-904 with PetscAXPY/2/3/4 of
 * include/petsc-private/petscaxpy.h:101-110: the first nv%4 vectors in one sweep, then fours. */
void orc_vec_maxpy(size_t n, int nv, const double *alpha, const double *const *y, double *x) {
  int rem = nv & 3, j;
  size_t i;
  switch (rem) {
  case 3: for (i = 0; i < n; i++) x[i] += alpha[0] * y[0][i] + alpha[1] * y[1][i] + alpha[2] * y[2][i]; break;
  case 2: for (i = 0; i < n; i++) x[i] += alpha[0] * y[0][i] + alpha[1] * y[1][i]; break;
  case 1: for (i = 0; i < n; i++) x[i] += alpha[0] * y[0][i]; break;
  }
  for (j = rem; j < nv; j += 4) {
    const double a0 = alpha[j], a1 = alpha[j + 1], a2 = alpha[j + 2], a3 = alpha[j + 3];
    const double *y0 = y[j], *y1 = y[j + 1], *y2 = y[j + 2], *y3 = y[j + 3];
    for (i = 0; i < n; i++) x[i] += a0 * y0[i] + a1 * y1[i] + a2 * y2[i] + a3 * y3[i];
  }
}

/* VecDot_Seq/VecTDot_Seq, src/vec/vec/impls/seq/bvec1.c:57,122: ddot_, netlib order (left to right) */
/* ---- the device's summation order (test aid).  The reference adds the terms of a dot product one after the other (BLAS
 * ddot / the loops of dvec2.c); the HIP reductions add the same terms in a fixed tree (csrc/vec_kernels.hip reduce_kernel:
 * min(ceil(n/4096),512) workgroups of 256 lanes, lane t of the grid takes the element pairs t, t+T, t+2T, ... in order,
 * a shuffle-down tree over each wavefront, the four wavefronts in order, then one workgroup adds the per-workgroup sums the
 * same way).  With orc_set_device_reduction_order(1) the oracle's reductions use that tree on the same terms, so that a whole
 * Krylov solve can be compared with the HIP path bit for bit: whatever differs then is not summation order. ---- */
static int dev_order = 0;
void orc_set_device_reduction_order(int on) { dev_order = on; }
typedef double (*orc_term_fn)(const double *a, const double *b, size_t i);
static double term_mul(const double *a, const double *b, size_t i) { return a[i] * b[i]; }
static double term_abs(const double *a, const double *b, size_t i) { (void)b; return fabs(a[i]); }
static double tree64(double *v) {
  for (int off = 32; off > 0; off >>= 1) for (int l = 0; l < off; l++) v[l] = v[l] + v[l + off];
  return v[0];
}
static double block256(double *lanes) {
  double w0 = tree64(lanes), w1 = tree64(lanes + 64), w2 = tree64(lanes + 128), w3 = tree64(lanes + 192);
  double s = w0; s = s + w1; s = s + w2; s = s + w3;
  return s;
}
static double dev_reduce(size_t n, orc_term_fn f, const double *a, const double *b) {
  /* csrc/vec_kernels.hip reduce_kernel / launch_reduce.  Vectors below 256 MiB: <= 512 workgroups (MI355X_REDUCE_GRID_CAP), grid-stride:
   * lane t of workgroup b meets the pairs b * 256 + t, + grid * 256, ...  Vectors of 256 MiB and more: tiles of 1024 pairs
   * (MI355X_TILE2) in contiguous runs, one run for each of <= 4096 workgroups (MI355X_REDUCE_GRID_CAP_BIG); lane t of a workgroup meets
   * its run's pairs s0 + t, s0 + t + 256, ... */
  const size_t n2 = n >> 1;
  const int runs = n * sizeof(double) >= ((size_t)256 << 20);
  size_t grid, per = 0;
  if (runs) {
    const size_t ntiles = (n2 + 1023) / 1024;
    grid = ntiles > 4096 ? 4096 : ntiles;
    per = (ntiles + grid - 1) / grid;
  } else {
    grid = (n + 4095) / 4096;
    if (grid < 1) grid = 1;
    if (grid > 512) grid = 512;
  }
  double *partial = (double *)malloc(sizeof(double) * grid), lanes[256], res;
  for (size_t blk = 0; blk < grid; blk++) {
    size_t first, step, end;
    if (runs) { first = blk * per * 1024; end = first + per * 1024; if (first > n2) first = n2; if (end > n2) end = n2; step = 256; }
    else { first = blk * 256; end = n2; step = grid * 256; }
    for (size_t t = 0; t < 256; t++) {
      double acc = 0.0;
      for (size_t i = first + t; i < end; i += step) { acc = acc + f(a, b, 2 * i); acc = acc + f(a, b, 2 * i + 1); }
      if ((n & 1) && blk == 0 && t == 0) acc = acc + f(a, b, n - 1);
      lanes[t] = acc;
    }
    partial[blk] = block256(lanes);
  }
  if (grid == 1) res = partial[0];
  else {
    for (size_t t = 0; t < 256; t++) {
      double acc = 0.0;
      for (size_t blk = t; blk < grid; blk += 256) acc = acc + partial[blk];
      lanes[t] = acc;
    }
    res = block256(lanes);
  }
  free(partial);
  return res;
}

double orc_vec_dot(size_t n, const double *x, const double *y) {
  if (dev_order) return dev_reduce(n, term_mul, x, y);
  double s = 0.0;
  for (size_t i = 0; i < n; i++) s = s + x[i] * y[i];
  return s;
}

/* One column of VecMDot_Seq, src/vec/vec/impls/seq/dvec2.c:146-342: the n%4 leading elements are
 * consumed highest index first (switch fall-through :168-183), then fours with
 * sum += x0*y0 + x1*y1 + x2*y2 + x3*y3.  The 4-vectors-at-a-time grouping does not change any
 * individual sum. */
static double mdot_one(size_t n, const double *x, const double *y) {
  double sum = 0.0;
  size_t rem = n & 3, j = n;
  switch (rem) {
  case 3: sum += x[2] * y[2]; /* fall through */
  case 2: sum += x[1] * y[1]; /* fall through */
  case 1: sum += x[0] * y[0]; /* fall through */
  case 0: x += rem; y += rem; j -= rem; break;
  }
  while (j > 0) {
    sum += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
    x += 4; y += 4; j -= 4;
  }
  return sum;
}
void orc_vec_mdot(size_t n, int nv, const double *x, const double *const *y, double *z) {
  for (int j = 0; j < nv; j++) z[j] = dev_order ? dev_reduce(n, term_mul, x, y[j]) : mdot_one(n, x, y[j]);
}

/* VecNorm_Seq, src/vec/vec/impls/seq/bvec2.c:605-642 */
void orc_vec_norm(size_t n, int type, const double *x, double *out) {
  size_t i;
  if (type == 1 || type == 2) {
    out[0] = sqrt(orc_vec_dot(n, x, x));
  } else if (type == 3) {
    double max = 0.0, tmp;
    for (i = 0; i < n; i++) {
      if ((tmp = fabs(x[i])) > max) max = tmp;
      if (tmp != tmp) { max = tmp; break; }
    }
    out[0] = max;
  } else if (type == 0) {
    double s = 0.0;
    if (dev_order) { out[0] = dev_reduce(n, term_abs, x, x); return; }
    for (i = 0; i < n; i++) s = s + fabs(x[i]);   /* dasum */
    out[0] = s;
  } else if (type == 4) {
    orc_vec_norm(n, 0, x, out);
    orc_vec_norm(n, 1, x, out + 1);
  }
}

/* VecDotNorm2 default branch, src/vec/vec/utils/vinv.c:1222-1234 */
void orc_vec_dotnorm2(size_t n, const double *s, const double *t, double *dp, double *nm) {
  double dpx = 0.0, nmx = 0.0;
  if (dev_order) { *dp = dev_reduce(n, term_mul, s, t); *nm = dev_reduce(n, term_mul, t, t); return; }
  for (size_t i = 0; i < n; i++) { dpx += s[i] * t[i]; nmx += t[i] * t[i]; }
  *dp = dpx; *nm = nmx;
}

/* ------------------------------------------------------------------ SeqAIJ */

/* MatMult_SeqAIJ, src/mat/impls/aij/seq/aij.c:1269-1277 with PetscSparseDensePlusDot aij.h:383-386 */
void orc_spmv_csr(int m, const int *ai, const int *aj, const double *aa, const double *x, double *y) {
  for (int i = 0; i < m; i++) {
    double sum = 0.0;
    for (int k = ai[i]; k < ai[i + 1]; k++) sum += aa[k] * x[aj[k]];
    y[i] = sum;
  }
}

/* MatMultAdd_SeqAIJ, src/mat/impls/aij/seq/aij.c:1334-1341 */
void orc_spmv_csr_add(int m, const int *ai, const int *aj, const double *aa, const double *x, const double *y, double *z) {
  for (int i = 0; i < m; i++) {
    double sum = y[i];
    for (int k = ai[i]; k < ai[i + 1]; k++) sum += aa[k] * x[aj[k]];
    z[i] = sum;
  }
}

/* MatMultTransposeAdd_SeqAIJ, src/mat/impls/aij/seq/aij.c:1078-1119: y = z; y[j[k]] += x[i]*a[k] */
void orc_spmv_csr_transpose_add(int m, int n, const int *ai, const int *aj, const double *aa, const double *x,
                                const double *z, double *y) {
  if (z != y) memmove(y, z, (size_t)n * sizeof(double));
  for (int i = 0; i < m; i++) {
    const double alpha = x[i];
    for (int k = ai[i]; k < ai[i + 1]; k++) y[aj[k]] += alpha * aa[k];
  }
}

/* MatMultTranspose_SeqAIJ, src/mat/impls/aij/seq/aij.c:1124-1133: VecSet(y,0) then the Add form */
void orc_spmv_csr_transpose(int m, int n, const int *ai, const int *aj, const double *aa, const double *x, double *y) {
  for (int j = 0; j < n; j++) y[j] = 0.0;
  orc_spmv_csr_transpose_add(m, n, ai, aj, aa, x, y, y);
}

/* MatGetDiagonal_SeqAIJ, src/mat/impls/aij/seq/aij.c:1040-1073: linear search of each row, 0 if absent */
/* MatDiagonalScale_SeqAIJ, aij.c:2055-2092: the left pass over all entries, then the right pass; either may be NULL */
void orc_csr_diagonal_scale(int m, const int *ai, const int *aj, double *aa, const double *l, const double *r) {
  if (l) for (int i = 0; i < m; i++) for (int k = ai[i]; k < ai[i + 1]; k++) aa[k] *= l[i];
  if (r) for (int k = 0; k < ai[m]; k++) aa[k] *= r[aj[k]];
}
void orc_csr_get_diagonal(int m, const int *ai, const int *aj, const double *aa, double *d) {
  for (int i = 0; i < m; i++) {
    d[i] = 0.0;
    for (int k = ai[i]; k < ai[i + 1]; k++) if (aj[k] == i) { d[i] = aa[k]; break; }
  }
}

/* Counting-sort transpose; stable, so row c of A^T lists (r, a_rc) in increasing r. */
void orc_csr_transpose(int m, int n, const int *ai, const int *aj, const double *aa, int *ti, int *tj, double *ta) {
  int nz = ai[m];
  for (int j = 0; j <= n; j++) ti[j] = 0;
  for (int k = 0; k < nz; k++) ti[aj[k] + 1]++;
  for (int j = 0; j < n; j++) ti[j + 1] += ti[j];
  int *next = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  for (int j = 0; j < n; j++) next[j] = ti[j];
  for (int i = 0; i < m; i++)
    for (int k = ai[i]; k < ai[i + 1]; k++) {
      int p = next[aj[k]]++;
      tj[p] = i;
      ta[p] = aa[k];
    }
  free(next);
}

/* ------------------------------------------------------------------ SeqBAIJ */

/* MatMult_SeqBAIJ_3/_4/_N, src/mat/impls/baij/seq/baij2.c:331-383, 387-436, 981-1032: per block
 * row, bs running sums; each block (column-major) adds sum_r += v[r] x0 + v[r+bs] x1 + ... as one
 * left-to-right expression (bs = 3, 4 spelled out in the reference; _N uses a dgemv-like loop,
 * restated here in the same column order). */
void orc_spmv_bsr(int mbs, int bs, const int *ai, const int *aj, const double *aa, const double *x, double *y) {
  const int bs2 = bs * bs;
  double sum[16];
  for (int i = 0; i < mbs; i++) {
    for (int r = 0; r < bs; r++) sum[r] = 0.0;
    for (int k = ai[i]; k < ai[i + 1]; k++) {
      const double *v = aa + (size_t)k * bs2;
      const double *xb = x + (size_t)aj[k] * bs;
      for (int r = 0; r < bs; r++) {
        double t = v[r] * xb[0];
        for (int c = 1; c < bs; c++) t = t + v[r + c * bs] * xb[c];
        sum[r] += t;
      }
    }
    for (int r = 0; r < bs; r++) y[(size_t)i * bs + r] = sum[r];
  }
}

/* ------------------------------------------------------------------ SeqAIJ, inode variant */

/* Mat_CheckInode, src/mat/impls/aij/seq/inode.c:3964-4034: consecutive rows with identical column
 * pattern are grouped, at most `limit` (default 5) rows per node.  ns[] receives the node sizes.
 * Returns the node count, or 0 when the reference would NOT switch to the inode routines
 * (node_count > 0.8 m). */
int orc_check_inode(int m, const int *ai, const int *aj, int limit, int *ns) {
  int i = 0, node_count = 0;
  const int *idx = aj;
  while (i < m) {
    int nzx = ai[i + 1] - ai[i], j, blk = 1;
    const int *idy = idx;
    for (j = i + 1; j < m && blk < limit; ++j, ++blk) {
      int nzy = ai[j + 1] - ai[j];
      if (nzy != nzx) break;
      idy += nzx;
      if (memcmp(idx, idy, (size_t)nzx * sizeof(int))) break;
    }
    ns[node_count++] = blk;
    idx += (size_t)blk * nzx;
    i = j;
  }
  if (!m || node_count > .8 * m) return 0;
  return node_count;
}

/* MatMult_SeqAIJ_Inode, src/mat/impls/aij/seq/inode.c:392-578: whatever the node size (cases 1..5), every
 * row's sum is built two products at a time, sum += v[0]*x0 + v[1]*x1, with a single trailing product
 * when the row length is odd -- a different rounding from MatMult_SeqAIJ's one-at-a-time loop. */
void orc_spmv_csr_inode(int m, const int *ai, const int *aj, const double *aa, const double *x, double *y) {
  for (int i = 0; i < m; i++) {
    const int s = ai[i], sz = ai[i + 1] - ai[i];
    double sum = 0.0;
    int n;
    for (n = 0; n < sz - 1; n += 2) sum += aa[s + n] * x[aj[s + n]] + aa[s + n + 1] * x[aj[s + n + 1]];
    if (n == sz - 1) sum += aa[s + n] * x[aj[s + n]];
    y[i] = sum;
  }
}
/* MatMultAdd_SeqAIJ_Inode, inode.c:583-760: the same loops with every row's sum started from z[row] */
void orc_spmv_csr_inode_add(int m, const int *ai, const int *aj, const double *aa, const double *x, const double *z, double *y) {
  for (int i = 0; i < m; i++) {
    const int s = ai[i], sz = ai[i + 1] - ai[i];
    double sum = z[i];
    int n;
    for (n = 0; n < sz - 1; n += 2) sum += aa[s + n] * x[aj[s + n]] + aa[s + n + 1] * x[aj[s + n + 1]];
    if (n == sz - 1) sum += aa[s + n] * x[aj[s + n]];
    y[i] = sum;
  }
}
/* What MatMult / MatMultAdd dispatch to for a MATSEQAIJ matrix (inode.use defaults to true, inode2.c:85-99): the inode
 * routines when Mat_CheckInode keeps them (node_count <= 0.8 m), the plain ones otherwise.  z == NULL: MatMult. */
int orc_matmult_seqaij(int m, const int *ai, const int *aj, const double *aa, const double *x, const double *z, double *y, int *ns_work) {
  const int nodes = orc_check_inode(m, ai, aj, 5, ns_work);
  if (nodes) { if (z) orc_spmv_csr_inode_add(m, ai, aj, aa, x, z, y); else orc_spmv_csr_inode(m, ai, aj, aa, x, y); }
  else       { if (z) orc_spmv_csr_add(m, ai, aj, aa, x, z, y); else orc_spmv_csr(m, ai, aj, aa, x, y); }
  return nodes;
}


/* ---- point-block Jacobi: src/ksp/pc/impls/pbjacobi/pbjacobi.c ----
 * PetscKernel_A_gets_inverse_A_N (src/mat/impls/baij/seq/dgefa.c, dgefa2.c .. dgefa7.c, dgedi.c): LINPACK dgefa (Gaussian
 * elimination with partial pivoting, multipliers stored negated) followed by dgedi (inverse(U), then inverse(U)*inverse(L),
 * column interchanges undone), on a column-major n x n block, in place; no shift.  Returns the 0-based zero-pivot row + 1, or 0. */
int orc_block_inverse(int n, double *a) {
  int ipvt[16];
  double work[16];
  if (n < 1 || n > 16) return -1;
#define A_(i, j) a[(i) + (j) * n]
  for (int k = 0; k < n - 1; k++) {
    int l = k;
    double max = fabs(A_(k, k));
    for (int i = k + 1; i < n; i++) { const double t = fabs(A_(i, k)); if (t > max) { max = t; l = i; } }
    ipvt[k] = l;
    if (A_(l, k) == 0.0) return k + 1;
    if (l != k) { const double t = A_(l, k); A_(l, k) = A_(k, k); A_(k, k) = t; }
    { const double t = -1. / A_(k, k); for (int i = k + 1; i < n; i++) A_(i, k) *= t; }
    for (int j = k + 1; j < n; j++) {
      const double t = A_(l, j);
      if (l != k) { A_(l, j) = A_(k, j); A_(k, j) = t; }
      for (int i = k + 1; i < n; i++) A_(i, j) += t * A_(i, k);
    }
  }
  ipvt[n - 1] = n - 1;
  if (A_(n - 1, n - 1) == 0.0) return n;
  for (int k = 0; k < n; k++) {                 /* inverse(U) */
    A_(k, k) = 1.0 / A_(k, k);
    { const double t = -A_(k, k); for (int i = 0; i < k; i++) A_(i, k) *= t; }
    for (int j = k + 1; j < n; j++) {
      const double t = A_(k, j);
      A_(k, j) = 0.0;
      for (int i = 0; i <= k; i++) A_(i, j) += t * A_(i, k);
    }
  }
  for (int kb = 1; kb < n; kb++) {              /* inverse(U) * inverse(L) */
    const int k = n - 1 - kb;
    for (int i = k + 1; i < n; i++) { work[i] = A_(i, k); A_(i, k) = 0.0; }
    for (int j = k + 1; j < n; j++) { const double t = work[j]; for (int i = 0; i < n; i++) A_(i, k) += t * A_(i, j); }
    const int l = ipvt[k];
    if (l != k) for (int i = 0; i < n; i++) { const double t = A_(i, k); A_(i, k) = A_(i, l); A_(i, l) = t; }
  }
#undef A_
  return 0;
}
/* MatInvertBlockDiagonal_SeqBAIJ (baij.c:13-160): the diagonal block of every block row, inverted; blocks column-major */
int orc_bsr_invert_block_diagonal(int mbs, int bs, const int *ai, const int *aj, const double *aa, double *idiag) {
  const int bs2 = bs * bs;
  for (int i = 0; i < mbs; i++) {
    int found = 0;
    for (int k = ai[i]; k < ai[i + 1]; k++) if (aj[k] == i) { memcpy(idiag + (size_t)i * bs2, aa + (size_t)k * bs2, sizeof(double) * (size_t)bs2); found = 1; break; }
    if (!found) return -(i + 1);
    const int z = orc_block_inverse(bs, idiag + (size_t)i * bs2);
    if (z) return i * bs + z;
  }
  return 0;
}
/* PCApply_PBJacobi_N (pbjacobi.c:20-200): y_i = D_i^-1 x_i, every row the left-to-right sum d[r]*x0 + d[r+bs]*x1 + ... */
void orc_pbjacobi_apply(int mbs, int bs, const double *idiag, const double *x, double *y) {
  const int bs2 = bs * bs;
  for (int i = 0; i < mbs; i++) {
    const double *d = idiag + (size_t)i * bs2, *xx = x + (size_t)i * bs;
    for (int r = 0; r < bs; r++) {
      double sum = d[r] * xx[0];
      for (int c = 1; c < bs; c++) sum += d[r + c * bs] * xx[c];
      y[(size_t)i * bs + r] = sum;
    }
  }
}
