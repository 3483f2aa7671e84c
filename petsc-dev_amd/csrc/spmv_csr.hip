// CSR SpMV for gfx950: y = A x and z = y + A x  (MatMult_SeqAIJ / MatMultAdd_SeqAIJ,
// reference src/mat/impls/aij/seq/aij.c:1225-1358).
//
// "Row-block streaming" layout of the work (HBM-bound, AI = 0.125 flop/B):
//   * host analysis cuts the rows into row blocks of <= 256 rows and <= 2046 nonzeros;
//   * one 256-thread workgroup per row block streams that block's val/col_idx slice with
//     fully coalesced 16-byte (val) / 8-byte (col) non-temporal loads, gathers x through
//     L1/L2 (x is the only reused operand, so val/col are kept out of the cache with `nt`),
//     multiplies, and parks the products in LDS.  All loads are issued unconditionally and
//     ahead of their use (clamped addresses instead of predication, see the idx8 kernel);
//   * after one barrier each row is summed from LDS: one lane per row, products added in
//     column order starting from 0.0 (or y[r]) -- the exact order of PetscSparseDensePlusDot
//     (aij.h:383-386), so the result is bit-identical to the reference's non-FMA C loop.
//     Row blocks with few, long rows use 2..64 lanes per row and a shuffle tree instead;
//   * a row longer than 2046 nonzeros gets a whole workgroup (strided partial sums + tree);
//   * blockIdx is remapped so that runs of SPMV_CH consecutive row blocks are dealt round-robin
//     to the 8 XCDs: all XCDs stream one window of the matrix while the x entries a 7-point
//     row needs (r, r+-1, r+-N, r+-N^2) are re-used out of one XCD's own 4 MiB L2;
//   * variants: offset-dictionary index compression (1 byte per nonzero instead of 4), an
//     x'y by-product for CG, and the BCSR form of the same structure.
#include "common.hpp"
#include <map>
#include <array>
#include <memory>
#include <vector>
#include <thread>
#include <functional>
#include <stdlib.h>
#include <string.h>

// tuning knobs (the defaults are the measured best on MI355X for the 7-point operator)
#ifndef SPMV_THREADS
#define SPMV_THREADS 256
#endif
#ifndef SPMV_NT
#define SPMV_NT 1            // non-temporal loads for the val/col streams
#endif
#ifndef SPMV_REMAP
#define SPMV_REMAP 2         // 0 = dispatch order; 1 = each XCD walks a contiguous eighth; 2 = runs of SPMV_CH row blocks dealt round-robin to the XCDs (same speed as 0, 18 % less fabric traffic: x lines stay in one XCD's L2)
#endif
#ifndef SPMV_CH
#define SPMV_CH 32
#endif
#ifndef SPMV_MINWAVES
#define SPMV_MINWAVES 1      // __launch_bounds__ second argument (waves per SIMD the register allocator must allow)
#endif
#ifndef SPMV_SEQ_AVG
#define SPMV_SEQ_AVG 16      // row blocks with <= this many nonzeros per row on average: one lane per row, reference summation order
#endif
#define SPMV_GJ_CAP 960      // grouped-row kernel: shared column indices per row block (LDS sized for 7 workgroups per CU)
#define SPMV_BLOCK_NNZ (8 * SPMV_THREADS)   // LDS stage (doubles): 4 pairs per lane
#define SPMV_BLOCK_CAP (SPMV_BLOCK_NNZ - 2)   // nonzeros per row block: any alignment of the first pair still fits
#define SPMV_BLOCK_ROWS SPMV_THREADS
#if SPMV_NT
#define SPMV_LOAD(p) __builtin_nontemporal_load(p)
#else
#define SPMV_LOAD(p) (*(p))
#endif

typedef double v2d __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));

struct mi355x_spmv_plan_s {
  int nrows;       // rows of the (possibly compressed) row pointer
  int nblocks;     // row blocks
  int nlong;       // of which single long rows
  int chunk;       // ceil(nblocks / NXCD)
  int2 *d_rowblk;  // nblocks+1 entries {first row, first nonzero}
  int *d_rows;     // compressed-row output indices or NULL
  // offset-dictionary index compression (col = row + table[idx8]); NULL when the matrix has > 256 distinct offsets
  unsigned char *d_idx8;
  int *d_offtab;
  int ntab;
  // row patterns (stencil matrices): the offset lists of the rows come from a small dictionary; per row the start of its
  // list in the table (2 bytes per ROW instead of 1 byte per nonzero)
  unsigned int *d_prow;
  int *d_pattab;       // SPMV_PAT_CAP ints
  int npat, use_pat;
  int ch;              // run length (row blocks) of the row-pattern kernel's block -> XCD map, from the operator's largest offset
  // value patterns (constant-coefficient operators): rows whose offsets AND values repeat; per row 2 bytes, the values
  // live in the table.  Valid only for the values they were derived from (mi355x_spmv_plan_value_patterns / _drop_)
  unsigned short *d_vrow;
  int *d_vpattab;      // SPMV_PAT_CAP ints   {length, offsets ...}
  double *d_vpatval;   // SPMV_PAT_CAP doubles, value q of an entry at the index of its offset q
  int nvpat, vtablen, vpat_valid, use_vpat;
  double *d_dotpart;   // per-row-block x'y values of mi355x_spmv_csr_dot (allocated on first use)
  int ndotpart;        // how many of them the last mi355x_spmv_csr_dot wrote
  // rows summed the way MatMult_SeqAIJ_Inode does (two products at a time, inode.c:392-578): set when the reference's
  // Mat_CheckInode would switch this matrix to its inode routines
  int pairsum;
  // grouped rows (the MI355X form of the reference's inodes): consecutive rows with one column pattern share ONE
  // stored column list; d_rowblk4 = {first row, first nonzero, first shared column index, 0} per row block
  int4 *d_rowblk4;
  int *d_goff;         // per row: where its group's column list starts in d_gj
  int *d_gj;           // the groups' column lists, one after the other
  int ngroups;
  long ngj;
};

// One lane's row sum out of the LDS product stage, 8 reads in flight.  pairsum == 0: products added one at a time in
// column order (PetscSparseDensePlusDot, aij.h:383-386).  pairsum != 0: two at a time, sum += p[n] + p[n+1], odd
// tail alone (MatMult_SeqAIJ_Inode / MatMultAdd_SeqAIJ_Inode, inode.c:430-440,619-631).
__device__ __forceinline__ double row_sum_lds(const double *prod, int rs, int re, double sum, int pairsum) {
  if (!pairsum) {
    for (int k = rs; k < re; k += 8) {
      double t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = prod[(k + j < re) ? k + j : re - 1];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const double u = sum + t[j]; sum = (k + j < re) ? u : sum; }
    }
  } else {
    for (int k = rs; k < re; k += 8) {       // rs + multiples of 8: pairs stay aligned with the row start
      double t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = prod[(k + j < re) ? k + j : re - 1];
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const double pr = t[j] + t[j + 1];
        const double inc = (k + j + 1 < re) ? pr : t[j];
        const double u = sum + inc;
        sum = (k + j < re) ? u : sum;
      }
    }
  }
  return sum;
}

// How a row's sum reaches y.  ADD == 0: y = A x.  ADD == 1: y = yin + A x (MatMultAdd).  ADD == 2: y = yin .* (A x), yin being a
// diagonal scaling (PCApply_Jacobi's VecPointwiseMult fused into the product: same bits as the two separate sweeps).
template <int ADD> __device__ __forceinline__ double spmv_out(double yv, double t) { return ADD == 1 ? yv + t : (ADD == 2 ? yv * t : t); }
template <int ADD> __device__ __forceinline__ double spmv_empty(double yv) { return ADD == 1 ? yv : (ADD == 2 ? yv * 0.0 : 0.0); }
// same, for a sum that was started from yv when ADD == 1
template <int ADD> __device__ __forceinline__ double spmv_fin(double yv, double s) { return ADD == 2 ? yv * s : s; }
static inline int spmv_y_streams(long nrows) { return (size_t)nrows * sizeof(double) >= ((size_t)256 << 20); }   // (vec_kernels.hip: vec_streams)

template <int ADD, bool CPROW, bool VEC>
__global__ __launch_bounds__(SPMV_THREADS, SPMV_MINWAVES) void spmv_csr_rowblock_kernel(
    const int2 *__restrict__ rowblk, int nblocks, int chunk, const int *__restrict__ ai, const int *__restrict__ aj,
    const double *__restrict__ aa, const double *__restrict__ x, const double *yin, double *yout,
    const int *__restrict__ rows, int pairsum) {
  __shared__ double prod[SPMV_BLOCK_NNZ];
  __shared__ double wsum[SPMV_THREADS / MI355X_WAVE];

  // XCD-aware remap: workgroups b, b+8, b+16.. share an XCD; give them consecutive row blocks
#if SPMV_REMAP == 1
  const int xcd = blockIdx.x % MI355X_NXCD;
  const int slot = blockIdx.x / MI355X_NXCD;
  const int lb = xcd * chunk + slot;
  if (slot >= chunk || lb >= nblocks) return;
#elif SPMV_REMAP == 2
  // interleaved: XCD x owns runs of SPMV_CH consecutive row blocks, runs dealt round-robin over the XCDs, so all
  // XCDs stream one window of 8*SPMV_CH blocks while each re-uses its own x lines
  const int xcd = blockIdx.x % MI355X_NXCD;
  const int slot = blockIdx.x / MI355X_NXCD;
  const int lb = ((slot / SPMV_CH) * MI355X_NXCD + xcd) * SPMV_CH + (slot % SPMV_CH);
  if (lb >= nblocks) return;
#else
  const int lb = blockIdx.x;
  if (lb >= nblocks) return;
#endif

  // {first row, first nonzero} of this row block and of the next: one round trip, no dependent chain
  const int2 b0 = rowblk[lb];
  const int2 b1 = rowblk[lb + 1];
  const int r0 = b0.x, r1 = b1.x, k0 = b0.y, k1 = b1.y;
  const int nnz = k1 - k0;
  const int nrows = r1 - r0;
  const int tid = threadIdx.x;

  if (nnz > SPMV_BLOCK_CAP) {
    // one long row: strided partial sums, then a fixed tree
    double s = 0.0;
    for (int k = k0 + tid; k < k1; k += SPMV_THREADS) {
      double v = SPMV_LOAD(aa + k);
      int c = SPMV_LOAD(aj + k);
      s += v * x[c];
    }
    s = wave_sum(s);
    if ((tid & (MI355X_WAVE - 1)) == 0) wsum[tid / MI355X_WAVE] = s;
    __syncthreads();
    if (tid == 0) {
      double t = wsum[0];
#pragma unroll
      for (int w = 1; w < SPMV_THREADS / MI355X_WAVE; ++w) t += wsum[w];
      const int orow = CPROW ? rows[r0] : r0;
      yout[orow] = spmv_out<ADD>(ADD ? yin[orow] : 0.0, t);
    }
    return;
  }
  if (nnz == 0) {   // only empty rows
    if (tid < nrows) {
      const int orow = CPROW ? rows[r0 + tid] : r0 + tid;
      yout[orow] = spmv_empty<ADD>(ADD ? yin[orow] : 0.0);
    }
    return;
  }

  // lanes per row: largest power of two with nrows*tpr <= 256, at most one wavefront; short rows
  // (the stencil case) get one lane per row and the reference's summation order
  int tpr = 1;
  while (tpr < MI355X_WAVE && nrows * (tpr * 2) <= SPMV_THREADS) tpr *= 2;
  if (nnz <= SPMV_SEQ_AVG * nrows) tpr = 1;

  // Every load below is unconditional (see the idx8 kernel further down for why): lanes past the last row re-read
  // the last row's extent, lanes past the last pair re-read the block's first pair.
  const int r = tid / tpr, sub = tid & (tpr - 1);
  const int rc = r < nrows ? r : nrows - 1;
  const int a0 = ai[r0 + rc], a1 = ai[r0 + rc + 1];
  const int orow = CPROW ? rows[r0 + rc] : r0 + rc;
  double ysum = 0.0;
  if (ADD) ysum = yin[orow];

  // ---- stream the block's nonzeros: product -> LDS -----------------------
  if (VEC) {
    constexpr int PAIRS = SPMV_BLOCK_NNZ / (2 * SPMV_THREADS);
    const int ka = k0 & ~1;  // 16-byte aligned start for val, 8-byte for col
    v2d v[PAIRS];
    v2i c[PAIRS];
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
      const int k = ka + 2 * tid + p * 2 * SPMV_THREADS;
      const int kk = (k < k1) ? k : ka;
      v[p] = SPMV_LOAD(reinterpret_cast<const v2d *>(aa + kk));
      c[p] = SPMV_LOAD(reinterpret_cast<const v2i *>(aj + kk));
    }
    double xa[PAIRS], xb[PAIRS];
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
      const int k = ka + 2 * tid + p * 2 * SPMV_THREADS;
      const bool in = k < k1;
      const bool v0 = in && k >= k0, v1 = in && (k + 1 < k1);
      // a slot whose own element lies outside the block gathers with the column of the pair's other element, an
      // idle lane (it holds the block's first pair) with that of element k0: always a column of this block
      const int s0 = v0 ? 0 : (in ? 1 : (k0 & 1));
      const int s1 = v1 ? 1 : (in ? 0 : (k0 & 1));
      xa[p] = x[s0 ? c[p].y : c[p].x];
      xb[p] = x[s1 ? c[p].y : c[p].x];
    }
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
      const int k = ka + 2 * tid + p * 2 * SPMV_THREADS;
      const double pa = v[p].x * xa[p], pb = v[p].y * xb[p];
      // products of elements outside the block go to the stage's last slot, which no block uses (nnz <= CAP)
      prod[(k >= k0 && k < k1) ? k - k0 : SPMV_BLOCK_NNZ - 1] = pa;
      prod[(k + 1 < k1) ? k + 1 - k0 : SPMV_BLOCK_NNZ - 1] = pb;
    }
  } else {
    for (int k = k0 + tid; k < k1; k += SPMV_THREADS)
      prod[k - k0] = SPMV_LOAD(aa + k) * x[SPMV_LOAD(aj + k)];
  }
  __syncthreads();

  // ---- per-row sums out of LDS -------------------------------------------
  const int rs = r < nrows ? a0 - k0 : 0, re = r < nrows ? a1 - k0 : 0;
  if (tpr == 1) {
    if (r < nrows) yout[orow] = spmv_fin<ADD>(ysum, row_sum_lds(prod, rs, re, ADD == 1 ? ysum : 0.0, pairsum));
  } else {
    double sum = 0.0;
    if (r < nrows) for (int k = rs + sub; k < re; k += tpr) sum += prod[k];
    for (int off = tpr >> 1; off > 0; off >>= 1) sum += __shfl_down(sum, off, MI355X_WAVE);
    if (r < nrows && sub == 0) yout[orow] = spmv_out<ADD>(ysum, sum);
  }
}

// ---------------------------------------------------------------------------------------------
// Index-compressed variant.  A matrix whose entries use at most 256 distinct offsets (col - row) -- every
// stencil operator: 7 for the 3-D Poisson matrix, 135 for 3-dof 27-point elasticity -- gets, at analysis time,
// one byte per nonzero (the position of its offset in a table) next to the unchanged CSR arrays.  The kernel
// streams val (8 B) + idx8 (1 B) instead of val + col (12 B): 25 % fewer matrix bytes.  The row of each nonzero,
// which the offset needs, comes from an LDS marker array written by the lanes that own the rows (they hold the
// row extents anyway).  Arithmetic and summation order are those of the plain kernel: same bits.
// Branch-free addressing: every lane issues all of its loads unconditionally -- a lane whose pair lies outside the
// block re-reads the block's first pair (same address as lane 0: merged by the coalescer), a pair cut by the block
// boundary is loaded whole (an aligned 16-byte access cannot leave the page its first half lies in) and the element
// outside the block is simply not written to LDS.  With no control flow between the loads the compiler keeps all of
// them in flight behind a single s_waitcnt; the earlier predicated form serialised them pair by pair.
template <int ADD, bool DOT>
__global__ __launch_bounds__(SPMV_THREADS, SPMV_MINWAVES) void spmv_csr_rowblock_idx8_kernel(
    const int2 *__restrict__ rowblk, int nblocks, const int *__restrict__ ai, const unsigned char *__restrict__ idx8,
    const int *__restrict__ offtab_g, int ntab, const double *__restrict__ aa, const double *__restrict__ x,
    const double *yin, double *yout, double *__restrict__ dotpart, int pairsum) {
  __shared__ double prod[SPMV_BLOCK_NNZ];
  __shared__ unsigned char rowof[SPMV_BLOCK_NNZ];
  __shared__ int offtab[256];
  __shared__ double wsum[SPMV_THREADS / MI355X_WAVE];
  static_assert(SPMV_THREADS == 256 || SPMV_THREADS == 128 || SPMV_THREADS == 512, "the offset table is staged 256 / SPMV_THREADS entries per lane");
  static_assert(!(DOT && ADD != 0), "the x'y by-product is provided for y = A x only");
#if SPMV_REMAP == 2
  const int xcd = blockIdx.x % MI355X_NXCD;
  const int slot = blockIdx.x / MI355X_NXCD;
  const int lb = ((slot / SPMV_CH) * MI355X_NXCD + xcd) * SPMV_CH + (slot % SPMV_CH);
#else
  const int lb = blockIdx.x;
#endif
  if (lb >= nblocks) return;
  const int2 b0 = rowblk[lb];
  const int2 b1 = rowblk[lb + 1];
  const int r0 = b0.x, r1 = b1.x, k0 = b0.y, k1 = b1.y;
  const int nnz = k1 - k0;
  const int nrows = r1 - r0;
  const int tid = threadIdx.x;
  const int tabv = offtab_g[tid & 255];   // 256 initialised entries (zeros past ntab)
  const int tabv2 = (SPMV_THREADS < 256) ? offtab_g[(tid + SPMV_THREADS) & 255] : 0;
  (void)ntab;

  if (nnz > SPMV_BLOCK_CAP) {   // one long row: every entry belongs to row r0
    if (tid < 256) offtab[tid] = tabv;
    if (SPMV_THREADS < 256) offtab[tid + SPMV_THREADS] = tabv2;
    __syncthreads();
    double s = 0.0;
    for (int k = k0 + tid; k < k1; k += SPMV_THREADS) s += SPMV_LOAD(aa + k) * x[r0 + offtab[SPMV_LOAD(idx8 + k)]];
    s = wave_sum(s);
    if ((tid & (MI355X_WAVE - 1)) == 0) wsum[tid / MI355X_WAVE] = s;
    __syncthreads();
    if (tid == 0) {
      double t = wsum[0];
#pragma unroll
      for (int w = 1; w < SPMV_THREADS / MI355X_WAVE; ++w) t += wsum[w];
      const double yv = spmv_out<ADD>(ADD ? yin[r0] : 0.0, t);
      yout[r0] = yv;
      if (DOT) dotpart[lb] = yv * x[r0];
    }
    return;
  }
  if (nnz == 0) {               // only empty rows
    if (tid < nrows) yout[r0 + tid] = spmv_empty<ADD>(ADD ? yin[r0 + tid] : 0.0);
    if (DOT && tid == 0) dotpart[lb] = 0.0;
    return;
  }

  int tpr = 1;
  while (tpr < MI355X_WAVE && nrows * (tpr * 2) <= SPMV_THREADS) tpr *= 2;
  if (nnz <= SPMV_SEQ_AVG * nrows) tpr = 1;
  const int r = tid / tpr, sub = tid & (tpr - 1);
  const int rc = r < nrows ? r : nrows - 1;
  const int a0 = ai[r0 + rc], a1 = ai[r0 + rc + 1];
  double ysum = 0.0;
  if (ADD) ysum = yin[r0 + rc];
  double xrow = 0.0;
  if (DOT) xrow = x[r0 + rc];   // square matrix: x entry of this lane's row, for the x'y by-product
  // this lane's slice of the value / index streams
  constexpr int PAIRS = SPMV_BLOCK_NNZ / (2 * SPMV_THREADS);
  const int ka = k0 & ~1;
  v2d v[PAIRS];
  unsigned int ix[PAIRS];
#pragma unroll
  for (int p = 0; p < PAIRS; ++p) {
    const int k = ka + 2 * tid + p * 2 * SPMV_THREADS;
    const int kk = (k < k1) ? k : ka;
    v[p] = SPMV_LOAD(reinterpret_cast<const v2d *>(aa + kk));
    ix[p] = SPMV_LOAD(reinterpret_cast<const unsigned short *>(idx8 + kk));
  }
  const int rs = r < nrows ? a0 - k0 : 0, re = r < nrows ? a1 - k0 : 0;
  // row markers: the lanes of row r tag its nonzeros
  for (int k = rs + sub; k < re; k += tpr) rowof[k] = (unsigned char)r;
  if (tid < 256) offtab[tid] = tabv;
  if (SPMV_THREADS < 256) offtab[tid + SPMV_THREADS] = tabv2;
  __syncthreads();
  double xa[PAIRS], xb[PAIRS];
#pragma unroll
  for (int p = 0; p < PAIRS; ++p) {
    const int k = ka + 2 * tid + p * 2 * SPMV_THREADS;
    const bool in = k < k1;                       // the pair touches the block
    const bool v0 = in && k >= k0, v1 = in && (k + 1 < k1);
    const int kk = in ? k : ka;
    // element to address x with when the slot's own element lies outside the block: the pair's other element,
    // or (idle lane, pair = the block's first) element k0
    const int s0 = v0 ? 0 : (in ? 1 : (k0 & 1));
    const int s1 = v1 ? 1 : (in ? 0 : (k0 & 1));
    xa[p] = x[r0 + rowof[kk + s0 - k0] + offtab[(ix[p] >> (8 * s0)) & 0xff]];
    xb[p] = x[r0 + rowof[kk + s1 - k0] + offtab[(ix[p] >> (8 * s1)) & 0xff]];
  }
#pragma unroll
  for (int p = 0; p < PAIRS; ++p) {
    const int k = ka + 2 * tid + p * 2 * SPMV_THREADS;
    const double pa = v[p].x * xa[p], pb = v[p].y * xb[p];
    // products of elements outside the block go to the stage's last slot, which no block uses (nnz <= CAP)
    prod[(k >= k0 && k < k1) ? k - k0 : SPMV_BLOCK_NNZ - 1] = pa;
    prod[(k + 1 < k1) ? k + 1 - k0 : SPMV_BLOCK_NNZ - 1] = pb;
  }
  __syncthreads();
  double yval = 0.0;             // this lane's row result (lanes without a row: 0)
  if (tpr == 1) {
    if (r < nrows) {
      const double sum = spmv_fin<ADD>(ysum, row_sum_lds(prod, rs, re, ADD == 1 ? ysum : 0.0, pairsum));
      yout[r0 + r] = sum;
      yval = sum;
    }
  } else {
    double sum = 0.0;
    if (r < nrows) for (int k = rs + sub; k < re; k += tpr) sum += prod[k];
    for (int off = tpr >> 1; off > 0; off >>= 1) sum += __shfl_down(sum, off, MI355X_WAVE);
    if (r < nrows && sub == 0) { yval = spmv_out<ADD>(ysum, sum); yout[r0 + r] = yval; }
  }
  if (DOT) {
    // x'y by-product: this block's sum of x_r y_r in a fixed order (lanes -> wavefront tree -> 4 wavefronts in order);
    // mi355x_spmv_csr_dot sums the per-block values in block order afterwards
    const bool mine = (r < nrows) && (sub == 0);
    double c = wave_sum(mine ? yval * xrow : 0.0);
    if ((tid & (MI355X_WAVE - 1)) == 0) wsum[tid / MI355X_WAVE] = c;
    __syncthreads();
    if (tid == 0) {
      double t = wsum[0];
#pragma unroll
      for (int w = 1; w < SPMV_THREADS / MI355X_WAVE; ++w) t += wsum[w];
      dotpart[lb] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Row-pattern variant for stencil matrices.  When the rows' offset lists (col - row, in column order) come from a small
// dictionary -- 27 lists of <= 7 offsets for the 7-point operator on a box: interior rows and the boundary cases -- the
// analysis stores, per ROW, one 4-byte word -- where its list starts in a table, and the row's first nonzero relative to its row
// block -- and nothing per nonzero: the kernel streams the values (8 B per nonzero) and 4 bytes per row; the row pointer is
// not read at all (a table entry carries its list's length).  Work layout: the block's values go to LDS with coalesced 16-byte loads; after ONE
// barrier lane r owns row r and gathers x[row + offset_q] itself -- for a fixed q the lanes of a wavefront read
// consecutive x entries (the rows are consecutive, the offsets equal), so the gathers are coalesced, which the per-nonzero
// layouts above cannot offer -- multiplies with the staged values and adds in column order (or two at a time, pairsum):
// the arithmetic and order of the other kernels, same bits.  No row markers, no per-nonzero index stream, one barrier less.
#define SPMV_PAT_CAP 512
// DOT: the block also leaves the sum of x_r y_r over its rows in dotpart[block] (square matrix; KSPSolve_CG's p'w from the pass that
// makes w = A p): lanes in row order, a fixed tree -- deterministic; mi355x_spmv_dot_finish adds the blocks' values in block order.
template <int ADD, bool DOT>
__global__ __launch_bounds__(SPMV_THREADS, SPMV_MINWAVES) void spmv_csr_rowblock_pat_kernel(
    const int2 *__restrict__ rowblk, int nblocks, const unsigned int *__restrict__ prow,
    const int *__restrict__ pattab_g, const double *__restrict__ aa, const double *__restrict__ x, const double *yin, double *yout,
    double *__restrict__ dotpart, int pairsum, int ch, int nty) {
  __shared__ double vs[SPMV_BLOCK_NNZ];
  __shared__ int pattab[SPMV_PAT_CAP];
  __shared__ double wdot[DOT ? SPMV_THREADS / MI355X_WAVE : 1];
  static_assert(SPMV_THREADS == 256 && SPMV_BLOCK_ROWS <= SPMV_THREADS, "one lane per row of the block");
#if SPMV_REMAP == 2
  const int xcd = blockIdx.x % MI355X_NXCD;
  const int slot = blockIdx.x / MI355X_NXCD;
  const int lb = ((slot / ch) * MI355X_NXCD + xcd) * ch + (slot % ch);     // runs of ch row blocks, see mi355x_spmv_plan_compress_indices
#else
  const int lb = blockIdx.x;
#endif
  if (lb >= nblocks) return;
  const int2 b0 = rowblk[lb];
  const int2 b1 = rowblk[lb + 1];
  const int r0 = b0.x, r1 = b1.x, k0 = b0.y, k1 = b1.y;
  const int nrows = r1 - r0;
  const int tid = threadIdx.x;
  const int t0 = pattab_g[tid], t1 = pattab_g[tid + SPMV_THREADS];     // SPMV_PAT_CAP initialised entries
  if (k1 == k0) {               // only empty rows
    if (tid < nrows) yout[r0 + tid] = spmv_empty<ADD>(ADD ? yin[r0 + tid] : 0.0);
    if (DOT && tid == 0) dotpart[lb] = 0.0;
    return;
  }
  const int rc = tid < nrows ? tid : nrows - 1;
  const unsigned int pw = prow[r0 + rc];          // {where the row's list starts in the table : 16, its first nonzero in the block : 16}
  const int pst = (int)(pw & 0xffffu), rs = (int)(pw >> 16);
  double ysum = 0.0;
  if (ADD) ysum = yin[r0 + rc];
  constexpr int PAIRS = SPMV_BLOCK_NNZ / (2 * SPMV_THREADS);
  const int ka = k0 & ~1;
  v2d v[PAIRS];
#pragma unroll
  for (int p = 0; p < PAIRS; ++p) {      // every load unconditional, see the idx8 kernel
    const int k = ka + 2 * tid + p * 2 * SPMV_THREADS;
    const int kk = (k < k1) ? k : ka;
    v[p] = SPMV_LOAD(reinterpret_cast<const v2d *>(aa + kk));
  }
  pattab[tid] = t0;
  pattab[tid + SPMV_THREADS] = t1;
#pragma unroll
  for (int p = 0; p < PAIRS; ++p) {
    const int k = ka + 2 * tid + p * 2 * SPMV_THREADS;
    vs[(k >= k0 && k < k1) ? k - k0 : SPMV_BLOCK_NNZ - 1] = v[p].x;     // elements outside the block: the spare last slot
    vs[(k + 1 < k1) ? k + 1 - k0 : SPMV_BLOCK_NNZ - 1] = v[p].y;
  }
  __syncthreads();
  if (!DOT && tid >= nrows) return;
  const int len = tid < nrows ? pattab[pst] : 0;   // table entry: {length, offsets ...}
  const long xbase = (long)r0 + rc;
  double sum = (ADD == 1) ? ysum : 0.0;
  for (int q0 = 0; q0 < len; q0 += 8) {
    double xv[8], av[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int qq = (q0 + j < len) ? q0 + j : len - 1;
      xv[j] = x[xbase + pattab[pst + 1 + qq]];
      av[j] = vs[rs + qq];
    }
    if (!pairsum) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const double u = sum + av[j] * xv[j]; sum = (q0 + j < len) ? u : sum; }
    } else {
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const double pa = av[j] * xv[j], pb = av[j + 1] * xv[j + 1];
        const double inc = (q0 + j + 1 < len) ? pa + pb : pa;
        const double u = sum + inc;
        sum = (q0 + j < len) ? u : sum;
      }
    }
  }
  const double yv = spmv_fin<ADD>(ysum, sum);
  // y beyond the Infinity Cache (vectors of >= 256 MiB) is written once and read by another kernel much later: a non-temporal store
  // there (+3 % on P7(512), three of three alternations on one box: profiles/r04_spmv_p7_512_ab.log; nothing either way at P7(256))
  if (tid < nrows) { if (nty) __builtin_nontemporal_store(yv, yout + r0 + tid); else yout[r0 + tid] = yv; }
  if (DOT) {
    const double c = wave_sum(tid < nrows ? yv * x[xbase] : 0.0);
    if ((tid & (MI355X_WAVE - 1)) == 0) wdot[tid / MI355X_WAVE] = c;
    __syncthreads();
    if (tid == 0) {
      double t = wdot[0];
#pragma unroll
      for (int w = 1; w < SPMV_THREADS / MI355X_WAVE; ++w) t += wdot[w];
      dotpart[lb] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Value-pattern variant for constant-coefficient operators (value indexing in the sense of CSR-VI, taken per row).  When
// whole rows repeat -- the same offsets AND bit-for-bit the same values: the 27 row kinds of the 7-point operator on a
// box, any stencil with constant coefficients -- the analysis keeps ONE copy of each distinct row {length, offsets, values}
// in a table of at most SPMV_PAT_CAP entries and 2 bytes per ROW saying which; the value array is not read at all.  The
// kernel streams 2 B + y per row and gathers x (coalesced: consecutive lanes own consecutive rows, see the row-pattern
// kernel); products and their order are those of the other kernels, so the result carries the same bits.  The table
// describes the values it was derived from: every path that changes values on the device drops it
// (mi355x_spmv_plan_drop_value_patterns), every upload derives it again.
// What bounds it is not memory: rocprofv3 counts 0.36 GB per launch on P7(256) (ideal 0.30: x once, y once, 2 B per row; x is
// fetched 1.45 times, neighbouring planes by more than one XCD's L2) at 3.3 TB/s.  The launch is a latency chain -- row word -> table -> gathers -> store -- run by as many wavefronts as a CU holds:
// probe builds take 0.046 ms with every global access removed and 0.02-0.04 ms more for each of the three phases.  Per nonzero
// the work is therefore kept minimal (table entries padded to a multiple of 8 slots and holding BYTE offsets, a gather's
// address = the uniform base of x + one 32-bit add, no index clamped; stores after all of a lane's rows, because loads and
// stores share one in-order completion counter).  Measured and not kept, all within +-10 % of this form or slower: an LDS copy
// of the x window around the block's rows for the near offsets, all of a lane's gathers issued before any is used, two
// adjacent rows per lane with 16-byte gathers, wavefront tiles of 512 rows with one 16-byte word load per lane, a persistent
// grid walking the rows, an XCD-contiguous block map, wavefront-uniform table entries through scalar registers.
#ifndef SPMV_VPAT_RPL
#define SPMV_VPAT_RPL 4       // rows per lane (inside the CG iteration on P7(256): 1 -> 0.127 ms, 2 -> 0.116, 4 -> 0.106, 8 -> 0.107)
#endif
#define SPMV_VPAT_ROWS (SPMV_VPAT_RPL * SPMV_THREADS)      // rows per workgroup
#define SPMV_VPAT_MAXROWS (1 << 28)                        // 32-bit byte offsets into x

// one row out of the table, the other kernels' arithmetic and order; rb8 = 8 * row, xb = x as bytes
__device__ __forceinline__ double vpat_row(const int *pattab, const double *patval, int ps, unsigned int rb8, const char *__restrict__ xb,
                                           double sum, int pairsum) {
  const int len = pattab[ps];                      // table entry: {length, byte offsets ... (padded to 8 k slots)}; values at the offsets' indices
  for (int q0 = 0; q0 < len; q0 += 8) {
    double xv[8], av[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {                  // the gathers are what the launch pays for (texture addresser 76 % busy): a slot beyond
      xv[i] = 0.0;                                 // the row's length is not loaded -- and not issued at all when no lane of the wavefront needs it
      if (q0 + i < len) xv[i] = *reinterpret_cast<const double *>(xb + (unsigned int)(rb8 + (unsigned int)pattab[ps + 1 + q0 + i]));
      av[i] = patval[ps + 1 + q0 + i];
    }
    if (!pairsum) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { const double u = sum + av[i] * xv[i]; sum = (q0 + i < len) ? u : sum; }
    } else {
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        const double pa = av[i] * xv[i], pb = av[i + 1] * xv[i + 1];
        const double inc = (q0 + i + 1 < len) ? pa + pb : pa;
        const double u = sum + inc;
        sum = (q0 + i < len) ? u : sum;
      }
    }
  }
  return sum;
}

template <int ADD, bool DOT>
__global__ __launch_bounds__(SPMV_THREADS) void spmv_csr_valpat_kernel(
    int nrows, const unsigned short *__restrict__ vrow, const int *__restrict__ pattab_g, const double *__restrict__ patval_g, int tablen,
    const double *__restrict__ x, const double *yin, double *yout, double *__restrict__ dotpart, int pairsum) {
  __shared__ int pattab[SPMV_PAT_CAP];
  __shared__ double patval[SPMV_PAT_CAP];
  __shared__ double wdot[DOT ? SPMV_THREADS / MI355X_WAVE : 1];
  const int tid = threadIdx.x;
  for (int t = tid; t < tablen; t += SPMV_THREADS) { pattab[t] = pattab_g[t]; patval[t] = patval_g[t]; }
  const long rbase = (long)blockIdx.x * SPMV_VPAT_ROWS + tid;          // the grid covers the rows exactly: no block without one
  int pst[SPMV_VPAT_RPL];
  double yv[SPMV_VPAT_RPL];
#pragma unroll
  for (int j = 0; j < SPMV_VPAT_RPL; ++j) {
    const long row = rbase + (long)j * SPMV_THREADS;
    const long rc = row < nrows ? row : (long)nrows - 1;
    pst[j] = vrow[rc];
    yv[j] = ADD == 2 ? __builtin_nontemporal_load(yin + rc) : (ADD ? yin[rc] : 0.0);   // ADD == 2: the Jacobi diagonal, a stream read once per product
  }
  __syncthreads();
  const char *xb = reinterpret_cast<const char *>(x);
  // the stores wait until every row of the lane has its sum: loads and stores share one in-order completion counter on this
  // ISA, so a store issued between two rows' gathers would put its acknowledgement (an HBM round trip) on the second row's path
  double res[SPMV_VPAT_RPL];
#pragma unroll
  for (int j = 0; j < SPMV_VPAT_RPL; ++j) {
    const long row = rbase + (long)j * SPMV_THREADS;
    res[j] = 0.0;
    if (row < nrows) res[j] = spmv_fin<ADD>(yv[j], vpat_row(pattab, patval, pst[j], (unsigned int)row * 8u, xb, (ADD == 1) ? yv[j] : 0.0, pairsum));
  }
#pragma unroll
  for (int j = 0; j < SPMV_VPAT_RPL; ++j) {
    const long row = rbase + (long)j * SPMV_THREADS;
    if (row < nrows) yout[row] = res[j];
  }
  if (DOT) {   // x_r y_r over the workgroup's rows: a lane's rows in order, then the fixed tree of the other kernels
    double c = 0.0;
#pragma unroll
    for (int j = 0; j < SPMV_VPAT_RPL; ++j) {
      const long row = rbase + (long)j * SPMV_THREADS;
      if (row < nrows) c += res[j] * x[row];
    }
    c = wave_sum(c);
    if ((tid & (MI355X_WAVE - 1)) == 0) wdot[tid / MI355X_WAVE] = c;
    __syncthreads();
    if (tid == 0) {
      double t = wdot[0];
#pragma unroll
      for (int w = 1; w < SPMV_THREADS / MI355X_WAVE; ++w) t += wdot[w];
      dotpart[blockIdx.x] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Grouped-row variant: the MI355X form of the reference's inodes (Mat_CheckInode inode.c:3964-4034,
// MatMult_SeqAIJ_Inode inode.c:392-578).  Consecutive rows with one and the same column pattern -- the dof rows of
// one node of a finite-element matrix -- form a group whose column list is stored ONCE (gj); the value array is the
// untouched CSR `a`.  A 3-dof matrix streams 8 + 4/3 bytes per nonzero instead of 12.  Work layout as above: row
// blocks of whole groups (<= 256 rows, <= 2046 nonzeros, <= SPMV_GJ_CAP shared column indices), values streamed with
// coalesced 16-byte loads, the block's column lists staged in LDS by one coalesced load, and the column of nonzero k
// looked up as gjs[rbase[row(k)] + k] with row(k) from the LDS marker array the row-owning lanes write (a per-nonzero
// 16-bit marker that saves one LDS level was measured 5 % slower: 0.252 against 0.239 ms on the FEM stand-in).  Row sums as in
// the other kernels (pairsum: the inode routine's two-at-a-time order, so the result carries the reference's bits).
template <int ADD>
__global__ __launch_bounds__(SPMV_THREADS, SPMV_MINWAVES) void spmv_csr_rowblock_inode_kernel(
    const int4 *__restrict__ rowblk, int nblocks, const int *__restrict__ ai, const int *__restrict__ goff,
    const int *__restrict__ gj, const double *__restrict__ aa, const double *__restrict__ x, const double *yin, double *yout,
    int pairsum) {
  __shared__ double prod[SPMV_BLOCK_NNZ];
  __shared__ int gjs[SPMV_GJ_CAP + 2];               // +2: parking slots for the halves of a pair outside the block
  __shared__ int rbase[SPMV_BLOCK_ROWS];             // per row: its nonzero e of the block has column gjs[rbase + e]
  __shared__ unsigned char rowof[SPMV_BLOCK_NNZ];    // per nonzero of the block: its row (written by the lanes that own the row)
  static_assert(SPMV_THREADS == 256, "two index pairs per lane cover SPMV_GJ_CAP entries for 256 lanes");
  static_assert(SPMV_GJ_CAP + 2 <= 4 * SPMV_THREADS, "index staging: two int2 loads per lane");
#if SPMV_REMAP == 2
  const int xcd = blockIdx.x % MI355X_NXCD;
  const int slot = blockIdx.x / MI355X_NXCD;
  const int lb = ((slot / SPMV_CH) * MI355X_NXCD + xcd) * SPMV_CH + (slot % SPMV_CH);
#else
  const int lb = blockIdx.x;
#endif
  if (lb >= nblocks) return;
  const int4 b0 = rowblk[lb];
  const int4 b1 = rowblk[lb + 1];
  const int r0 = b0.x, r1 = b1.x, k0 = b0.y, k1 = b1.y, g0 = b0.z, g1 = b1.z;
  const int nnz = k1 - k0;
  const int nrows = r1 - r0;
  const int tid = threadIdx.x;
  if (nnz == 0) {               // only empty rows
    if (tid < nrows) yout[r0 + tid] = spmv_empty<ADD>(ADD ? yin[r0 + tid] : 0.0);
    return;
  }
  int tpr = 1;
  while (tpr < MI355X_WAVE && nrows * (tpr * 2) <= SPMV_THREADS) tpr *= 2;
  if (nnz <= SPMV_SEQ_AVG * nrows) tpr = 1;
  const int r = tid / tpr, sub = tid & (tpr - 1);
  const int rc = r < nrows ? r : nrows - 1;
  const int a0 = ai[r0 + rc], a1 = ai[r0 + rc + 1];
  const int go = goff[r0 + rc];
  double ysum = 0.0;
  if (ADD) ysum = yin[r0 + rc];
  // this lane's slices of the value stream and of the block's shared column lists; every load unconditional
  constexpr int PAIRS = SPMV_BLOCK_NNZ / (2 * SPMV_THREADS);
  const int ka = k0 & ~1, ga = g0 & ~1;
  v2d v[PAIRS];
  v2i gv[2];
#pragma unroll
  for (int p = 0; p < PAIRS; ++p) {
    const int k = ka + 2 * tid + p * 2 * SPMV_THREADS;
    const int kk = (k < k1) ? k : ka;
    v[p] = SPMV_LOAD(reinterpret_cast<const v2d *>(aa + kk));
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int g = ga + 2 * tid + q * 2 * SPMV_THREADS;
    const int gg = (g < g1) ? g : ga;
    gv[q] = SPMV_LOAD(reinterpret_cast<const v2i *>(gj + gg));
  }
  const int rs = r < nrows ? a0 - k0 : 0, re = r < nrows ? a1 - k0 : 0;
  for (int k = rs + sub; k < re; k += tpr) rowof[k] = (unsigned char)r;
  if (r < nrows && sub == 0) rbase[r] = (go - g0) - rs;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int g = ga + 2 * tid + q * 2 * SPMV_THREADS;
    gjs[(g >= g0 && g < g1) ? g - g0 : SPMV_GJ_CAP] = gv[q].x;
    gjs[(g + 1 < g1) ? g + 1 - g0 : SPMV_GJ_CAP + 1] = gv[q].y;
  }
  __syncthreads();
  double xa[PAIRS], xb[PAIRS];
#pragma unroll
  for (int p = 0; p < PAIRS; ++p) {
    const int k = ka + 2 * tid + p * 2 * SPMV_THREADS;
    const bool in = k < k1;
    const bool v0 = in && k >= k0, v1 = in && (k + 1 < k1);
    const int kk = in ? k : ka;
    const int e0 = kk + (v0 ? 0 : (in ? 1 : (k0 & 1))) - k0;   // an element of this block to take the column from
    const int e1 = kk + (v1 ? 1 : (in ? 0 : (k0 & 1))) - k0;
    xa[p] = x[gjs[rbase[rowof[e0]] + e0]];
    xb[p] = x[gjs[rbase[rowof[e1]] + e1]];
  }
#pragma unroll
  for (int p = 0; p < PAIRS; ++p) {
    const int k = ka + 2 * tid + p * 2 * SPMV_THREADS;
    const double pa = v[p].x * xa[p], pb = v[p].y * xb[p];
    prod[(k >= k0 && k < k1) ? k - k0 : SPMV_BLOCK_NNZ - 1] = pa;
    prod[(k + 1 < k1) ? k + 1 - k0 : SPMV_BLOCK_NNZ - 1] = pb;
  }
  __syncthreads();
  if (tpr == 1) {
    if (r < nrows) yout[r0 + r] = spmv_fin<ADD>(ysum, row_sum_lds(prod, rs, re, ADD == 1 ? ysum : 0.0, pairsum));
  } else {
    double sum = 0.0;
    if (r < nrows) for (int k = rs + sub; k < re; k += tpr) sum += prod[k];
    for (int off = tpr >> 1; off > 0; off >>= 1) sum += __shfl_down(sum, off, MI355X_WAVE);
    if (r < nrows && sub == 0) yout[r0 + r] = spmv_out<ADD>(ysum, sum);
  }
}

// ---------------------------------------------------------------------------------------------
// BCSR (MatMult_SeqBAIJ_3/_4/_N, reference src/mat/impls/baij/seq/baij2.c:331-436,981) with the same
// row-block streaming structure: the plan is built over the block-row pointer scaled by bs*bs (so it
// counts values), a workgroup streams <= 2046 values of consecutive block rows with 16-byte loads,
// looks the block column up once per value (e / bs^2), multiplies by x[col*bs + c] and parks the
// products in LDS; point row (br, r) then owns the LDS entries s + bs*j + r, j < nblocks*bs
// (blocks are column-major, baij.h:13-30) and is summed by 1..64 lanes + a shuffle tree.
// XLDS: the x entries of the row block's block columns are staged in LDS ONCE per block (bs doubles per stored block, requested
// together with the value stream, before the barrier) and the products read them from there: bs^2 values share bs staged entries,
// so the global gathers drop from one per value to one per bs values, and none of them sits behind the barrier.
#ifndef MI355X_BSR_XLDS_DEFAULT
#define MI355X_BSR_XLDS_DEFAULT 1   // measured at 128^3 nodes (profiles/r03_cfg5.log): bs = 3 0.781 -> 0.728 ms, bs = 4 1.420 -> 1.290 ms; same bits
#endif
template <int BS, bool XLDS>
__global__ __launch_bounds__(SPMV_THREADS) void bsr_rowblock_kernel(const int2 *__restrict__ rowblk, int nblocks,
                                                                   const int *__restrict__ ai, const int *__restrict__ aj,
                                                                   const double *__restrict__ aa,
                                                                   const double *__restrict__ x, const double *yin, double *y) {
  // yin != NULL: y = yin + A x (MatMultAdd_SeqBAIJ_N, baij2.c:1168-1480; y may alias yin: every point row is read and written by one lane)
  __shared__ double prod[SPMV_BLOCK_NNZ];
  __shared__ int ajs[XLDS ? 1 : SPMV_BLOCK_NNZ / 4 + 1];   // block columns of the row block (bs >= 2: at most NNZ/4 blocks)
  __shared__ double xs[XLDS ? (SPMV_BLOCK_NNZ / (BS * BS) + 1) * BS : 1];   // XLDS: x[bs * col .. + bs) of every stored block
  constexpr int BS2 = BS * BS;
  // interleaved block -> XCD map of the CSR kernels: each XCD walks runs of consecutive row blocks, so a block column's x
  // entries are pulled into ONE XCD's L2 instead of all eight (PMC at 128^3 nodes: 4.75 GB fetched for 4.35 GB without it)
#if SPMV_REMAP == 2
  const int xcd = blockIdx.x % MI355X_NXCD;
  const int slot = blockIdx.x / MI355X_NXCD;
  const int lb = ((slot / SPMV_CH) * MI355X_NXCD + xcd) * SPMV_CH + (slot % SPMV_CH);
#else
  const int lb = blockIdx.x;
#endif
  if (lb >= nblocks) return;
  const int2 b0 = rowblk[lb];
  const int2 b1 = rowblk[lb + 1];
  const int r0 = b0.x, r1 = b1.x, k0 = b0.y, k1 = b1.y;   // block rows [r0,r1), values [k0,k1)
  const int tid = threadIdx.x;
  const int nv = (r1 - r0) * BS;                           // point rows of this workgroup
  if (k1 - k0 > SPMV_BLOCK_CAP) {
    // one block row wider than the LDS stage: strided partial sums per point row, tree at the end
    double acc[BS];
#pragma unroll
    for (int r = 0; r < BS; ++r) acc[r] = 0.0;
    for (int e = k0 + tid; e < k1; e += SPMV_THREADS) {
      const int blk = e / BS2, q = e - blk * BS2, c = q / BS, r = q - c * BS;
      const double p = SPMV_LOAD(aa + e) * x[(long)aj[blk] * BS + c];
#pragma unroll
      for (int rr = 0; rr < BS; ++rr) acc[rr] += (rr == r) ? p : 0.0;
    }
    __shared__ double part[SPMV_THREADS / MI355X_WAVE][BS];
#pragma unroll
    for (int r = 0; r < BS; ++r) { double v = wave_sum(acc[r]); if ((tid & 63) == 0) part[tid / 64][r] = v; }
    __syncthreads();
    if (tid < BS) { double t = part[0][tid]; for (int w = 1; w < SPMV_THREADS / MI355X_WAVE; ++w) t += part[w][tid]; y[(long)r0 * BS + tid] = yin ? yin[(long)r0 * BS + tid] + t : t; }
    return;
  }
  if (k1 == k0) {   // only empty block rows
    if (tid < nv) y[(long)r0 * BS + tid] = yin ? yin[(long)r0 * BS + tid] : 0.0;
    return;
  }
  int tpr = 1;
  while (tpr < MI355X_WAVE && nv * (tpr * 2) <= SPMV_THREADS) tpr *= 2;
  // extents of this lane's point row, requested before the stream; all loads unconditional (see the idx8 kernel)
  const int v = tid / tpr, sub = tid & (tpr - 1);
  const int vc = v < nv ? v : nv - 1;
  const int br = r0 + vc / BS;
  const int rr_ = vc - (vc / BS) * BS;
  const int a0 = ai[br], a1 = ai[br + 1];
  // block columns of this row block: one coalesced load into LDS instead of a global gather per value (a block's bs^2
  // values share one entry); k0 is a multiple of bs^2 because row blocks start at block-row boundaries
  const int kb0 = k0 / BS2, nblk = (k1 - k0) / BS2;
  constexpr int PAIRS = SPMV_BLOCK_NNZ / (2 * SPMV_THREADS);
  const int ka = k0 & ~1;
  v2d vv[PAIRS];
  if (XLDS) {
    // the value stream first (it is the long pole), then one block column per lane and its bs x entries
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
      const int k = ka + 2 * tid + p * 2 * SPMV_THREADS;
      const int kk = (k < k1) ? k : ka;
      vv[p] = SPMV_LOAD(reinterpret_cast<const v2d *>(aa + kk));
    }
    for (int b = tid; b < nblk; b += SPMV_THREADS) {
      const long c = (long)aj[kb0 + b] * BS;
#pragma unroll
      for (int q = 0; q < BS; ++q) xs[b * BS + q] = x[c + q];
    }
  } else {
    for (int b = tid; b < nblk; b += SPMV_THREADS) ajs[b] = aj[kb0 + b];
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
      const int k = ka + 2 * tid + p * 2 * SPMV_THREADS;
      const int kk = (k < k1) ? k : ka;
      vv[p] = SPMV_LOAD(reinterpret_cast<const v2d *>(aa + kk));
    }
  }
  __syncthreads();
  double xa[PAIRS], xb[PAIRS];
#pragma unroll
  for (int p = 0; p < PAIRS; ++p) {
    const int k = ka + 2 * tid + p * 2 * SPMV_THREADS;
    const bool in = k < k1;
    const bool v0 = in && k >= k0, v1 = in && (k + 1 < k1);
    const int kk = in ? k : ka;
    // value index each slot takes its block column from: its own, or a neighbour inside the block
    const int f0 = kk + (v0 ? 0 : (in ? 1 : (k0 & 1))) - k0;
    const int f1 = kk + (v1 ? 1 : (in ? 0 : (k0 & 1))) - k0;
    const int blk0 = f0 / BS2, blk1 = f1 / BS2;
    if (XLDS) {
      xa[p] = xs[blk0 * BS + (f0 - blk0 * BS2) / BS];
      xb[p] = xs[blk1 * BS + (f1 - blk1 * BS2) / BS];
    } else {
      xa[p] = x[(long)ajs[blk0] * BS + (f0 - blk0 * BS2) / BS];
      xb[p] = x[(long)ajs[blk1] * BS + (f1 - blk1 * BS2) / BS];
    }
  }
#pragma unroll
  for (int p = 0; p < PAIRS; ++p) {
    const int k = ka + 2 * tid + p * 2 * SPMV_THREADS;
    const double pa = vv[p].x * xa[p], pb = vv[p].y * xb[p];
    prod[(k >= k0 && k < k1) ? k - k0 : SPMV_BLOCK_NNZ - 1] = pa;
    prod[(k + 1 < k1) ? k + 1 - k0 : SPMV_BLOCK_NNZ - 1] = pb;
  }
  __syncthreads();
  const int s = a0 * BS2 - k0;
  const int cnt = v < nv ? (a1 - a0) * BS : 0;
  double sum = 0.0;
  for (int j = sub; j < cnt; j += tpr) sum += prod[s + BS * j + rr_];
  for (int off = tpr >> 1; off > 0; off >>= 1) sum += __shfl_down(sum, off, MI355X_WAVE);
  if (v < nv && sub == 0) y[(long)br * BS + rr_] = yin ? yin[(long)br * BS + rr_] + sum : sum;
  // block rows with very few blocks: the row block can hold more point rows than the workgroup has lanes
  // (<= 256 block rows x bs); the remaining ones are summed the same way, one lane per point row (tpr is 1 here)
  for (int v2 = tid + SPMV_THREADS; v2 < nv; v2 += SPMV_THREADS) {
    const int br2 = r0 + v2 / BS, rr2 = v2 - (v2 / BS) * BS;
    const int b0_ = ai[br2], b1_ = ai[br2 + 1];
    const int s2 = b0_ * BS2 - k0, cnt2 = (b1_ - b0_) * BS;
    double sum2 = 0.0;
    for (int j = 0; j < cnt2; ++j) sum2 += prod[s2 + BS * j + rr2];
    y[(long)br2 * BS + rr2] = yin ? yin[(long)br2 * BS + rr2] + sum2 : sum2;
  }
}

// sum of the per-row-block x'y values in block order: 1024 lanes stride over them, fixed tree
__global__ __launch_bounds__(1024) void dot_partials_kernel(const double *__restrict__ part, int n, double *out) {
  __shared__ double lds[1024 / MI355X_WAVE];
  // eight loads in flight per lane (one dependent load per step took 22 us for the 57 K values of P7(256)); a fixed order of
  // additions whatever n is: lane t owns values t, t + 1024, ..., added eight accumulators wide, the accumulators in a fixed tree
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0, a5 = 0.0, a6 = 0.0, a7 = 0.0;
  int i = threadIdx.x;
  for (; i + 7 * 1024 < n; i += 8 * 1024) {
    const double v0 = part[i], v1 = part[i + 1024], v2 = part[i + 2 * 1024], v3 = part[i + 3 * 1024];
    const double v4 = part[i + 4 * 1024], v5 = part[i + 5 * 1024], v6 = part[i + 6 * 1024], v7 = part[i + 7 * 1024];
    a0 += v0; a1 += v1; a2 += v2; a3 += v3; a4 += v4; a5 += v5; a6 += v6; a7 += v7;
  }
  for (; i < n; i += 1024) a0 += part[i];
  double s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  s = wave_sum(s);
  if ((threadIdx.x & (MI355X_WAVE - 1)) == 0) lds[threadIdx.x / MI355X_WAVE] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = lds[0];
#pragma unroll
    for (int w = 1; w < 1024 / MI355X_WAVE; ++w) t += lds[w];
    out[0] = t;
  }
}

// Value assembly through a precomputed map (MatSetValuesBatch with an unchanged pattern): nonzero `segslot[s]` receives
// the contributions v[order[k]], k in [segptr[s], segptr[s+1]), added to its current value one after the other in the
// order the reference's loop of MatSetValues(ADD_VALUES) would add them (matrix.c:1715-1718) -- one lane per nonzero,
// no atomics, same bits as the host loop.
__global__ __launch_bounds__(MI355X_BLOCK) void csr_assemble_kernel(int nseg, const int *__restrict__ segptr, const int *__restrict__ segslot,
                                                                   const int *__restrict__ order, const double *__restrict__ v, double *aa) {
  const int s = blockIdx.x * MI355X_BLOCK + threadIdx.x;
  if (s >= nseg) return;
  const int slot = segslot[s];
  double sum = aa[slot];
  for (int k = segptr[s]; k < segptr[s + 1]; ++k) sum += v[order[k]];
  aa[slot] = sum;
}

// MatDiagonalScale_SeqAIJ (aij.c:2055-2092): a[k] = (a[k] * l[row]) * r[col]; either vector may be absent.  One lane per row.
__global__ __launch_bounds__(MI355X_BLOCK) void csr_diagscale_kernel(int m, const int *__restrict__ ai, const int *__restrict__ aj,
                                                                    double *aa, const double *__restrict__ l, const double *__restrict__ r) {
  const int row = blockIdx.x * MI355X_BLOCK + threadIdx.x;
  if (row >= m) return;
  const double lv = l ? l[row] : 1.0;
  for (int k = ai[row]; k < ai[row + 1]; ++k) {
    double v = aa[k];
    if (l) v = v * lv;
    if (r) v = v * r[aj[k]];
    aa[k] = v;
  }
}

__global__ __launch_bounds__(MI355X_BLOCK) void csr_diag_kernel(int m, const int *__restrict__ ai,
                                                               const int *__restrict__ aj,
                                                               const double *__restrict__ aa, double *d) {
  int r = blockIdx.x * MI355X_BLOCK + threadIdx.x;
  if (r >= m) return;
  double v = 0.0;
  for (int k = ai[r]; k < ai[r + 1]; ++k) {
    if (aj[k] == r) { v = aa[k]; break; }
  }
  d[r] = v;
}

template <int ADD>
static int launch_spmv(mi355x_handle_t h, mi355x_spmv_plan_t p, const int *ai, const int *aj, const double *aa,
                       const double *x, const double *yin, double *yout) {
  if (p->nblocks == 0) return 0;
  const bool vec = mi355x_aligned16(aa) && ((((uintptr_t)aj) & 7u) == 0);
  const bool cprow = p->d_rows != nullptr;
  if (p->vpat_valid && p->use_vpat && !cprow) {
    const int nb = (p->nrows + SPMV_VPAT_ROWS - 1) / SPMV_VPAT_ROWS;
    hipLaunchKernelGGL((spmv_csr_valpat_kernel<ADD, false>), dim3(nb), dim3(SPMV_THREADS), 0, h->stream, p->nrows, p->d_vrow, p->d_vpattab,
                       p->d_vpatval, p->vtablen, x, yin, yout, (double *)nullptr, p->pairsum);
    MI355X_LAUNCH_CHECK();
    return 0;
  }
  if (p->d_gj && !cprow && mi355x_aligned16(aa)) {
#if SPMV_REMAP == 2
    const int perg = MI355X_NXCD * SPMV_CH;
    const int gg = ((p->nblocks + perg - 1) / perg) * perg;
#else
    const int gg = p->nblocks;
#endif
    hipLaunchKernelGGL((spmv_csr_rowblock_inode_kernel<ADD>), dim3(gg), dim3(SPMV_THREADS), 0, h->stream, p->d_rowblk4, p->nblocks,
                       ai, p->d_goff, p->d_gj, aa, x, yin, yout, p->pairsum);
    MI355X_LAUNCH_CHECK();
    return 0;
  }
  if (p->d_prow && p->use_pat && !cprow && mi355x_aligned16(aa)) {
#if SPMV_REMAP == 2
    const int perp = MI355X_NXCD * p->ch;
    const int gp = ((p->nblocks + perp - 1) / perp) * perp;
#else
    const int gp = p->nblocks;
#endif
    hipLaunchKernelGGL((spmv_csr_rowblock_pat_kernel<ADD, false>), dim3(gp), dim3(SPMV_THREADS), 0, h->stream, p->d_rowblk, p->nblocks,
                       p->d_prow, p->d_pattab, aa, x, yin, yout, (double *)nullptr, p->pairsum, p->ch, spmv_y_streams(p->nrows));
    MI355X_LAUNCH_CHECK();
    return 0;
  }
  if (p->d_idx8 && !cprow && mi355x_aligned16(aa)) {
#if SPMV_REMAP == 2
    const int per8 = MI355X_NXCD * SPMV_CH;
    const int g8 = ((p->nblocks + per8 - 1) / per8) * per8;
#else
    const int g8 = p->nblocks;
#endif
    hipLaunchKernelGGL((spmv_csr_rowblock_idx8_kernel<ADD, false>), dim3(g8), dim3(SPMV_THREADS), 0, h->stream, p->d_rowblk,
                       p->nblocks, ai, p->d_idx8, p->d_offtab, p->ntab, aa, x, yin, yout, (double *)nullptr, p->pairsum);
    MI355X_LAUNCH_CHECK();
    return 0;
  }
  #if SPMV_REMAP == 2
  const int per = MI355X_NXCD * SPMV_CH;
  dim3 grid(((p->nblocks + per - 1) / per) * per), block(SPMV_THREADS);
#else
  dim3 grid(p->chunk * MI355X_NXCD), block(SPMV_THREADS);
#endif
#define SPMV_GO(C, V)                                                                                               \
  hipLaunchKernelGGL((spmv_csr_rowblock_kernel<ADD, C, V>), grid, block, 0, h->stream, p->d_rowblk, p->nblocks,    \
                     p->chunk, ai, aj, aa, x, yin, yout, p->d_rows, p->pairsum)
  if (cprow) { if (vec) SPMV_GO(true, true); else SPMV_GO(true, false); }
  else       { if (vec) SPMV_GO(false, true); else SPMV_GO(false, false); }
#undef SPMV_GO
  MI355X_LAUNCH_CHECK();
  return 0;
}

extern "C" {

int mi355x_spmv_plan_create(mi355x_handle_t h, int nrows, const int *ai_host, const int *rows_host,
                            mi355x_spmv_plan_t *plan) {
  mi355x_spmv_plan_s *p = new mi355x_spmv_plan_s();
  p->nrows = nrows;
  p->d_rowblk = nullptr;
  p->d_rows = nullptr;
  p->d_idx8 = nullptr;
  p->d_offtab = nullptr;
  p->ntab = 0;
  p->d_prow = nullptr; p->d_pattab = nullptr; p->npat = 0; p->use_pat = 1; p->ch = SPMV_CH;
  p->d_vrow = nullptr; p->d_vpattab = nullptr; p->d_vpatval = nullptr; p->nvpat = 0; p->vtablen = 0; p->vpat_valid = 0; p->use_vpat = 1;
  p->nlong = 0;
  p->d_dotpart = nullptr;
  p->pairsum = 0;
  p->d_rowblk4 = nullptr;
  p->d_goff = nullptr;
  p->d_gj = nullptr;
  p->ngroups = 0;
  p->ngj = 0;
  std::vector<int2> rb;
  rb.reserve((size_t)nrows / 128 + 2);
  rb.push_back(make_int2(0, ai_host[0]));
  int r = 0;
  while (r < nrows) {
    const int start = r;
    int nnz = 0;
    while (r < nrows && (r - start) < SPMV_BLOCK_ROWS) {
      const int len = ai_host[r + 1] - ai_host[r];
      if (nnz + len > SPMV_BLOCK_CAP) break;
      nnz += len;
      ++r;
    }
    if (r == start) {  // a single row longer than the LDS stage
      ++r;
      p->nlong++;
    }
    rb.push_back(make_int2(r, ai_host[r]));
  }
  p->nblocks = (int)rb.size() - 1;
  p->chunk = (p->nblocks + MI355X_NXCD - 1) / MI355X_NXCD;
  MI355X_TRY(hipMalloc((void **)&p->d_rowblk, sizeof(int2) * rb.size()));
  MI355X_TRY(hipMemcpyAsync(p->d_rowblk, rb.data(), sizeof(int2) * rb.size(), hipMemcpyHostToDevice, h->stream));
  if (rows_host) {
    MI355X_TRY(hipMalloc((void **)&p->d_rows, sizeof(int) * (size_t)(nrows > 0 ? nrows : 1)));
    MI355X_TRY(hipMemcpyAsync(p->d_rows, rows_host, sizeof(int) * (size_t)nrows, hipMemcpyHostToDevice, h->stream));
  }
  MI355X_TRY(hipStreamSynchronize(h->stream));  // rb is a local
  *plan = p;
  return 0;
}

// host threads of the pattern analyses: up to `cap`, one below 400 000 rows; MI355X_ANALYSIS_THREADS=<n> in the environment
// forces a count (tests: the merge of the chunks' tables on small matrices), never more than one thread per row
static int analysis_threads(int m, int cap) {
  int nth = mi355x_host_threads(cap);
  if (m < 400000) nth = 1;
  const char *e = getenv("MI355X_ANALYSIS_THREADS");
  if (e && atoi(e) > 0) nth = atoi(e) > 64 ? 64 : atoi(e);
  if (nth > m) nth = m > 0 ? m : 1;
  return nth;
}

// Offset-dictionary analysis: idx8[k] = position of (aj[k] - row) in a table of <= 256 distinct offsets.
// Returns 0 and leaves the plan uncompressed when the matrix has more distinct offsets.
int mi355x_spmv_plan_compress_indices(mi355x_handle_t h, mi355x_spmv_plan_t p, const int *ai_host, const int *aj_host) {
  if (!p || p->d_rows || p->d_idx8 || p->nrows == 0) return 0;
  if (SPMV_BLOCK_ROWS > 256) return 0;   // row markers are bytes
  const int m = p->nrows;
  const long nnz = ai_host[m];
  // The rows are analysed in contiguous chunks by host threads (one pass over the column indices of P7(256) on one thread: 0.3 s
  // before the first product): every chunk collects its offsets / its rows' slot lists in order of first appearance, the chunks'
  // tables are merged in chunk order -- which gives the table of a single pass over all rows -- and the chunks renumber their part.
  int nth = analysis_threads(m, 16);
  auto chunk_lo = [&](int k) { return (int)((long)m * k / nth); };
  auto run_chunks = [&](auto fn) {
    if (nth == 1) { fn(0); return; }
    std::vector<std::thread> th;
    for (int k = 1; k < nth; ++k) th.emplace_back(fn, k);
    fn(0);
    for (auto &t : th) t.join();
  };
  std::unique_ptr<unsigned char[]> idx(new unsigned char[(size_t)(nnz > 0 ? nnz : 1)]);
  struct OffTab { int tab[256]; int n = 0; bool ok = true; };
  std::vector<OffTab> lt((size_t)nth);
  run_chunks([&](int k) {
    OffTab &t = lt[(size_t)k];
    int last = 0;      // (offsets of a stencil matrix repeat row after row, so the slot after the previous one is tried first)
    for (int r = chunk_lo(k); r < chunk_lo(k + 1); ++r) {
      for (int q = ai_host[r]; q < ai_host[r + 1]; ++q) {
        const int off = aj_host[q] - r;
        int slot = -1;
        if (t.n && t.tab[last] == off) slot = last;
        else for (int e = 0; e < t.n; ++e) if (t.tab[e] == off) { slot = e; break; }
        if (slot < 0) {
          if (t.n == 256) { t.ok = false; return; }   // too many distinct offsets: keep plain CSR
          t.tab[t.n] = off;
          slot = t.n++;
        }
        idx[(size_t)q] = (unsigned char)slot;
        last = (slot + 1 < t.n) ? slot + 1 : 0;
      }
    }
  });
  for (auto &t : lt) if (!t.ok) return 0;
  int tab[256];
  int ntab = 0;
  std::vector<std::array<unsigned char, 256>> remap((size_t)nth);
  std::vector<char> identity((size_t)nth, 1);
  for (int k = 0; k < nth; ++k) {
    for (int e = 0; e < lt[(size_t)k].n; ++e) {
      int g = -1;
      for (int f = 0; f < ntab; ++f) if (tab[f] == lt[(size_t)k].tab[e]) { g = f; break; }
      if (g < 0) { if (ntab == 256) return 0; tab[ntab] = lt[(size_t)k].tab[e]; g = ntab++; }
      remap[(size_t)k][(size_t)e] = (unsigned char)g;
      if (g != e) identity[(size_t)k] = 0;
    }
  }
  run_chunks([&](int k) {
    if (identity[(size_t)k]) return;
    const unsigned char *mp = remap[(size_t)k].data();
    for (long q = ai_host[chunk_lo(k)]; q < ai_host[chunk_lo(k + 1)]; ++q) idx[(size_t)q] = mp[idx[(size_t)q]];
  });
  MI355X_TRY(hipMalloc((void **)&p->d_idx8, (size_t)(nnz > 0 ? nnz : 1) + 16));
  MI355X_TRY(hipMalloc((void **)&p->d_offtab, sizeof(int) * 256));
  MI355X_TRY(hipMemsetAsync(p->d_offtab, 0, sizeof(int) * 256, h->stream));
  MI355X_TRY(hipMemcpyAsync(p->d_idx8, idx.get(), (size_t)nnz, hipMemcpyHostToDevice, h->stream));
  MI355X_TRY(hipMemcpyAsync(p->d_offtab, tab, sizeof(int) * (size_t)ntab, hipMemcpyHostToDevice, h->stream));
  p->ntab = ntab;
  // Run length of the block -> XCD map for this operator.  XCD x owns runs of `ch` consecutive row blocks, dealt round-robin, so the
  // rows a run covers come back to the SAME XCD every 8 * ch blocks.  With 8 * ch * 256 rows = the operator's largest offset (the plane
  // of a 3-D stencil) the x entries a row reads at that offset were fetched into this XCD's own L2 when it worked on the plane before:
  // P7(256) ch = 32 (the value tuned by hand in round 2), P7(512) ch = 128: 1.96 -> 1.77 ms per product there
  // (profiles/r04_spmv_map_sweep.log); any other ch leaves those gathers to the Infinity Cache.  MI355X_SPMV_CH=<n> overrides.
  {
    long maxoff = 0;
    for (int e = 0; e < ntab; ++e) { const long o = tab[e] < 0 ? -(long)tab[e] : (long)tab[e]; if (o > maxoff) maxoff = o; }
    long ch = (maxoff + (long)MI355X_NXCD * SPMV_BLOCK_ROWS / 2) / ((long)MI355X_NXCD * SPMV_BLOCK_ROWS);
    if (ch < 8) ch = 8;
    if (ch > 1024) ch = 1024;
    const char *e_ = getenv("MI355X_SPMV_CH");
    if (e_ && atoi(e_) > 0) ch = atoi(e_);
    p->ch = (int)ch;
  }
  // row patterns: the rows' slot lists as a dictionary of at most SPMV_PAT_CAP table entries in all (stencil operators: a
  // handful of lists); rows too long for the row block's one-lane-per-row sums (> SPMV_BLOCK_CAP never happens here) or a
  // table that would not fit leave the plan at the per-nonzero bytes
  {
    // first nonzero of every row block (the rows carry their offset from it in 16 bits: a block holds <= 2046 nonzeros)
    std::vector<int2> blk((size_t)p->nblocks + 1);
    MI355X_TRY(hipMemcpyAsync(blk.data(), p->d_rowblk, sizeof(int2) * ((size_t)p->nblocks + 1), hipMemcpyDeviceToHost, h->stream));
    MI355X_TRY(hipStreamSynchronize(h->stream));          // (also: idx and tab have left the host)
    std::unique_ptr<unsigned int[]> prow(new unsigned int[(size_t)m]);
    struct PatDict { std::vector<int> ptab; std::vector<int> starts; std::vector<std::vector<unsigned char>> keys; bool ok = true; };
    std::vector<PatDict> pd((size_t)nth);
    run_chunks([&](int k) {
      PatDict &d = pd[(size_t)k];
      std::map<std::vector<unsigned char>, int> dict;
      std::vector<unsigned char> cur, prev;
      int prev_start = -1;
      const int r0 = chunk_lo(k), r1 = chunk_lo(k + 1);
      int b = 0;
      { int lo = 0, hi = p->nblocks;                       // the block that holds row r0: the last one starting at or before it
        while (lo < hi) { const int mid = (lo + hi + 1) / 2; if (blk[(size_t)mid].x <= r0) lo = mid; else hi = mid - 1; }
        b = lo; }
      for (int r = r0; r < r1; ++r) {
        while (b + 1 <= p->nblocks && blk[(size_t)b + 1].x <= r) ++b;
        const int len = ai_host[r + 1] - ai_host[r];
        const int rs = ai_host[r] - blk[(size_t)b].y;
        if (len > SPMV_BLOCK_CAP || rs < 0 || rs > 0xffff) { d.ok = false; return; }
        cur.assign(idx.get() + ai_host[r], idx.get() + ai_host[r + 1]);
        int start;
        if (prev_start >= 0 && cur == prev) start = prev_start;
        else {
          auto it = dict.find(cur);
          if (it != dict.end()) start = it->second;
          else {
            start = (int)d.ptab.size();
            if (start + 1 + len > SPMV_PAT_CAP) { d.ok = false; return; }
            d.ptab.push_back(len);                                          // table entry: {length, offsets ...}
            for (int q = 0; q < len; ++q) d.ptab.push_back(tab[cur[(size_t)q]]);
            dict.emplace(cur, start);
            d.starts.push_back(start); d.keys.push_back(cur);
          }
          prev = cur; prev_start = start;
        }
        prow[(size_t)r] = (unsigned int)start | ((unsigned int)rs << 16);
      }
    });
    bool ok = true;
    for (auto &d : pd) ok = ok && d.ok;
    std::vector<int> ptab;
    std::map<std::vector<unsigned char>, int> gdict;
    std::vector<std::vector<int>> to((size_t)nth);
    std::vector<char> same((size_t)nth, 1);
    for (int k = 0; k < nth && ok; ++k) {
      const PatDict &d = pd[(size_t)k];
      to[(size_t)k].assign((size_t)SPMV_PAT_CAP, -1);
      for (size_t e = 0; e < d.starts.size() && ok; ++e) {
        const int ls = d.starts[e], len = d.ptab[(size_t)ls];
        int g;
        auto it = gdict.find(d.keys[e]);
        if (it != gdict.end()) g = it->second;
        else {
          g = (int)ptab.size();
          if (g + 1 + len > SPMV_PAT_CAP) { ok = false; break; }
          ptab.insert(ptab.end(), d.ptab.begin() + ls, d.ptab.begin() + ls + 1 + len);
          gdict.emplace(d.keys[e], g);
        }
        to[(size_t)k][(size_t)ls] = g;
        if (g != ls) same[(size_t)k] = 0;
      }
    }
    if (ok) {
      run_chunks([&](int k) {
        if (same[(size_t)k]) return;
        const int *t = to[(size_t)k].data();
        for (int r = chunk_lo(k); r < chunk_lo(k + 1); ++r) prow[(size_t)r] = (prow[(size_t)r] & 0xffff0000u) | (unsigned int)t[prow[(size_t)r] & 0xffffu];
      });
      ptab.resize(SPMV_PAT_CAP, 0);
      MI355X_TRY(hipMalloc((void **)&p->d_prow, sizeof(unsigned int) * (size_t)m + 16));
      MI355X_TRY(hipMalloc((void **)&p->d_pattab, sizeof(int) * SPMV_PAT_CAP));
      MI355X_TRY(hipMemcpyAsync(p->d_prow, prow.get(), sizeof(unsigned int) * (size_t)m, hipMemcpyHostToDevice, h->stream));
      MI355X_TRY(hipMemcpyAsync(p->d_pattab, ptab.data(), sizeof(int) * SPMV_PAT_CAP, hipMemcpyHostToDevice, h->stream));
      MI355X_TRY(hipStreamSynchronize(h->stream));
      p->npat = (int)gdict.size();
    }
  }
  return 0;
}

// A/B switch for the row-pattern kernel (on by default when the analysis found a dictionary); *npat: its size, 0 if none
int mi355x_spmv_plan_use_patterns(mi355x_spmv_plan_t p, int on, int *npat) {
  if (!p) return (int)hipErrorInvalidValue;
  if (on >= 0) p->use_pat = on ? 1 : 0;
  if (npat) *npat = p->d_prow ? p->npat : 0;
  return 0;
}

// Value-pattern analysis (see spmv_csr_valpat_kernel): one table entry per distinct row {length, offsets, values},
// values compared bit for bit.  Gives up as soon as the table would exceed SPMV_PAT_CAP entries -- after a few dozen
// rows for a matrix with varying coefficients -- and then leaves the plan as it was.  To be called with the values that
// are (about to be) on the device, after every change of them.  *nvpat: distinct rows found, 0 when not applicable.
int mi355x_spmv_plan_value_patterns(mi355x_handle_t h, mi355x_spmv_plan_t p, const int *ai_host, const int *aj_host,
                                    const double *aa_host, int *nvpat) {
  if (nvpat) *nvpat = 0;
  if (!p) return (int)hipErrorInvalidValue;
  p->vpat_valid = 0;
  if (p->d_rows || p->nrows == 0 || !p->use_vpat) return 0;
  const int m = p->nrows;
  if (m > SPMV_VPAT_MAXROWS) return 0;              // byte offsets into x are 32-bit
  std::vector<unsigned short> vrow((size_t)m);
  // table entry: {length, byte offsets of the columns relative to the row ..., padded to a multiple of 8 slots with copies of the
  // first offset}; values at the offsets' indices (pads 0.0, never used in a sum).  Entries in order of first appearance.
  // The rows are analysed in contiguous chunks by a few host threads, each with a dictionary of its own (a pass over 0.94 GB of
  // values on one thread cost 0.2 s at every upload of P7(256)); the chunks' dictionaries are then merged in chunk order -- the
  // merged table is the one a single pass over all rows builds -- and the rows' entries renumbered.
  struct Dict { std::vector<int> ptab; std::vector<double> pval; std::vector<int> starts; bool ok = true; };
  auto analyse = [&](int r0, int r1, Dict &d) {
    d.ptab.reserve(SPMV_PAT_CAP); d.pval.reserve(SPMV_PAT_CAP);
    auto same = [&](int s_, int r, int len) {
      if (d.ptab[(size_t)s_] != len) return false;
      const int k0 = ai_host[r];
      for (int q = 0; q < len; ++q) if ((aj_host[k0 + q] - r) * 8 != d.ptab[(size_t)s_ + 1 + q]) return false;
      return len == 0 || memcmp(aa_host + k0, d.pval.data() + s_ + 1, sizeof(double) * (size_t)len) == 0;
    };
    int prev = -1;
    for (int r = r0; r < r1; ++r) {
      const int len = ai_host[r + 1] - ai_host[r];
      int start = -1;
      if (prev >= 0 && same(prev, r, len)) start = prev;
      else for (size_t e = 0; e < d.starts.size(); ++e) if (same(d.starts[e], r, len)) { start = d.starts[e]; break; }
      if (start < 0) {
        const int slots = (len + 7) / 8 * 8;
        start = (int)d.ptab.size();
        if (start + 1 + slots > SPMV_PAT_CAP) { d.ok = false; return; }      // not a constant-coefficient operator
        d.ptab.push_back(len); d.pval.push_back(0.0);
        for (int q = 0; q < slots; ++q) {
          const int col = aj_host[ai_host[r] + (q < len ? q : 0)];
          if (col >= SPMV_VPAT_MAXROWS) { d.ok = false; return; }
          d.ptab.push_back((col - r) * 8);
          d.pval.push_back(q < len ? aa_host[ai_host[r] + q] : 0.0);
        }
        d.starts.push_back(start);
      }
      prev = start;
      vrow[(size_t)r] = (unsigned short)start;
    }
  };
  const int nth = analysis_threads(m, 8);
  std::vector<Dict> dicts((size_t)nth);
  if (nth == 1) analyse(0, m, dicts[0]);
  else {
    std::vector<std::thread> th;
    for (int k = 0; k < nth; ++k) th.emplace_back(analyse, (int)((long)m * k / nth), (int)((long)m * (k + 1) / nth), std::ref(dicts[(size_t)k]));
    for (auto &t : th) t.join();
  }
  for (auto &d : dicts) if (!d.ok) return 0;
  std::vector<int> ptab(std::move(dicts[0].ptab)), starts(std::move(dicts[0].starts));
  std::vector<double> pval(std::move(dicts[0].pval));
  for (int k = 1; k < nth; ++k) {
    const Dict &d = dicts[(size_t)k];
    std::vector<int> to((size_t)SPMV_PAT_CAP, -1);             // this chunk's entry (by its start) -> the merged table's
    for (size_t e = 0; e < d.starts.size(); ++e) {
      const int ls = d.starts[e], len = d.ptab[(size_t)ls], slots = (len + 7) / 8 * 8;
      int g = -1;
      for (size_t f = 0; f < starts.size() && g < 0; ++f) {
        const int gs = starts[f];
        if (ptab[(size_t)gs] == len && memcmp(&ptab[(size_t)gs + 1], &d.ptab[(size_t)ls + 1], sizeof(int) * (size_t)len) == 0 &&
            (len == 0 || memcmp(&pval[(size_t)gs + 1], &d.pval[(size_t)ls + 1], sizeof(double) * (size_t)len) == 0)) g = gs;
      }
      if (g < 0) {
        g = (int)ptab.size();
        if (g + 1 + slots > SPMV_PAT_CAP) return 0;
        ptab.insert(ptab.end(), d.ptab.begin() + ls, d.ptab.begin() + ls + 1 + slots);
        pval.insert(pval.end(), d.pval.begin() + ls, d.pval.begin() + ls + 1 + slots);
        starts.push_back(g);
      }
      to[(size_t)ls] = g;
    }
    const int r0 = (int)((long)m * k / nth), r1 = (int)((long)m * (k + 1) / nth);
    for (int r = r0; r < r1; ++r) vrow[(size_t)r] = (unsigned short)to[(size_t)vrow[(size_t)r]];
  }
  if (!p->d_vrow) {
    MI355X_TRY(hipMalloc((void **)&p->d_vrow, sizeof(unsigned short) * (size_t)m + 16));
    MI355X_TRY(hipMalloc((void **)&p->d_vpattab, sizeof(int) * SPMV_PAT_CAP));
    MI355X_TRY(hipMalloc((void **)&p->d_vpatval, sizeof(double) * SPMV_PAT_CAP));
  }
  MI355X_TRY(hipMemcpyAsync(p->d_vrow, vrow.data(), sizeof(unsigned short) * (size_t)m, hipMemcpyHostToDevice, h->stream));
  MI355X_TRY(hipMemcpyAsync(p->d_vpattab, ptab.data(), sizeof(int) * ptab.size(), hipMemcpyHostToDevice, h->stream));
  MI355X_TRY(hipMemcpyAsync(p->d_vpatval, pval.data(), sizeof(double) * pval.size(), hipMemcpyHostToDevice, h->stream));
  MI355X_TRY(hipStreamSynchronize(h->stream));   // the vectors are locals
  p->vtablen = (int)ptab.size();
  p->nvpat = (int)starts.size();
  p->vpat_valid = 1;
  if (nvpat) *nvpat = p->nvpat;
  return 0;
}

// the values on the device no longer are the ones the table was derived from
int mi355x_spmv_plan_drop_value_patterns(mi355x_spmv_plan_t p) {
  if (p) p->vpat_valid = 0;
  return 0;
}

// A/B switch (on by default); on < 0 only queries.  *nvpat: size of the dictionary in use, 0 if none
int mi355x_spmv_plan_use_value_patterns(mi355x_spmv_plan_t p, int on, int *nvpat) {
  if (!p) return (int)hipErrorInvalidValue;
  if (on >= 0) { p->use_vpat = on ? 1 : 0; if (!on) p->vpat_valid = 0; }
  if (nvpat) *nvpat = p->vpat_valid ? p->nvpat : 0;
  return 0;
}

// Row grouping (the analysis half of the reference's inode machinery).  The caller passes the node sizes the
// reference's Mat_CheckInode finds (ns[nnodes], consecutive rows with identical column lists, at most `limit` rows
// each; inode.c:3981-3998) -- the host library computes them with the reference's loop so that they can be compared
// with it.  This routine stores one column list per group and rebuilds the row blocks from whole groups.  It leaves
// the plan as it is (returns 0) when grouping would not pay (shared indices > 2/3 of the nonzeros), when a row has
// more than SPMV_GJ_CAP entries, or for compressed-row / index-compressed plans.
int mi355x_spmv_plan_group_rows(mi355x_handle_t h, mi355x_spmv_plan_t p, const int *ai_host, const int *aj_host, int nnodes,
                                const int *ns) {
  if (!p || p->d_rows || p->d_idx8 || p->d_gj || p->nrows == 0 || nnodes <= 0) return 0;
  const int m = p->nrows;
  const long nnz = ai_host[m];
  // groups: a node whose rows together exceed the LDS stage is cut into smaller groups
  std::vector<int> gstart;          // first row of each group (+ m at the end)
  gstart.reserve((size_t)nnodes + 1);
  long ngj = 0;
  int row = 0;
  for (int g = 0; g < nnodes; ++g) {
    const int nc = ai_host[row + 1] - ai_host[row];
    if (nc > SPMV_GJ_CAP) return 0;
    int left = ns[g];
    if (left < 1 || row + left > m) return (int)hipErrorInvalidValue;
    int per = left;
    while ((long)per * nc > SPMV_BLOCK_CAP) --per;          // nc <= SPMV_GJ_CAP: per >= 2
    while (left > 0) {
      const int take = left < per ? left : per;
      gstart.push_back(row);
      ngj += nc;
      row += take;
      left -= take;
    }
  }
  if (row != m) return (int)hipErrorInvalidValue;
  gstart.push_back(m);
  if (3 * ngj > 2 * nnz) return 0;                            // not enough shared structure to pay for the lookups
  const int ngroups = (int)gstart.size() - 1;
  std::vector<int> goff((size_t)m), gjh((size_t)(ngj > 0 ? ngj : 1));
  std::vector<int4> rb;
  rb.reserve((size_t)m / 64 + 2);
  rb.push_back(make_int4(0, ai_host[0], 0, 0));
  long gpos = 0;
  int brows = 0, bnnz = 0, bgj = 0;
  for (int g = 0; g < ngroups; ++g) {
    const int ra = gstart[g], rbn = gstart[g + 1];
    const int nc = ai_host[ra + 1] - ai_host[ra];
    const int gn = (rbn - ra) * nc;
    if (brows + (rbn - ra) > SPMV_BLOCK_ROWS || bnnz + gn > SPMV_BLOCK_CAP || bgj + nc > SPMV_GJ_CAP) {
      rb.push_back(make_int4(ra, ai_host[ra], (int)gpos, 0));
      brows = bnnz = bgj = 0;
    }
    for (int r = ra; r < rbn; ++r) goff[(size_t)r] = (int)gpos;
    for (int c = 0; c < nc; ++c) gjh[(size_t)(gpos + c)] = aj_host[ai_host[ra] + c];
    gpos += nc;
    brows += rbn - ra; bnnz += gn; bgj += nc;
  }
  rb.push_back(make_int4(m, ai_host[m], (int)gpos, 0));
  MI355X_TRY(hipMalloc((void **)&p->d_rowblk4, sizeof(int4) * rb.size()));
  MI355X_TRY(hipMalloc((void **)&p->d_goff, sizeof(int) * (size_t)m));
  MI355X_TRY(hipMalloc((void **)&p->d_gj, sizeof(int) * (size_t)(ngj > 0 ? ngj : 1) + 16));   // +16: the last pair load may straddle the end
  MI355X_TRY(hipMemcpyAsync(p->d_rowblk4, rb.data(), sizeof(int4) * rb.size(), hipMemcpyHostToDevice, h->stream));
  MI355X_TRY(hipMemcpyAsync(p->d_goff, goff.data(), sizeof(int) * (size_t)m, hipMemcpyHostToDevice, h->stream));
  MI355X_TRY(hipMemcpyAsync(p->d_gj, gjh.data(), sizeof(int) * (size_t)ngj, hipMemcpyHostToDevice, h->stream));
  MI355X_TRY(hipStreamSynchronize(h->stream));
  p->nblocks = (int)rb.size() - 1;
  p->chunk = (p->nblocks + MI355X_NXCD - 1) / MI355X_NXCD;
  p->nlong = 0;
  p->ngroups = ngroups;
  p->ngj = ngj;
  return 0;
}

int mi355x_spmv_plan_set_pairsum(mi355x_spmv_plan_t p, int on) {
  if (p) p->pairsum = on ? 1 : 0;
  return 0;
}

int mi355x_spmv_plan_group_info(mi355x_spmv_plan_t p, int *ngroups, long *nshared_indices, int *pairsum) {
  if (ngroups) *ngroups = p->d_gj ? p->ngroups : 0;
  if (nshared_indices) *nshared_indices = p->d_gj ? p->ngj : 0;
  if (pairsum) *pairsum = p->pairsum;
  return 0;
}

int mi355x_spmv_plan_destroy(mi355x_spmv_plan_t p) {
  if (!p) return 0;
  hipFree(p->d_rowblk);
  if (p->d_idx8) hipFree(p->d_idx8);
  if (p->d_offtab) hipFree(p->d_offtab);
  if (p->d_prow) hipFree(p->d_prow);
  if (p->d_pattab) hipFree(p->d_pattab);
  if (p->d_vrow) hipFree(p->d_vrow);
  if (p->d_vpattab) hipFree(p->d_vpattab);
  if (p->d_vpatval) hipFree(p->d_vpatval);
  if (p->d_rows) hipFree(p->d_rows);
  if (p->d_dotpart) hipFree(p->d_dotpart);
  if (p->d_rowblk4) hipFree(p->d_rowblk4);
  if (p->d_goff) hipFree(p->d_goff);
  if (p->d_gj) hipFree(p->d_gj);
  delete p;
  return 0;
}

int mi355x_spmv_plan_is_compressed(mi355x_spmv_plan_t p, int *ntab) {
  if (ntab) *ntab = p->d_idx8 ? p->ntab : 0;
  return 0;
}

// would mi355x_spmv_csr_dot run on this plan with this value array?
int mi355x_spmv_plan_dot_available(mi355x_spmv_plan_t p, const double *aa, int *yes) {
  const bool vpat = p->vpat_valid && p->use_vpat && !p->d_rows;
  const bool pat = p->d_prow && p->use_pat && !p->d_rows && mi355x_aligned16(aa);
  if (yes) *yes = (vpat || pat || (p->d_idx8 && !p->d_rows && mi355x_aligned16(aa))) ? 1 : 0;
  return 0;
}

int mi355x_spmv_plan_info(mi355x_spmv_plan_t p, int *nblocks, int *nlong, size_t *workspace_bytes) {
  if (nblocks) *nblocks = p->nblocks;
  if (nlong) *nlong = p->nlong;
  if (workspace_bytes) *workspace_bytes = sizeof(int) * (2 * ((size_t)p->nblocks + 1) + (p->d_rows ? (size_t)p->nrows : 0));
  return 0;
}

int mi355x_spmv_csr(mi355x_handle_t h, mi355x_spmv_plan_t plan, const int *ai, const int *aj, const double *aa,
                    const double *x, double *y) {
  return launch_spmv<0>(h, plan, ai, aj, aa, x, nullptr, y);
}

// y = d .* (A x): MatMult followed by PCApply_Jacobi's VecPointwiseMult (jacobi.c:266) in the product's epilogue
int mi355x_spmv_csr_scaled(mi355x_handle_t h, mi355x_spmv_plan_t plan, const int *ai, const int *aj, const double *aa,
                           const double *x, const double *d, double *y) {
  return launch_spmv<2>(h, plan, ai, aj, aa, x, d, y);
}

// y = A x and x'y from one pass over the matrix (KSPSolve_CG's w = A p, dpi = p'w): every row block leaves its sum of
// x_r y_r in the plan; mi355x_spmv_dot_finish adds them in block order (a second, tiny launch).  Needs the
// index-compressed plan and a square matrix; hipErrorNotSupported otherwise (the caller then uses mi355x_spmv_csr +
// mi355x_vec_dot).
int mi355x_spmv_csr_dot(mi355x_handle_t h, mi355x_spmv_plan_t p, const int *ai, const int *aj, const double *aa,
                        const double *x, double *y) {
  (void)aj;
  // the same choice of kernel as mi355x_spmv_csr makes (value patterns, row patterns, 8-bit offsets), each with the per-block sums
  const bool vpat = p->vpat_valid && p->use_vpat && !p->d_rows;
  const bool pat = p->d_prow && p->use_pat && !p->d_rows && mi355x_aligned16(aa);
  if (!vpat && !pat && (!p->d_idx8 || p->d_rows || !mi355x_aligned16(aa))) return (int)hipErrorNotSupported;
  if (p->nblocks == 0) return 0;
  const int nvb = (p->nrows + SPMV_VPAT_ROWS - 1) / SPMV_VPAT_ROWS;
  { const size_t need = (size_t)(p->nblocks > nvb ? p->nblocks : nvb);
    if (!p->d_dotpart) MI355X_TRY(hipMalloc((void **)&p->d_dotpart, sizeof(double) * need)); }
#if SPMV_REMAP == 2
  const int per8 = MI355X_NXCD * SPMV_CH;
  const int g8 = ((p->nblocks + per8 - 1) / per8) * per8;
#else
  const int g8 = p->nblocks;
#endif
  if (vpat) {
    hipLaunchKernelGGL((spmv_csr_valpat_kernel<0, true>), dim3(nvb), dim3(SPMV_THREADS), 0, h->stream, p->nrows, p->d_vrow, p->d_vpattab,
                       p->d_vpatval, p->vtablen, x, (const double *)nullptr, y, p->d_dotpart, p->pairsum);
    p->ndotpart = nvb;
  } else if (pat) {
    const int perp = MI355X_NXCD * p->ch;
    const int gp = SPMV_REMAP == 2 ? ((p->nblocks + perp - 1) / perp) * perp : p->nblocks;
    hipLaunchKernelGGL((spmv_csr_rowblock_pat_kernel<0, true>), dim3(gp), dim3(SPMV_THREADS), 0, h->stream, p->d_rowblk, p->nblocks,
                       p->d_prow, p->d_pattab, aa, x, (const double *)nullptr, y, p->d_dotpart, p->pairsum, p->ch, spmv_y_streams(p->nrows));
    p->ndotpart = p->nblocks;
  } else {
    hipLaunchKernelGGL((spmv_csr_rowblock_idx8_kernel<0, true>), dim3(g8), dim3(SPMV_THREADS), 0, h->stream, p->d_rowblk,
                       p->nblocks, ai, p->d_idx8, p->d_offtab, p->ntab, aa, x, (const double *)nullptr, y, p->d_dotpart, p->pairsum);
    p->ndotpart = p->nblocks;
  }
  MI355X_LAUNCH_CHECK();
  return 0;
}
int mi355x_spmv_dot_finish(mi355x_handle_t h, mi355x_spmv_plan_t p, double *out) {
  if (p->nblocks == 0 || !p->d_dotpart || p->ndotpart <= 0) { MI355X_TRY(hipMemsetAsync(out, 0, sizeof(double), h->stream)); return 0; }
  hipLaunchKernelGGL(dot_partials_kernel, dim3(1), dim3(1024), 0, h->stream, p->d_dotpart, p->ndotpart, out);
  MI355X_LAUNCH_CHECK();
  return 0;
}

int mi355x_spmv_csr_add(mi355x_handle_t h, mi355x_spmv_plan_t plan, const int *ai, const int *aj, const double *aa,
                        const double *x, const double *y, double *z) {
  return launch_spmv<1>(h, plan, ai, aj, aa, x, y, z);
}

static int spmv_bsr_planned_impl(mi355x_handle_t h, mi355x_spmv_plan_t p, int bs, const int *ai, const int *aj,
                                 const double *aa, const double *x, const double *yin, double *y, bool xlds) {
  if (p->nblocks == 0) return 0;
#if SPMV_REMAP == 2
  const int perb = MI355X_NXCD * SPMV_CH;
  dim3 grid(((p->nblocks + perb - 1) / perb) * perb), block(SPMV_THREADS);
#else
  dim3 grid(p->nblocks), block(SPMV_THREADS);
#endif
#define BSR_GO(B) do { if (xlds) hipLaunchKernelGGL((bsr_rowblock_kernel<B, true>), grid, block, 0, h->stream, p->d_rowblk, p->nblocks, ai, aj, aa, x, yin, y); \
                       else hipLaunchKernelGGL((bsr_rowblock_kernel<B, false>), grid, block, 0, h->stream, p->d_rowblk, p->nblocks, ai, aj, aa, x, yin, y); } while (0)
  switch (bs) {
    case 2: BSR_GO(2); break;
    case 3: BSR_GO(3); break;
    case 4: BSR_GO(4); break;
    case 5: BSR_GO(5); break;
    case 6: BSR_GO(6); break;
    case 7: BSR_GO(7); break;
    case 8: BSR_GO(8); break;
    default: return (int)hipErrorInvalidValue;
  }
#undef BSR_GO
  MI355X_LAUNCH_CHECK();
  return 0;
}
// development / A-B entry points: the two forms of the row-block BCSR kernel side by side (tests/tools/cfg5_baij.py)
int mi355x_spmv_bsr_planned_form(mi355x_handle_t h, mi355x_spmv_plan_t p, int bs, int x_in_lds, const int *ai, const int *aj,
                                 const double *aa, const double *x, double *y) {
  return spmv_bsr_planned_impl(h, p, bs, ai, aj, aa, x, nullptr, y, x_in_lds != 0);
}
int mi355x_spmv_bsr_planned(mi355x_handle_t h, mi355x_spmv_plan_t p, int bs, const int *ai, const int *aj,
                            const double *aa, const double *x, double *y) {
  return spmv_bsr_planned_impl(h, p, bs, ai, aj, aa, x, nullptr, y, MI355X_BSR_XLDS_DEFAULT != 0);
}
// z = y + A x (MatMultAdd_SeqBAIJ_N, baij2.c:1168-1480); z may alias y
int mi355x_spmv_bsr_planned_add(mi355x_handle_t h, mi355x_spmv_plan_t p, int bs, const int *ai, const int *aj,
                                const double *aa, const double *x, const double *y, double *z) {
  return spmv_bsr_planned_impl(h, p, bs, ai, aj, aa, x, y, z, MI355X_BSR_XLDS_DEFAULT != 0);
}

int mi355x_csr_assemble(mi355x_handle_t h, int nseg, const int *segptr, const int *segslot, const int *order, const double *v, double *aa) {
  if (nseg <= 0) return 0;
  hipLaunchKernelGGL(csr_assemble_kernel, dim3((nseg + MI355X_BLOCK - 1) / MI355X_BLOCK), dim3(MI355X_BLOCK), 0, h->stream, nseg, segptr, segslot, order, v, aa);
  MI355X_LAUNCH_CHECK();
  return 0;
}

int mi355x_csr_diagonal_scale(mi355x_handle_t h, int m, const int *ai, const int *aj, double *aa, const double *l, const double *r) {
  if (m <= 0 || (!l && !r)) return 0;
  hipLaunchKernelGGL(csr_diagscale_kernel, dim3((m + MI355X_BLOCK - 1) / MI355X_BLOCK), dim3(MI355X_BLOCK), 0, h->stream, m, ai, aj, aa, l, r);
  MI355X_LAUNCH_CHECK();
  return 0;
}

int mi355x_csr_get_diagonal(mi355x_handle_t h, int m, const int *ai, const int *aj, const double *aa, double *d) {
  if (m <= 0) return 0;
  hipLaunchKernelGGL(csr_diag_kernel, dim3((m + MI355X_BLOCK - 1) / MI355X_BLOCK), dim3(MI355X_BLOCK), 0, h->stream, m,
                     ai, aj, aa, d);
  MI355X_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
