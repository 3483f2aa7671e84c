// CSR SpMV for gfx950: y = A x and z = y + A x  (MatMult_SeqAIJ / MatMultAdd_SeqAIJ,
// reference src/mat/impls/aij/seq/aij.c:1225-1358).
//
// "Row-block streaming" layout of the work (HBM-bound, AI = 0.125 flop/B):
//   * host analysis cuts the rows into row blocks of <= 256 rows and <= 2048 nonzeros;
//   * one 256-thread workgroup per row block streams that block's val/col_idx slice with
//     fully coalesced 16-byte (val) / 8-byte (col) non-temporal loads, gathers x through
//     L1/L2 (x is the only reused operand, so val/col are kept out of the cache with `nt`),
//     multiplies, and parks the products in LDS;
//   * after one barrier each row is summed from LDS: one lane per row, products added in
//     column order starting from 0.0 (or y[r]) -- the exact order of PetscSparseDensePlusDot
//     (aij.h:383-386), so the result is bit-identical to the reference's non-FMA C loop.
//     Row blocks with few, long rows use 2..64 lanes per row and a shuffle tree instead;
//   * a row longer than 2048 nonzeros gets a whole workgroup (strided partial sums + tree);
//   * blockIdx is remapped so that each XCD walks one contiguous eighth of the row blocks:
//     the x entries a 7-point row needs (r, r+-1, r+-N, r+-N^2) are then re-used out of
//     that XCD's own 4 MiB L2 instead of being fetched into all eight.
#include "common.hpp"
#include <vector>

#define SPMV_BLOCK_NNZ 2048
#define SPMV_BLOCK_ROWS 256
#define SPMV_LONG_FLAG 0x40000000

typedef double v2d __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));

struct mi355x_spmv_plan_s {
  int nrows;       // rows of the (possibly compressed) row pointer
  int nblocks;     // row blocks
  int nlong;       // of which single long rows
  int chunk;       // ceil(nblocks / NXCD)
  int *d_rowblk;   // nblocks+1 row boundaries
  int *d_rows;     // compressed-row output indices or NULL
};

template <bool ADD, bool CPROW, bool VEC>
__global__ __launch_bounds__(MI355X_BLOCK) void spmv_csr_rowblock_kernel(
    const int *__restrict__ rowblk, int nblocks, int chunk, const int *__restrict__ ai, const int *__restrict__ aj,
    const double *__restrict__ aa, const double *__restrict__ x, const double *yin, double *yout,
    const int *__restrict__ rows) {
  __shared__ double prod[SPMV_BLOCK_NNZ];
  __shared__ double wsum[MI355X_BLOCK / MI355X_WAVE];

  // XCD-aware remap: workgroups b, b+8, b+16.. share an XCD; give them consecutive row blocks
  const int xcd = blockIdx.x % MI355X_NXCD;
  const int slot = blockIdx.x / MI355X_NXCD;
  const int lb = xcd * chunk + slot;
  if (slot >= chunk || lb >= nblocks) return;

  const int r0 = rowblk[lb];
  const int r1 = rowblk[lb + 1];
  const int k0 = ai[r0];
  const int k1 = ai[r1];
  const int nnz = k1 - k0;
  const int nrows = r1 - r0;
  const int tid = threadIdx.x;

  if (nnz > SPMV_BLOCK_NNZ) {
    // one long row: strided partial sums, then a fixed tree
    double s = 0.0;
    for (int k = k0 + tid; k < k1; k += MI355X_BLOCK) {
      double v = __builtin_nontemporal_load(aa + k);
      int c = __builtin_nontemporal_load(aj + k);
      s += v * x[c];
    }
    s = wave_sum(s);
    if ((tid & (MI355X_WAVE - 1)) == 0) wsum[tid / MI355X_WAVE] = s;
    __syncthreads();
    if (tid == 0) {
      double t = wsum[0];
#pragma unroll
      for (int w = 1; w < MI355X_BLOCK / MI355X_WAVE; ++w) t += wsum[w];
      const int orow = CPROW ? rows[r0] : r0;
      yout[orow] = ADD ? (yin[orow] + t) : t;
    }
    return;
  }

  // ---- stream the block's nonzeros: product -> LDS -----------------------
  if (VEC) {
    const int ka = k0 & ~1;  // 16-byte aligned start for val, 8-byte for col
    for (int k = ka + 2 * tid; k < k1; k += 2 * MI355X_BLOCK) {
      if (k >= k0 && k + 1 < k1) {
        v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(aa + k));
        v2i c = __builtin_nontemporal_load(reinterpret_cast<const v2i *>(aj + k));
        double x0 = x[c.x];
        double x1 = x[c.y];
        prod[k - k0] = v.x * x0;
        prod[k - k0 + 1] = v.y * x1;
      } else {
        if (k >= k0) prod[k - k0] = __builtin_nontemporal_load(aa + k) * x[__builtin_nontemporal_load(aj + k)];
        if (k + 1 >= k0 && k + 1 < k1)
          prod[k + 1 - k0] = __builtin_nontemporal_load(aa + k + 1) * x[__builtin_nontemporal_load(aj + k + 1)];
      }
    }
  } else {
    for (int k = k0 + tid; k < k1; k += MI355X_BLOCK)
      prod[k - k0] = __builtin_nontemporal_load(aa + k) * x[__builtin_nontemporal_load(aj + k)];
  }
  __syncthreads();

  // ---- per-row sums out of LDS -------------------------------------------
  // lanes per row: largest power of two with nrows*tpr <= 256, at most one wavefront
  int tpr = 1;
  while (tpr < MI355X_WAVE && nrows * (tpr * 2) <= MI355X_BLOCK) tpr *= 2;
  if (nnz <= 16 * nrows) tpr = 1;  // short rows: one lane per row, reference summation order

  if (tpr == 1) {
    for (int r = tid; r < nrows; r += MI355X_BLOCK) {
      const int row = r0 + r;
      const int s = ai[row] - k0;
      const int e = ai[row + 1] - k0;
      const int orow = CPROW ? rows[row] : row;
      double sum = ADD ? yin[orow] : 0.0;
      for (int k = s; k < e; ++k) sum += prod[k];
      yout[orow] = sum;
    }
  } else {
    const int r = tid / tpr;
    const int sub = tid & (tpr - 1);
    double sum = 0.0;
    int orow = 0;
    if (r < nrows) {
      const int row = r0 + r;
      const int s = ai[row] - k0;
      const int e = ai[row + 1] - k0;
      orow = CPROW ? rows[row] : row;
      for (int k = s + sub; k < e; k += tpr) sum += prod[k];
    }
    for (int off = tpr >> 1; off > 0; off >>= 1) sum += __shfl_down(sum, off, MI355X_WAVE);
    if (r < nrows && sub == 0) yout[orow] = ADD ? (yin[orow] + sum) : sum;
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

template <bool ADD>
static int launch_spmv(mi355x_handle_t h, mi355x_spmv_plan_t p, const int *ai, const int *aj, const double *aa,
                       const double *x, const double *yin, double *yout) {
  if (p->nblocks == 0) return 0;
  const bool vec = mi355x_aligned16(aa) && ((((uintptr_t)aj) & 7u) == 0);
  const bool cprow = p->d_rows != nullptr;
  dim3 grid(p->chunk * MI355X_NXCD), block(MI355X_BLOCK);
#define SPMV_GO(C, V)                                                                                               \
  hipLaunchKernelGGL((spmv_csr_rowblock_kernel<ADD, C, V>), grid, block, 0, h->stream, p->d_rowblk, p->nblocks,    \
                     p->chunk, ai, aj, aa, x, yin, yout, p->d_rows)
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
  p->nlong = 0;
  std::vector<int> rb;
  rb.reserve((size_t)nrows / 128 + 2);
  rb.push_back(0);
  int r = 0;
  while (r < nrows) {
    const int start = r;
    int nnz = 0;
    while (r < nrows && (r - start) < SPMV_BLOCK_ROWS) {
      const int len = ai_host[r + 1] - ai_host[r];
      if (nnz + len > SPMV_BLOCK_NNZ) break;
      nnz += len;
      ++r;
    }
    if (r == start) {  // a single row longer than the LDS stage
      ++r;
      p->nlong++;
    }
    rb.push_back(r);
  }
  p->nblocks = (int)rb.size() - 1;
  p->chunk = (p->nblocks + MI355X_NXCD - 1) / MI355X_NXCD;
  MI355X_TRY(hipMalloc((void **)&p->d_rowblk, sizeof(int) * rb.size()));
  MI355X_TRY(hipMemcpyAsync(p->d_rowblk, rb.data(), sizeof(int) * rb.size(), hipMemcpyHostToDevice, h->stream));
  if (rows_host) {
    MI355X_TRY(hipMalloc((void **)&p->d_rows, sizeof(int) * (size_t)(nrows > 0 ? nrows : 1)));
    MI355X_TRY(hipMemcpyAsync(p->d_rows, rows_host, sizeof(int) * (size_t)nrows, hipMemcpyHostToDevice, h->stream));
  }
  MI355X_TRY(hipStreamSynchronize(h->stream));  // rb is a local
  *plan = p;
  return 0;
}

int mi355x_spmv_plan_destroy(mi355x_spmv_plan_t p) {
  if (!p) return 0;
  hipFree(p->d_rowblk);
  if (p->d_rows) hipFree(p->d_rows);
  delete p;
  return 0;
}

int mi355x_spmv_plan_info(mi355x_spmv_plan_t p, int *nblocks, int *nlong, size_t *workspace_bytes) {
  if (nblocks) *nblocks = p->nblocks;
  if (nlong) *nlong = p->nlong;
  if (workspace_bytes) *workspace_bytes = sizeof(int) * ((size_t)p->nblocks + 1 + (p->d_rows ? (size_t)p->nrows : 0));
  return 0;
}

int mi355x_spmv_csr(mi355x_handle_t h, mi355x_spmv_plan_t plan, const int *ai, const int *aj, const double *aa,
                    const double *x, double *y) {
  return launch_spmv<false>(h, plan, ai, aj, aa, x, nullptr, y);
}

int mi355x_spmv_csr_add(mi355x_handle_t h, mi355x_spmv_plan_t plan, const int *ai, const int *aj, const double *aa,
                        const double *x, const double *y, double *z) {
  return launch_spmv<true>(h, plan, ai, aj, aa, x, y, z);
}

int mi355x_csr_get_diagonal(mi355x_handle_t h, int m, const int *ai, const int *aj, const double *aa, double *d) {
  if (m <= 0) return 0;
  hipLaunchKernelGGL(csr_diag_kernel, dim3((m + MI355X_BLOCK - 1) / MI355X_BLOCK), dim3(MI355X_BLOCK), 0, h->stream, m,
                     ai, aj, aa, d);
  MI355X_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
