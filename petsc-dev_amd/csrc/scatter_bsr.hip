// Halo pack/unpack (VecScatter Pack_1/UnPack_1, reference src/vec/vec/utils/vpscat.c:493-534)
// and the BCSR SpMV (MatMult_SeqBAIJ_3/_4/_N, reference src/mat/impls/baij/seq/baij2.c:331-436,981).
#include "common.hpp"

__global__ __launch_bounds__(MI355X_BLOCK) void pack_kernel(size_t n, const int *__restrict__ idx,
                                                           const double *__restrict__ x, double *__restrict__ buf) {
  const size_t stride = (size_t)gridDim.x * MI355X_BLOCK;
  for (size_t k = (size_t)blockIdx.x * MI355X_BLOCK + threadIdx.x; k < n; k += stride) buf[k] = x[idx[k]];
}

template <bool ADD>
__global__ __launch_bounds__(MI355X_BLOCK) void unpack_kernel(size_t n, const int *__restrict__ idx,
                                                             const double *__restrict__ buf, double *y) {
  const size_t stride = (size_t)gridDim.x * MI355X_BLOCK;
  for (size_t k = (size_t)blockIdx.x * MI355X_BLOCK + threadIdx.x; k < n; k += stride) {
    const size_t dst = idx ? (size_t)idx[k] : k;
    if (ADD) y[dst] = y[dst] + buf[k];
    else     y[dst] = buf[k];
  }
}

// One wavefront per block row.  The block row's values (nb*bs*bs doubles, blocks stored
// column-major as in baij.h) are read as one contiguous, fully coalesced stream; element e
// belongs to block e/(bs*bs), column (e%(bs*bs))/bs, row e%bs.  Each lane requests its first
// four elements (value, block column, x entry) before consuming any, so a wavefront keeps
// 2 KB of values in flight instead of 512 B; bs accumulators per lane; a shuffle tree
// finishes the bs row sums.
template <int BS>
__global__ __launch_bounds__(MI355X_BLOCK) void bsr_wave_kernel(int mbs, const int *__restrict__ ai,
                                                               const int *__restrict__ aj,
                                                               const double *__restrict__ aa,
                                                               const double *__restrict__ x, double *__restrict__ y) {
  const int lane = threadIdx.x & (MI355X_WAVE - 1);
  const int brow = (blockIdx.x * MI355X_BLOCK + threadIdx.x) / MI355X_WAVE;
  if (brow >= mbs) return;
  constexpr int BS2 = BS * BS;
  constexpr int UNR = 4;
  const long e0 = (long)ai[brow] * BS2, e1 = (long)ai[brow + 1] * BS2;
  double acc[BS];
#pragma unroll
  for (int r = 0; r < BS; ++r) acc[r] = 0.0;

  double av[UNR], xv[UNR];
  int bc[UNR], qq[UNR];
  bool ok[UNR];
#pragma unroll
  for (int t = 0; t < UNR; ++t) {
    const long e = e0 + lane + (long)t * MI355X_WAVE;
    ok[t] = e < e1;
    if (ok[t]) {
      const long blk = e / BS2;
      qq[t] = (int)(e - blk * BS2);
      av[t] = __builtin_nontemporal_load(aa + e);
      bc[t] = aj[blk];
    }
  }
#pragma unroll
  for (int t = 0; t < UNR; ++t)
    if (ok[t]) xv[t] = x[(long)bc[t] * BS + qq[t] / BS];
#pragma unroll
  for (int t = 0; t < UNR; ++t) {
    if (ok[t]) {
      const int r = qq[t] % BS;
      const double p = av[t] * xv[t];
#pragma unroll
      for (int rr = 0; rr < BS; ++rr) acc[rr] += (rr == r) ? p : 0.0;
    }
  }
  for (long e = e0 + lane + (long)UNR * MI355X_WAVE; e < e1; e += MI355X_WAVE) {   // block rows beyond 256/bs^2 blocks
    const long blk = e / BS2;
    const int q = (int)(e - blk * BS2);
    const int c = q / BS;
    const int r = q - c * BS;
    const double p = __builtin_nontemporal_load(aa + e) * x[(long)aj[blk] * BS + c];
#pragma unroll
    for (int rr = 0; rr < BS; ++rr) acc[rr] += (rr == r) ? p : 0.0;
  }
#pragma unroll
  for (int r = 0; r < BS; ++r) acc[r] = wave_sum(acc[r]);
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < BS; ++r) y[(long)brow * BS + r] = acc[r];
  }
}

template <int BS>
static int launch_bsr(mi355x_handle_t h, int mbs, const int *ai, const int *aj, const double *aa, const double *x,
                      double *y) {
  const long threads = (long)mbs * MI355X_WAVE;
  const int grid = (int)((threads + MI355X_BLOCK - 1) / MI355X_BLOCK);
  hipLaunchKernelGGL((bsr_wave_kernel<BS>), dim3(grid), dim3(MI355X_BLOCK), 0, h->stream, mbs, ai, aj, aa, x, y);
  MI355X_LAUNCH_CHECK();
  return 0;
}

extern "C" {

int mi355x_pack(mi355x_handle_t h, size_t n, const int *idx, const double *x, double *buf) {
  if (!n) return 0;
  hipLaunchKernelGGL(pack_kernel, dim3(mi355x_grid_for(n, 4)), dim3(MI355X_BLOCK), 0, h->stream, n, idx, x, buf);
  MI355X_LAUNCH_CHECK();
  return 0;
}
int mi355x_unpack_insert(mi355x_handle_t h, size_t n, const int *idx, const double *buf, double *y) {
  if (!n) return 0;
  hipLaunchKernelGGL((unpack_kernel<false>), dim3(mi355x_grid_for(n, 4)), dim3(MI355X_BLOCK), 0, h->stream, n, idx,
                     buf, y);
  MI355X_LAUNCH_CHECK();
  return 0;
}
int mi355x_unpack_add(mi355x_handle_t h, size_t n, const int *idx, const double *buf, double *y) {
  if (!n) return 0;
  hipLaunchKernelGGL((unpack_kernel<true>), dim3(mi355x_grid_for(n, 4)), dim3(MI355X_BLOCK), 0, h->stream, n, idx,
                     buf, y);
  MI355X_LAUNCH_CHECK();
  return 0;
}

int mi355x_spmv_bsr(mi355x_handle_t h, int mbs, int bs, const int *ai, const int *aj, const double *aa,
                    const double *x, double *y) {
  if (mbs <= 0) return 0;
  switch (bs) {
    case 1: return launch_bsr<1>(h, mbs, ai, aj, aa, x, y);
    case 2: return launch_bsr<2>(h, mbs, ai, aj, aa, x, y);
    case 3: return launch_bsr<3>(h, mbs, ai, aj, aa, x, y);
    case 4: return launch_bsr<4>(h, mbs, ai, aj, aa, x, y);
    case 5: return launch_bsr<5>(h, mbs, ai, aj, aa, x, y);
    case 6: return launch_bsr<6>(h, mbs, ai, aj, aa, x, y);
    case 7: return launch_bsr<7>(h, mbs, ai, aj, aa, x, y);
    case 8: return launch_bsr<8>(h, mbs, ai, aj, aa, x, y);
    default: return (int)hipErrorInvalidValue;
  }
}

}  // extern "C"
