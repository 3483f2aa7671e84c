// Halo pack/unpack (VecScatter Pack_1/UnPack_1, reference src/vec/vec/utils/vpscat.c:493-534)
// and the BCSR SpMV (MatMult_SeqBAIJ_3/_4/_N, reference src/mat/impls/baij/seq/baij2.c:331-436,981).
#include "common.hpp"

__global__ __launch_bounds__(MI355X_BLOCK) void pack_kernel(size_t n, const int *__restrict__ idx,
                                                           const double *__restrict__ x, double *__restrict__ buf) {
  const size_t stride = (size_t)gridDim.x * MI355X_BLOCK;
  for (size_t k = (size_t)blockIdx.x * MI355X_BLOCK + threadIdx.x; k < n; k += stride) buf[k] = x[idx[k]];
}

// UnPack_1 (vpscat.c:503-534): MODE 0 INSERT_VALUES, 1 ADD_VALUES, 2 MAX_VALUES (PetscMax(y, v) = (y < v) ? v : y, petscmath.h)
template <int MODE>
__global__ __launch_bounds__(MI355X_BLOCK) void unpack_kernel(size_t n, const int *__restrict__ idx,
                                                             const double *__restrict__ buf, double *y) {
  const size_t stride = (size_t)gridDim.x * MI355X_BLOCK;
  for (size_t k = (size_t)blockIdx.x * MI355X_BLOCK + threadIdx.x; k < n; k += stride) {
    const size_t dst = idx ? (size_t)idx[k] : k;
    if (MODE == 1) y[dst] = y[dst] + buf[k];
    else if (MODE == 2) { const double a = y[dst], b = buf[k]; y[dst] = (a < b) ? b : a; }
    else y[dst] = buf[k];
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

// ---------------------------------------------------------------------------------------------
// bs = 4 on the matrix cores (BASELINE configs[4]: "MatSeqBAIJ SpMV with MFMA 4x4 tile path"; MatMult_SeqBAIJ_4,
// baij2.c:387-436).  v_mfma_f64_4x4x4_4b_f64 gives every lane one A, one B and one C/D value; measured lane maps on
// gfx950 (tests/tools/probe/mfma_f64_probe.hip): A lane = 16 k + 4 blk + i, B lane = 16 k + 4 blk + j, D lane =
// 16 i + 4 blk + j.  Read with B(lane) = "the x entry that belongs to A(lane)", the instruction is a fused
// multiply + 4-lane strided reduction: D(16 i + 4 blk + i) += sum_k a(16 k + 4 blk + i) * x(16 k + 4 blk + i), i.e.
// lanes l, l+16, l+32, l+48 are summed, whatever they hold -- as long as those four hold entries of ONE matrix row.
// One wavefront per block row streams its blocks (column-major 4x4, baij.h) as one contiguous run:
//   WIDE  : 16 bytes per lane and step, 8 blocks per step: lane l holds elements 2l, 2l+1 of the step's 128 values
//           (block l/8, column (l%8)/2, rows 2(l%2) and 2(l%2)+1): two MFMAs share one x gather;
//   !WIDE : 8 bytes per lane and step, 4 blocks per step: lane l holds element l (block l/16, column (l/4)%4, row l%4).
// Lanes l and l+16 hold the same (column, row) of different blocks in both layouts.  No LDS, no barrier; all loads of
// up to 4 steps are issued before the first MFMA.  The MFMA fuses multiply and add and the final sums over the four
// columns use a shuffle tree, so the result agrees with the reference's loop to rounding (tests: 1e-12 * sum|a x|),
// not bit for bit.
typedef double v2d_b __attribute__((ext_vector_type(2)));
// ROWS block rows per wavefront, all of their loads issued before the first MFMA (ROWS = 2: twice the bytes in flight
// per wavefront at 5 instead of 8 wavefronts per SIMD -- measured slower, only ROWS = 1 is instantiated)
template <bool WIDE, int ROWS>
__global__ __launch_bounds__(MI355X_BLOCK) void bsr4_mfma_kernel(int mbs, const int *__restrict__ ai, const int *__restrict__ aj,
                                                                const double *__restrict__ aa, const double *__restrict__ x,
                                                                double *__restrict__ y) {
  const int lane = threadIdx.x & (MI355X_WAVE - 1);
  // workgroups b, b+8, .. share an XCD: give each XCD runs of 32 consecutive workgroups (128 block rows), so that a block
  // column's x entries are pulled into one L2 instead of eight (the map of the CSR kernels, spmv_csr.hip)
  const int xcd = blockIdx.x % MI355X_NXCD, slot = blockIdx.x / MI355X_NXCD;
  const int wg = ((slot / 32) * MI355X_NXCD + xcd) * 32 + (slot % 32);
  const int br0 = (wg * (MI355X_BLOCK / MI355X_WAVE) + (threadIdx.x >> 6)) * ROWS;
  if (br0 >= mbs) return;
  constexpr int BPS = WIDE ? 8 : 4;           // blocks per step
  constexpr int STEPS = 4;                    // steps in flight
  const int myblk = WIDE ? (lane >> 3) : (lane >> 4);
  const int mycol = WIDE ? ((lane & 7) >> 1) : ((lane >> 2) & 3);
  int a0[ROWS], a1[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int br = (br0 + r < mbs) ? br0 + r : br0;
    a0[r] = ai[br]; a1[r] = (br0 + r < mbs) ? ai[br + 1] : a0[r];
  }
  double acc0[ROWS], acc1[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) { acc0[r] = 0.0; acc1[r] = 0.0; }
  int longest = 0;
#pragma unroll
  for (int r = 0; r < ROWS; ++r) longest = (a1[r] - a0[r] > longest) ? a1[r] - a0[r] : longest;
  for (int off = 0; off < longest; off += BPS * STEPS) {
    v2d_b v[ROWS][STEPS]; double xv[ROWS][STEPS]; int col[ROWS][STEPS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        const int blk = a0[r] + off + BPS * s + myblk;
        const bool ok = blk < a1[r];
        const int bb = ok ? blk : a0[0];
        col[r][s] = aj[bb];
        if (WIDE) {
          const v2d_b t = __builtin_nontemporal_load(reinterpret_cast<const v2d_b *>(aa + (long)bb * 16) + (lane & 7));
          v[r][s].x = ok ? t.x : 0.0; v[r][s].y = ok ? t.y : 0.0;
        } else {
          const double t = __builtin_nontemporal_load(aa + (long)bb * 16 + (lane & 15));
          v[r][s].x = ok ? t : 0.0; v[r][s].y = 0.0;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        const bool ok = a0[r] + off + BPS * s + myblk < a1[r];
        const double t = x[(long)col[r][s] * 4 + mycol];
        xv[r][s] = ok ? t : 0.0;               // a lane past the row contributes 0 * 0, whatever x holds
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        if (a0[r] + off + BPS * s < a1[r]) {   // wave-uniform
          acc0[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[r][s].x, xv[r][s], acc0[r], 0, 0, 0);
          if (WIDE) acc1[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[r][s].y, xv[r][s], acc1[r], 0, 0, 0);
        }
      }
    }
  }
  // the sums sit on the "diagonal" D lanes 16 i + 4 blk + i (class 4 blk + i = lane % 16 of the contributing lanes)
  const bool diag = (lane & 3) == (lane >> 4);
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    if (br0 + r >= mbs) break;
    if (WIDE) {
      // class cls = 4 blk + i holds column (cls % 8) / 2 and rows 2 (cls % 2) [acc0], 2 (cls % 2) + 1 [acc1]
      const int odd = (lane >> 4) & 1;         // i % 2 == cls % 2 on a diagonal lane
      const double r0 = wave_sum(diag && !odd ? acc0[r] : 0.0), r1 = wave_sum(diag && !odd ? acc1[r] : 0.0);
      const double r2 = wave_sum(diag && odd ? acc0[r] : 0.0), r3 = wave_sum(diag && odd ? acc1[r] : 0.0);
      if (lane == 0) {
        v2d_b *yo = reinterpret_cast<v2d_b *>(y + (long)(br0 + r) * 4);
        v2d_b lo, hi; lo.x = r0; lo.y = r1; hi.x = r2; hi.y = r3;
        yo[0] = lo; yo[1] = hi;
      }
    } else {
      // class cls = 4 c + i: row i = lane / 16 on a diagonal lane; the four columns are lanes 4 c apart
      double q = diag ? acc0[r] : 0.0;
      q += __shfl_xor(q, 4, MI355X_WAVE);
      q += __shfl_xor(q, 8, MI355X_WAVE);
      if ((lane & 15) == (lane >> 4)) y[(long)(br0 + r) * 4 + (lane >> 4)] = q;
    }
  }
}

// PCApply_PBJacobi_N (src/ksp/pc/impls/pbjacobi/pbjacobi.c:20-200): y_i = D_i^-1 x_i with the inverted bs x bs diagonal blocks
// stored column-major, one lane per point row, the row's products added left to right (d[r] x0 + d[r+bs] x1 + ...): the
// reference's expression order, same bits.  (8 bs + 16) B per point row: HBM-bound.
__global__ __launch_bounds__(MI355X_BLOCK) void pbjacobi_apply_kernel(long n, int bs, const double *__restrict__ idiag,
                                                                     const double *__restrict__ x, double *__restrict__ y) {
  const long stride = (long)gridDim.x * MI355X_BLOCK;
  for (long row = (long)blockIdx.x * MI355X_BLOCK + threadIdx.x; row < n; row += stride) {
    const long blk = row / bs;
    const int r = (int)(row - blk * bs);
    const double *d = idiag + blk * bs * bs + r, *xx = x + blk * bs;
    double sum = d[0] * xx[0];
    for (int c = 1; c < bs; ++c) sum += d[(long)c * bs] * xx[c];
    y[row] = sum;
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
  hipLaunchKernelGGL((unpack_kernel<0>), dim3(mi355x_grid_for(n, 4)), dim3(MI355X_BLOCK), 0, h->stream, n, idx,
                     buf, y);
  MI355X_LAUNCH_CHECK();
  return 0;
}
int mi355x_unpack_add(mi355x_handle_t h, size_t n, const int *idx, const double *buf, double *y) {
  if (!n) return 0;
  hipLaunchKernelGGL((unpack_kernel<1>), dim3(mi355x_grid_for(n, 4)), dim3(MI355X_BLOCK), 0, h->stream, n, idx,
                     buf, y);
  MI355X_LAUNCH_CHECK();
  return 0;
}

int mi355x_unpack_max(mi355x_handle_t h, size_t n, const int *idx, const double *buf, double *y) {
  if (!n) return 0;
  hipLaunchKernelGGL((unpack_kernel<2>), dim3(mi355x_grid_for(n, 4)), dim3(MI355X_BLOCK), 0, h->stream, n, idx,
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

int mi355x_pbjacobi_apply(mi355x_handle_t h, int mbs, int bs, const double *idiag, const double *x, double *y) {
  if (mbs <= 0) return 0;
  if (bs < 1) return (int)hipErrorInvalidValue;
  const long n = (long)mbs * bs;
  hipLaunchKernelGGL(pbjacobi_apply_kernel, dim3(mi355x_grid_for((size_t)n, 4)), dim3(MI355X_BLOCK), 0, h->stream, n, bs, idiag, x, y);
  MI355X_LAUNCH_CHECK();
  return 0;
}

// MatMult_SeqBAIJ_4 on the matrix cores; variant 0: 16-byte loads, 8 blocks per step; 1: 8-byte loads, 4 blocks per step;
// (two block rows per wavefront, ROWS = 2, was measured 8 % slower: 1.407 against 1.295 ms at 128^3 nodes).
// aa must be 16-byte aligned (blocks are 128 bytes).
int mi355x_spmv_bsr4_mfma(mi355x_handle_t h, int mbs, int variant, const int *ai, const int *aj, const double *aa, const double *x,
                          double *y) {
  if (mbs <= 0) return 0;
  if (!mi355x_aligned16(aa) || !mi355x_aligned16(y)) return (int)hipErrorInvalidValue;
  const int wpb = MI355X_BLOCK / MI355X_WAVE;
  const int per = MI355X_NXCD * 32;                                   // whole runs for every XCD (bsr4_mfma_kernel's workgroup map)
  const int grid = (((mbs + wpb - 1) / wpb + per - 1) / per) * per;
  if (variant == 0) hipLaunchKernelGGL((bsr4_mfma_kernel<true, 1>), dim3(grid), dim3(MI355X_BLOCK), 0, h->stream, mbs, ai, aj, aa, x, y);
  else hipLaunchKernelGGL((bsr4_mfma_kernel<false, 1>), dim3(grid), dim3(MI355X_BLOCK), 0, h->stream, mbs, ai, aj, aa, x, y);
  MI355X_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
