// Shared definitions of the MI355X kernel library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "mi355x_kernels.h"

#define MI355X_WAVE 64            // CDNA wavefront width
#define MI355X_BLOCK 256          // 4 wavefronts per workgroup, one per SIMD
#define MI355X_NXCD 8             // XCDs (each with a private 4 MiB L2)
#define MI355X_MAX_GRID 2048      // 256 CUs x 8 resident 256-thread workgroups
#define MI355X_REDUCE_GRID_CAP 512   // workgroups of a reduction launch over vectors that fit the Infinity Cache (vec_kernels.hip launch_reduce): grid-stride
#define MI355X_REDUCE_GRID_CAP_BIG 4096   // ... over vectors of >= 256 MiB: contiguous runs of tiles, one run per workgroup
#define MI355X_TILE2 1024            // double2's of a reduction tile: 256 lanes x 4 (vec_kernels.hip)
#define MI355X_MAP_TILE2 512         // double2's of an element-wise tile: 256 lanes x 2
#define MI355X_MAX_RED 32         // max simultaneous reduction outputs (MDot chunk)
#define MI355X_SCRATCH_DOUBLES 64

#define MI355X_TRY(expr)                                   \
  do {                                                     \
    hipError_t e_ = (expr);                                \
    if (e_ != hipSuccess) return (int)e_;                  \
  } while (0)

#define MI355X_LAUNCH_CHECK()                              \
  do {                                                     \
    hipError_t e_ = hipGetLastError();                     \
    if (e_ != hipSuccess) return (int)e_;                  \
  } while (0)

struct mi355x_handle_s {
  hipStream_t stream;
  double *partials;           // MI355X_REDUCE_GRID_CAP_BIG * MI355X_MAX_RED doubles (HBM)
  unsigned int *ticket;       // arrival counter for the single-launch reductions
  double *host_scratch;       // pinned + mapped, MI355X_SCRATCH_DOUBLES
  double *dev_scratch;        // HBM, MI355X_SCRATCH_DOUBLES
  unsigned long long seq;     // sequence number of the last reduction whose result goes to host_scratch
  volatile unsigned long long *host_seq;  // pinned word the finishing workgroup stores that number to (after the result)
};

struct mi355x_event_s {
  hipEvent_t ev;
};

static inline int mi355x_grid_for(size_t n, int per_thread) {
  size_t per_block = (size_t)MI355X_BLOCK * (size_t)per_thread;
  size_t nb = (n + per_block - 1) / per_block;
  if (nb < 1) nb = 1;
  if (nb > MI355X_MAX_GRID) nb = MI355X_MAX_GRID;
  return (int)nb;
}

static inline bool mi355x_aligned16(const void *p) { return (((uintptr_t)p) & 15u) == 0; }

// ---- device helpers ----------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = MI355X_WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, MI355X_WAVE);
  return v;
}
// max that propagates NaN the way VecNorm_Seq's NORM_INFINITY loop does (bvec2.c:628-630)
__device__ __forceinline__ double nanmax(double a, double b) {
  if (a != a) return a;
  if (b != b) return b;
  return a > b ? a : b;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = MI355X_WAVE / 2; off > 0; off >>= 1) v = nanmax(v, __shfl_down(v, off, MI355X_WAVE));
  return v;
}
