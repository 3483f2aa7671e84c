// Vec BLAS-1 kernels for gfx950.  All of them are HBM-bound streams of 16-byte (double2) loads / stores per lane.
// How the index space is dealt to workgroups decides the rate (profiles/r04_stream_probe*.log, n = 2^27 doubles per vector, beyond the
// 256 MiB Infinity Cache): a grid-stride loop over 2048 resident workgroups copies at 4.9 TB/s, ONE CONTIGUOUS TILE PER WORKGROUP
// ("flat": as many workgroups as tiles, the dispatcher hands out the next tile when a workgroup retires, so the tiles in flight
// form one compact window that slides through memory) at 6.0, with non-temporal accesses at 6.3-6.7; inside the cache (2^24) 7.7 ->
// 7.8-8.4 and the non-temporal hint costs 20 %.  So: element-wise kernels take one tile of 256 x 2 double2 per workgroup;
// reductions, whose workgroup count is also the number of partials the last workgroup has to add, keep <= 512 grid-striding
// workgroups for vectors that fit the cache (inside the cache the geometry makes no difference to a pure reader) and take contiguous
// runs of tiles of 256 x 4 double2, one run each for <= 4096 workgroups, for vectors of >= 256 MiB; those are also streamed
// non-temporally.  Compiled with -ffp-contract=off so a*x+y is a rounded multiply then a rounded add, as in the
// reference's C loops (src/vec/vec/impls/seq/{bvec1,bvec2,dvec2}.c).
#include "common.hpp"

// Streaming accesses for operands nobody reads again soon.  In a CG iteration x and r are touched by the update sweep only and z
// by the AYPX that follows it only; loaded / stored non-temporally they stop evicting p, w and z from the L2 / Infinity Cache,
// which the neighbouring kernels re-read (measured on P7(256): fused update 0.183 -> 0.166 ms, whole iteration 0.414 -> 0.374 ms).
// The Jacobi diagonal, read once per iteration by the update alone, is a stream of the same kind (+3 % more); w, read here for the
// last time but still cached from the dot before, is better left alone.  Same values, same bits.
typedef double vk_v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 nt_load2(const double2 *p) {
  const vk_v2d t = __builtin_nontemporal_load(reinterpret_cast<const vk_v2d *>(p));
  double2 r; r.x = t.x; r.y = t.y; return r;
}
__device__ __forceinline__ void nt_store2(double2 *p, double2 v) {
  vk_v2d t; t.x = v.x; t.y = v.y;
  __builtin_nontemporal_store(t, reinterpret_cast<vk_v2d *>(p));
}
// Gram-Schmidt sweeps (VecMDot, VecMAXPY and their fused forms): when the nv basis vectors of a sweep are larger than the caches
// they are pure streams, and loaded non-temporally they leave the ONE vector the sweep shares with its neighbours (w: written by
// the product, read by MDot, read and written by MAXPY, scaled, gathered by the next product) in the L2 / Infinity Cache:
// GMRES(30)+Jacobi on P7(256) 1.045 -> 0.955 ms per iteration.  A basis that fits (P7(64): 16 x 2 MB) must stay cacheable:
// there the hint costs 20 %.  Threshold: half of the 256 MiB Infinity Cache.
static inline int gs_streams(size_t n, int nv) { return (size_t)nv * n * sizeof(double) > ((size_t)128 << 20); }

// ------------------------------------------------------------------------
// element-wise map:  out[i] = op(a[i], b[i], c[i])   (inputs may alias out)
// ------------------------------------------------------------------------
// big: vectors of >= 256 MiB cannot be cache-resident between two kernels: stream them (measured: +5-12 % there, -20 % inside the cache)
static inline int vec_streams(size_t n) { return n * sizeof(double) >= ((size_t)256 << 20); }
template <bool NT> __device__ __forceinline__ double2 ld2(const double2 *p) { return NT ? nt_load2(p) : *p; }
template <bool NT> __device__ __forceinline__ void st2(double2 *p, double2 v) { if (NT) nt_store2(p, v); else *p = v; }
// the same choice at run time (a functor member, uniform over the launch)
__device__ __forceinline__ double2 ldq(const double2 *p, int nt) { return nt ? nt_load2(p) : *p; }
__device__ __forceinline__ void stq(double2 *p, double2 v, int nt) { if (nt) nt_store2(p, v); else *p = v; }

// vec_ok: workgroup b owns the tile [b * MI355X_MAP_TILE2, (b + 1) * MI355X_MAP_TILE2) of double2's, lane t its entries t and t + 256
template <int NIN, class Op, bool NT>
__global__ __launch_bounds__(MI355X_BLOCK) void map_kernel(Op op, const double *a, const double *b, const double *c,
                                                          double *out, size_t n, int vec_ok) {
  const size_t tid = (size_t)blockIdx.x * MI355X_BLOCK + threadIdx.x;
  if (vec_ok) {
    const size_t n2 = n >> 1;
    const double2 *a2 = reinterpret_cast<const double2 *>(a);
    const double2 *b2 = reinterpret_cast<const double2 *>(b);
    const double2 *c2 = reinterpret_cast<const double2 *>(c);
    double2 *o2 = reinterpret_cast<double2 *>(out);
    const size_t stride = MI355X_BLOCK;
    const size_t i = (size_t)blockIdx.x * MI355X_MAP_TILE2 + threadIdx.x;
    if (i + stride < n2) {
      double2 av0 = {0, 0}, bv0 = {0, 0}, cv0 = {0, 0}, av1 = {0, 0}, bv1 = {0, 0}, cv1 = {0, 0};
      if (NIN >= 1) { av0 = ld2<NT>(a2 + i); av1 = ld2<NT>(a2 + i + stride); }
      if (NIN >= 2) { bv0 = ld2<NT>(b2 + i); bv1 = ld2<NT>(b2 + i + stride); }
      if (NIN >= 3) { cv0 = ld2<NT>(c2 + i); cv1 = ld2<NT>(c2 + i + stride); }
      double2 r0, r1;
      r0.x = op(av0.x, bv0.x, cv0.x); r0.y = op(av0.y, bv0.y, cv0.y);
      r1.x = op(av1.x, bv1.x, cv1.x); r1.y = op(av1.y, bv1.y, cv1.y);
      st2<NT>(o2 + i, r0);
      st2<NT>(o2 + i + stride, r1);
    } else if (i < n2) {
      double2 av = {0, 0}, bv = {0, 0}, cv = {0, 0};
      if (NIN >= 1) av = a2[i];
      if (NIN >= 2) bv = b2[i];
      if (NIN >= 3) cv = c2[i];
      double2 r;
      r.x = op(av.x, bv.x, cv.x); r.y = op(av.y, bv.y, cv.y);
      o2[i] = r;
    }
    if ((n & 1) && tid == 0) {
      const size_t k = n - 1;
      out[k] = op(NIN >= 1 ? a[k] : 0.0, NIN >= 2 ? b[k] : 0.0, NIN >= 3 ? c[k] : 0.0);
    }
  } else {
    const size_t stride = (size_t)gridDim.x * MI355X_BLOCK;
    for (size_t i = tid; i < n; i += stride)
      out[i] = op(NIN >= 1 ? a[i] : 0.0, NIN >= 2 ? b[i] : 0.0, NIN >= 3 ? c[i] : 0.0);
  }
}

// workgroups of an element-wise launch: one per tile of MI355X_MAP_TILE2 double2's (an unaligned view takes the scalar grid-stride loop)
static inline unsigned int map_grid(size_t n, int vec_ok) {
  if (!vec_ok) return (unsigned int)mi355x_grid_for(n, 4);
  const size_t n2 = n >> 1, nt = (n2 + MI355X_MAP_TILE2 - 1) / MI355X_MAP_TILE2;
  return (unsigned int)(nt ? nt : 1);
}
template <int NIN, class Op>
static int launch_map(mi355x_handle_t h, Op op, const double *a, const double *b, const double *c, double *out, size_t n) {
  if (n == 0) return 0;
  int vec_ok = mi355x_aligned16(out) && (NIN < 1 || mi355x_aligned16(a)) && (NIN < 2 || mi355x_aligned16(b)) &&
               (NIN < 3 || mi355x_aligned16(c));
  const unsigned int grid = map_grid(n, vec_ok);
  if (vec_streams(n)) hipLaunchKernelGGL((map_kernel<NIN, Op, true>), dim3(grid), dim3(MI355X_BLOCK), 0, h->stream, op, a, b, c, out, n, vec_ok);
  else hipLaunchKernelGGL((map_kernel<NIN, Op, false>), dim3(grid), dim3(MI355X_BLOCK), 0, h->stream, op, a, b, c, out, n, vec_ok);
  MI355X_LAUNCH_CHECK();
  return 0;
}

struct OpSet      { double al; __device__ double operator()(double, double, double) const { return al; } };
struct OpCopy     { __device__ double operator()(double x, double, double) const { return x; } };
struct OpScale    { double al; __device__ double operator()(double x, double, double) const { return al * x; } };
struct OpAxpy     { double al; __device__ double operator()(double x, double y, double) const { return y + al * x; } };
struct OpAypx     { double al; __device__ double operator()(double x, double y, double) const { return x + al * y; } };
struct OpXmY      { __device__ double operator()(double x, double y, double) const { return x - y; } };
struct OpYmX      { __device__ double operator()(double x, double y, double) const { return y - x; } };
struct OpXpY      { __device__ double operator()(double x, double y, double) const { return y + x; } };
struct OpAxpby    { double al, be; __device__ double operator()(double x, double y, double) const { return al * x + be * y; } };
struct OpMul      { __device__ double operator()(double x, double y, double) const { return x * y; } };
struct OpDiv      { __device__ double operator()(double x, double y, double) const { return x / y; } };
struct OpRecip    { __device__ double operator()(double x, double, double) const { return x != 0.0 ? 1.0 / x : x; } };
struct OpJacInv   { __device__ double operator()(double x, double, double) const { return x == 0.0 ? 1.0 : 1.0 / x; } };
// z = al x + be y + ga z with the reference's four variants (bvec1.c:430-449)
struct OpAxpbypczA1 { double be, ga; __device__ double operator()(double x, double y, double z) const { return x + be * y + ga * z; } };
struct OpAxpbypczG1 { double al, be; __device__ double operator()(double x, double y, double z) const { return al * x + be * y + z; } };
struct OpAxpbypczG0 { double al, be; __device__ double operator()(double x, double y, double) const { return al * x + be * y; } };
struct OpAxpbypcz   { double al, be, ga; __device__ double operator()(double x, double y, double z) const { return al * x + be * y + ga * z; } };
struct OpTriad    { double al; __device__ double operator()(double b, double c, double) const { return b + al * c; } };

// swap needs two outputs
__global__ __launch_bounds__(MI355X_BLOCK) void swap_kernel(double *x, double *y, size_t n) {
  const size_t stride = (size_t)gridDim.x * MI355X_BLOCK;
  for (size_t i = (size_t)blockIdx.x * MI355X_BLOCK + threadIdx.x; i < n; i += stride) {
    double t = x[i];
    x[i] = y[i];
    y[i] = t;
  }
}

// ------------------------------------------------------------------------
// MAXPY: x += (a0 y0 + .. ) [group 0, G0 vectors] ; x += (..) [group 1, G1 vectors]
// groups and left-to-right sums as petscaxpy.h:101-110 / dvec2.c:853-900
// ------------------------------------------------------------------------
struct MaxpyArgs {
  const double *y[32];
  double a[32];
  int nt;
};

template <int G>
__device__ __forceinline__ double group_sum(const double *a, const double *v) {
  double s = a[0] * v[0];
#pragma unroll
  for (int j = 1; j < G; ++j) s = s + a[j] * v[j];
  return s;
}

// x += group 0 (G0 = 1..4 vectors), then NG4 groups of four, each group summed left to right and added to x in turn:
// the association of petscaxpy.h:101-110.  Up to 32 vectors per sweep, so x is read and written once per 32
// (GMRES(30)'s largest update in one sweep: 0.79 ms instead of 0.86 ms at n = 2^24).
template <int G0, int NG4>
__device__ __forceinline__ double maxpy_elem(double xv, const double *a, const double *v) {
  xv = xv + group_sum<G0>(a, v);
#pragma unroll
  for (int g = 0; g < NG4; ++g) xv = xv + group_sum<4>(a + G0 + 4 * g, v + G0 + 4 * g);
  return xv;
}

template <int G0, int NG4>
__global__ __launch_bounds__(MI355X_BLOCK) void maxpy_kernel(MaxpyArgs args, double *x, size_t n, int vec_ok) {
  const size_t tid = (size_t)blockIdx.x * MI355X_BLOCK + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * MI355X_BLOCK;
  constexpr int NV = G0 + 4 * NG4;
  if (vec_ok) {
    const size_t n2 = n >> 1;
    double2 *x2 = reinterpret_cast<double2 *>(x);
    const size_t i = tid;                            // one double2 of every stream per lane, one contiguous tile per workgroup (grid = all tiles)
    if (i < n2) {
      double2 yv[NV];
      if (args.nt) {                               // see gs_streams()
#pragma unroll
        for (int j = 0; j < NV; ++j) yv[j] = nt_load2(reinterpret_cast<const double2 *>(args.y[j]) + i);
      } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) yv[j] = reinterpret_cast<const double2 *>(args.y[j])[i];
      }
      double2 xv = x2[i];
      double lo[NV], hi[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) { lo[j] = yv[j].x; hi[j] = yv[j].y; }
      xv.x = maxpy_elem<G0, NG4>(xv.x, args.a, lo);
      xv.y = maxpy_elem<G0, NG4>(xv.y, args.a, hi);
      x2[i] = xv;
    }
    if ((n & 1) && tid == 0) {
      const size_t k = n - 1;
      double v[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j] = args.y[j][k];
      x[k] = maxpy_elem<G0, NG4>(x[k], args.a, v);
    }
  } else {
    for (size_t k = tid; k < n; k += stride) {
      double v[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j] = args.y[j][k];
      x[k] = maxpy_elem<G0, NG4>(x[k], args.a, v);
    }
  }
}

template <int G0, int NG4>
static int launch_maxpy(mi355x_handle_t h, const MaxpyArgs &args, double *x, size_t n, int vec_ok) {
  const size_t nt2 = ((n >> 1) + MI355X_BLOCK - 1) / MI355X_BLOCK;
  const unsigned int grid = vec_ok ? (unsigned int)(nt2 ? nt2 : 1) : (unsigned int)mi355x_grid_for(n, 2);
  hipLaunchKernelGGL((maxpy_kernel<G0, NG4>), dim3(grid), dim3(MI355X_BLOCK), 0, h->stream, args, x, n, vec_ok);
  MI355X_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------
// reductions: single launch, fixed tree, last-arriving workgroup finishes
// ------------------------------------------------------------------------
enum { RED_SUM = 0, RED_MAX = 1 };

template <int NOUT, int MODE, bool PUBLISH = false>
__device__ __forceinline__ void block_reduce_store(double (&acc)[NOUT], double *dst /* NOUT doubles */, double (*lds)[NOUT]) {
  const int lane = threadIdx.x & (MI355X_WAVE - 1);
  const int wave = threadIdx.x / MI355X_WAVE;
#pragma unroll
  for (int j = 0; j < NOUT; ++j) {
    double v = (MODE == RED_MAX) ? wave_max(acc[j]) : wave_sum(acc[j]);
    if (lane == 0) lds[wave][j] = v;
  }
  __syncthreads();
  if (threadIdx.x < NOUT) {
    double s = lds[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < MI355X_BLOCK / MI355X_WAVE; ++w)
      s = (MODE == RED_MAX) ? nanmax(s, lds[w][threadIdx.x]) : s + lds[w][threadIdx.x];
    // a partial that another workgroup will read is stored write-through (sc1): it reaches the memory side
    // without an L2 write-back fence per workgroup
    if (PUBLISH) __hip_atomic_store(dst + threadIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else dst[threadIdx.x] = s;
  }
}

// After the result has been stored: make it visible to the host, then store the completion number the host polls.
__device__ __forceinline__ void publish_to_host(unsigned long long *host_seq, unsigned long long seq) {
  __syncthreads();                       // the result stores of lanes < NOUT are done
  if (host_seq && threadIdx.x == 0) {
    __threadfence_system();
    __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// y = x + (num/den) y with the scalar's numerator still in device memory (KSPSolve_CG: b = beta_new/beta_old, beta_new
// being the z'r the previous kernel on the stream has just reduced).  VecAYPX_Seq's special case alpha == 0 -> copy
// (dvec2.c:980) is kept; alpha == +-1 need no special form (x + 1*y and x + (-1)*y are the bits of x + y and x - y).
template <bool NT>
__global__ __launch_bounds__(MI355X_BLOCK) void aypx_dev_kernel(const double *num, double den, const double *x, double *y, size_t n, int vec_ok) {
  const double alpha = *num / den;
  const bool copy = (alpha == 0.0);
  const size_t tid = (size_t)blockIdx.x * MI355X_BLOCK + threadIdx.x;
  if (vec_ok) {                                    // one tile of MI355X_MAP_TILE2 double2's per workgroup (see map_kernel)
    const size_t n2 = n >> 1;
    const double2 *x2 = reinterpret_cast<const double2 *>(x);
    double2 *y2 = reinterpret_cast<double2 *>(y);
    const size_t stride = MI355X_BLOCK;
    const size_t i = (size_t)blockIdx.x * MI355X_MAP_TILE2 + threadIdx.x;
    if (i + stride < n2) {
      double2 xv0 = nt_load2(x2 + i), xv1 = nt_load2(x2 + i + stride), yv0 = ld2<NT>(y2 + i), yv1 = ld2<NT>(y2 + i + stride), r0, r1;   // x = z: its last reader (see nt_load2)
      r0.x = copy ? xv0.x : xv0.x + alpha * yv0.x; r0.y = copy ? xv0.y : xv0.y + alpha * yv0.y;
      r1.x = copy ? xv1.x : xv1.x + alpha * yv1.x; r1.y = copy ? xv1.y : xv1.y + alpha * yv1.y;
      st2<NT>(y2 + i, r0);
      st2<NT>(y2 + i + stride, r1);
    } else if (i < n2) {
      double2 xv = nt_load2(x2 + i), yv = y2[i], r;
      r.x = copy ? xv.x : xv.x + alpha * yv.x; r.y = copy ? xv.y : xv.y + alpha * yv.y;
      y2[i] = r;
    }
    if ((n & 1) && tid == 0) y[n - 1] = copy ? x[n - 1] : x[n - 1] + alpha * y[n - 1];
  } else {
    const size_t stride = (size_t)gridDim.x * MI355X_BLOCK;
    for (size_t i = tid; i < n; i += stride) y[i] = copy ? x[i] : x[i] + alpha * y[i];
  }
}

// copy <= 64 doubles from device memory to the handle's pinned scratch, then store the completion number the host
// polls (mi355x_handle_wait_result): how a result that had to stay on the device first (RCCL all-reduce, a later kernel
// reading it) reaches the host without a stream synchronisation
__global__ void publish_kernel(const double *src, double *host_dst, int count, unsigned long long *host_seq, unsigned long long seq) {
  if ((int)threadIdx.x < count) host_dst[threadIdx.x] = src[threadIdx.x];
  publish_to_host(host_seq, seq);
}

// functors that need a once-per-lane hook before the sweep specialise this
template <class F> struct has_prologue { static constexpr bool value = false; };

// F::accum(i2 or i, acc): adds element contributions
// RUNS == false: grid-stride (lane t of workgroup b meets double2's b * 256 + t, + grid * 256, ...): vectors that live in the caches.
// RUNS == true: workgroup b owns a contiguous run of tiles of MI355X_TILE2 double2's, [s0, s1), and lane t meets s0 + t, s0 + t + 256,
// ...: vectors of >= 256 MiB, where one compact window of tiles in flight streams 6.0-7.0 TB/s and 512 strided workgroups 4.9-6.1
// (profiles/r04_stream_probe*.log).  The functors' sweeps take a first index, a step and an end, so they serve both.  The oracle's
// device-order emulation (oracle/vecmat_oracle.c dev_reduce) restates both geometries and the size that separates them: change together.
template <int NOUT, int MODE, class F, bool RUNS>
__global__ __launch_bounds__(MI355X_BLOCK) void reduce_kernel(F f, size_t n, int vec_ok, double *partials,
                                                             unsigned int *ticket, double *out, double *host_copy,
                                                             unsigned long long *host_seq, unsigned long long seq) {
  __shared__ double lds[MI355X_BLOCK / MI355X_WAVE][NOUT];
  __shared__ int is_last;
  const size_t tid = (size_t)blockIdx.x * MI355X_BLOCK + threadIdx.x;
  double acc[NOUT];
#pragma unroll
  for (int j = 0; j < NOUT; ++j) acc[j] = 0.0;
  if constexpr (has_prologue<F>::value) f.prologue(tid, acc);
  if (vec_ok) {
    const size_t n2 = n >> 1;
    if (RUNS) {
      const size_t ntiles = (n2 + MI355X_TILE2 - 1) / MI355X_TILE2;
      const size_t per = (ntiles + gridDim.x - 1) / gridDim.x;
      size_t s0 = (size_t)blockIdx.x * per * MI355X_TILE2, s1 = s0 + per * MI355X_TILE2;
      if (s0 > n2) s0 = n2;
      if (s1 > n2) s1 = n2;
      f.template sweep<NOUT>(s0 + threadIdx.x, (size_t)MI355X_BLOCK, s1, acc);
    } else {
      f.template sweep<NOUT>(tid, (size_t)gridDim.x * MI355X_BLOCK, n2, acc);
    }
    if ((n & 1) && tid == 0) f.accum1(n - 1, acc);
  } else {
    const size_t stride = (size_t)gridDim.x * MI355X_BLOCK;
    for (size_t i = tid; i < n; i += stride) f.accum1(i, acc);
  }
  if (gridDim.x == 1) {  // single workgroup: no hand-off needed
    block_reduce_store<NOUT, MODE>(acc, out, lds);
    if (host_copy) { __syncthreads(); if (threadIdx.x < NOUT) host_copy[threadIdx.x] = out[threadIdx.x]; }
    publish_to_host(host_seq, seq);
    return;
  }
  block_reduce_store<NOUT, MODE, true>(acc, partials + (size_t)blockIdx.x * NOUT, lds);
  // Publish: the partials were stored write-through (sc1) by lanes of wavefront 0; once those stores have left
  // (vmcnt(0)) lane 0 of the SAME wavefront takes a ticket with a relaxed agent-scope add.  The workgroup that draws
  // the last ticket acquires (invalidates its CU's L1) and reads every partial with sc1 loads.  This is the
  // write-through form of the hand-off MI355X_MICROARCH.md lists as valid on gfx950 ("sc1 slab stores need no release
  // fence -> every storing wave s_waitcnt vmcnt(0) -> one lane's relaxed agent fetch_add; the reducer acquires");
  // it is a property of this ISA, not of the HIP memory model.  -DMI355X_REDUCE_RELEASE=1 builds the model-conformant
  // form (agent-scope release fence before the ticket: an L2 write-back per workgroup, dot 0.044 -> 0.106 ms at n = 2^24).
  if (threadIdx.x < MI355X_WAVE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
#if defined(MI355X_REDUCE_RELEASE) && MI355X_REDUCE_RELEASE
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int last = (t == gridDim.x - 1);
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      is_last = last;
    }
  }
  __syncthreads();
  if (!is_last) return;
  // last-arriving workgroup: sum the per-workgroup partials in workgroup order
#pragma unroll
  for (int j = 0; j < NOUT; ++j) acc[j] = 0.0;
  // (a lane's partials b = t, t + 256, ... are added in that order; up to four workgroups' worth requested together)
  for (unsigned int b0 = threadIdx.x; b0 < gridDim.x; b0 += 4 * MI355X_BLOCK) {
    double v[4][NOUT];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned int b = b0 + u * MI355X_BLOCK;
#pragma unroll
      for (int j = 0; j < NOUT; ++j) v[u][j] = __hip_atomic_load(partials + (size_t)(b < gridDim.x ? b : b0) * NOUT + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (b0 + u * MI355X_BLOCK < gridDim.x) {
#pragma unroll
        for (int j = 0; j < NOUT; ++j) acc[j] = (MODE == RED_MAX) ? nanmax(acc[j], v[u][j]) : acc[j] + v[u][j];
      }
    }
  }
  __syncthreads();
  block_reduce_store<NOUT, MODE>(acc, out, lds);
  if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // a result that stays on the device for the next kernel can be handed to the host as well (same lanes that stored it)
  if (host_copy) { __syncthreads(); if (threadIdx.x < NOUT) host_copy[threadIdx.x] = out[threadIdx.x]; }
  publish_to_host(host_seq, seq);
}

template <int NOUT, int MODE, class F>
static int launch_reduce(mi355x_handle_t h, F f, size_t n, int vec_ok, double *out, bool also_to_host = false) {
  // vectors that fit the caches: at most 512 workgroups, 2 per CU (measured on the CG iteration at n = 2^24, one box, same process
  // order: 256 -> 0.541 ms, 384 -> 0.535, 512 -> 0.523, 768 -> 0.529, 1024 -> 0.534, 2048 -> 0.536; 8192 with one tile each: 0.56),
  // grid-stride; the last one sums <= 512 partials.  Vectors of >= 256 MiB: contiguous runs of tiles for <= 4096 workgroups.  The
  // oracle's device-order emulation (oracle/vecmat_oracle.c dev_reduce) restates both: change together.
  const bool runs = vec_streams(n) != 0;
  int grid;
  if (runs) {
    const size_t ntiles = ((n >> 1) + MI355X_TILE2 - 1) / MI355X_TILE2;
    grid = (int)(ntiles > MI355X_REDUCE_GRID_CAP_BIG ? MI355X_REDUCE_GRID_CAP_BIG : ntiles);
  } else {
    grid = mi355x_grid_for(n, 16);
    if (grid > MI355X_REDUCE_GRID_CAP) grid = MI355X_REDUCE_GRID_CAP;
  }
  // a result that goes to the handle's pinned scratch is followed by a completion number (mi355x_handle_wait_result)
  unsigned long long *hs = nullptr, seq = 0;
  double *host_copy = nullptr;
  if (out >= h->host_scratch && out < h->host_scratch + MI355X_SCRATCH_DOUBLES) {
    hs = const_cast<unsigned long long *>(h->host_seq);
    seq = ++h->seq;
  } else if (also_to_host) {   // result to `out` (device) AND to the first NOUT pinned slots, then the completion number
    hs = const_cast<unsigned long long *>(h->host_seq);
    seq = ++h->seq;
    host_copy = h->host_scratch;
  }
  if (runs) hipLaunchKernelGGL((reduce_kernel<NOUT, MODE, F, true>), dim3(grid), dim3(MI355X_BLOCK), 0, h->stream, f, n, vec_ok,
                               h->partials, h->ticket, out, host_copy, hs, seq);
  else hipLaunchKernelGGL((reduce_kernel<NOUT, MODE, F, false>), dim3(grid), dim3(MI355X_BLOCK), 0, h->stream, f, n, vec_ok,
                          h->partials, h->ticket, out, host_copy, hs, seq);
  MI355X_LAUNCH_CHECK();
  return 0;
}

// default sweep: 4 grid-stride iterations' loads issued together, consumed in order
#define DEFAULT_SWEEP()                                                                              \
  template <int NOUT_>                                                                               \
  __device__ __forceinline__ void sweep(size_t tid, size_t stride, size_t n2, double (&a)[NOUT_]) const { \
    size_t i = tid;                                                                                  \
    for (; i + 3 * stride < n2; i += 4 * stride) accum2x4(i, stride, a);                             \
    for (; i < n2; i += stride) accum2(i, a);                                                        \
  }

struct DotF {
  const double *x, *y;
  int nt = 0;                  // vectors too large for the cache: streamed (vec_streams)
  DEFAULT_SWEEP()
  __device__ void accum2x4(size_t i, size_t st, double (&a)[1]) const {
    const double2 *x2 = reinterpret_cast<const double2 *>(x), *y2 = reinterpret_cast<const double2 *>(y);
    double2 xv0 = ldq(x2 + i, nt), xv1 = ldq(x2 + i + st, nt), xv2 = ldq(x2 + i + 2 * st, nt), xv3 = ldq(x2 + i + 3 * st, nt);
    double2 yv0 = ldq(y2 + i, nt), yv1 = ldq(y2 + i + st, nt), yv2 = ldq(y2 + i + 2 * st, nt), yv3 = ldq(y2 + i + 3 * st, nt);
    a[0] += xv0.x * yv0.x; a[0] += xv0.y * yv0.y; a[0] += xv1.x * yv1.x; a[0] += xv1.y * yv1.y;
    a[0] += xv2.x * yv2.x; a[0] += xv2.y * yv2.y; a[0] += xv3.x * yv3.x; a[0] += xv3.y * yv3.y;
  }
  __device__ void accum1(size_t i, double (&a)[1]) const { a[0] += x[i] * y[i]; }
  __device__ void accum2(size_t i, double (&a)[1]) const {
    double2 xv = reinterpret_cast<const double2 *>(x)[i], yv = reinterpret_cast<const double2 *>(y)[i];
    a[0] += xv.x * yv.x;
    a[0] += xv.y * yv.y;
  }
};
struct SumSqF {
  const double *x;
  int nt = 0;
  DEFAULT_SWEEP()
  __device__ void accum2x4(size_t i, size_t st, double (&a)[1]) const {
    const double2 *x2 = reinterpret_cast<const double2 *>(x);
    double2 v0 = ldq(x2 + i, nt), v1 = ldq(x2 + i + st, nt), v2 = ldq(x2 + i + 2 * st, nt), v3 = ldq(x2 + i + 3 * st, nt);
    a[0] += v0.x * v0.x; a[0] += v0.y * v0.y; a[0] += v1.x * v1.x; a[0] += v1.y * v1.y;
    a[0] += v2.x * v2.x; a[0] += v2.y * v2.y; a[0] += v3.x * v3.x; a[0] += v3.y * v3.y;
  }
  __device__ void accum1(size_t i, double (&a)[1]) const { a[0] += x[i] * x[i]; }
  __device__ void accum2(size_t i, double (&a)[1]) const {
    double2 xv = reinterpret_cast<const double2 *>(x)[i];
    a[0] += xv.x * xv.x;
    a[0] += xv.y * xv.y;
  }
};
struct SumAbsF {
  const double *x;
  DEFAULT_SWEEP()
  __device__ void accum2x4(size_t i, size_t st, double (&a)[1]) const { accum2(i, a); accum2(i + st, a); accum2(i + 2 * st, a); accum2(i + 3 * st, a); }
  __device__ void accum1(size_t i, double (&a)[1]) const { a[0] += fabs(x[i]); }
  __device__ void accum2(size_t i, double (&a)[1]) const {
    double2 xv = reinterpret_cast<const double2 *>(x)[i];
    a[0] += fabs(xv.x);
    a[0] += fabs(xv.y);
  }
};
struct MaxAbsF {
  const double *x;
  DEFAULT_SWEEP()
  __device__ void accum2x4(size_t i, size_t st, double (&a)[1]) const { accum2(i, a); accum2(i + st, a); accum2(i + 2 * st, a); accum2(i + 3 * st, a); }
  __device__ void accum1(size_t i, double (&a)[1]) const { a[0] = nanmax(a[0], fabs(x[i])); }
  __device__ void accum2(size_t i, double (&a)[1]) const {
    double2 xv = reinterpret_cast<const double2 *>(x)[i];
    a[0] = nanmax(a[0], fabs(xv.x));
    a[0] = nanmax(a[0], fabs(xv.y));
  }
};
struct Norm12F {
  const double *x;
  DEFAULT_SWEEP()
  __device__ void accum2x4(size_t i, size_t st, double (&a)[2]) const { accum2(i, a); accum2(i + st, a); accum2(i + 2 * st, a); accum2(i + 3 * st, a); }
  __device__ void accum1(size_t i, double (&a)[2]) const { a[0] += fabs(x[i]); a[1] += x[i] * x[i]; }
  __device__ void accum2(size_t i, double (&a)[2]) const {
    double2 xv = reinterpret_cast<const double2 *>(x)[i];
    a[0] += fabs(xv.x); a[1] += xv.x * xv.x;
    a[0] += fabs(xv.y); a[1] += xv.y * xv.y;
  }
};
struct DotNorm2F {
  const double *s, *t;
  DEFAULT_SWEEP()
  __device__ void accum2x4(size_t i, size_t st, double (&a)[2]) const { accum2(i, a); accum2(i + st, a); accum2(i + 2 * st, a); accum2(i + 3 * st, a); }
  __device__ void accum1(size_t i, double (&a)[2]) const { a[0] += s[i] * t[i]; a[1] += t[i] * t[i]; }
  __device__ void accum2(size_t i, double (&a)[2]) const {
    double2 sv = reinterpret_cast<const double2 *>(s)[i], tv = reinterpret_cast<const double2 *>(t)[i];
    a[0] += sv.x * tv.x; a[1] += tv.x * tv.x;
    a[0] += sv.y * tv.y; a[1] += tv.y * tv.y;
  }
};
// One CG update sweep (KSPSolve_CG cg.c:206-232 with PCApply_Jacobi jacobi.c:266-277 in between):
//   x += a p ; r += (-a) w ; z = r .* d ; out = { sum z*z , sum z*r , sum r*r }
// Element-wise arithmetic is that of the separate kernels (OpAxpy, OpAxpy, OpMul) and each lane meets its elements in
// the same order as SumSqF / DotF do under launch_reduce, so x, r, z and both sums carry the same bits as the five
// separate launches; HBM passes drop from 12 to 8.
struct CGUpdateF {
  double a, ma;
  const double *p, *w, *d;
  double *x, *r, *z;
  int big = 0;                 // vectors too large for the cache: p, w and z are streams as well (vec_streams)
  template <int NOUT_>
  __device__ __forceinline__ void sweep(size_t tid, size_t stride, size_t n2, double (&acc)[NOUT_]) const {
    size_t i = tid;
    for (; i + stride < n2; i += 2 * stride) {
      const double2 *p2 = reinterpret_cast<const double2 *>(p), *w2 = reinterpret_cast<const double2 *>(w);
      const double2 *d2 = reinterpret_cast<const double2 *>(d);
      double2 *x2 = reinterpret_cast<double2 *>(x), *r2 = reinterpret_cast<double2 *>(r), *z2 = reinterpret_cast<double2 *>(z);
      const double2 one2 = {1.0, 1.0};
      double2 pv0 = ldq(p2 + i, big), pv1 = ldq(p2 + i + stride, big), wv0 = ldq(w2 + i, big), wv1 = ldq(w2 + i + stride, big), dv0 = d ? nt_load2(d2 + i) : one2, dv1 = d ? nt_load2(d2 + i + stride) : one2;
      double2 xv0 = nt_load2(x2 + i), xv1 = nt_load2(x2 + i + stride), rv0 = nt_load2(r2 + i), rv1 = nt_load2(r2 + i + stride), zv0, zv1;
      step(pv0.x, wv0.x, dv0.x, xv0.x, rv0.x, zv0.x, acc); step(pv0.y, wv0.y, dv0.y, xv0.y, rv0.y, zv0.y, acc);
      step(pv1.x, wv1.x, dv1.x, xv1.x, rv1.x, zv1.x, acc); step(pv1.y, wv1.y, dv1.y, xv1.y, rv1.y, zv1.y, acc);
      nt_store2(x2 + i, xv0); nt_store2(r2 + i, rv0); stq(z2 + i, zv0, big);
      nt_store2(x2 + i + stride, xv1); nt_store2(r2 + i + stride, rv1); stq(z2 + i + stride, zv1, big);
    }
    for (; i < n2; i += stride) {
      const double2 one2 = {1.0, 1.0};
      double2 pv = reinterpret_cast<const double2 *>(p)[i], wv = reinterpret_cast<const double2 *>(w)[i];
      double2 dv = d ? nt_load2(reinterpret_cast<const double2 *>(d) + i) : one2;
      double2 xv = nt_load2(reinterpret_cast<double2 *>(x) + i), rv = nt_load2(reinterpret_cast<double2 *>(r) + i), zv;
      step(pv.x, wv.x, dv.x, xv.x, rv.x, zv.x, acc); step(pv.y, wv.y, dv.y, xv.y, rv.y, zv.y, acc);
      nt_store2(reinterpret_cast<double2 *>(x) + i, xv); nt_store2(reinterpret_cast<double2 *>(r) + i, rv); reinterpret_cast<double2 *>(z)[i] = zv;
    }
  }
  __device__ __forceinline__ void step(double pv, double wv, double dv, double &xv, double &rv, double &zv, double (&acc)[3]) const {
    xv = xv + a * pv;
    rv = rv + ma * wv;
    zv = rv * dv;
    acc[0] += zv * zv;
    acc[1] += zv * rv;
    acc[2] += rv * rv;         // VecNorm(R) for KSP_NORM_UNPRECONDITIONED (cg.c:246)
  }
  __device__ void accum1(size_t i, double (&acc)[3]) const {
    double xv = x[i], rv = r[i], zv;
    step(p[i], w[i], d ? d[i] : 1.0, xv, rv, zv, acc);
    x[i] = xv; r[i] = rv; z[i] = zv;
  }
};
// The same sweep with the step length computed on the device: a = beta / dpi where dpi = p'w is still in device memory
// (the VecTDot kernel, all-reduced in place over RCCL on several ranks, wrote it there), so the host does not have to
// wait for the dot before it can launch the update -- one host synchronisation per CG iteration instead of two.
// KSPSolve_CG's break-down tests on dpi (cg.c:196-199) are evaluated here too: when one fires nothing is modified, and
// the host, which receives dpi in out[3], takes the reference's exit with x, r, z untouched.  out[3] carries dpi
// through the reduction tree unchanged (lane 0 of workgroup 0 contributes it, every other lane +0.0).
struct CGUpdateDevF {
  double beta, dpiold;
  int check_sign;
  const double *dpi_ptr;
  const double *p, *w, *d;
  double *x, *r, *z;
  int big = 0;                 // see CGUpdateF
  __device__ __forceinline__ bool scalars(double &a) const {
    const double dpi = *dpi_ptr;
    const bool bad = !(dpi == dpi) || fabs(dpi) == __builtin_huge_val() || dpi == 0.0 || (check_sign && dpi * dpiold <= 0.0);
    a = bad ? 0.0 : beta / dpi;
    return !bad;
  }
  __device__ __forceinline__ void prologue(size_t tid, double (&acc)[4]) const {
    if (tid == 0) acc[3] = *dpi_ptr;
  }
  __device__ __forceinline__ void step(double a, double pv, double wv, double dv, double &xv, double &rv, double &zv, double (&acc)[4]) const {
    if (a != 0.0) {            // VecAXPY leaves y alone for alpha == 0 (bvec1.c:253)
      xv = xv + a * pv;
      rv = rv + (-a) * wv;
    }
    zv = rv * dv;
    acc[0] += zv * zv;
    acc[1] += zv * rv;
    acc[2] += rv * rv;
  }
  template <int NOUT_>
  __device__ __forceinline__ void sweep(size_t tid, size_t stride, size_t n2, double (&acc)[NOUT_]) const {
    double a;
    if (!scalars(a)) return;
    const double2 *p2 = reinterpret_cast<const double2 *>(p), *w2 = reinterpret_cast<const double2 *>(w);
    const double2 *d2 = reinterpret_cast<const double2 *>(d);
    const double2 one2 = {1.0, 1.0};
    double2 *x2 = reinterpret_cast<double2 *>(x), *r2 = reinterpret_cast<double2 *>(r), *z2 = reinterpret_cast<double2 *>(z);
    size_t i = tid;
    for (; i + stride < n2; i += 2 * stride) {
      double2 pv0 = ldq(p2 + i, big), pv1 = ldq(p2 + i + stride, big), wv0 = ldq(w2 + i, big), wv1 = ldq(w2 + i + stride, big), dv0 = d ? nt_load2(d2 + i) : one2, dv1 = d ? nt_load2(d2 + i + stride) : one2;
      double2 xv0 = nt_load2(x2 + i), xv1 = nt_load2(x2 + i + stride), rv0 = nt_load2(r2 + i), rv1 = nt_load2(r2 + i + stride), zv0, zv1;
      step(a, pv0.x, wv0.x, dv0.x, xv0.x, rv0.x, zv0.x, acc); step(a, pv0.y, wv0.y, dv0.y, xv0.y, rv0.y, zv0.y, acc);
      step(a, pv1.x, wv1.x, dv1.x, xv1.x, rv1.x, zv1.x, acc); step(a, pv1.y, wv1.y, dv1.y, xv1.y, rv1.y, zv1.y, acc);
      nt_store2(x2 + i, xv0); nt_store2(r2 + i, rv0); stq(z2 + i, zv0, big);
      nt_store2(x2 + i + stride, xv1); nt_store2(r2 + i + stride, rv1); stq(z2 + i + stride, zv1, big);
    }
    for (; i < n2; i += stride) {
      double2 pv = p2[i], wv = w2[i], dv = d ? nt_load2(d2 + i) : one2, xv = nt_load2(x2 + i), rv = nt_load2(r2 + i), zv;
      step(a, pv.x, wv.x, dv.x, xv.x, rv.x, zv.x, acc); step(a, pv.y, wv.y, dv.y, xv.y, rv.y, zv.y, acc);
      nt_store2(x2 + i, xv); nt_store2(r2 + i, rv); z2[i] = zv;
    }
  }
  __device__ void accum1(size_t i, double (&acc)[4]) const {
    double a;
    if (!scalars(a)) return;
    double xv = x[i], rv = r[i], zv;
    step(a, p[i], w[i], d ? d[i] : 1.0, xv, rv, zv, acc);
    x[i] = xv; r[i] = rv; z[i] = zv;
  }
};
template <> struct has_prologue<CGUpdateDevF> { static constexpr bool value = true; };

// ---- fused forms for KSPSolve_BCGS (bcgs.c:43-160); each keeps the element-wise arithmetic of the calls it replaces and
// the per-lane summation order of DotF / DotNorm2F / SumSqF under launch_reduce, so results carry the same bits.
// w = x .* d (PCApply_Jacobi; d == NULL: identity, PCNONE's copy) and Sum w*y (VecDot(w, y)): 4 passes instead of 5.
struct PMultDotF {
  const double *x, *d, *y;
  double *w;
  DEFAULT_SWEEP()
  __device__ void accum2x4(size_t i, size_t st, double (&a)[1]) const { accum2(i, a); accum2(i + st, a); accum2(i + 2 * st, a); accum2(i + 3 * st, a); }
  __device__ void accum1(size_t i, double (&a)[1]) const {
    const double wv = x[i] * (d ? d[i] : 1.0);
    w[i] = wv;
    a[0] += wv * y[i];
  }
  __device__ void accum2(size_t i, double (&a)[1]) const {
    const double2 one2 = {1.0, 1.0};
    double2 xv = reinterpret_cast<const double2 *>(x)[i], dv = d ? reinterpret_cast<const double2 *>(d)[i] : one2;
    double2 yv = reinterpret_cast<const double2 *>(y)[i], wv;
    wv.x = xv.x * dv.x; wv.y = xv.y * dv.y;
    reinterpret_cast<double2 *>(w)[i] = wv;
    a[0] += wv.x * yv.x;
    a[0] += wv.y * yv.y;
  }
};
// w = x .* d and VecDotNorm2(s, w): Sum s*w, Sum w*w
struct PMultDotNorm2F {
  const double *x, *d, *s;
  double *w;
  DEFAULT_SWEEP()
  __device__ void accum2x4(size_t i, size_t st, double (&a)[2]) const { accum2(i, a); accum2(i + st, a); accum2(i + 2 * st, a); accum2(i + 3 * st, a); }
  __device__ void accum1(size_t i, double (&a)[2]) const {
    const double wv = x[i] * (d ? d[i] : 1.0);
    w[i] = wv;
    a[0] += s[i] * wv; a[1] += wv * wv;
  }
  __device__ void accum2(size_t i, double (&a)[2]) const {
    const double2 one2 = {1.0, 1.0};
    double2 xv = reinterpret_cast<const double2 *>(x)[i], dv = d ? reinterpret_cast<const double2 *>(d)[i] : one2;
    double2 sv = reinterpret_cast<const double2 *>(s)[i], wv;
    wv.x = xv.x * dv.x; wv.y = xv.y * dv.y;
    reinterpret_cast<double2 *>(w)[i] = wv;
    a[0] += sv.x * wv.x; a[1] += wv.x * wv.x;
    a[0] += sv.y * wv.y; a[1] += wv.y * wv.y;
  }
};
// x = alpha p + omega s + x (VecAXPBYPCZ, gamma == 1 form, bvec1.c:436), r = s + (-omega) t (VecWAXPY, dvec2.c:1094),
// Sum r*r (VecNorm), Sum r*rp (next iteration's VecDot(R,RP)): 7 passes instead of 10, one reduction instead of two
struct BcgsUpdateF {
  double alpha, omega, momega;
  const double *p, *s, *t, *rp;
  double *x, *r;
  DEFAULT_SWEEP()
  __device__ void accum2x4(size_t i, size_t st, double (&a)[2]) const { accum2(i, a); accum2(i + st, a); accum2(i + 2 * st, a); accum2(i + 3 * st, a); }
  __device__ __forceinline__ void one(double pv, double sv, double tv, double rpv, double &xv, double &rv, double (&a)[2]) const {
    xv = alpha * pv + omega * sv + xv;
    rv = (momega == 0.0) ? sv : sv + momega * tv;      // VecWAXPY copies for alpha == 0
    a[0] += rv * rv;
    a[1] += rv * rpv;
  }
  __device__ void accum1(size_t i, double (&a)[2]) const {
    double xv = x[i], rv;
    one(p[i], s[i], t[i], rp[i], xv, rv, a);
    x[i] = xv; r[i] = rv;
  }
  __device__ void accum2(size_t i, double (&a)[2]) const {
    // x is touched here only, s and t have their last reader here: streaming accesses (see nt_load2)
    double2 pv = reinterpret_cast<const double2 *>(p)[i], sv = nt_load2(reinterpret_cast<const double2 *>(s) + i);
    double2 tv = nt_load2(reinterpret_cast<const double2 *>(t) + i), qv = reinterpret_cast<const double2 *>(rp)[i];
    double2 xv = nt_load2(reinterpret_cast<double2 *>(x) + i), rv;
    one(pv.x, sv.x, tv.x, qv.x, xv.x, rv.x, a);
    one(pv.y, sv.y, tv.y, qv.y, xv.y, rv.y, a);
    nt_store2(reinterpret_cast<double2 *>(x) + i, xv);
    reinterpret_cast<double2 *>(r)[i] = rv;
  }
};

template <int NV>
struct MDotF {
  const double *x;
  const double *y[NV];
  int nt;
  template <int NOUT_>
  __device__ __forceinline__ void sweep(size_t tid, size_t stride, size_t n2, double (&a)[NOUT_]) const {
    size_t i = tid;
    if (NV <= 4) for (; i + stride < n2; i += 2 * stride) { accum2(i, a); accum2(i + stride, a); }
    for (; i < n2; i += stride) accum2(i, a);
  }
  __device__ void accum1(size_t i, double (&a)[NV]) const {
    double xv = x[i];
#pragma unroll
    for (int j = 0; j < NV; ++j) a[j] += xv * y[j][i];
  }
  __device__ void accum2(size_t i, double (&a)[NV]) const {
    double2 xv = reinterpret_cast<const double2 *>(x)[i];
    double2 yv[NV];
    if (nt) {                                      // basis larger than the caches: see gs_streams()
#pragma unroll
      for (int j = 0; j < NV; ++j) yv[j] = nt_load2(reinterpret_cast<const double2 *>(y[j]) + i);
    } else {
#pragma unroll
      for (int j = 0; j < NV; ++j) yv[j] = reinterpret_cast<const double2 *>(y[j])[i];
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      a[j] += xv.x * yv[j].x;
      a[j] += xv.y * yv[j].y;
    }
  }
};


// ---- fused forms for KSPGMRESCycle (gmres.c:118-209) ------------------------------------------------------------------
// The Gram-Schmidt update x += sum_j (-h_j) y_j (VecMAXPY of borthog2.c:64 with the coefficients VecMDot has just left in
// device memory, negated as borthog2.c:63 does) and Sum x_new^2 (the VecNorm inside gmres.c:146's VecNormalize) in ONE
// sweep: the grouping of maxpy_elem (= petscaxpy.h:101-110) and the per-lane order of SumSqF, so x and the norm carry the
// bits of the separate calls.  Up to 32 vectors per sweep.
template <int G0, int NG4>
struct MaxpyNormF {
  static constexpr int NV = G0 + 4 * NG4;
  const double *y[NV];
  int nt;
  const double *adev;   // coefficients in device memory ...
  double sign;          // ... times +-1
  double *x;
  double a[NV];
  __device__ __forceinline__ void prologue(size_t, double (&)[1]) {
#pragma unroll
    for (int j = 0; j < NV; ++j) a[j] = sign * adev[j];   // (-1) * h: the bits of -h
  }
  template <int NOUT_>
  __device__ __forceinline__ void sweep(size_t tid, size_t stride, size_t n2, double (&acc)[NOUT_]) const {
    for (size_t i = tid; i < n2; i += stride) accum2(i, acc);
  }
  __device__ __forceinline__ void accum1(size_t i, double (&acc)[1]) const {
    double v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = y[j][i];
    const double xv = maxpy_elem<G0, NG4>(x[i], a, v);
    x[i] = xv;
    acc[0] += xv * xv;
  }
  __device__ __forceinline__ void accum2(size_t i, double (&acc)[1]) const {
    double2 yv[NV];
    if (nt) {                                      // basis larger than the caches: see gs_streams()
#pragma unroll
      for (int j = 0; j < NV; ++j) yv[j] = nt_load2(reinterpret_cast<const double2 *>(y[j]) + i);
    } else {
#pragma unroll
      for (int j = 0; j < NV; ++j) yv[j] = reinterpret_cast<const double2 *>(y[j])[i];
    }
    double2 xv = reinterpret_cast<double2 *>(x)[i];
    double lo[NV], hi[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) { lo[j] = yv[j].x; hi[j] = yv[j].y; }
    xv.x = maxpy_elem<G0, NG4>(xv.x, a, lo);
    xv.y = maxpy_elem<G0, NG4>(xv.y, a, hi);
    reinterpret_cast<double2 *>(x)[i] = xv;
    acc[0] += xv.x * xv.x;
    acc[0] += xv.y * xv.y;
  }
};
template <int G0, int NG4> struct has_prologue<MaxpyNormF<G0, NG4>> { static constexpr bool value = true; };

template <int G0, int NG4>
static int launch_maxpy_norm(mi355x_handle_t h, size_t n, const double *adev, double sign, const double *const *y, double *x, double *out) {
  MaxpyNormF<G0, NG4> f;
  int vec_ok = mi355x_aligned16(x);
  for (int j = 0; j < G0 + 4 * NG4; ++j) { f.y[j] = y[j]; vec_ok = vec_ok && mi355x_aligned16(y[j]); }
  f.adev = adev; f.sign = sign; f.x = x;
  f.nt = gs_streams(n, G0 + 4 * NG4);
  return launch_reduce<1, RED_SUM>(h, f, n, vec_ok, out);
}

// x *= 1/sqrt(*norm2) with the cases of VecNormalize (rvector.c:308-314: a zero norm leaves x alone, so does a norm of one)
// and of VecScale_Seq (bvec1.c:183: alpha == 0 sets zero); sqrt and the division are IEEE-exact on the device as on the host
__global__ __launch_bounds__(MI355X_BLOCK) void scale_rnorm_dev_kernel(const double *norm2, double *x, size_t n, int vec_ok) {
  const double nrm = sqrt(*norm2);
  if (nrm == 0.0 || nrm == 1.0) return;
  const double alpha = 1.0 / nrm;
  if (alpha == 1.0) return;
  const size_t tid = (size_t)blockIdx.x * MI355X_BLOCK + threadIdx.x;
  if (vec_ok) {                                    // one tile of MI355X_MAP_TILE2 double2's per workgroup (see map_kernel)
    const size_t n2 = n >> 1;
    double2 *x2 = reinterpret_cast<double2 *>(x);
    const size_t stride = MI355X_BLOCK;
    const size_t i = (size_t)blockIdx.x * MI355X_MAP_TILE2 + threadIdx.x;
    if (i + stride < n2) {
      double2 v0 = x2[i], v1 = x2[i + stride];
      v0.x = alpha == 0.0 ? 0.0 : v0.x * alpha; v0.y = alpha == 0.0 ? 0.0 : v0.y * alpha;
      v1.x = alpha == 0.0 ? 0.0 : v1.x * alpha; v1.y = alpha == 0.0 ? 0.0 : v1.y * alpha;
      x2[i] = v0; x2[i + stride] = v1;
    } else if (i < n2) { double2 v = x2[i]; v.x = alpha == 0.0 ? 0.0 : v.x * alpha; v.y = alpha == 0.0 ? 0.0 : v.y * alpha; x2[i] = v; }
    if ((n & 1) && tid == 0) x[n - 1] = alpha == 0.0 ? 0.0 : x[n - 1] * alpha;
  } else {
    const size_t stride = (size_t)gridDim.x * MI355X_BLOCK;
    for (size_t i = tid; i < n; i += stride) x[i] = alpha == 0.0 ? 0.0 : x[i] * alpha;
  }
}

template <int NV>
static int launch_mdot(mi355x_handle_t h, size_t n, const double *x, const double *const *y, double *out) {
  MDotF<NV> f;
  f.x = x;
  f.nt = gs_streams(n, NV);
  int vec_ok = mi355x_aligned16(x);
  for (int j = 0; j < NV; ++j) {
    f.y[j] = y[j];
    vec_ok = vec_ok && mi355x_aligned16(y[j]);
  }
  return launch_reduce<NV, RED_SUM>(h, f, n, vec_ok, out);
}

extern "C" {

int mi355x_vec_set(mi355x_handle_t h, size_t n, double alpha, double *x) {
  return launch_map<0>(h, OpSet{alpha}, nullptr, nullptr, nullptr, x, n);
}
int mi355x_vec_copy(mi355x_handle_t h, size_t n, const double *x, double *y) {
  if (x == y) return 0;
  return launch_map<1>(h, OpCopy{}, x, nullptr, nullptr, y, n);
}
int mi355x_vec_scale(mi355x_handle_t h, size_t n, double alpha, double *x) {
  // VecScale_Seq (bvec1.c:183): alpha==0 -> set 0, alpha==1 -> nothing, else dscal
  if (alpha == 0.0) return mi355x_vec_set(h, n, 0.0, x);
  if (alpha == 1.0) return 0;
  return launch_map<1>(h, OpScale{alpha}, x, nullptr, nullptr, x, n);
}
int mi355x_vec_swap(mi355x_handle_t h, size_t n, double *x, double *y) {
  if (x == y || n == 0) return 0;
  hipLaunchKernelGGL(swap_kernel, dim3(mi355x_grid_for(n, 4)), dim3(MI355X_BLOCK), 0, h->stream, x, y, n);
  MI355X_LAUNCH_CHECK();
  return 0;
}
int mi355x_vec_axpy(mi355x_handle_t h, size_t n, double alpha, const double *x, double *y) {
  if (alpha == 0.0) return 0;  // bvec1.c:253
  return launch_map<2>(h, OpAxpy{alpha}, x, y, nullptr, y, n);
}
int mi355x_vec_aypx(mi355x_handle_t h, size_t n, double alpha, const double *x, double *y) {
  // dvec2.c:980-1009
  if (alpha == 0.0) return mi355x_vec_copy(h, n, x, y);
  if (alpha == 1.0) return mi355x_vec_axpy(h, n, 1.0, x, y);
  if (alpha == -1.0) return launch_map<2>(h, OpXmY{}, x, y, nullptr, y, n);
  return launch_map<2>(h, OpAypx{alpha}, x, y, nullptr, y, n);
}
int mi355x_vec_aypx_dev(mi355x_handle_t h, size_t n, const double *num_dev, double den, const double *x, double *y) {
  if (n == 0) return 0;
  int vec_ok = mi355x_aligned16(x) && mi355x_aligned16(y);
  if (vec_streams(n)) hipLaunchKernelGGL(aypx_dev_kernel<true>, dim3(map_grid(n, vec_ok)), dim3(MI355X_BLOCK), 0, h->stream, num_dev, den, x, y, n, vec_ok);
  else hipLaunchKernelGGL(aypx_dev_kernel<false>, dim3(map_grid(n, vec_ok)), dim3(MI355X_BLOCK), 0, h->stream, num_dev, den, x, y, n, vec_ok);
  MI355X_LAUNCH_CHECK();
  return 0;
}
int mi355x_handle_publish_at(mi355x_handle_t h, const double *src_dev, int count, int dst_offset) {
  if (count < 0 || dst_offset < 0 || dst_offset + count > MI355X_SCRATCH_DOUBLES) return (int)hipErrorInvalidValue;
  const unsigned long long seq = ++h->seq;
  hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(MI355X_WAVE), 0, h->stream, src_dev, h->host_scratch + dst_offset, count,
                     const_cast<unsigned long long *>(h->host_seq), seq);
  MI355X_LAUNCH_CHECK();
  return 0;
}
int mi355x_handle_publish(mi355x_handle_t h, const double *src_dev, int count) { return mi355x_handle_publish_at(h, src_dev, count, 0); }
int mi355x_vec_axpby(mi355x_handle_t h, size_t n, double alpha, double beta, const double *x, double *y) {
  // bvec1.c:329-356
  if (alpha == 0.0) return mi355x_vec_scale(h, n, beta, y);
  if (beta == 1.0) return mi355x_vec_axpy(h, n, alpha, x, y);
  if (alpha == 1.0) return mi355x_vec_aypx(h, n, beta, x, y);
  if (beta == 0.0) return launch_map<1>(h, OpScale{alpha}, x, nullptr, nullptr, y, n);
  return launch_map<2>(h, OpAxpby{alpha, beta}, x, y, nullptr, y, n);
}
int mi355x_vec_waxpy(mi355x_handle_t h, size_t n, double alpha, const double *x, const double *y, double *w) {
  // dvec2.c:1094-1109
  if (alpha == 1.0) return launch_map<2>(h, OpXpY{}, x, y, nullptr, w, n);
  if (alpha == -1.0) return launch_map<2>(h, OpYmX{}, x, y, nullptr, w, n);
  if (alpha == 0.0) return mi355x_vec_copy(h, n, y, w);
  return launch_map<2>(h, OpAxpy{alpha}, x, y, nullptr, w, n);
}
int mi355x_vec_axpbypcz(mi355x_handle_t h, size_t n, double alpha, double beta, double gamma, const double *x,
                        const double *y, double *z) {
  if (alpha == 1.0) return launch_map<3>(h, OpAxpbypczA1{beta, gamma}, x, y, z, z, n);
  if (gamma == 1.0) return launch_map<3>(h, OpAxpbypczG1{alpha, beta}, x, y, z, z, n);
  if (gamma == 0.0) return launch_map<2>(h, OpAxpbypczG0{alpha, beta}, x, y, nullptr, z, n);
  return launch_map<3>(h, OpAxpbypcz{alpha, beta, gamma}, x, y, z, z, n);
}
int mi355x_vec_pointwise_mult(mi355x_handle_t h, size_t n, const double *x, const double *y, double *w) {
  return launch_map<2>(h, OpMul{}, x, y, nullptr, w, n);
}
int mi355x_vec_pointwise_divide(mi355x_handle_t h, size_t n, const double *x, const double *y, double *w) {
  return launch_map<2>(h, OpDiv{}, x, y, nullptr, w, n);
}
int mi355x_vec_reciprocal(mi355x_handle_t h, size_t n, double *x) {
  return launch_map<1>(h, OpRecip{}, x, nullptr, nullptr, x, n);
}
int mi355x_vec_jacobi_invert(mi355x_handle_t h, size_t n, double *d, int *nzero_dev) {
  (void)nzero_dev;
  return launch_map<1>(h, OpJacInv{}, d, nullptr, nullptr, d, n);
}
int mi355x_stream_triad(mi355x_handle_t h, size_t n, double alpha, const double *b, const double *c, double *a) {
  return launch_map<2>(h, OpTriad{alpha}, b, c, nullptr, a, n);
}

int mi355x_vec_maxpy(mi355x_handle_t h, size_t n, int nv, const double *alpha, const double *const *y, double *x) {
  if (nv <= 0 || n == 0) return 0;
  int pos = 0;
  const int rem = nv & 3;
  while (pos < nv) {
    MaxpyArgs args;
    // first group: the remainder nv % 4 if there is one (dvec2.c:853-877 handles it first), else four; then up to
    // seven more groups of four in the same sweep
    const int g0 = (pos == 0 && rem) ? rem : 4;
    int ng4 = (nv - pos - g0) / 4;
    if (ng4 > 7) ng4 = 7;
    const int cnt = g0 + 4 * ng4;
    int vec_ok = mi355x_aligned16(x);
    for (int j = 0; j < 32; ++j) {
      args.y[j] = (j < cnt) ? y[pos + j] : nullptr;
      args.a[j] = (j < cnt) ? alpha[pos + j] : 0.0;
      if (j < cnt) vec_ok = vec_ok && mi355x_aligned16(y[pos + j]);
    }
    args.nt = gs_streams(n, cnt);
    int rc = 0;
#define MAXPY_CASE(G, N4) case (G) * 10 + (N4): rc = launch_maxpy<G, N4>(h, args, x, n, vec_ok); break
#define MAXPY_ROW(G) MAXPY_CASE(G, 0); MAXPY_CASE(G, 1); MAXPY_CASE(G, 2); MAXPY_CASE(G, 3); MAXPY_CASE(G, 4); MAXPY_CASE(G, 5); MAXPY_CASE(G, 6); MAXPY_CASE(G, 7)
    switch (g0 * 10 + ng4) {
      MAXPY_ROW(1); MAXPY_ROW(2); MAXPY_ROW(3); MAXPY_ROW(4);
      default: return (int)hipErrorInvalidValue;
    }
#undef MAXPY_ROW
#undef MAXPY_CASE
    if (rc) return rc;
    pos += cnt;
  }
  return 0;
}

int mi355x_vec_dot(mi355x_handle_t h, size_t n, const double *x, const double *y, double *out) {
  DotF f{x, y};
  f.nt = vec_streams(n);
  return launch_reduce<1, RED_SUM>(h, f, n, mi355x_aligned16(x) && mi355x_aligned16(y), out);
}
int mi355x_vec_norm(mi355x_handle_t h, size_t n, int type, const double *x, double *out) {
  int v = mi355x_aligned16(x);
  switch (type) {
    case 0: return launch_reduce<1, RED_SUM>(h, SumAbsF{x}, n, v, out);
    case 1:
    case 2: return launch_reduce<1, RED_SUM>(h, SumSqF{x, vec_streams(n)}, n, v, out);
    case 3: return launch_reduce<1, RED_MAX>(h, MaxAbsF{x}, n, v, out);
    case 4: return launch_reduce<2, RED_SUM>(h, Norm12F{x}, n, v, out);
    default: return (int)hipErrorInvalidValue;
  }
}
int mi355x_vec_dotnorm2(mi355x_handle_t h, size_t n, const double *s, const double *t, double *out) {
  DotNorm2F f{s, t};
  return launch_reduce<2, RED_SUM>(h, f, n, mi355x_aligned16(s) && mi355x_aligned16(t), out);
}
int mi355x_vec_cg_update(mi355x_handle_t h, size_t n, double a, const double *p, const double *w, const double *d,
                         double *x, double *r, double *z, double *out) {
  CGUpdateF f{a, -a, p, w, d, x, r, z};
  f.big = vec_streams(n);
  int v = mi355x_aligned16(p) && mi355x_aligned16(w) && mi355x_aligned16(d) && mi355x_aligned16(x) && mi355x_aligned16(r) &&
          mi355x_aligned16(z);   /* d == NULL (identity preconditioner) counts as aligned */
  return launch_reduce<3, RED_SUM>(h, f, n, v, out);
}
int mi355x_vec_cg_update_dev(mi355x_handle_t h, size_t n, double beta, const double *dpi_dev, double dpiold, int check_sign,
                             const double *p, const double *w, const double *d, double *x, double *r, double *z, double *out,
                             int also_to_host) {
  CGUpdateDevF f{beta, dpiold, check_sign, dpi_dev, p, w, d, x, r, z};
  f.big = vec_streams(n);
  int v = mi355x_aligned16(p) && mi355x_aligned16(w) && mi355x_aligned16(d) && mi355x_aligned16(x) && mi355x_aligned16(r) &&
          mi355x_aligned16(z);   /* d == NULL (identity preconditioner) counts as aligned */
  return launch_reduce<4, RED_SUM>(h, f, n, v, out, also_to_host != 0);
}
int mi355x_vec_pmult_dot(mi355x_handle_t h, size_t n, const double *x, const double *d, const double *y, double *w, double *out) {
  PMultDotF f{x, d, y, w};
  return launch_reduce<1, RED_SUM>(h, f, n, mi355x_aligned16(x) && mi355x_aligned16(d) && mi355x_aligned16(y) && mi355x_aligned16(w), out);
}
int mi355x_vec_pmult_dotnorm2(mi355x_handle_t h, size_t n, const double *x, const double *d, const double *s, double *w, double *out) {
  PMultDotNorm2F f{x, d, s, w};
  return launch_reduce<2, RED_SUM>(h, f, n, mi355x_aligned16(x) && mi355x_aligned16(d) && mi355x_aligned16(s) && mi355x_aligned16(w), out);
}
int mi355x_vec_bcgs_update(mi355x_handle_t h, size_t n, double alpha, double omega, const double *p, const double *s, const double *t,
                           const double *rp, double *x, double *r, double *out) {
  BcgsUpdateF f{alpha, omega, -omega, p, s, t, rp, x, r};
  int v = mi355x_aligned16(p) && mi355x_aligned16(s) && mi355x_aligned16(t) && mi355x_aligned16(rp) && mi355x_aligned16(x) && mi355x_aligned16(r);
  return launch_reduce<2, RED_SUM>(h, f, n, v, out);
}

// borthog2.c:63-64 + the norm of gmres.c:146: x += sum_j sign * coef_dev[j] * y_j, *out = sum x_new^2 (out: device or the
// handle's pinned scratch); coef_dev must not be written while this runs
int mi355x_vec_maxpy_dev_norm2(mi355x_handle_t h, size_t n, int nv, const double *coef_dev, double sign, const double *const *y,
                               double *x, double *out) {
  if (nv <= 0) return mi355x_vec_norm(h, n, 2, x, out);
  if (n == 0) return launch_reduce<1, RED_SUM>(h, SumSqF{x}, n, 1, out);
  int pos = 0;
  const int rem = nv & 3;
  while (pos < nv) {
    const int g0 = (pos == 0 && rem) ? rem : 4;
    int ng4 = (nv - pos - g0) / 4;
    if (ng4 > 7) ng4 = 7;
    const int cnt = g0 + 4 * ng4;
    const bool last = pos + cnt == nv;
    // an earlier sweep's sum is of no use: it goes to the last device scratch slot
    double *o = last ? out : h->dev_scratch + (MI355X_SCRATCH_DOUBLES - 1);
    int rc = 0;
#define MN_CASE(G, N4) case (G) * 10 + (N4): rc = launch_maxpy_norm<G, N4>(h, n, coef_dev + pos, sign, y + pos, x, o); break
#define MN_ROW(G) MN_CASE(G, 0); MN_CASE(G, 1); MN_CASE(G, 2); MN_CASE(G, 3); MN_CASE(G, 4); MN_CASE(G, 5); MN_CASE(G, 6); MN_CASE(G, 7)
    switch (g0 * 10 + ng4) {
      MN_ROW(1); MN_ROW(2); MN_ROW(3); MN_ROW(4);
      default: return (int)hipErrorInvalidValue;
    }
#undef MN_ROW
#undef MN_CASE
    if (rc) return rc;
    pos += cnt;
  }
  return 0;
}
int mi355x_vec_scale_rnorm_dev(mi355x_handle_t h, size_t n, const double *norm2_dev, double *x) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(scale_rnorm_dev_kernel, dim3(map_grid(n, mi355x_aligned16(x))), dim3(MI355X_BLOCK), 0, h->stream, norm2_dev, x, n, mi355x_aligned16(x));
  MI355X_LAUNCH_CHECK();
  return 0;
}
int mi355x_vec_mdot(mi355x_handle_t h, size_t n, int nv, const double *x, const double *const *y, double *out) {
  int pos = 0;
  static int width = 0;   // vectors per pass over x: MI355X_MDOT_WIDTH (16 or 32)
  if (!width) { const char *e = getenv("MI355X_MDOT_WIDTH"); width = (e && atoi(e) == 32) ? 32 : 16; }
  while (pos < nv) {
    int left = nv - pos;
    int cnt = left > width ? width : left;
    if (width == 16 && left > 16 && left < 24) cnt = (left + 1) / 2;   // two passes of similar width instead of 16 + a few
    int rc;
    switch (cnt) {
#define MDOT_CASE(N) case N: rc = launch_mdot<N>(h, n, x, y + pos, out + pos); break
      MDOT_CASE(1); MDOT_CASE(2); MDOT_CASE(3); MDOT_CASE(4); MDOT_CASE(5); MDOT_CASE(6); MDOT_CASE(7); MDOT_CASE(8);
      MDOT_CASE(9); MDOT_CASE(10); MDOT_CASE(11); MDOT_CASE(12); MDOT_CASE(13); MDOT_CASE(14); MDOT_CASE(15); MDOT_CASE(16);
      MDOT_CASE(17); MDOT_CASE(18); MDOT_CASE(19); MDOT_CASE(20); MDOT_CASE(21); MDOT_CASE(22); MDOT_CASE(23); MDOT_CASE(24);
      MDOT_CASE(25); MDOT_CASE(26); MDOT_CASE(27); MDOT_CASE(28); MDOT_CASE(29); MDOT_CASE(30); MDOT_CASE(31); MDOT_CASE(32);
#undef MDOT_CASE
      default: return (int)hipErrorInvalidValue;
    }
    if (rc) return rc;
    pos += cnt;
  }
  return 0;
}

}  // extern "C"
