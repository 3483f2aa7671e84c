// Probe for a dense form of the column-tiled SpMV's inner loop: a wavefront streams (value, packed index) pairs with full 16-byte /
// 8-byte loads (two consecutive entries per lane), gathers x from an LDS tile and adds the product to the row's accumulator in LDS with
// ds_add_f64 -- consecutive entries mostly share a row (2.4 entries per row and tile on the stand-in), so lanes of one instruction
// collide on the accumulator.  Questions: (1) the rate, against the plain two-stream read of multistream_probe; (2) in which order one
// instruction's colliding lanes are added (the sum is reproducible only if that order is fixed, and a host restatement needs to know it).
//   hipcc -O3 --offload-arch=gfx950 dense_atomic_probe.hip -o dense_atomic_probe && ./dense_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void lds_add(double *p, double v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// MODE 0: loads only; 1: + x gather from LDS; 2: + ds_add_f64 into the row accumulators; 3: as 2 but x gathered from GLOBAL memory
// (a window of 2^18 doubles = 2 MiB that all workgroups share: the long-range entries of the product, one L2 request per entry)
template <int MODE, int U, int W>
__global__ __launch_bounds__(W * 64) void dense_kernel(const v2d *__restrict__ val, const v2u *__restrict__ idx, long per2, int nblocks, double *out, const double *__restrict__ xg = nullptr) {
  extern __shared__ double lds[];
  double *xt = lds, *acc = lds + 8192;                       // two 4096-column tiles, then 2048 + 1 accumulators
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8192 + 2049; i += W * 64) lds[i] = 1.0 + i;
  __syncthreads();
  const long base = ((long)blockIdx.x * W + w) * per2;       // in pairs of entries
  double s0 = 0.0;
  for (int b = 0; b < nblocks; b += U) {
    v2d v[U]; v2u ix[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u] = __builtin_nontemporal_load(val + base + (long)(b + u) * 64 + lane);
      ix[u] = __builtin_nontemporal_load(idx + base + (long)(b + u) * 64 + lane);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (MODE == 0) { s0 += v[u].x + v[u].y + (double)(ix[u].x ^ ix[u].y); continue; }
      if (MODE == 3) continue;
      const double x0 = xt[ix[u].x & 0xffffu], x1 = xt[ix[u].y & 0xffffu];
      const double p0 = v[u].x * x0, p1 = v[u].y * x1;
      if (MODE == 1) { s0 += p0 + p1; continue; }
      lds_add(acc + (ix[u].x >> 16), p0);
      lds_add(acc + (ix[u].y >> 16), p1);
    }
    if (MODE == 3) {                                     // the gathers of all U blocks go out together, then the products
      double g0[U], g1[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        // a pseudo-random column of the window from the word (the probe's words hold 12-bit columns: spread them over 2^18)
        const unsigned int c0 = (ix[u].x * 2654435761u) >> 14, c1 = (ix[u].y * 2246822519u) >> 14;
        g0[u] = xg[c0]; g1[u] = xg[c1];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        lds_add(acc + (ix[u].x >> 16), v[u].x * g0[u]);
        lds_add(acc + (ix[u].y >> 16), v[u].y * g1[u]);
      }
    }
  }
  __syncthreads();
  if (MODE == 2) s0 = acc[threadIdx.x];
  if (s0 == 1.2345e300) out[threadIdx.x] = s0;
}

static const double *g_xg = nullptr;
template <int MODE, int U, int W>
static void run(const char *name, const v2d *val, const v2u *idx, long nent, double *out, int G) {
  const size_t ldsb = 80 * 1024;
  const long per2 = nent / 2 / ((long)G * W);
  const int nblocks = (int)(per2 / 64) / U * U;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(dense_kernel<MODE, U, W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((dense_kernel<MODE, U, W>), dim3(G), dim3(W * 64), ldsb, 0, val, idx, per2, nblocks, out, g_xg);
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((dense_kernel<MODE, U, W>), dim3(G), dim3(W * 64), ldsb, 0, val, idx, per2, nblocks, out, g_xg);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double entries = (double)G * W * nblocks * 128.0;
  printf("%-64s G=%4d W=%d  %7.3f ms  %6.2f TB/s of 12-byte entries (%.1f M entries)\n", name, G, W, ms, entries * 12.0 / ms * 1e-9, entries * 1e-6);
  fflush(stdout);
}

// one instruction, all 64 lanes on one accumulator: which order?  acc = 0; lane l adds 2^(60 - l) ... the low bits that survive tell.
__global__ void order_kernel(double *out) {
  __shared__ double a[4];
  const int lane = threadIdx.x;
  if (lane == 0) { a[0] = 0.0; a[1] = 0.0; }
  __syncthreads();
  // lane 0 adds 1e16, lanes 1..63 add 1.0 each: ascending order -> every 1.0 is lost (1e16 + 1 rounds to 1e16), descending -> 63 survives
  lds_add(&a[0], lane == 0 ? 1.0e16 : 1.0);
  // lane 63 adds 1e16, others 1.0: ascending -> 63 is added first and survives (1e16 + 63 = exactly representable? 1e16 has ulp 2: 63 -> 64)
  lds_add(&a[1], lane == 63 ? 1.0e16 : 1.0);
  __syncthreads();
  if (lane == 0) { out[0] = a[0] - 1.0e16; out[1] = a[1] - 1.0e16; }
}

int main() {
  const long nent = 96L << 20;                                 // 100 M entries: 805 MB of values + 403 MB of packed indices
  std::vector<unsigned int> hidx((size_t)nent);
  unsigned long long s = 88172645463325252ULL;
  double rowf = 0.0;
  for (long e = 0; e < nent; ++e) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const unsigned int lcol = (unsigned int)(s % 4096);
    rowf += 1.0 / 2.4;                                         // 2.4 consecutive entries per row
    const unsigned int lrow = (unsigned int)((long)rowf % 2048);
    hidx[(size_t)e] = (lrow << 16) | lcol;
  }
  double *val, *out; unsigned int *idx;
  CK(hipMalloc(&val, nent * 8 + 4096)); CK(hipMalloc(&idx, nent * 4 + 4096)); CK(hipMalloc(&out, 8192));
  CK(hipMemset(val, 0, nent * 8 + 4096));
  CK(hipMemcpy(idx, hidx.data(), nent * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(order_kernel, dim3(1), dim3(64), 0, 0, out);
  double ho[2]; CK(hipMemcpy(ho, out, 16, hipMemcpyDeviceToHost));
  printf("one ds_add_f64 instruction, 64 lanes on one accumulator: big addend in lane 0 leaves %.0f of 63 ones, big addend in lane 63 leaves %.0f\n", ho[0], ho[1]);
  printf("  (ascending lane order: 0 and 64 [63 rounded to the ulp of 2]; descending: 64 and 0)\n");
  const v2d *v = (const v2d *)val; const v2u *ix = (const v2u *)idx;
  run<0, 2, 8>("loads only, 2 blocks of 128 entries in flight", v, ix, nent, out, 512);
  run<0, 4, 8>("loads only, 4 blocks in flight", v, ix, nent, out, 512);
  run<1, 4, 8>("+ x gathered from LDS, 4 blocks in flight", v, ix, nent, out, 512);
  run<2, 2, 8>("+ ds_add_f64 into the row sums, 2 blocks in flight", v, ix, nent, out, 512);
  run<2, 4, 8>("+ ds_add_f64 into the row sums, 4 blocks in flight", v, ix, nent, out, 512);
  run<2, 4, 4>("+ ds_add_f64, 4 blocks in flight, 4 wavefronts per workgroup", v, ix, nent, out, 512);
  run<2, 8, 4>("+ ds_add_f64, 8 blocks in flight, 4 wavefronts per workgroup", v, ix, nent, out, 512);
  {
    double *xg; CK(hipMalloc(&xg, 8u << 18)); CK(hipMemset(xg, 0, 8u << 18)); g_xg = xg;
    printf("-- x gathered from a 2 MiB window in global memory instead of LDS (one L2 request per entry), 20 M entries' worth would be 1/5 of these times\n");
    run<3, 2, 8>("global gathers + ds_add_f64, 2 blocks in flight", v, ix, nent, out, 256);
    run<3, 4, 8>("global gathers + ds_add_f64, 4 blocks in flight", v, ix, nent, out, 256);
    run<3, 4, 8>("global gathers + ds_add_f64, 4 blocks in flight", v, ix, nent, out, 512);
    run<3, 8, 8>("global gathers + ds_add_f64, 8 blocks in flight", v, ix, nent, out, 256);
    run<3, 4, 15>("global gathers + ds_add_f64, 4 blocks in flight, 15 wavefronts", v, ix, nent, out, 256);
  }
  return 0;
}
