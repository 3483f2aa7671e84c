// Probe: what does the memory system give MANY CONCURRENT SEQUENTIAL STREAMS of small steps -- the access pattern of the column-tiled
// SpMV (csrc/spmv_tiled.hip): 512 resident workgroups of 8 wavefronts, every wavefront walking its own stream of 8-byte values and
// 2-byte column numbers in steps of <= 64 entries?  Variants: where a wavefront's steps lie (its own contiguous chunk / interleaved
// with the other wavefronts of its workgroup so that the workgroup reads 4 KB bursts / one flat front over the whole chip), steps of
// 64 entries (aligned) or 53 (the stand-in's average: every step straddles lines), with and without the 2-byte stream, loads in
// flight per wavefront.
//   hipcc -O3 --offload-arch=gfx950 multistream_probe.hip -o multistream_probe && ./multistream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MODE 0: wavefront (b, w) owns entries [ (b*W + w) * per, ... + per ), step s at + s*ADV
// MODE 1: workgroup b owns [b*W*per, ...), wavefront w's step s at + (s*W + w)*ADV   (the workgroup's wavefronts read one 8-step burst side by side)
// MODE 2: flat front: step s of wavefront (b, w) at ((s * G + b) * W + w) * ADV       (the whole chip reads one moving window)
template <int MODE, int ADV, int LCOL, int U, int W>
__global__ __launch_bounds__(W * 64) void ms_kernel(const double *__restrict__ val, const unsigned short *__restrict__ lcol, long per, int nsteps, double *out) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long b = blockIdx.x, G = gridDim.x;
  double s0 = 0.0;
  unsigned int c0 = 0;
  for (int s = 0; s < nsteps; s += U) {
    double v[U]; unsigned short c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long e;
      if (MODE == 0) e = (b * W + w) * per + (long)(s + u) * ADV;
      else if (MODE == 1) e = b * W * per + ((long)(s + u) * W + w) * ADV;
      else e = (((long)(s + u) * G + b) * W + w) * ADV;
      v[u] = __builtin_nontemporal_load(val + e + lane);
      if (LCOL) c[u] = __builtin_nontemporal_load(lcol + e + lane);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { s0 += v[u]; if (LCOL) c0 += c[u]; }
  }
  if (s0 == 1.2345e300 || c0 == 0xdeadbeefu) out[threadIdx.x] = s0 + lds[lane];
}

template <int MODE, int ADV, int LCOL, int U>
static void run(const char *name, const double *val, const unsigned short *lcol, long nent, double *out, int G, size_t ldsb) {
  constexpr int W = 8;
  const long per = nent / ((long)G * W);                       // entries per wavefront
  const int nsteps = (int)(per / 64) / U * U;                  // (ADV <= 64: stays inside the chunk)
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(ms_kernel<MODE, ADV, LCOL, U, W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((ms_kernel<MODE, ADV, LCOL, U, W>), dim3(G), dim3(W * 64), ldsb, 0, val, lcol, per, nsteps, out);
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((ms_kernel<MODE, ADV, LCOL, U, W>), dim3(G), dim3(W * 64), ldsb, 0, val, lcol, per, nsteps, out);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  const double bytes = (double)G * W * nsteps * ADV * (8.0 + (LCOL ? 2.0 : 0.0));
  printf("%-58s G=%5d  %7.3f ms  %6.2f TB/s (useful bytes)\n", name, G, ms, bytes / ms * 1e-9);
  fflush(stdout);
}

int main() {
  const long nent = 100L << 20;                                // 105 M entries: 839 MB of values + 210 MB of column numbers
  double *val, *out; unsigned short *lcol;
  CK(hipMalloc(&val, nent * 8 + 4096)); CK(hipMalloc(&lcol, nent * 2 + 4096)); CK(hipMalloc(&out, 8192));
  CK(hipMemset(val, 0, nent * 8 + 4096)); CK(hipMemset(lcol, 0, nent * 2 + 4096));
  const size_t big = 80 * 1024;                                // 80 KB of LDS: 2 workgroups per CU, as the product kernel
  printf("-- 512 resident workgroups x 8 wavefronts (80 KB LDS each), every wavefront its own stream\n");
  run<0, 64, 1, 4>("own chunk, steps of 64, 4 steps in flight", val, lcol, nent, out, 512, big);
  run<0, 64, 1, 8>("own chunk, steps of 64, 8 steps in flight", val, lcol, nent, out, 512, big);
  run<0, 64, 1, 16>("own chunk, steps of 64, 16 steps in flight", val, lcol, nent, out, 512, big);
  run<0, 53, 1, 8>("own chunk, steps of 53 (unaligned), 8 in flight", val, lcol, nent, out, 512, big);
  run<0, 64, 0, 8>("own chunk, steps of 64, values only, 8 in flight", val, lcol, nent, out, 512, big);
  run<0, 53, 0, 8>("own chunk, steps of 53, values only, 8 in flight", val, lcol, nent, out, 512, big);
  run<1, 64, 1, 8>("interleaved in the workgroup, steps of 64, 8 in flight", val, lcol, nent, out, 512, big);
  run<1, 53, 1, 8>("interleaved in the workgroup, steps of 53, 8 in flight", val, lcol, nent, out, 512, big);
  run<2, 64, 1, 8>("one front over the chip, steps of 64, 8 in flight", val, lcol, nent, out, 512, big);
  run<2, 53, 1, 8>("one front over the chip, steps of 53, 8 in flight", val, lcol, nent, out, 512, big);
  printf("-- the same with 1024 workgroups (two rounds of the slots)\n");
  run<0, 64, 1, 8>("own chunk, steps of 64, 8 in flight", val, lcol, nent, out, 1024, big);
  run<0, 53, 1, 8>("own chunk, steps of 53, 8 in flight", val, lcol, nent, out, 1024, big);
  printf("-- no LDS: as many workgroups per CU as wavefront slots allow\n");
  run<0, 64, 1, 8>("own chunk, steps of 64, 8 in flight", val, lcol, nent, out, 1024, 0);
  run<0, 53, 1, 8>("own chunk, steps of 53, 8 in flight", val, lcol, nent, out, 1024, 0);
  run<2, 64, 1, 8>("one front over the chip, steps of 64, 8 in flight", val, lcol, nent, out, 1024, 0);
  run<0, 64, 1, 8>("own chunk, steps of 64, 8 in flight", val, lcol, nent, out, 2048, 0);
  run<0, 64, 1, 8>("own chunk, steps of 64, 8 in flight", val, lcol, nent, out, 8192, 0);
  run<0, 53, 1, 8>("own chunk, steps of 53, 8 in flight", val, lcol, nent, out, 8192, 0);
  return 0;
}
