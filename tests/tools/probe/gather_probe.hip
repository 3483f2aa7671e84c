// Probe: what does one batch of gathers of the sync-free triangular solves cost?  One wavefront per workgroup issues B loads of 8
// bytes per lane at scattered addresses inside a window of the solution vector, waits for them, and repeats (every round's
// addresses depend on the previous round's data, as a solve's do not -- this is the pure round trip).  Variants: the cache-scope
// bits of the loads (plain / sc0 / sc1 / sc0 sc1), the window (recent levels: 64 KB; the whole 12 MB vector), workgroups in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int SC> __device__ __forceinline__ double ldx(const double *p) {
  if (SC == 0) return *(const volatile double *)p;
  if (SC == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (SC == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int SC, int B>
__global__ void gather(const double *w, const int *idx, int rounds, long window, long n, long long *out) {
  const int lane = threadIdx.x;
  long base = ((long)blockIdx.x * 7919 * 64) % (n - window);
  const int *ix = idx + (size_t)blockIdx.x * rounds * B * 64;
  double acc = 0.0;
  long long t = 0;
  for (int r = 0; r < rounds; ++r) {
    int c[B];
#pragma unroll
    for (int j = 0; j < B; ++j) c[j] = ix[(size_t)(r * B + j) * 64 + lane];      // (reduced to the window on the host)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t0 = wall_clock64();
    double v[B];
#pragma unroll
    for (int j = 0; j < B; ++j) v[j] = ldx<SC>(w + base + c[j]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t += wall_clock64() - t0;
#pragma unroll
    for (int j = 0; j < B; ++j) acc += v[j];
    base = (base + 4096 + ((long)acc & 1));
    if (base >= n - window) base -= n - window;
  }
  if (lane == 0) { out[blockIdx.x * 2] = t; out[blockIdx.x * 2 + 1] = (long long)acc; }
}

template <int SC, int B>
static void run(const char *name, int wgs, long window, const double *w, const int *idx, long n, long long *out) {
  const int rounds = 200;
  hipLaunchKernelGGL((gather<SC, B>), dim3(wgs), dim3(64), 0, 0, w, idx, rounds, window, n, out);
  hipDeviceSynchronize();
  std::vector<long long> h(2 * wgs);
  hipMemcpy(h.data(), out, sizeof(long long) * 2 * wgs, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < wgs; ++i) s += (double)h[2 * i];
  printf("%-12s B=%d  workgroups %4d  window %8ld doubles: %7.0f ns per batch\n", name, B, wgs, window, s / wgs * 10.0 / rounds);
}

int main() {
  const long n = 1600000;
  double *w; int *idx; long long *out;
  hipMalloc(&w, sizeof(double) * n); hipMalloc(&out, sizeof(long long) * 4096);
  std::vector<double> hw(n, 1.0); hipMemcpy(w, hw.data(), sizeof(double) * n, hipMemcpyHostToDevice);
  const size_t ni = (size_t)256 * 200 * 8 * 64;
  std::vector<int> hi(ni); unsigned x = 12345; for (size_t i = 0; i < ni; ++i) { x = x * 1664525u + 1013904223u; hi[i] = (int)(x >> 4); }
  hipMalloc(&idx, sizeof(int) * ni); hipMemcpy(idx, hi.data(), sizeof(int) * ni, hipMemcpyHostToDevice);
  for (long window : {8192L, 1500000L}) for (int wgs : {1, 16, 64, 256}) {
    if (wgs == 1) { std::vector<int> hr(ni); for (size_t i = 0; i < ni; ++i) hr[i] = (int)(hi[i] % window); hipMemcpy(idx, hr.data(), sizeof(int) * ni, hipMemcpyHostToDevice); }
    run<0, 8>("plain", wgs, window, w, idx, n, out);
    run<1, 8>("sc0", wgs, window, w, idx, n, out);
    run<2, 8>("sc1", wgs, window, w, idx, n, out);
    run<3, 8>("sc0 sc1", wgs, window, w, idx, n, out);
  }
  { std::vector<int> hr(ni); for (size_t i = 0; i < ni; ++i) hr[i] = (int)(hi[i] % 8192); hipMemcpy(idx, hr.data(), sizeof(int) * ni, hipMemcpyHostToDevice); }
  for (int wgs : {16}) { run<2, 1>("sc1", wgs, 8192, w, idx, n, out); run<2, 2>("sc1", wgs, 8192, w, idx, n, out); run<2, 4>("sc1", wgs, 8192, w, idx, n, out); }
  return 0;
}
