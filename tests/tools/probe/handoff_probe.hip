// Probe: the latency of one producer -> consumer hand-off through a polled memory slot, between workgroups of one XCD and between
// workgroups of different XCDs, for the cache-scope bits a load / store can carry (gfx950: sc0 = workgroup, sc1 = agent).
// A chain of N links: link k polls slot[k-1] and then stores slot[k]; the links are dealt round-robin to the participating
// workgroups (one lane each), so that every hand-off crosses workgroups.  Prints ns per link.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define SENT 0x7ff8dead7ff8deadULL

template <int LS, int SS> __device__ __forceinline__ unsigned long long ld(const unsigned long long *p) {
  if (LS == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (LS == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <int LS, int SS> __device__ __forceinline__ void st(unsigned long long *p, unsigned long long v) {
  if (SS == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else if (SS == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// xcd_mode: -1 every workgroup takes part; >= 0: only the workgroups of the XCD whose id the first arriving workgroup carries
template <int LS, int SS>
__global__ void chain(unsigned long long *slot, int n, int stride, unsigned int *ctl, int one_xcd, int want, int *xcc_seen) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 15u;
  if (threadIdx.x != 0) return;
  if (one_xcd) {
    unsigned int lead = atomicCAS(ctl + 0, 0xffffffffu, xcc);
    if (lead != 0xffffffffu && lead != xcc) return;
  }
  const unsigned int r = atomicAdd(ctl + 32, 1u);
  if ((int)r >= want) return;
  xcc_seen[r] = (int)xcc;
  // wait until all `want` participants have arrived (they are all resident: want <= CUs)
  long spins = 0;
  while (__hip_atomic_load(ctl + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)want && ++spins < 20000000) __builtin_amdgcn_s_sleep(1);
  for (int k = (int)r; k < n; k += want) {
    unsigned long long v = 0;
    if (k > 0) {
      long sp = 0;
      while ((v = ld<LS, SS>(slot + (size_t)(k - 1) * stride)) == SENT) { if (++sp > 4000000) { ctl[64] = 1; return; } }
    }
    st<LS, SS>(slot + (size_t)k * stride, v + 1);
  }
}

template <int LS, int SS>
static void run(const char *name, int one_xcd, int want, int n, int stride) {
  unsigned long long *slot; unsigned int *ctl; int *seen;
  hipMalloc(&slot, sizeof(unsigned long long) * (size_t)n * stride);
  hipMalloc(&ctl, 4 * 128); hipMalloc(&seen, 4 * 1024);
  std::vector<unsigned long long> h((size_t)n * stride, SENT);
  float best = 1e30f; unsigned long long last = 0; unsigned int ab = 0; int hs[1024];
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(slot, h.data(), sizeof(unsigned long long) * h.size(), hipMemcpyHostToDevice);
    unsigned int c[128]; for (int i = 0; i < 128; ++i) c[i] = 0; c[0] = 0xffffffffu;
    hipMemcpy(ctl, c, sizeof(c), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((chain<LS, SS>), dim3(one_xcd ? 16 * want : want), dim3(64), 0, 0, slot, n, stride, ctl, one_xcd, want, seen);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    hipMemcpy(&last, slot + (size_t)(n - 1) * stride, 8, hipMemcpyDeviceToHost);
    hipMemcpy(c, ctl, sizeof(c), hipMemcpyDeviceToHost); ab = c[64];
    hipMemcpy(hs, seen, 4 * (want < 1024 ? want : 1024), hipMemcpyDeviceToHost);
  }
  int nx = 0, seenx[16] = {0}; for (int i = 0; i < want && i < 1024; ++i) if (hs[i] >= 0 && hs[i] < 16 && !seenx[hs[i]]++) nx++;
  printf("%-34s workgroups %3d on %d XCD(s): %8.1f ns per hand-off   (chain %s%s)\n", name, want, nx, best * 1e6 / n,
         last == (unsigned long long)n ? "complete" : "INCOMPLETE", ab ? ", gave up" : "");
  hipFree(slot); hipFree(ctl); hipFree(seen);
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 20000;
  for (int stride : {1, 16}) {
    printf("slot stride %d doubles\n", stride);
    for (int want : {2, 8, 32}) {
      run<1, 1>("all XCDs, load sc1 store sc1", 0, want, n, stride);
      run<1, 1>("one XCD,  load sc1 store sc1", 1, want, n, stride);
      run<1, 0>("one XCD,  load sc1 store sc0", 1, want, n, stride);
      run<0, 0>("one XCD,  load sc0 store sc0", 1, want, n, stride);
      run<0, 1>("one XCD,  load sc0 store sc1", 1, want, n, stride);
      run<2, 2>("all XCDs, load sys store sys", 0, want, n, stride);
    }
  }
  return 0;
}
