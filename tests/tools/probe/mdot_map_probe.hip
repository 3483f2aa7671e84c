// Which element -> lane mapping streams 17 (MDot) / 18 (MAXPY) vectors fastest?  Development probe, not part of the library.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off mdot_map_probe.hip -o mdot_map_probe && ./mdot_map_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int NV = 16;
struct Args { const double *y[NV]; };

// MAP 0: grid-stride over double2 (current library mapping); 1: each workgroup owns a contiguous slab, walks it 256 double2 at a time;
// 2: grid-stride, two adjacent double2 per lane (32 B per lane, 2 KB per wave and stream); 3: slab per WAVE (64 double2 at a time)
template <int MAP>
__global__ __launch_bounds__(256) void mdot(Args a, const double *x, size_t n2, double *out) {
  double acc[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) acc[j] = 0.0;
  const double2 *x2 = reinterpret_cast<const double2 *>(x);
  auto body = [&](size_t i) {
    double2 xv = x2[i], yv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) yv[j] = reinterpret_cast<const double2 *>(a.y[j])[i];
#pragma unroll
    for (int j = 0; j < NV; ++j) { acc[j] += xv.x * yv[j].x; acc[j] += xv.y * yv[j].y; }
  };
  if (MAP == 0) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) body(i);
  } else if (MAP == 1) {
    const size_t per = (n2 + gridDim.x - 1) / gridDim.x;
    const size_t b = per * blockIdx.x, e = (b + per < n2) ? b + per : n2;
    for (size_t i = b + threadIdx.x; i < e; i += 256) body(i);
  } else if (MAP == 2) {
    const size_t stride = (size_t)gridDim.x * 512;
    for (size_t i = (size_t)blockIdx.x * 512 + 2 * threadIdx.x; i + 1 < n2; i += stride) { body(i); body(i + 1); }
  } else {
    const size_t nw = (size_t)gridDim.x * 4, w = (size_t)blockIdx.x * 4 + threadIdx.x / 64;
    const size_t per = (n2 + nw - 1) / nw;
    const size_t b = per * w, e = (b + per < n2) ? b + per : n2;
    for (size_t i = b + (threadIdx.x & 63); i < e; i += 64) body(i);
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < NV; ++j) s += acc[j];
  if (s == 1.2345) out[0] = s;
}

template <int MAP>
__global__ __launch_bounds__(256) void maxpy(Args a, double *x, size_t n2, double c) {
  double2 *x2 = reinterpret_cast<double2 *>(x);
  auto body = [&](size_t i) {
    double2 xv = x2[i], yv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) yv[j] = reinterpret_cast<const double2 *>(a.y[j])[i];
#pragma unroll
    for (int j = 0; j < NV; ++j) { xv.x += c * yv[j].x; xv.y += c * yv[j].y; }
    x2[i] = xv;
  };
  if (MAP == 0) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) body(i);
  } else if (MAP == 1) {
    const size_t per = (n2 + gridDim.x - 1) / gridDim.x;
    const size_t b = per * blockIdx.x, e = (b + per < n2) ? b + per : n2;
    for (size_t i = b + threadIdx.x; i < e; i += 256) body(i);
  } else if (MAP == 2) {
    const size_t stride = (size_t)gridDim.x * 512;
    for (size_t i = (size_t)blockIdx.x * 512 + 2 * threadIdx.x; i + 1 < n2; i += stride) { body(i); body(i + 1); }
  } else {
    const size_t nw = (size_t)gridDim.x * 4, w = (size_t)blockIdx.x * 4 + threadIdx.x / 64;
    const size_t per = (n2 + nw - 1) / nw;
    const size_t b = per * w, e = (b + per < n2) ? b + per : n2;
    for (size_t i = b + (threadIdx.x & 63); i < e; i += 64) body(i);
  }
}

// ST 0: plain in-place store; 1: nontemporal store; 2: out of place (z != x); 3: in place, two iterations' loads before the stores
template <int ST>
__global__ __launch_bounds__(256) void maxpy_st(Args a, double *__restrict__ x, double *__restrict__ z, size_t n2, double c) {
  const size_t stride = (size_t)gridDim.x * 256;
  auto comp = [&](size_t i, double2 xv) {
    double2 yv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) yv[j] = reinterpret_cast<const double2 *>(a.y[j])[i];
#pragma unroll
    for (int j = 0; j < NV; ++j) { xv.x += c * yv[j].x; xv.y += c * yv[j].y; }
    return xv;
  };
  double2 *x2 = reinterpret_cast<double2 *>(x), *z2 = reinterpret_cast<double2 *>(z);
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (ST == 3) {
    for (; i + stride < n2; i += 2 * stride) {
      double2 r0 = comp(i, x2[i]), r1 = comp(i + stride, x2[i + stride]);
      x2[i] = r0; x2[i + stride] = r1;
    }
  }
  for (; i < n2; i += stride) {
    double2 r = comp(i, x2[i]);
    if (ST == 1) { __builtin_nontemporal_store(r.x, &x[2 * i]); __builtin_nontemporal_store(r.y, &x[2 * i + 1]); }
    else if (ST == 2) z2[i] = r;
    else x2[i] = r;
  }
}
template <int ST> float run_st(int grid, Args a, double *x, double *z, size_t n2) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 20;
  for (int r = 0; r < 3 + reps; ++r) {
    if (r == 3) CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(maxpy_st<ST>, dim3(grid), dim3(256), 0, 0, a, x, z, n2, 1e-9);
  }
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}
__global__ void fill(double *p, size_t n) { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 1e-3 * (double)((i * 2654435761u) % 1000003) - 300.0; }

template <int MAP> float run(int which, int grid, Args a, double *x, size_t n2, double *out) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 20;
  for (int r = 0; r < 3 + reps; ++r) {
    if (r == 3) CK(hipEventRecord(e0, 0));
    if (which == 0) hipLaunchKernelGGL(mdot<MAP>, dim3(grid), dim3(256), 0, 0, a, x, n2, out);
    else hipLaunchKernelGGL(maxpy<MAP>, dim3(grid), dim3(256), 0, 0, a, x, n2, 1e-9);
  }
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  const size_t n = (size_t)1 << 24, n2 = n / 2;
  for (size_t skew : {(size_t)0}) {
    const size_t stride = n * 8 + skew;
    char *base; CK(hipMalloc((void **)&base, stride * (NV + 1) + 64));
    CK(hipMemset(base, 0, stride * (NV + 1)));
    Args a; for (int j = 0; j < NV; ++j) a.y[j] = (const double *)(base + stride * j);
    double *x = (double *)(base + stride * NV), *out; CK(hipMalloc((void **)&out, 64));
    for (int which = 0; which < 2; ++which) {
      const double gb = (which == 0 ? (NV + 1) : (NV + 2)) * 8.0 * n / 1e9;
      for (int grid : {1024}) {
        float t0 = run<0>(which, grid, a, x, n2, out), t1 = run<1>(which, grid, a, x, n2, out), t2 = run<2>(which, grid, a, x, n2, out), t3 = run<3>(which, grid, a, x, n2, out);
        printf("skew %6zu %s grid %4d: stride %.4f ms %.0f GB/s | wg-slab %.4f ms %.0f GB/s | stride32B %.4f ms %.0f GB/s | wave-slab %.4f ms %.0f GB/s\n", skew,
               which ? "maxpy16" : "mdot16 ", grid, t0, gb / t0 * 1e3, t1, gb / t1 * 1e3, t2, gb / t2 * 1e3, t3, gb / t3 * 1e3);
        fflush(stdout);
      }
    }
    if (skew == 0) {
      double *z; CK(hipMalloc((void **)&z, n * 8));
      const double gb = (NV + 2) * 8.0 * n / 1e9;
      for (int rnd = 0; rnd < 2; ++rnd) {
        if (rnd) { hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (double *)base, stride * (NV + 1) / 8); CK(hipDeviceSynchronize()); }
        for (int grid : {1024, 2048}) {
          float t0 = run_st<0>(grid, a, x, z, n2), t1 = run_st<1>(grid, a, x, z, n2), t2 = run_st<2>(grid, a, x, z, n2), t3 = run_st<3>(grid, a, x, z, n2);
          float tm = run<0>(0, grid, a, x, n2, out);
          printf("%s grid %4d maxpy16: plain %.4f ms %.0f | nontemporal %.4f ms %.0f | out-of-place %.4f ms %.0f | 2-deep %.4f ms %.0f GB/s ; mdot16 %.4f ms %.0f GB/s\n", rnd ? "random" : "zeros ",
                 grid, t0, gb / t0 * 1e3, t1, gb / t1 * 1e3, t2, gb / t2 * 1e3, t3, gb / t3 * 1e3, tm, (NV + 1) * 8.0 * n / 1e9 / tm * 1e3);
          fflush(stdout);
        }
      }
      CK(hipFree(z));
    }
    CK(hipFree(base)); CK(hipFree(out));
  }
  return 0;
}
