// Probe: how do element-wise streams of NR read + NW written double vectors have to be issued to reach the rate the memory system
// gives a plain copy (MI355X_MICROARCH.md: 6.29 TB/s float4 copy), at sizes inside (2^24 doubles = 134 MB per vector) and beyond
// (2^27 = 1 GiB per vector) the 256 MiB Infinity Cache?  Shapes: copy (1R 1W), AYPX (2R 1W, one of them in place), the fused CG update
// (5R 3W, x r in place), read-only (dot: 2R).  Variants: how the index space is dealt to workgroups and how many 16-byte accesses a
// lane keeps in flight.
//   hipcc -O3 --offload-arch=gfx950 stream_probe.hip -o stream_probe && ./stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

typedef double v2d __attribute__((ext_vector_type(2)));
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Ptrs { const v2d *in[5]; v2d *out[3]; };

template <int NT> __device__ __forceinline__ v2d ld(const v2d *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <int NT> __device__ __forceinline__ void st(v2d *p, v2d v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// one "element" of work: NR loads, a little arithmetic, NW stores (out[j] may alias in[j]: in place)
template <int NR, int NW, int U, int NTL, int NTS>
__device__ __forceinline__ void work(const Ptrs &P, const size_t (&idx)[U], const bool (&ok)[U]) {
  v2d v[U][NR];
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int j = 0; j < NR; ++j) v[u][j] = ld<NTL>(P.in[j] + (ok[u] ? idx[u] : 0));
#pragma unroll
  for (int u = 0; u < U; ++u) {
    v2d s = v[u][0];
#pragma unroll
    for (int j = 1; j < NR; ++j) s = s + 1.0000001 * v[u][j];
#pragma unroll
    for (int j = 0; j < NW; ++j) if (ok[u]) st<NTS>(P.out[j] + idx[u], s + (double)j);
  }
}

// MODE 0: grid-stride (what csrc/vec_kernels.hip map_kernel does), U accesses per stream in flight per lane
// MODE 1: flat -- workgroup b owns the contiguous tile [b*T*U, (b+1)*T*U) of double2's, lane l takes l, l+T, ..; grid = all tiles
// MODE 2: segment -- the grid's G workgroups each own ONE contiguous segment of n2/G double2's and walk it tile by tile
// MODE 3: XCD segments -- the 8 XCDs (blockIdx % 8) each own a contiguous eighth of the vector; inside it workgroups take tiles in turn
template <int MODE, int NR, int NW, int U, int T, int NTL, int NTS>
__global__ __launch_bounds__(T) void stream_kernel(Ptrs P, size_t n2) {
  const size_t tile = (size_t)T * U;
  if (MODE == 0) {
    const size_t stride = (size_t)gridDim.x * T;
    for (size_t i = (size_t)blockIdx.x * T + threadIdx.x; i < n2; i += U * stride) {
      size_t idx[U]; bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { idx[u] = i + u * stride; ok[u] = idx[u] < n2; }
      work<NR, NW, U, NTL, NTS>(P, idx, ok);
    }
  } else if (MODE == 1) {
    const size_t base = (size_t)blockIdx.x * tile + threadIdx.x;
    size_t idx[U]; bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { idx[u] = base + (size_t)u * T; ok[u] = idx[u] < n2; }
    work<NR, NW, U, NTL, NTS>(P, idx, ok);
  } else if (MODE == 2) {
    const size_t ntiles = (n2 + tile - 1) / tile;
    const size_t per = (ntiles + gridDim.x - 1) / gridDim.x;
    const size_t t0 = (size_t)blockIdx.x * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
    for (size_t t = t0; t < t1; ++t) {
      size_t idx[U]; bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { idx[u] = t * tile + (size_t)u * T + threadIdx.x; ok[u] = idx[u] < n2; }
      work<NR, NW, U, NTL, NTS>(P, idx, ok);
    }
  } else {
    const size_t ntiles = (n2 + tile - 1) / tile;
    const size_t per_x = (ntiles + 7) / 8;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const size_t x0 = (size_t)xcd * per_x, x1 = x0 + per_x < ntiles ? x0 + per_x : ntiles;
    for (size_t t = x0 + slot; t < x1; t += nslot) {
      size_t idx[U]; bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { idx[u] = t * tile + (size_t)u * T + threadIdx.x; ok[u] = idx[u] < n2; }
      work<NR, NW, U, NTL, NTS>(P, idx, ok);
    }
  }
}

// read-only: NR streams summed (a dot's traffic)
template <int MODE, int NR, int U, int T, int NTL>
__global__ __launch_bounds__(T) void read_kernel(Ptrs P, size_t n2, double *out) {
  const size_t tile = (size_t)T * U;
  v2d acc = {0.0, 0.0};
  auto body = [&](size_t base, size_t step) {
    v2d v[U][NR];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < NR; ++j) { const size_t i = base + (size_t)u * step; v[u][j] = ld<NTL>(P.in[j] + (i < n2 ? i : 0)); }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < NR; ++j) acc = acc + v[u][j];
  };
  if (MODE == 0) {
    const size_t stride = (size_t)gridDim.x * T;
    for (size_t i = (size_t)blockIdx.x * T + threadIdx.x; i < n2; i += U * stride) body(i, stride);
  } else if (MODE == 1) {
    body((size_t)blockIdx.x * tile + threadIdx.x, T);
  } else {
    const size_t ntiles = (n2 + tile - 1) / tile;
    const size_t per = (ntiles + gridDim.x - 1) / gridDim.x;
    const size_t t0 = (size_t)blockIdx.x * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
    for (size_t t = t0; t < t1; ++t) body(t * tile + threadIdx.x, T);
  }
  if (acc.x + acc.y == 1.2345e300) out[0] = acc.x;
}

static double *bufs[8];
static size_t cap2;

template <class F> static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) launch();
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms / reps;
}

template <int MODE, int NR, int NW, int U, int T, int NTL, int NTS>
static void run(const char *shape, const char *variant, size_t n, int grid_fixed) {
  Ptrs P;
  // in place where the shape says so: AYPX writes its 2nd input; the CG update writes inputs 0,1 and one fresh vector
  for (int j = 0; j < 5; ++j) P.in[j] = (const v2d *)bufs[j];
  if (NR == 1) P.out[0] = (v2d *)bufs[5];                                   // copy
  else if (NR == 2) P.out[0] = (v2d *)bufs[1];                              // aypx: y = x + a y
  else { P.out[0] = (v2d *)bufs[0]; P.out[1] = (v2d *)bufs[1]; P.out[2] = (v2d *)bufs[5]; }
  const size_t n2 = n / 2;
  const size_t tile = (size_t)T * U;
  int grid = MODE == 1 ? (int)((n2 + tile - 1) / tile) : grid_fixed;
  const int reps = n >= ((size_t)1 << 26) ? 10 : 40;
  double ms = time_ms([&] { hipLaunchKernelGGL((stream_kernel<MODE, NR, NW, U, T, NTL, NTS>), dim3(grid), dim3(T), 0, 0, P, n2); }, reps);
  const double bytes = 8.0 * (double)n * (NR + NW);
  printf("%-6s n=2^%-2d %-44s grid %8d  %8.4f ms  %6.2f TB/s\n", shape, (int)__builtin_ctzll(n), variant, grid, ms, bytes / ms / 1e9);
  fflush(stdout);
}
template <int MODE, int NR, int U, int T, int NTL>
static void run_read(const char *variant, size_t n, int grid_fixed) {
  Ptrs P;
  for (int j = 0; j < 5; ++j) P.in[j] = (const v2d *)bufs[j];
  const size_t n2 = n / 2, tile = (size_t)T * U;
  int grid = MODE == 1 ? (int)((n2 + tile - 1) / tile) : grid_fixed;
  const int reps = n >= ((size_t)1 << 26) ? 10 : 40;
  double ms = time_ms([&] { hipLaunchKernelGGL((read_kernel<MODE, NR, U, T, NTL>), dim3(grid), dim3(T), 0, 0, P, n2, bufs[6]); }, reps);
  printf("%-6s n=2^%-2d %-44s grid %8d  %8.4f ms  %6.2f TB/s\n", NR == 1 ? "read1" : "read2", (int)__builtin_ctzll(n), variant, grid, ms, 8.0 * (double)n * NR / ms / 1e9);
  fflush(stdout);
}

template <int NR, int NW> static void shape(const char *name, size_t n) {
  run<0, NR, NW, 2, 256, 0, 0>(name, "grid-stride U=2 (library today)", n, 2048);
  run<0, NR, NW, 4, 256, 0, 0>(name, "grid-stride U=4", n, 2048);
  run<0, NR, NW, 2, 256, 0, 0>(name, "grid-stride U=2, 4096 workgroups", n, 4096);
  run<1, NR, NW, 1, 256, 0, 0>(name, "flat U=1 T=256", n, 0);
  run<1, NR, NW, 2, 256, 0, 0>(name, "flat U=2 T=256", n, 0);
  run<1, NR, NW, 4, 256, 0, 0>(name, "flat U=4 T=256", n, 0);
  run<1, NR, NW, 8, 256, 0, 0>(name, "flat U=8 T=256", n, 0);
  run<1, NR, NW, 4, 512, 0, 0>(name, "flat U=4 T=512", n, 0);
  run<1, NR, NW, 4, 1024, 0, 0>(name, "flat U=4 T=1024", n, 0);
  run<1, NR, NW, 4, 256, 1, 1>(name, "flat U=4 T=256 nt loads+stores", n, 0);
  run<1, NR, NW, 4, 256, 0, 1>(name, "flat U=4 T=256 nt stores", n, 0);
  run<1, NR, NW, 4, 256, 1, 0>(name, "flat U=4 T=256 nt loads", n, 0);
  run<2, NR, NW, 4, 256, 0, 0>(name, "segment per workgroup U=4, 2048 wg", n, 2048);
  run<2, NR, NW, 4, 256, 0, 0>(name, "segment per workgroup U=4, 1024 wg", n, 1024);
  run<2, NR, NW, 4, 256, 1, 1>(name, "segment per workgroup U=4, 2048 wg, nt", n, 2048);
  run<2, NR, NW, 8, 256, 0, 0>(name, "segment per workgroup U=8, 2048 wg", n, 2048);
  run<2, NR, NW, 4, 1024, 0, 0>(name, "segment per workgroup U=4 T=1024, 512 wg", n, 512);
  run<3, NR, NW, 4, 256, 0, 0>(name, "XCD eighths, tiles in turn U=4, 2048 wg", n, 2048);
  run<3, NR, NW, 4, 256, 1, 1>(name, "XCD eighths, tiles in turn U=4, 2048 wg, nt", n, 2048);
}

// second table: segments dealt to MORE workgroups than are resident (between "flat" and "one segment per resident workgroup"), and the
// non-temporal forms the first table left out
template <int NR, int NW> static void shape2(const char *name, size_t n) {
  run<1, NR, NW, 4, 256, 0, 0>(name, "flat U=4 T=256", n, 0);
  run<1, NR, NW, 4, 256, 1, 1>(name, "flat U=4 T=256 nt loads+stores", n, 0);
  run<1, NR, NW, 2, 256, 1, 1>(name, "flat U=2 T=256 nt loads+stores", n, 0);
  run<1, NR, NW, 8, 256, 1, 1>(name, "flat U=8 T=256 nt loads+stores", n, 0);
  run<1, NR, NW, 1, 256, 1, 1>(name, "flat U=1 T=256 nt loads+stores", n, 0);
  for (int g : {4096, 8192, 16384, 32768}) {
    char buf[96];
    snprintf(buf, sizeof buf, "segment per workgroup U=4, %d wg", g);
    run<2, NR, NW, 4, 256, 0, 0>(name, buf, n, g);
    snprintf(buf, sizeof buf, "segment per workgroup U=4, %d wg, nt", g);
    run<2, NR, NW, 4, 256, 1, 1>(name, buf, n, g);
  }
  run<2, NR, NW, 2, 256, 1, 1>(name, "segment per workgroup U=2, 8192 wg, nt", n, 8192);
  run<2, NR, NW, 8, 256, 1, 1>(name, "segment per workgroup U=8, 8192 wg, nt", n, 8192);
}

int main(int argc, char **argv) {
  const size_t nmax = (size_t)1 << 27;
  cap2 = nmax / 2;
  for (int j = 0; j < 7; ++j) { CK(hipMalloc(&bufs[j], 8 * nmax + 64)); CK(hipMemset(bufs[j], 0, 8 * nmax)); }
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
  if (argc > 1 && std::string(argv[1]) == "2") {
    for (size_t n : {(size_t)1 << 22, (size_t)1 << 24, (size_t)1 << 27}) {
      shape2<1, 1>("copy", n);
      shape2<2, 1>("aypx", n);
      shape2<5, 3>("cgupd", n);
      run_read<1, 2, 4, 256, 0>("flat U=4", n, 0);
      run_read<1, 2, 4, 256, 1>("flat U=4 nt", n, 0);
      run_read<1, 2, 8, 256, 1>("flat U=8 nt", n, 0);
      for (int g : {2048, 4096, 8192, 16384}) {
        char buf[96];
        snprintf(buf, sizeof buf, "segment U=4, %d wg", g); run_read<2, 2, 4, 256, 0>(buf, n, g);
        snprintf(buf, sizeof buf, "segment U=4, %d wg, nt", g); run_read<2, 2, 4, 256, 1>(buf, n, g);
      }
      run_read<2, 2, 8, 256, 1>("segment U=8, 4096 wg, nt", n, 4096);
      run_read<2, 1, 4, 256, 1>("segment U=4, 4096 wg, nt", n, 4096);
      run_read<2, 1, 4, 256, 0>("segment U=4, 4096 wg", n, 4096);
    }
    return 0;
  }
  for (size_t n : {(size_t)1 << 24, (size_t)1 << 27}) {
    shape<1, 1>("copy", n);
    shape<2, 1>("aypx", n);
    shape<5, 3>("cgupd", n);
    run_read<0, 2, 4, 256, 0>("grid-stride U=4, 512 wg (library today)", n, 512);
    run_read<0, 2, 4, 256, 0>("grid-stride U=4, 2048 wg", n, 2048);
    run_read<1, 2, 4, 256, 0>("flat U=4", n, 0);
    run_read<2, 2, 4, 256, 0>("segment U=4, 2048 wg", n, 2048);
    run_read<2, 2, 4, 256, 1>("segment U=4, 2048 wg, nt", n, 2048);
    run_read<0, 1, 4, 256, 0>("grid-stride U=4, 2048 wg", n, 2048);
  }
  return 0;
}
