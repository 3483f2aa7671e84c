// Row plans of the sync-free triangular solves laid out ON THE DEVICE.
//
// The host route (trisolve.hip: trisolve_plan_fill) sorts 16.7 M rows, permutes and scatters 1.4 GB of factor entries with host
// threads and uploads the result: 0.45 s per plan for the ILU(0) factor of P7(256), most of the preconditioner's set-up.  All of
// it is sorting, gathering and scanning, i.e. device work: the factor's arrays go up AS THEY ARE (row starts, row lengths, the
// range of column indices / values the rows name, the dependency level of every row -- the one thing that is inherently
// sequential and stays on the host), and
//   1. a stable radix sort of the rows by (level, longer rows first)             hipcub::DeviceRadixSort, only the key bits in use
//   2. level boundaries out of the sorted order -> first position of every level  (host: nlev numbers, the by-level alignment)
//   3. row -> position, position -> row
//   4. one wavefront per slice: (length, sub-step) words, inverted diagonals in position order, slice width, sub-steps
//   5. exclusive scan of the slice widths -> slice offsets                        hipcub::DeviceScan, 64-bit, checked against int
//   6. every position's lane writes its row into the sliced-ELL arrays, dependencies translated to positions
// produce the arrays of the host route bit for bit (tests: both routes behind the same solves; MI355X_TRISOLVE_BUILD=host).
// Column order only: the by-level entry order and the node plans stay with the host route.
#include "trisolve_plan.hpp"
#include <hipcub/hipcub.hpp>
#include <chrono>
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace {

__device__ __forceinline__ int wave_imax(int v) {
#pragma unroll
  for (int off = MI355X_WAVE / 2; off > 0; off >>= 1) { const int o = __shfl_xor(v, off, MI355X_WAVE); v = o > v ? o : v; }
  return v;
}
__device__ __forceinline__ int wave_imin(int v) {
#pragma unroll
  for (int off = MI355X_WAVE / 2; off > 0; off >>= 1) { const int o = __shfl_xor(v, off, MI355X_WAVE); v = o < v ? o : v; }
  return v;
}

// ext[0] = longest row, ext[1] = one past the last entry any row names, ext[2] = first entry any row names (ext preset to 0, 0, INT_MAX);
// ext[3] |= 1 for a negative length / start or a level outside [0, nlev)
__global__ __launch_bounds__(MI355X_BLOCK) void tri_extent_kernel(int n, int nlev, const int *__restrict__ lev, const int *__restrict__ rp,
                                                                 const int *__restrict__ rl, int *ext) {
  int mx = 0, hi = 0, lo = 0x7fffffff, bad = 0;
  for (long i = (long)blockIdx.x * MI355X_BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * MI355X_BLOCK) {
    const int l = rl[i], s = rp[i], v = lev[i];
    if (l < 0 || s < 0 || v < 0 || v >= nlev || (long)s + l > 2147483000L) { bad = 1; continue; }
    mx = l > mx ? l : mx;
    if (l > 0) { hi = s + l > hi ? s + l : hi; lo = s < lo ? s : lo; }
  }
  mx = wave_imax(mx); hi = wave_imax(hi); lo = wave_imin(lo); bad = wave_imax(bad);
  if ((threadIdx.x & (MI355X_WAVE - 1)) == 0) {
    atomicMax(ext + 0, mx); atomicMax(ext + 1, hi); atomicMin(ext + 2, lo);
    if (bad) atomicOr(ext + 3, 1);
  }
}

// sort key: level major, longer rows first inside a level; the sort is stable, so equal keys stay in row order
__global__ __launch_bounds__(MI355X_BLOCK) void tri_keys_kernel(int n, int maxlen, const int *__restrict__ lev, const int *__restrict__ rl,
                                                               unsigned long long *__restrict__ keys, int *__restrict__ rows) {
  for (long i = (long)blockIdx.x * MI355X_BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * MI355X_BLOCK) {
    keys[i] = (unsigned long long)lev[i] * (unsigned long long)(maxlen + 1) + (unsigned long long)(maxlen - rl[i]);
    rows[i] = (int)i;
  }
}

// levptr[l] = index in the sorted order of level l's first row (levptr preset to -1: a level without rows stays -1)
__global__ __launch_bounds__(MI355X_BLOCK) void tri_levptr_kernel(int n, const int *__restrict__ order, const int *__restrict__ lev, int *__restrict__ levptr) {
  for (long t = (long)blockIdx.x * MI355X_BLOCK + threadIdx.x; t < n; t += (long)gridDim.x * MI355X_BLOCK) {
    const int l = lev[order[t]];
    if (t == 0 || lev[order[t - 1]] != l) levptr[l] = (int)t;
  }
}

__global__ __launch_bounds__(MI355X_BLOCK) void tri_positions_kernel(int n, const int *__restrict__ order, const int *__restrict__ lev,
                                                                    const int *__restrict__ levptr, const int *__restrict__ levbase,
                                                                    int *__restrict__ pos, int *__restrict__ rowof) {
  for (long t = (long)blockIdx.x * MI355X_BLOCK + threadIdx.x; t < n; t += (long)gridDim.x * MI355X_BLOCK) {
    const int i = order[t], l = lev[i];
    const int P = levbase[l] + ((int)t - levptr[l]);
    pos[i] = P;
    rowof[P] = i;
  }
}

// one wavefront per slice (rowof preset to -1 = padding position)
__global__ __launch_bounds__(MI355X_BLOCK) void tri_slices_kernel(int nslices, const int *__restrict__ rowof, const int *__restrict__ lev,
                                                                 const int *__restrict__ rl, const double *__restrict__ dinv_row,
                                                                 const double *__restrict__ rsc_row, int *__restrict__ info,
                                                                 double *__restrict__ dinv, double *__restrict__ rsc,
                                                                 long long *__restrict__ width, unsigned char *__restrict__ nsub, int *bad) {
  const int lane = threadIdx.x & (MI355X_WAVE - 1);
  const long s = ((long)blockIdx.x * MI355X_BLOCK + threadIdx.x) / MI355X_WAVE;
  if (s >= nslices) return;
  const long P = s * MI355X_WAVE + lane;
  const int i = rowof[P];
  const bool valid = i >= 0;
  const int l = valid ? lev[i] : 0x7fffffff, len = valid ? rl[i] : 0;
  const int l0 = wave_imin(l);                       // positions are in level order: the slice's lowest level is its first row's
  const int sub = valid ? l - l0 : 0;
  if (sub > 255) { atomicOr(bad, 2); }
  info[P] = valid ? ((len << 8) | (sub & 255)) : 0;
  if (dinv) dinv[P] = (valid && dinv_row) ? dinv_row[i] : 1.0;
  if (rsc) rsc[P] = (valid && rsc_row) ? rsc_row[i] : 1.0;
  const int mx = wave_imax(len), ns = wave_imax(valid ? sub + 1 : 1);
  if (lane == 0) { width[s] = (long long)mx * MI355X_WAVE; nsub[s] = (unsigned char)(ns > 255 ? 255 : ns); }
}

__global__ __launch_bounds__(MI355X_BLOCK) void tri_narrow_kernel(int n, const long long *__restrict__ in, int *__restrict__ out) {
  for (long i = (long)blockIdx.x * MI355X_BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * MI355X_BLOCK) out[i] = (int)in[i];
}

// every position's lane copies its row: entry q of the row at position P = 64 s + lane goes to ptr[s] + 64 q + lane, its column
// translated to the POSITION of that row.  cj / cv are the uploaded range [ext_lo, ext_hi) of the host arrays.
__global__ __launch_bounds__(MI355X_BLOCK) void tri_fill_kernel(long np, int n, const int *__restrict__ rowof, const int *__restrict__ info,
                                                               const int *__restrict__ ptr, const int *__restrict__ rp, int ext_lo,
                                                               const int *__restrict__ cj, const double *__restrict__ cv,
                                                               const int *__restrict__ pos, int *__restrict__ col, double *__restrict__ val, int *bad) {
  const long P = (long)blockIdx.x * MI355X_BLOCK + threadIdx.x;
  if (P >= np) return;
  const int i = rowof[P];
  if (i < 0) return;
  const int lane = (int)(P & (MI355X_WAVE - 1)), len = info[P] >> 8;
  const long base = (long)ptr[P / MI355X_WAVE] + lane;
  const long r0 = (long)rp[i] - ext_lo;
  for (int q = 0; q < len; ++q) {
    const int dep = cj[r0 + q];
    if ((unsigned)dep >= (unsigned)n) { atomicOr(bad, 4); return; }
    const int pd = pos[dep];
    if (pd >= P) { atomicOr(bad, 8); return; }              // a dependency must come earlier
    col[base + (long)q * MI355X_WAVE] = pd;
    val[base + (long)q * MI355X_WAVE] = cv[r0 + q];
  }
}

struct DevTmp {                       // temporaries of one construction, released together
  std::vector<void *> v;
  ~DevTmp() { for (void *q : v) (void)hipFree(q); }
  template <class T> int get(T **out, size_t count) {
    void *q = nullptr;
    const hipError_t e = hipMalloc(&q, sizeof(T) * (count ? count : 1));
    if (e != hipSuccess) { *out = nullptr; return (int)e; }
    v.push_back(q); *out = (T *)q; return 0;
  }
};

}  // namespace

#define B_TRY(expr) do { const int e__ = (int)(expr); if (e__) { (void)hipStreamSynchronize(st); return e__; } } while (0)
#define B_FAIL() do { (void)hipStreamSynchronize(st); return (int)hipErrorInvalidValue; } while (0)
#define B_GRID(cnt) dim3(mi355x_grid_for((size_t)(cnt), 4)), dim3(MI355X_BLOCK), 0, st

int trisolve_plan_fill_device(mi355x_handle_t h, mi355x_trisolve_plan_s *p, int n, int nlev, const int *lev, const int *rp, const int *rl, const int *cj,
                              const double *cv, const double *dinv_host, const double *rscale_host, int by_level) {
  hipStream_t st = h->stream;
  const int W = MI355X_WAVE;
  const bool timing = getenv("MI355X_TRISOLVE_TIMING") != nullptr, upper = dinv_host != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tlast = now();
  auto tick = [&](const char *what) { if (timing) { (void)hipStreamSynchronize(st); const double t = now(); fprintf(stderr, "[mi355x trisolve plan %s, device] %-22s %.3f s\n", upper ? "U" : "L", what, t - tlast); tlast = t; } };
  if (by_level || n <= 0 || nlev <= 0) return (int)hipErrorInvalidValue;
  p->n = n; p->upper = upper; p->nlev = nlev;
  DevTmp tmp;
  int *d_lev, *d_rp, *d_rl, *d_ext;
  double *d_dinvrow = nullptr, *d_rscrow = nullptr;
  B_TRY(tmp.get(&d_lev, (size_t)n)); B_TRY(tmp.get(&d_rp, (size_t)n)); B_TRY(tmp.get(&d_rl, (size_t)n)); B_TRY(tmp.get(&d_ext, 4));
  B_TRY(hipMemcpyAsync(d_lev, lev, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
  B_TRY(hipMemcpyAsync(d_rp, rp, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
  B_TRY(hipMemcpyAsync(d_rl, rl, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
  if (dinv_host) { B_TRY(tmp.get(&d_dinvrow, (size_t)n)); B_TRY(hipMemcpyAsync(d_dinvrow, dinv_host, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st)); }
  if (dinv_host && rscale_host) { B_TRY(tmp.get(&d_rscrow, (size_t)n)); B_TRY(hipMemcpyAsync(d_rscrow, rscale_host, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st)); }
  int ext[4] = {0, 0, 0x7fffffff, 0};
  B_TRY(hipMemcpyAsync(d_ext, ext, sizeof(ext), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(tri_extent_kernel, B_GRID(n), n, nlev, d_lev, d_rp, d_rl, d_ext);
  B_TRY(hipGetLastError());
  B_TRY(hipMemcpyAsync(ext, d_ext, sizeof(ext), hipMemcpyDeviceToHost, st));
  B_TRY(hipStreamSynchronize(st));
  if (ext[3]) B_FAIL();
  const int maxlen = ext[0], ext_hi = ext[1], ext_lo = ext[1] > 0 ? ext[2] : 0;
  tick("row arrays up, extents");
  // the entries the rows name: one contiguous range of the host's column / value arrays
  const size_t nent = ext_hi > ext_lo ? (size_t)(ext_hi - ext_lo) : 0;
  int *d_cj; double *d_cv;
  B_TRY(tmp.get(&d_cj, nent)); B_TRY(tmp.get(&d_cv, nent));
  if (nent) {
    B_TRY(hipMemcpyAsync(d_cj, cj + ext_lo, sizeof(int) * nent, hipMemcpyHostToDevice, st));
    B_TRY(hipMemcpyAsync(d_cv, cv + ext_lo, sizeof(double) * nent, hipMemcpyHostToDevice, st));
  }
  tick("factor entries up");
  // 1. rows by (level, longer first), stable in the row number
  unsigned long long *d_k0, *d_k1; int *d_r0, *d_order;
  B_TRY(tmp.get(&d_k0, (size_t)n)); B_TRY(tmp.get(&d_k1, (size_t)n)); B_TRY(tmp.get(&d_r0, (size_t)n)); B_TRY(tmp.get(&d_order, (size_t)n));
  hipLaunchKernelGGL(tri_keys_kernel, B_GRID(n), n, maxlen, d_lev, d_rl, d_k0, d_r0);
  B_TRY(hipGetLastError());
  { int bits = 1;
    const unsigned long long top = (unsigned long long)nlev * (unsigned long long)(maxlen + 1);
    while (bits < 64 && (top >> bits)) ++bits;
    size_t tb = 0;
    B_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, d_k0, d_k1, d_r0, d_order, n, 0, bits, st));
    unsigned char *d_t; B_TRY(tmp.get(&d_t, tb));
    B_TRY(hipcub::DeviceRadixSort::SortPairs(d_t, tb, d_k0, d_k1, d_r0, d_order, n, 0, bits, st)); }
  // 2. level boundaries; the first position of every level
  int *d_levptr, *d_levbase;
  B_TRY(tmp.get(&d_levptr, (size_t)nlev)); B_TRY(tmp.get(&d_levbase, (size_t)nlev));
  B_TRY(hipMemsetAsync(d_levptr, 0xFF, sizeof(int) * (size_t)nlev, st));
  hipLaunchKernelGGL(tri_levptr_kernel, B_GRID(n), n, d_order, d_lev, d_levptr);
  B_TRY(hipGetLastError());
  std::vector<int> levptr((size_t)nlev + 1), levbase((size_t)nlev);
  B_TRY(hipMemcpyAsync(levptr.data(), d_levptr, sizeof(int) * (size_t)nlev, hipMemcpyDeviceToHost, st));
  B_TRY(hipStreamSynchronize(st));
  levptr[(size_t)nlev] = n;
  if (levptr[0] != 0) B_FAIL();
  for (int l = 0; l < nlev; ++l) if (levptr[(size_t)l] < 0) B_FAIL();       // a level without rows
  long cur = 0;
  p->levpos = (int *)malloc(sizeof(int) * 2 * (size_t)nlev);
  if (!p->levpos) B_FAIL();
  for (int l = 0; l < nlev; ++l) {
    const int sz = levptr[(size_t)l + 1] - levptr[(size_t)l];
    if (sz <= 0) B_FAIL();                                // every level holds a row
    if (by_level && sz >= TRI_ALIGN_MIN && (cur % W)) cur += W - cur % W;
    if (cur + sz > 2147483000L) B_FAIL();
    levbase[(size_t)l] = (int)cur;
    p->levpos[2 * l] = (int)cur; p->levpos[2 * l + 1] = (int)cur + sz;
    cur += sz;
  }
  p->nslices = (int)((cur + W - 1) / W);
  p->nchunks = (p->nslices + 3) / 4;
  const size_t np = (size_t)p->nslices * W;
  B_TRY(hipMemcpyAsync(d_levbase, levbase.data(), sizeof(int) * (size_t)nlev, hipMemcpyHostToDevice, st));
  tick("sort, levels");
  // 3. positions
  B_TRY(hipMalloc((void **)&p->d_pos, sizeof(int) * (size_t)n));
  B_TRY(hipMalloc((void **)&p->d_row, sizeof(int) * np));
  B_TRY(hipMemsetAsync(p->d_row, 0xFF, sizeof(int) * np, st));
  hipLaunchKernelGGL(tri_positions_kernel, B_GRID(n), n, d_order, d_lev, d_levptr, d_levbase, p->d_pos, p->d_row);
  B_TRY(hipGetLastError());
  // 4. per-position words, per-slice widths
  B_TRY(hipMalloc((void **)&p->d_info, sizeof(int) * np));
  B_TRY(hipMalloc((void **)&p->d_nsub, (size_t)p->nslices));
  if (dinv_host) B_TRY(hipMalloc((void **)&p->d_dinv, sizeof(double) * np));
  if (dinv_host && rscale_host) B_TRY(hipMalloc((void **)&p->d_rscale, sizeof(double) * np));
  long long *d_width, *d_ptr64; int *d_bad;
  B_TRY(tmp.get(&d_width, (size_t)p->nslices + 1)); B_TRY(tmp.get(&d_ptr64, (size_t)p->nslices + 1)); B_TRY(tmp.get(&d_bad, 1));
  B_TRY(hipMemsetAsync(d_bad, 0, sizeof(int), st));
  B_TRY(hipMemsetAsync(d_width + p->nslices, 0, sizeof(long long), st));
  { const long threads = (long)p->nslices * W;
    hipLaunchKernelGGL(tri_slices_kernel, dim3((unsigned)((threads + MI355X_BLOCK - 1) / MI355X_BLOCK)), dim3(MI355X_BLOCK), 0, st, p->nslices, p->d_row, d_lev,
                       d_rl, d_dinvrow, d_rscrow, p->d_info, p->d_dinv, p->d_rscale, d_width, p->d_nsub, d_bad);
    B_TRY(hipGetLastError()); }
  // 5. slice offsets
  { size_t tb = 0;
    B_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, d_width, d_ptr64, p->nslices + 1, st));
    unsigned char *d_t; B_TRY(tmp.get(&d_t, tb));
    B_TRY(hipcub::DeviceScan::ExclusiveSum(d_t, tb, d_width, d_ptr64, p->nslices + 1, st)); }
  long long total = 0; int bad = 0;
  B_TRY(hipMemcpyAsync(&total, d_ptr64 + p->nslices, sizeof(long long), hipMemcpyDeviceToHost, st));
  B_TRY(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, st));
  B_TRY(hipStreamSynchronize(st));
  if (bad || total > 2147483000LL) B_FAIL();              // a slice spanning more than 255 levels / offsets beyond int: as the host route
  B_TRY(hipMalloc((void **)&p->d_ptr, sizeof(int) * ((size_t)p->nslices + 1)));
  hipLaunchKernelGGL(tri_narrow_kernel, B_GRID(p->nslices + 1), p->nslices + 1, d_ptr64, p->d_ptr);
  B_TRY(hipGetLastError());
  tick("positions, slices");
  // 6. the sliced-ELL arrays (padding entries: zero, never read)
  const size_t ntot = (size_t)(total > 0 ? total : 1);
  B_TRY(hipMalloc((void **)&p->d_col, sizeof(int) * ntot));
  B_TRY(hipMalloc((void **)&p->d_val, sizeof(double) * ntot));
  B_TRY(hipMemsetAsync(p->d_col, 0, sizeof(int) * ntot, st));
  B_TRY(hipMemsetAsync(p->d_val, 0, sizeof(double) * ntot, st));
  if (np) {
    hipLaunchKernelGGL(tri_fill_kernel, dim3((unsigned)((np + MI355X_BLOCK - 1) / MI355X_BLOCK)), dim3(MI355X_BLOCK), 0, st, (long)np, n, p->d_row, p->d_info,
                       p->d_ptr, d_rp, ext_lo, d_cj, d_cv, p->d_pos, p->d_col, p->d_val, d_bad);
    B_TRY(hipGetLastError());
  }
  B_TRY(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, st));
  B_TRY(hipStreamSynchronize(st));
  if (bad) B_FAIL();
  tick("fill");
  B_TRY(trisolve_plan_finish(h, p, nlev, by_level));
  B_TRY(hipStreamSynchronize(st));
  tick("finish");
  return 0;
}
