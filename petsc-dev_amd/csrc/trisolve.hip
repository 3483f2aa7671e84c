// Level-scheduled sparse triangular solves for the ILU(0) factors (MatSolve_SeqAIJ_NaturalOrdering,
// reference src/mat/impls/aij/seq/aijfact.c:3126-3172, factor layout of :1628-1700: L rows forward, U rows
// stored from the last row backwards, each followed by its inverted diagonal at bdiag[i]).
// Rows of one level are independent; one lane per row subtracts its products in column order, exactly the
// order of PetscSparseDenseMinusDot (aij.h:337-339), so every x[i] carries the reference's bits.
#include "common.hpp"
#include <vector>
#include <algorithm>
#include <thread>
#include <atomic>
#include <chrono>
#include <memory>
#include <string.h>
#include <stdlib.h>

__global__ __launch_bounds__(MI355X_BLOCK) void ilu0_lower_level_kernel(int nrows, const int *__restrict__ rows,
                                                                       const int *__restrict__ bi,
                                                                       const int *__restrict__ bj,
                                                                       const double *__restrict__ ba,
                                                                       const double *b, double *x) {
  const int t = blockIdx.x * MI355X_BLOCK + threadIdx.x;
  if (t >= nrows) return;
  const int i = rows[t];
  double sum = b[i];
  for (int q = bi[i]; q < bi[i + 1]; ++q) sum -= ba[q] * x[bj[q]];
  x[i] = sum;
}

__global__ __launch_bounds__(MI355X_BLOCK) void ilu0_upper_level_kernel(int nrows, const int *__restrict__ rows,
                                                                       const int *__restrict__ bj,
                                                                       const double *__restrict__ ba,
                                                                       const int *__restrict__ bdiag, double *x) {
  const int t = blockIdx.x * MI355X_BLOCK + threadIdx.x;
  if (t >= nrows) return;
  const int i = rows[t];
  const int s0 = bdiag[i + 1] + 1, nz = bdiag[i] - bdiag[i + 1] - 1;
  double sum = x[i];
  for (int q = 0; q < nz; ++q) sum -= ba[s0 + q] * x[bj[s0 + q]];
  x[i] = sum * ba[s0 + nz];
}

extern "C" {

int mi355x_ilu0_lower_level(mi355x_handle_t h, int nrows, const int *rows, const int *bi, const int *bj,
                            const double *ba, const double *b, double *x) {
  if (nrows <= 0) return 0;
  hipLaunchKernelGGL(ilu0_lower_level_kernel, dim3((nrows + MI355X_BLOCK - 1) / MI355X_BLOCK), dim3(MI355X_BLOCK), 0,
                     h->stream, nrows, rows, bi, bj, ba, b, x);
  MI355X_LAUNCH_CHECK();
  return 0;
}

int mi355x_ilu0_upper_level(mi355x_handle_t h, int nrows, const int *rows, const int *bj, const double *ba,
                            const int *bdiag, double *x) {
  if (nrows <= 0) return 0;
  hipLaunchKernelGGL(ilu0_upper_level_kernel, dim3((nrows + MI355X_BLOCK - 1) / MI355X_BLOCK), dim3(MI355X_BLOCK), 0,
                     h->stream, nrows, rows, bj, ba, bdiag, x);
  MI355X_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Sync-free ("point-to-point") triangular solves: ONE launch per solve instead of one per dependency level.
//
// Layout built once on the host (mi355x_trisolve_plan_create): the rows of a factor are sorted by dependency level
// (ties: longer rows first) and stored in slices of 64 consecutive positions, column-major inside a slice
// (sliced ELL, C = one wavefront): entry q of the row at position t = 64 s + lane sits at ptr[s] + 64 q + lane, so a
// wavefront's q-th entries are one coalesced 512-byte load.  Column indices are POSITIONS in the same order, and the
// solution is kept in that order too (w[t]), so that a wavefront stores 64 consecutive doubles.
//
// Hand-off: w is initialised with a sentinel bit pattern; the lane that owns position t stores w[t] once, with a
// write-through (sc1) 8-byte store, and a consumer polls the value itself with sc1 loads until it is not the sentinel
// (data-tagged granule, MI355X_MICROARCH.md "handoff-1to1": no flag, no fence; an aligned 8-byte store is not torn).
// Workgroups pull chunks of 4 slices in position order from 8 interleaved queues (chunk c belongs to queue c % 8); a
// chunk only depends on earlier positions, every queue hands its chunks out in increasing order, and the grid is
// sized to be fully resident, so the smallest unfinished chunk never waits for anything unfinished: every wait ends.
// A slice that spans several levels (small levels) walks them in sub-steps: the lanes of a later level poll what the
// lanes of an earlier level of the same wavefront have already stored.
// Arithmetic: one lane per row, products subtracted in column order -- MatSolve_SeqAIJ_NaturalOrdering's bits
// (aijfact.c:3126-3172), as in the level kernels above.
// Every spin is bounded: a lane that gives up raises *abort_flag (pinned host memory) and all spinners drain.
#include "trisolve_plan.hpp"

__device__ __forceinline__ double tri_poll(const double *p, int *abort_flag, int sleep_cap) {
  double v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int spins = 0;
  while (__double_as_longlong(v) == (long long)TRI_SENTINEL) {
    // back off: a wavefront far ahead of the solve's front must not flood the L2 with polls (thousands of spinning
    // wavefronts slowed the producers 3x); the wait grows from 128 clocks by 128 per poll up to sleep_cap x 128
    { const int k = spins < sleep_cap ? spins + 1 : sleep_cap;
      for (int z = 0; z < k; ++z) __builtin_amdgcn_s_sleep(2); }
    if ((++spins & 255) == 0) {
      if (spins > TRI_SPIN_LIMIT || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
        __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
    }
    v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return v;
}

// UPPER == false: w[t] = b[row] - L(row,:) w          (b in natural order)
// UPPER == true : w[t] = (src[spos[row]] - U(row,:) w) * dinv[t] ;  y[row] = w[t]   (src = the lower solve's w)
// `reset`: the OTHER solve's w, returned to the sentinel for its next run -- lower: reset[t] (coalesced; the upper
// solve of the previous application is complete), upper: src[spos[row]] right after it has been read (only this lane
// reads that entry).  other_queue: the other solve's queue counters, zeroed by workgroup 0.
#ifdef MI355X_TRI_TRACE
// development build (csrc/variants/build_tri_trace.sh): lane 0 of every wavefront leaves four timestamps per slice (100 MHz
// wall clock): slice start, last batch's entries looked at, last dependency arrived, result stored
__device__ long long *tri_trace_buf = nullptr;
extern "C" int mi355x_trisolve_debug_trace(long long *dev_buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(tri_trace_buf), &dev_buf, sizeof(dev_buf)); }
#define TRI_STAMP(k) do { if (tri_trace_buf && lane == 0) tri_trace_buf[(UPPER ? 8000000L : 0L) + (long)s * 4 + (k)] = wall_clock64(); } while (0)
#else
#define TRI_STAMP(k) do { } while (0)
#endif

#ifndef TRI_POLL_TOGETHER
#define TRI_POLL_TOGETHER 1
#endif
template <bool UPPER>
__global__ __launch_bounds__(MI355X_BLOCK) void trisolve_syncfree_kernel(
    int nslices, int nchunks, const int *__restrict__ ptr, const int *__restrict__ info, const int *__restrict__ rowof,
    const int *__restrict__ col, const double *__restrict__ val, const double *__restrict__ dinv,
    const unsigned char *__restrict__ nsub, const double *src, const int *__restrict__ spos, double *w, double *y,
    double *reset, int reset_n, unsigned int *queue, unsigned int *other_queue, int *abort_flag, int sleep_cap,
    const double *__restrict__ rscale) {
  __shared__ int chunk_s[2];
  const int tid = threadIdx.x, lane = tid & (MI355X_WAVE - 1), wave = tid / MI355X_WAVE;
  if (blockIdx.x == 0 && tid < TRI_QUEUES) other_queue[tid * TRI_QSTRIDE] = 0u;
  if (!UPPER) {   // the other solve's vector may be longer than this one's (its own level padding): the tail is re-armed here
    for (long i = (long)nslices * MI355X_WAVE + (long)blockIdx.x * MI355X_BLOCK + tid; i < reset_n; i += (long)gridDim.x * MI355X_BLOCK)
      reset[i] = __longlong_as_double((long long)TRI_SENTINEL);
  }
  const int q = blockIdx.x % TRI_QUEUES;
  for (int it = 0;; ++it) {
    if (tid == 0) {
      const unsigned int k = __hip_atomic_fetch_add(queue + q * TRI_QSTRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      chunk_s[it & 1] = (int)(k * TRI_QUEUES + q);
    }
    __syncthreads();                       // (the slot written two iterations ago is free again: one barrier per iteration)
    const int chunk = chunk_s[it & 1];
    if (chunk >= nchunks || chunk < 0) break;
    const int s = chunk * (MI355X_BLOCK / MI355X_WAVE) + wave;
    if (s >= nslices) continue;
    const int t = s * MI355X_WAVE + lane;
    const int base = ptr[s], slen = (ptr[s + 1] - base) / MI355X_WAVE;
    const int inf = info[t], row = rowof[t];
    const int mylen = inf >> 8, mysub = inf & 255;
    const int ns = nsub[s];
    double sum = 0.0;
    if (row >= 0) {
      if (UPPER) {
        const int p = spos[row];
        sum = src[p];
        if (rscale) sum = sum * rscale[t];          // MatSolve_SeqSBAIJ_1_NaturalOrdering: x[i] = xi * (1/D(i)) between the two sweeps
        reset[p] = __longlong_as_double((long long)TRI_SENTINEL);
      }
      else sum = src[row];
    }
    if (!UPPER && t < reset_n) reset[t] = __longlong_as_double((long long)TRI_SENTINEL);
    const double di = UPPER ? dinv[t] : 1.0;
    TRI_STAMP(0);
    for (int step = 0; step < ns; ++step) {
      if (row >= 0 && mysub == step) {
        // entries in batches of 8: the loads of a batch are issued together, consumed in stored order
        for (int q0 = 0; q0 < mylen; q0 += 8) {
          int c[8]; double a[8], v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int qq = (q0 + j < mylen) ? q0 + j : q0;
            c[j] = col[base + qq * MI355X_WAVE + lane];
            a[j] = val[base + qq * MI355X_WAVE + lane];
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            v[j] = 0.0;
            if (q0 + j < mylen) v[j] = __hip_atomic_load(w + c[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (q0 + 8 >= mylen) TRI_STAMP(1);
#if TRI_POLL_TOGETHER
          // values that were not there yet: ALL of the batch's pending values are requested again together, round after round -- one
          // memory round trip per round however many are pending (a row's dependencies of the previous level complete at about the
          // same time: polled one after the other, each of them cost a round trip of its own ON the dependency chain: P7(256),
          // 3 entries per row, 2.83 ms per application that way, 2.67 so); backed off and bounded as tri_poll is.  Two probes in
          // flight half a round trip apart were measured too: 3.1-3.5 ms -- more poll traffic slows the producers down
          { bool pend = false;
#pragma unroll
            for (int j = 0; j < 8; ++j) pend = pend || (q0 + j < mylen && __double_as_longlong(v[j]) == (long long)TRI_SENTINEL);
            for (int spins = 0; pend;) {
              { const int kz = spins < sleep_cap ? spins + 1 : sleep_cap;
                for (int z = 0; z < kz; ++z) __builtin_amdgcn_s_sleep(2); }
              if ((++spins & 255) == 0) {
                if (spins > TRI_SPIN_LIMIT || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
                  __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                  break;
                }
              }
#pragma unroll
              for (int j = 0; j < 8; ++j)
                if (q0 + j < mylen && __double_as_longlong(v[j]) == (long long)TRI_SENTINEL) v[j] = __hip_atomic_load(w + c[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              pend = false;
#pragma unroll
              for (int j = 0; j < 8; ++j) pend = pend || (q0 + j < mylen && __double_as_longlong(v[j]) == (long long)TRI_SENTINEL);
            } }
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (q0 + j < mylen) sum -= a[j] * v[j];
#else
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (q0 + j < mylen) {
              double xv = v[j];
              if (__double_as_longlong(xv) == (long long)TRI_SENTINEL) xv = tri_poll(w + c[j], abort_flag, sleep_cap);
              sum -= a[j] * xv;
            }
          }
#endif
        }
        TRI_STAMP(2);
        const double r = UPPER ? sum * di : sum;
        __hip_atomic_store(w + t, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        TRI_STAMP(3);
        if (UPPER) y[row] = r;
      }
    }
    (void)slen;
  }
}


// ---------------------------------------------------------------------------------------------
// Node-blocked sync-free solves: the MI355X form of MatSolve_SeqAIJ_Inode (src/mat/impls/aij/seq/inode.c:2327-2760), the routine
// the reference runs on the ILU factor of a matrix with inodes (every 3-dof FEM matrix).  A position of the sliced layout holds a
// NODE: nsz <= NB consecutive rows whose factor rows share one column list (the first row's in L, the last row's in U) and are
// coupled to each other by a dense triangle.  One lane owns a node: it walks the shared columns ONCE -- one index, one gather /
// poll of the solution value, NB values per column -- then solves the nsz x nsz triangle in registers.  Against the row-granular
// kernel above: dependency levels / nsz (a node's rows were nsz consecutive levels), index traffic and polls / nsz.
// Summation order = the reference routine's: every row's sum takes the shared columns two at a time,
// sum -= v[j] t0 + v[j+1] t1 (the two products added to each other first), an odd last column alone, then the couplings inside the
// node row after row -- bit for bit MatSolve_SeqAIJ_Inode while the columns are stored in column order (by_level = 0).
// The index / value loads of the NEXT batch of columns are issued before the current batch's polls: they do not depend on any
// solution value, and behind the polls they would add a memory round trip per batch to the dependency chain.
#ifdef MI355X_TRI_TRACE
// trace build: lane 0 of every wavefront leaves eight timestamps per slice: start, after each of the first five batches, last
// dependency consumed, results stored
#define TRI_NSTAMP(k) do { if (POLL && tri_trace_buf && lane == 0) tri_trace_buf[(UPPER ? 8000000L : 0L) + (long)(t / MI355X_WAVE) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define TRI_NSTAMP(k) do { } while (0)
#endif
template <int NB, bool UPPER, bool POLL, bool BLK>
__device__ __forceinline__ void tri_node_solve(const int t, const int lane, const int np, const int base, const int off, const int ncol, const int row0, const int nsz,
                                               const int *__restrict__ col, const double *__restrict__ val, const double *__restrict__ din,
                                               const double *src, const int *__restrict__ spos, double *w, double *y, double *reset,
                                               int *abort_flag, const int sleep_cap, const double *__restrict__ rscale) {
#ifndef TRI_NODE_B
#define TRI_NODE_B 8
#endif
  static_assert(TRI_NODE_B >= 2 && TRI_NODE_B % 2 == 0 && TRI_NODE_B <= 16, "TRI_NODE_B: an even number of columns per batch (the inode routine's pairs), at most 16 (registers)");
  // BLK (block columns): the shared list consists of WHOLE dependency nodes, all nodes have NB rows.  The list then stores one
  // entry per dependency NODE (its position), the solution lives node by node (w[position * NB + row]) so that a dependency's
  // NB values are one contiguous gather, and a batch is a few whole nodes: 4.5 batches of 8 columns become 3 batches of 4
  // nodes on a 3-dof FEM factor -- fewer memory round trips on the dependency chain.  Same column sequence, same pairs, same bits.
  constexpr int B = BLK ? (NB == 2 ? 8 : (NB == 3 ? 12 : (NB == 4 ? 8 : 10))) : (NB <= 3 ? TRI_NODE_B : 4);  // columns per batch
  constexpr int CB = BLK ? B / NB : B;         // list entries per batch
  constexpr int NT = NB * (NB - 1) / 2;
  double sum[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    sum[k] = 0.0;
    if (k < nsz) {
      if (UPPER) {                             // sum[k] belongs to the node's row nsz - 1 - k (k = distance from the LAST row, as the reference counts)
        const int p = spos[row0 + nsz - 1 - k];
        sum[k] = src[p];
        if (rscale) sum[k] = sum[k] * rscale[t];      // single-row plans of an incomplete Cholesky factor: D^-1 between the two sweeps
        if (POLL) reset[p] = __longlong_as_double((long long)TRI_SENTINEL);
      } else sum[k] = src[row0 + k];
    }
  }
  // (without block columns a lane's list sits at the END of the slice's slots, trisolve_plan_fill_nodes: `off` slots of padding first)
  const double *vbase = val + ((size_t)base + (size_t)off * MI355X_WAVE) * NB + lane;
  const int *cbase = col + (BLK ? base / NB : base + off * MI355X_WAVE) + lane;
#define TRI_XADDR(cX, j) (BLK ? w + (size_t)cX[(j) / NB] * NB + ((j) % NB) : w + cX[(j)])
  TRI_NSTAMP(0);
  // the node's own triangle (and inverted diagonals): requested now, needed after the last dependency has arrived
  double dn[NT + (UPPER ? NB : 0) + 1];
#pragma unroll
  for (int e = 0; e < NT + (UPPER ? NB : 0); ++e) dn[e] = din[(size_t)e * np + t];
  constexpr int qstart = 0;
  int cA[CB]; double aA[B * NB];
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    const int q = BLK ? j * NB : qstart + j, qq = (q >= 0 && q < ncol) ? q : 0;
    cA[j] = ncol > 0 ? cbase[(BLK ? qq / NB : qq) * MI355X_WAVE] : 0;
  }
#pragma unroll
  for (int j = 0; j < B; ++j) {
    const int q = qstart + j, qq = (q >= 0 && q < ncol) ? q : 0;
#pragma unroll
    for (int k = 0; k < NB; ++k) aA[j * NB + k] = ncol > 0 ? vbase[(size_t)(qq * NB + k) * MI355X_WAVE] : 0.0;
  }
  for (int q0 = qstart; q0 < ncol; q0 += B) {
    double v[B];
#pragma unroll
    for (int j = 0; j < B; ++j) {
      v[j] = 0.0;
      if (q0 + j >= 0 && q0 + j < ncol) v[j] = POLL ? __hip_atomic_load(TRI_XADDR(cA, j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *TRI_XADDR(cA, j);
    }
    // the next batch's indices and values go out BEHIND this batch's gathers (loads return in issue order: the gathers must not
    // queue behind a round trip to HBM) and are in flight while this batch waits for its dependencies
    int cN[CB]; double aN[B * NB];
    const bool more = q0 + B < ncol;
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      const int qn = q0 + B + (BLK ? j * NB : j), qq = (qn >= 0 && qn < ncol) ? qn : 0;   // (an index outside the list re-reads entry 0: never out of bounds)
      cN[j] = more ? cbase[(BLK ? qq / NB : qq) * MI355X_WAVE] : 0;
    }
#pragma unroll
    for (int j = 0; j < B; ++j) {
      const int qn = q0 + B + j, qq = (qn >= 0 && qn < ncol) ? qn : 0;
#pragma unroll
      for (int k = 0; k < NB; ++k) aN[j * NB + k] = more ? vbase[(size_t)(qq * NB + k) * MI355X_WAVE] : 0.0;
    }
    // A value that was not there yet is polled for; when it arrives, the batch's other pending values are requested again
    // TOGETHER (one round trip): the rows of a node are stored by one lane at one time, and a poll per value would put a
    // memory round trip per value on the dependency chain
#define TRI_REFRESH(from)                                                                                              \
  do {                                                                                                                 \
    _Pragma("unroll") for (int jj = (from); jj < B; ++jj)                                                              \
      if (q0 + jj >= 0 && q0 + jj < ncol && __double_as_longlong(v[jj]) == (long long)TRI_SENTINEL)                    \
        v[jj] = __hip_atomic_load(TRI_XADDR(cA, jj), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                      \
  } while (0)
#pragma unroll
    for (int j = 0; j < B; j += 2) {
      if (q0 + j >= 0 && q0 + j < ncol) {
        double x0 = v[j];
        if (POLL && __double_as_longlong(x0) == (long long)TRI_SENTINEL) { x0 = tri_poll(TRI_XADDR(cA, j), abort_flag, sleep_cap); TRI_REFRESH(j + 1); }
        if (q0 + j + 1 < ncol) {
          double x1 = v[j + 1];
          if (POLL && __double_as_longlong(x1) == (long long)TRI_SENTINEL) { x1 = tri_poll(TRI_XADDR(cA, j + 1), abort_flag, sleep_cap); TRI_REFRESH(j + 2); }
          if (NB == 1) { sum[0] -= aA[j] * x0; sum[0] -= aA[j + 1] * x1; }     // single rows: one product after the other (aijfact.c:3126)
          else {
#pragma unroll
            for (int k = 0; k < NB; ++k) sum[k] -= aA[j * NB + k] * x0 + aA[(j + 1) * NB + k] * x1;
          }
        } else {
#pragma unroll
          for (int k = 0; k < NB; ++k) sum[k] -= aA[j * NB + k] * x0;
        }
      }
    }
#undef TRI_REFRESH
#pragma unroll
    for (int j = 0; j < CB; ++j) cA[j] = cN[j];
#pragma unroll
    for (int j = 0; j < B; ++j) {
#pragma unroll
      for (int k = 0; k < NB; ++k) aA[j * NB + k] = aN[j * NB + k];
    }
    if ((q0 - qstart) / B < 5) TRI_NSTAMP(1 + (q0 - qstart) / B);
  }
  TRI_NSTAMP(6);
  // the couplings inside the node
  if (!UPPER) {
#pragma unroll
    for (int k = 1; k < NB; ++k) {
      if (k < nsz) {
#pragma unroll
        for (int l = 0; l < k; ++l) sum[k] -= dn[k * (k - 1) / 2 + l] * sum[l];
      }
    }
#pragma unroll
    for (int k = 0; k < NB; ++k)
      if (k < nsz) {
        double *dst = BLK ? w + (size_t)t * NB + k : w + (size_t)k * np + t;
        if (POLL) __hip_atomic_store(dst, sum[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *dst = sum[k];
      }
  } else {
    double xr[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      xr[k] = 0.0;
      if (k < nsz) {
#pragma unroll
        for (int l = 0; l < k; ++l) sum[k] -= dn[k * (k - 1) / 2 + (k - 1 - l)] * xr[l];   // nearest row last (inode.c:2604-2610)
        xr[k] = sum[k] * dn[NT + k];
        const int kk = nsz - 1 - k;            // the row's slot counts from the node's FIRST row
        double *dst = BLK ? w + (size_t)t * NB + kk : w + (size_t)kk * np + t;
        if (POLL) __hip_atomic_store(dst, xr[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *dst = xr[k];
        y[row0 + kk] = xr[k];
      }
    }
  }
  TRI_NSTAMP(7);
#undef TRI_XADDR
}

template <int NB, bool UPPER, bool BLK>
__global__ __launch_bounds__(MI355X_BLOCK) void trisolve_node_kernel(
    int nslices, int nchunks, int np, const int *__restrict__ ptr, const int *__restrict__ info, const int *__restrict__ rowof,
    const unsigned char *__restrict__ nszof, const int *__restrict__ col, const double *__restrict__ val, const double *__restrict__ din,
    const unsigned char *__restrict__ nsub, const double *src, const int *__restrict__ spos, double *w, double *y,
    double *reset, int reset_n, unsigned int *queue, unsigned int *other_queue, int *abort_flag, int sleep_cap, const double *__restrict__ rscale) {
  __shared__ int chunk_s[2];
  const int tid = threadIdx.x, lane = tid & (MI355X_WAVE - 1), wave = tid / MI355X_WAVE;
  if (blockIdx.x == 0 && tid < TRI_QUEUES) other_queue[tid * TRI_QSTRIDE] = 0u;
  if (!UPPER) {   // the upper solve's slots beyond this plan's own positions are re-armed here
    for (long i = (long)NB * np + (long)blockIdx.x * blockDim.x + tid; i < reset_n; i += (long)gridDim.x * blockDim.x)
      reset[i] = __longlong_as_double((long long)TRI_SENTINEL);
  }
  const int q = blockIdx.x % TRI_QUEUES;
  for (int it = 0;; ++it) {
    if (tid == 0) {
      const unsigned int k = __hip_atomic_fetch_add(queue + q * TRI_QSTRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      chunk_s[it & 1] = (int)(k * TRI_QUEUES + q);
    }
    __syncthreads();
    const int chunk = chunk_s[it & 1];
    if (chunk >= nchunks || chunk < 0) break;
    const int s = chunk * (int)(blockDim.x / MI355X_WAVE) + wave;
    if (s >= nslices) continue;
    const int t = s * MI355X_WAVE + lane;
    const int inf = info[t], row0 = rowof[t], nsz = nszof[t];
    const int ncol = inf >> 8, mysub = inf & 255;
    const int ns = nsub[s];
    if (!UPPER) {   // re-arm the other (upper) solve's slots of this position: its previous application is complete
#pragma unroll
      for (int k = 0; k < NB; ++k) { const long i = BLK ? (long)t * NB + k : (long)k * np + t; if (i < reset_n) reset[i] = __longlong_as_double((long long)TRI_SENTINEL); }
    }
    for (int step = 0; step < ns; ++step)
      if (row0 >= 0 && mysub == step)
        tri_node_solve<NB, UPPER, true, BLK>(t, lane, np, ptr[s], BLK ? 0 : (((ptr[s + 1] - ptr[s]) / MI355X_WAVE - ncol) & ~1), ncol, row0, nsz, col, val, din, src, spos, w, y,
                                             reset, abort_flag, sleep_cap, rscale);
  }
}


// ---------------------------------------------------------------------------------------------
// Split-role form of the node-blocked sync-free solve.  A wavefront's vector-memory operations return in issue order, so in the
// kernel above every poll of a solution value waits behind the index / value loads issued before it -- a round trip to HBM under
// load (2-3 us measured, profiles/r03_tri_variants.log) per batch of columns, on the dependency chain.  Here a workgroup is two
// wavefronts: the LOADER dequeues slices and streams their headers, indices and values from HBM into a ring of batches in LDS,
// running ahead; the SOLVER's only vector-memory operations are the gathers / polls of solution values and its result stores, so
// a poll costs one L2 / fabric round trip (0.3-0.6 us, tests/tools/probe/handoff_probe.hip).  Same plan arrays, same column
// sequence, same pairs, same triangle: same bits as tri_node_solve.  Slices with dependent sub-steps (small levels packed into one
// slice) are solved by the solver with tri_node_solve itself.
// LDS: [64 B counters][2 headers][R batch stages]; counters: 0 headers produced, 1 headers consumed, 2 batches produced,
// 3 batches consumed (monotonic; each has one writer).  LDS operations of a wavefront execute in order: data, s_waitcnt, counter.
// A solved value becomes visible to the polling wavefronts.  All XCDs: an agent-scope store (written through to the memory side,
// 0.54 us to a poller on another XCD).  One XCD (every participating workgroup runs on the same XCD, see the kernel): the XCD's L2 is
// the point of coherence for all of them, a workgroup-scope store reaches it and an agent-scope load reads it there: 0.27 us
// (tests/tools/probe/handoff_probe.hip, profiles/r03_tri_variants.log).
__device__ __forceinline__ void tri_publish(double *p, const double v, const int one_xcd) {
  // agent scope always: a solution value is polled by OTHER workgroups, and only an agent-scope store is visible to them by the memory
  // model.  (Round 3 stored with workgroup scope in the one-XCD form -- the line then stays in the XCD's L2, 13.4 instead of 14.0 ms on
  // the FEM stand-in -- which is correct only as long as L1 writes through and every poller shares that L2: not kept.)
  (void)one_xcd;
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int NB> struct TriSplitGeom {
#ifndef TRI_SPLIT_B3
#define TRI_SPLIT_B3 16
#endif
  // columns per batch.  Nodes of 3 rows (3-dof FEM): 16 -- a node's newest dependencies (several nodes x 3 columns) then sit in ONE batch:
  // FEM stand-in 14.0 -> 13.4 ms in level order, 30.4 -> 26.0 ms in the reference routine's column order (with the one-XCD form below);
  // single rows: 8 (16: ICC(0) 35 -> 43 ms, row-granular ILU(0) 28 -> 40 ms); wider nodes: 4 (registers)
#ifndef TRI_SPLIT_B1
#define TRI_SPLIT_B1 8
#endif
  static constexpr int B = NB == 3 ? TRI_SPLIT_B3 : (NB == 1 ? TRI_SPLIT_B1 : (NB == 2 ? 8 : 4));
  static constexpr int NT = NB * (NB - 1) / 2, ND = NT + NB;
  static constexpr int HB = 64 + 3 * 256 + (NB + ND) * 512;     // header bytes: meta, info / first row / rows per lane, right-hand sides, triangle
  static constexpr int SB = B * 256 + B * NB * 512;             // stage bytes: B index rows, B * NB value rows
  // legal ranges of the build knobs (csrc/variants/build_tri.sh passes arbitrary -D options: a variant outside them must not build):
  // pairs of the inode summation stay pairs (even batches), and the smallest ring -- look-ahead + 3 batches -- fits the LDS asked for
  static_assert(B >= 2 && B % 2 == 0 && B <= 32, "TRI_SPLIT_B*: an even number of columns per batch, at most 32");
};
#define TRI_SPLIT_LDS_BYTES (150 * 1024)
#define TRI_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#ifndef TRI_LOOKAHEAD
// batches the solver's gathers run ahead of its products.  0: a batch's values are requested when the batch is due -- a value
// requested early is mostly a stale sentinel that costs a poll round later (FEM stand-in, profiles/r03_tri_variants.log: 13.9 ms
// at 0, 19.3 at 1, 22.7 at 3; in the inode routine's column order 30.0 / 35.5 / 44.7)
#define TRI_LOOKAHEAD 0
#endif
static_assert(TRI_LOOKAHEAD >= 0 && TRI_LOOKAHEAD <= 8, "TRI_LOOKAHEAD: 0..8 batches");
static_assert(64 + 2 * TriSplitGeom<1>::HB + (TRI_LOOKAHEAD + 3) * TriSplitGeom<1>::SB <= TRI_SPLIT_LDS_BYTES &&
              64 + 2 * TriSplitGeom<2>::HB + (TRI_LOOKAHEAD + 3) * TriSplitGeom<2>::SB <= TRI_SPLIT_LDS_BYTES &&
              64 + 2 * TriSplitGeom<3>::HB + (TRI_LOOKAHEAD + 3) * TriSplitGeom<3>::SB <= TRI_SPLIT_LDS_BYTES &&
              64 + 2 * TriSplitGeom<4>::HB + (TRI_LOOKAHEAD + 3) * TriSplitGeom<4>::SB <= TRI_SPLIT_LDS_BYTES &&
              64 + 2 * TriSplitGeom<5>::HB + (TRI_LOOKAHEAD + 3) * TriSplitGeom<5>::SB <= TRI_SPLIT_LDS_BYTES,
              "the smallest batch ring of the split-role kernels (look-ahead + 3 batches) must fit the LDS they ask for");
// wait until the LDS counter reaches `need`; bounded like every other wait of these kernels: a wavefront that gives up raises the
// abort flag and leaves, its partner's waits then run out the same way, and the application falls back to the level-by-level kernels
__device__ __forceinline__ bool tri_lds_wait(volatile int *c, const int need, int *abort_flag) {
  int spins = 0;
  while (*c < need) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1 << 25)) { __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return false; }
  }
  asm volatile("" ::: "memory");
  return true;
}

template <int NB, bool UPPER>
__device__ __forceinline__ void tri_split_loader(unsigned char *lds, const int lane, const int nslices, const int np, const int R,
                                                 const int *__restrict__ ptr, const int *__restrict__ info, const int *__restrict__ rowof,
                                                 const unsigned char *__restrict__ nszof, const int *__restrict__ col, const double *__restrict__ val,
                                                 const double *__restrict__ din, const unsigned char *__restrict__ nsub, const double *src,
                                                 const int *__restrict__ spos, double *reset, const int reset_n, unsigned int *queue, int *abort_flag,
                                                 const double *__restrict__ rscale, const int q) {
  using G = TriSplitGeom<NB>;
  constexpr int B = G::B, NT = G::NT;
  volatile int *ctl = (volatile int *)lds;
  int hp = 0, bp = 0;
  for (;;) {
    unsigned int k = 0;
    if (lane == 0) k = __hip_atomic_fetch_add(queue + q * TRI_QSTRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    k = (unsigned int)__builtin_amdgcn_readfirstlane((int)k);
    long sl = (long)k * TRI_QUEUES + q;
    const int s = sl < nslices ? (int)sl : -1;
    if (!tri_lds_wait(ctl + 1, hp - 1, abort_flag)) return;
    unsigned char *H = lds + 64 + (hp & 1) * G::HB;
    int *Hm = (int *)H, *Hi = (int *)(H + 64);
    double *Hd = (double *)(H + 64 + 3 * 256);
    if (s < 0) {
      if (lane == 0) Hm[0] = -1;
      TRI_LDS_FENCE();
      ctl[0] = ++hp;
      break;
    }
    const int t = s * MI355X_WAVE + lane;
    const int base = ptr[s], maxcol = (ptr[s + 1] - base) / MI355X_WAVE, ns = nsub[s];
    const int nbatch = ns == 1 ? (maxcol + B - 1) / B : 0;
    if (!UPPER) {   // re-arm the other (upper) solve's slots of this position: its previous application is complete
#pragma unroll
      for (int kk = 0; kk < NB; ++kk) { const long i = (long)kk * np + t; if (i < reset_n) reset[i] = __longlong_as_double((long long)TRI_SENTINEL); }
    }
    if (ns == 1) {
      const int inf = info[t], row0 = rowof[t], nsz = nszof[t];
      double sum[NB], dn[NT + NB];
#pragma unroll
      for (int kk = 0; kk < NB; ++kk) {
        sum[kk] = 0.0;
        if (row0 >= 0 && kk < nsz) {
          if (UPPER) {
            const int p = spos[row0 + nsz - 1 - kk];
            sum[kk] = src[p];
            if (rscale) sum[kk] = sum[kk] * rscale[t];
            reset[p] = __longlong_as_double((long long)TRI_SENTINEL);
          } else sum[kk] = src[row0 + kk];
        }
      }
#pragma unroll
      for (int e = 0; e < NT + (UPPER ? NB : 0); ++e) dn[e] = din[(size_t)e * np + t];
      Hi[lane] = inf; Hi[64 + lane] = row0; Hi[128 + lane] = nsz;
#pragma unroll
      for (int kk = 0; kk < NB; ++kk) Hd[kk * 64 + lane] = sum[kk];
#pragma unroll
      for (int e = 0; e < NT + (UPPER ? NB : 0); ++e) Hd[(NB + e) * 64 + lane] = dn[e];
    }
    if (lane == 0) { Hm[0] = s; Hm[1] = ns; Hm[2] = nbatch; Hm[3] = maxcol; }
    TRI_LDS_FENCE();
    ctl[0] = ++hp;
    if (nbatch == 0) continue;
    // the slice's batches, the loads of batch i + 1 in flight while batch i goes to LDS
    const int pad = nbatch * B - maxcol;
    const int *cbase = col + base + lane;
    const double *vbase = val + (size_t)base * NB + lane;
    int cA[B], cB[B]; double aA[B * NB], aB[B * NB];
#define TRI_LD_BATCH(i, cX, aX)                                                                                          \
  do {                                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < B; ++j) {                                                                      \
      const int qq = (i) * B + j - pad;               /* the slot: batches are aligned to the END of the slice's slots */  \
      cX[j] = qq >= 0 ? cbase[(size_t)qq * MI355X_WAVE] : 0;                                                             \
      _Pragma("unroll") for (int kk = 0; kk < NB; ++kk) aX[j * NB + kk] = qq >= 0 ? vbase[(size_t)(qq * NB + kk) * MI355X_WAVE] : 0.0; \
    }                                                                                                                    \
  } while (0)
#define TRI_PUT_BATCH(cX, aX)                                                                                            \
  do {                                                                                                                   \
    if (!tri_lds_wait(ctl + 3, bp - R + 1, abort_flag)) return;                                                          \
    unsigned char *S = lds + 64 + 2 * G::HB + (size_t)(bp % R) * G::SB;                                                  \
    int *Si = (int *)S; double *Sv = (double *)(S + B * 256);                                                            \
    _Pragma("unroll") for (int j = 0; j < B; ++j) Si[j * 64 + lane] = cX[j];                                             \
    _Pragma("unroll") for (int j = 0; j < B * NB; ++j) Sv[j * 64 + lane] = aX[j];                                        \
    TRI_LDS_FENCE();                                                                                                     \
    ctl[2] = ++bp;                                                                                                       \
  } while (0)
    TRI_LD_BATCH(0, cA, aA);
    for (int i = 0; i < nbatch; i += 2) {
      if (i + 1 < nbatch) TRI_LD_BATCH(i + 1, cB, aB);
      TRI_PUT_BATCH(cA, aA);
      if (i + 1 < nbatch) {
        if (i + 2 < nbatch) TRI_LD_BATCH(i + 2, cA, aA);
        TRI_PUT_BATCH(cB, aB);
      }
    }
#undef TRI_LD_BATCH
#undef TRI_PUT_BATCH
  }
}

template <int NB, bool UPPER>
__device__ __forceinline__ void tri_split_solver(unsigned char *lds, const int lane, const int np, const int R, const int *__restrict__ ptr,
                                                 const int *__restrict__ info, const int *__restrict__ rowof, const unsigned char *__restrict__ nszof,
                                                 const int *__restrict__ col, const double *__restrict__ val, const double *__restrict__ din,
                                                 const double *src, const int *__restrict__ spos, double *w, double *y, double *reset,
                                                 int *abort_flag, const int sleep_cap, const double *__restrict__ rscale, const int one_xcd) {
  using G = TriSplitGeom<NB>;
  constexpr int B = G::B, NT = G::NT;
  volatile int *ctl = (volatile int *)lds;
  int hb = 0, bb = 0;
  for (;;) {
    if (!tri_lds_wait(ctl + 0, hb + 1, abort_flag)) return;
    const unsigned char *H = lds + 64 + (hb & 1) * G::HB;
    const int *Hm = (const int *)H, *Hi = (const int *)(H + 64);
    const double *Hd = (const double *)(H + 64 + 3 * 256);
    const int s = Hm[0];
    if (s < 0) break;
    const int ns = Hm[1], nbatch = Hm[2], width = Hm[3];
    const int t = s * MI355X_WAVE + lane;
    if (ns != 1) {   // dependent sub-steps inside the slice: the one-wavefront routine, step by step
      TRI_LDS_FENCE();
      ctl[1] = ++hb;
      const int inf = info[t], row0 = rowof[t], nsz = nszof[t];
      const int ncol = inf >> 8, mysub = inf & 255;
      for (int step = 0; step < ns; ++step)
        if (row0 >= 0 && mysub == step)
          tri_node_solve<NB, UPPER, true, false>(t, lane, np, ptr[s], ((ptr[s + 1] - ptr[s]) / MI355X_WAVE - ncol) & ~1, ncol, row0, nsz, col, val, din, src, spos, w, y, reset,
                                                 abort_flag, sleep_cap, rscale);
      continue;
    }
    const int inf = Hi[lane], row0 = Hi[64 + lane], nsz = Hi[128 + lane];
    double sum[NB], dn[NT + NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) sum[k] = Hd[k * 64 + lane];
#pragma unroll
    for (int e = 0; e < NT + (UPPER ? NB : 0); ++e) dn[e] = Hd[(NB + e) * 64 + lane];
    TRI_LDS_FENCE();
    ctl[1] = ++hb;
    const int ncol = row0 >= 0 ? inf >> 8 : 0;
    // this lane's column q sits in slot q + (width - ncol & ~1); batch i, entry j is slot i B + j - (nbatch B - width): the last
    // batch holds the newest dependencies of every lane
    const int shift = nbatch * B - width + ((width - ncol) & ~1);
    int cA[B], cB[B]; double vA[B], vB[B];
#if TRI_LOOKAHEAD == 3
    int cC[B], cD[B]; double vC[B], vD[B];
#endif
#ifdef MI355X_TRI_TRACE
#define TRI_SSTAMP(k) do { if (tri_trace_buf && lane == 0) tri_trace_buf[(UPPER ? 8000000L : 0L) + (long)s * 8 + (k)] = wall_clock64(); } while (0)
#else
#define TRI_SSTAMP(k) do { } while (0)
#endif
    TRI_SSTAMP(0);
    // gathers of batch i: its indices out of LDS, one request per column of this lane
#define TRI_GATHER(i, cX, vX)                                                                                            \
  do {                                                                                                                   \
    if (!tri_lds_wait(ctl + 2, bb + (i) + 1, abort_flag)) return;                                                        \
    const int *Si = (const int *)(lds + 64 + 2 * G::HB + (size_t)((bb + (i)) % R) * G::SB);                              \
    _Pragma("unroll") for (int j = 0; j < B; ++j) {                                                                      \
      cX[j] = Si[j * 64 + lane];                                                                                         \
      vX[j] = 0.0;                                                                                                       \
      if ((i) * B + j >= shift && (i) * B + j - shift < ncol) vX[j] = __hip_atomic_load(w + cX[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          \
    }                                                                                                                    \
  } while (0)
    // batch i into the sums, two columns at a time
#define TRI_CONSUME(i, cX, vX)                                                                                           \
  do {                                                                                                                   \
    const double *Sv = (const double *)(lds + 64 + 2 * G::HB + (size_t)((bb + (i)) % R) * G::SB + B * 256);              \
    const int q0 = (i) * B - shift;                                                                                              \
    /* values that were not there yet: ALL of the batch's pending values are requested again together, round after round   \
     * (one memory round trip per round, however many of them are pending; bounded as tri_poll is) */                      \
    bool pend = false;                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < B; ++j) pend = pend || (q0 + j >= 0 && q0 + j < ncol && __double_as_longlong(vX[j]) == (long long)TRI_SENTINEL); \
    for (int spins = 0; pend;) {                                                                                         \
      { const int kz = spins < sleep_cap ? spins + 1 : sleep_cap;                                                        \
        for (int z = 0; z < kz; ++z) __builtin_amdgcn_s_sleep(2); }                                                      \
      if ((++spins & 255) == 0) {                                                                                        \
        if (spins > TRI_SPIN_LIMIT || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {      \
          __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);                                \
          break;                                                                                                         \
        }                                                                                                                \
      }                                                                                                                  \
      _Pragma("unroll") for (int j = 0; j < B; ++j)                                                                      \
        if (q0 + j >= 0 && q0 + j < ncol && __double_as_longlong(vX[j]) == (long long)TRI_SENTINEL)                                     \
          vX[j] = __hip_atomic_load(w + cX[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                              \
      pend = false;                                                                                                      \
      _Pragma("unroll") for (int j = 0; j < B; ++j) pend = pend || (q0 + j >= 0 && q0 + j < ncol && __double_as_longlong(vX[j]) == (long long)TRI_SENTINEL); \
    }                                                                                                                    \
    _Pragma("unroll") for (int j = 0; j < B; j += 2) {                                                                   \
      if (q0 + j >= 0 && q0 + j + 1 < ncol) {                                                                                           \
        if (NB == 1) { sum[0] -= Sv[j * 64 + lane] * vX[j]; sum[0] -= Sv[(j + 1) * 64 + lane] * vX[j + 1]; }                             \
        else { _Pragma("unroll") for (int k = 0; k < NB; ++k) sum[k] -= Sv[(j * NB + k) * 64 + lane] * vX[j] + Sv[((j + 1) * NB + k) * 64 + lane] * vX[j + 1]; } \
      } else if (q0 + j >= 0 && q0 + j < ncol) {                                                                                        \
        _Pragma("unroll") for (int k = 0; k < NB; ++k) sum[k] -= Sv[(j * NB + k) * 64 + lane] * vX[j];                   \
      }                                                                                                                  \
    }                                                                                                                    \
    TRI_LDS_FENCE();                                                                                                     \
    ctl[3] = bb + (i) + 1;                                                                                               \
    if ((i) < 5) TRI_SSTAMP(1 + (i));                                                                                    \
  } while (0)
#if TRI_LOOKAHEAD == 3
    // the gathers run three batches ahead of the products: behind a wait for a dependency the following batches' values (older
    // dependencies in column order) are then already in registers
    if (nbatch > 0) TRI_GATHER(0, cA, vA);
    if (nbatch > 1) TRI_GATHER(1, cB, vB);
    if (nbatch > 2) TRI_GATHER(2, cC, vC);
    for (int i = 0; i < nbatch; i += 4) {
      if (i + 3 < nbatch) TRI_GATHER(i + 3, cD, vD);
      TRI_CONSUME(i, cA, vA);
      if (i + 1 < nbatch) { if (i + 4 < nbatch) TRI_GATHER(i + 4, cA, vA); TRI_CONSUME(i + 1, cB, vB); }
      if (i + 2 < nbatch) { if (i + 5 < nbatch) TRI_GATHER(i + 5, cB, vB); TRI_CONSUME(i + 2, cC, vC); }
      if (i + 3 < nbatch) { if (i + 6 < nbatch) TRI_GATHER(i + 6, cC, vC); TRI_CONSUME(i + 3, cD, vD); }
    }
#elif TRI_LOOKAHEAD == 0
    for (int i = 0; i < nbatch; ++i) { TRI_GATHER(i, cA, vA); TRI_CONSUME(i, cA, vA); }
#else
    if (nbatch > 0) TRI_GATHER(0, cA, vA);
    for (int i = 0; i < nbatch; i += 2) {
      if (i + 1 < nbatch) TRI_GATHER(i + 1, cB, vB);
      TRI_CONSUME(i, cA, vA);
      if (i + 1 < nbatch) {
        if (i + 2 < nbatch) TRI_GATHER(i + 2, cA, vA);
        TRI_CONSUME(i + 1, cB, vB);
      }
    }
#endif
#undef TRI_GATHER
#undef TRI_CONSUME
    bb += nbatch;
    TRI_SSTAMP(6);
    // the couplings inside the node (as tri_node_solve)
    if (row0 < 0) {
    } else if (!UPPER) {
#pragma unroll
      for (int k = 1; k < NB; ++k) {
        if (k < nsz) {
#pragma unroll
          for (int l = 0; l < k; ++l) sum[k] -= dn[k * (k - 1) / 2 + l] * sum[l];
        }
      }
#pragma unroll
      for (int k = 0; k < NB; ++k)
        if (k < nsz) tri_publish(w + (size_t)k * np + t, sum[k], one_xcd);
    } else {
      double xr[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        xr[k] = 0.0;
        if (k < nsz) {
#pragma unroll
          for (int l = 0; l < k; ++l) sum[k] -= dn[k * (k - 1) / 2 + (k - 1 - l)] * xr[l];
          xr[k] = sum[k] * dn[NT + k];
          const int kk = nsz - 1 - k;
          tri_publish(w + (size_t)kk * np + t, xr[k], one_xcd);
          y[row0 + kk] = xr[k];
        }
      }
    }
    TRI_SSTAMP(7);
  }
}

template <int NB, bool UPPER>
__global__ __launch_bounds__(2 * MI355X_WAVE) void trisolve_node_split_kernel(
    int nslices, int np, int R, const int *__restrict__ ptr, const int *__restrict__ info, const int *__restrict__ rowof,
    const unsigned char *__restrict__ nszof, const int *__restrict__ col, const double *__restrict__ val, const double *__restrict__ din,
    const unsigned char *__restrict__ nsub, const double *src, const int *__restrict__ spos, double *w, double *y,
    double *reset, int reset_n, unsigned int *queue, unsigned int *other_queue, int *abort_flag, int sleep_cap, const double *__restrict__ rscale,
    int one_xcd) {
  extern __shared__ __align__(16) unsigned char tri_split_lds[];
  const int tid = threadIdx.x, lane = tid & (MI355X_WAVE - 1), wave = tid / MI355X_WAVE;
  if (tid < 16) ((int *)tri_split_lds)[tid] = 0;
  if (blockIdx.x == 0 && tid < TRI_QUEUES) other_queue[tid * TRI_QSTRIDE] = 0u;
  if (blockIdx.x == 0 && tid == 0) { other_queue[TRI_XCD_WORD] = 0xffffffffu; other_queue[TRI_XCD_WORD + 1] = 0u; }
  if (!UPPER) {   // the upper solve's slots beyond this plan's own positions are re-armed here
    for (long i = (long)NB * np + (long)blockIdx.x * blockDim.x + tid; i < reset_n; i += (long)gridDim.x * blockDim.x)
      reset[i] = __longlong_as_double((long long)TRI_SENTINEL);
  }
  __syncthreads();
  int q = blockIdx.x % TRI_QUEUES;
  if (one_xcd) {
    // ONE XCD: the dependency chain of a factor with narrow levels is a chain of hand-offs, and a hand-off inside one XCD (through
    // its L2) takes half the time of one across XCDs (through the memory side).  The launch holds 8 x the workgroups the plan asks
    // for; the XCD of the first workgroup to arrive is the one that solves, workgroups of the other XCDs leave at once; the ones that
    // stay take tickets, which deal them to the queues.  (Bandwidth: these solves move ~1 GB in ~6 ms, a fraction of one XCD's share.)
    int *me = (int *)tri_split_lds + 8;
    if (tid == 0) {
      unsigned int xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      xcc &= 15u;
      const unsigned int lead = atomicCAS(queue + TRI_XCD_WORD, 0xffffffffu, xcc);
      const int in = lead == 0xffffffffu || lead == xcc;
      me[0] = in;
      me[1] = in ? (int)(atomicAdd(queue + TRI_XCD_WORD + 1, 1u) % TRI_QUEUES) : 0;
    }
    __syncthreads();
    if (!me[0]) return;
    q = me[1];
  }
  if (wave == 1) tri_split_loader<NB, UPPER>(tri_split_lds, lane, nslices, np, R, ptr, info, rowof, nszof, col, val, din, nsub, src, spos, reset, reset_n, queue, abort_flag, rscale, q);
  else tri_split_solver<NB, UPPER>(tri_split_lds, lane, np, R, ptr, info, rowof, nszof, col, val, din, src, spos, w, y, reset, abort_flag, sleep_cap, rscale, one_xcd);
}

template <int NB, bool UPPER, bool BLK>
__global__ __launch_bounds__(MI355X_BLOCK) void trisolve_node_level_kernel(int p0, int p1, int np, const int *__restrict__ ptr, const int *__restrict__ info,
                                                                           const int *__restrict__ rowof, const unsigned char *__restrict__ nszof,
                                                                           const int *__restrict__ col, const double *__restrict__ val,
                                                                           const double *__restrict__ din, const double *src,
                                                                           const int *__restrict__ spos, double *w, double *y, const double *__restrict__ rscale) {
  const int t = p0 + blockIdx.x * MI355X_BLOCK + threadIdx.x;
  if (t >= p1) return;
  const int row0 = rowof[t];
  if (row0 < 0) return;
  const int sl = t / MI355X_WAVE, nc = info[t] >> 8;
  tri_node_solve<NB, UPPER, false, BLK>(t, t % MI355X_WAVE, np, ptr[sl], BLK ? 0 : (((ptr[sl + 1] - ptr[sl]) / MI355X_WAVE - nc) & ~1), nc, row0, nszof[t], col, val, din, src, spos, w, y,
                                   (double *)nullptr, (int *)nullptr, 0, rscale);
}

extern "C" int mi355x_trisolve_plan_destroy(mi355x_trisolve_plan_t p);

// The same rows, one dependency level per launch, no polling: what an application falls back to after a sync-free solve gave
// up (bounded spins), and the reference point the sync-free kernels are measured against.  Same layout, same per-row
// order of the products, same bits.
template <bool UPPER>
__global__ __launch_bounds__(MI355X_BLOCK) void trisolve_level_kernel(int p0, int p1, const int *__restrict__ ptr, const int *__restrict__ info,
                                                                      const int *__restrict__ rowof, const int *__restrict__ col,
                                                                      const double *__restrict__ val, const double *__restrict__ dinv,
                                                                      const double *src, const int *__restrict__ spos, double *w, double *y,
                                                                      const double *__restrict__ rscale) {
  const int t = p0 + blockIdx.x * MI355X_BLOCK + threadIdx.x;
  if (t >= p1) return;
  const int row = rowof[t];
  if (row < 0) return;
  const int s = t / MI355X_WAVE, lane = t % MI355X_WAVE, base = ptr[s], mylen = info[t] >> 8;
  double sum;
  if (UPPER) { sum = src[spos[row]]; if (rscale) sum = sum * rscale[t]; }
  else sum = src[row];
  for (int q = 0; q < mylen; ++q) sum -= val[base + q * MI355X_WAVE + lane] * w[col[base + q * MI355X_WAVE + lane]];
  const double r = UPPER ? sum * dinv[t] : sum;
  w[t] = r;
  if (UPPER) y[row] = r;
}

// a host array that is not initialised on allocation (zeroed = true: zero pages from the allocator), and a loop over [0, n) dealt
// to a few host threads in contiguous ranges (one thread below `grain` items per thread)
template <class T> struct HostBuf {
  T *p; size_t n;
  explicit HostBuf(size_t n_, bool zeroed = false) : p((T *)(zeroed ? calloc(n_ ? n_ : 1, sizeof(T)) : malloc((n_ ? n_ : 1) * sizeof(T)))), n(n_) {}
  ~HostBuf() { free(p); }
  HostBuf(const HostBuf &) = delete;
  HostBuf &operator=(const HostBuf &) = delete;
  T &operator[](size_t i) { return p[i]; }
  const T &operator[](size_t i) const { return p[i]; }
  size_t size() const { return n; }
  T *data() { return p; }
};
template <class F> static void host_parallel_for(long n, long grain, F f) {
  long nth = mi355x_host_threads(8);
  if (nth > n / (grain > 0 ? grain : 1)) nth = n / (grain > 0 ? grain : 1);
  if (nth <= 1) { f(0L, n); return; }
  std::vector<std::thread> th;
  for (long k = 1; k < nth; ++k) th.emplace_back(f, n * k / nth, n * (k + 1) / nth);
  f(0L, n / nth);
  for (auto &t : th) t.join();
}

__global__ __launch_bounds__(MI355X_BLOCK) void tri_arm_kernel(size_t n, double *w) {
  for (size_t i = (size_t)blockIdx.x * MI355X_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * MI355X_BLOCK) w[i] = __longlong_as_double((long long)TRI_SENTINEL);
}

// What a row plan needs beside its sliced-ELL arrays (p->nslices, p->nchunks set): the solution vector armed with the sentinel,
// the queue counters, the abort flag in pinned memory, the launch geometry.
int trisolve_plan_finish(mi355x_handle_t h, mi355x_trisolve_plan_s *p, int nlev, int by_level) {
  const size_t np = (size_t)p->nslices * MI355X_WAVE, npa = np > 0 ? np : 1;
  MI355X_TRY(hipMalloc((void **)&p->d_w, sizeof(double) * npa));
  hipLaunchKernelGGL(tri_arm_kernel, dim3(mi355x_grid_for(npa, 4)), dim3(MI355X_BLOCK), 0, h->stream, npa, p->d_w);
  MI355X_LAUNCH_CHECK();
  MI355X_TRY(hipMalloc((void **)&p->d_queue, sizeof(unsigned int) * (TRI_QUEUES * TRI_QSTRIDE + 32)));
  MI355X_TRY(hipMemsetAsync(p->d_queue, 0, sizeof(unsigned int) * (TRI_QUEUES * TRI_QSTRIDE + 32), h->stream));
  MI355X_TRY(hipMemsetAsync(p->d_queue + TRI_XCD_WORD, 0xFF, sizeof(unsigned int), h->stream));
  MI355X_TRY(hipHostMalloc((void **)&p->abort_flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
  *p->abort_flag = 0;
  // fully resident grid: the occupancy the runtime reports, at most 4 workgroups per CU (MI355X_MICROARCH.md:
  // the query can over-report by one; 4 of 256 threads is well inside what this kernel's registers admit)
  int dev = 0, ncu = 256, per_cu = 0;
  hipDeviceProp_t prop;
  MI355X_TRY(hipGetDevice(&dev));
  MI355X_TRY(hipGetDeviceProperties(&prop, dev));
  ncu = prop.multiProcessorCount;
  MI355X_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trisolve_syncfree_kernel<true>, MI355X_BLOCK, 0));
  p->by_level = by_level;
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) return (int)hipErrorInvalidValue;
  p->grid = ncu * per_cu;
  // ... and no larger than a few levels' worth of chunks: workgroups further ahead of the front could only spin
  { const char *e = getenv("MI355X_TRISOLVE_AHEAD");      // development knobs; the defaults are the measured best
    const long ahead = e ? atol(e) : 4;
    const long per_level = ((long)p->nchunks + nlev - 1) / (nlev > 0 ? nlev : 1);
    long g = ahead * per_level;
    if (g < TRI_QUEUES) g = TRI_QUEUES;
    if (g < p->grid) p->grid = (int)g; }
  if (p->grid > p->nchunks) p->grid = p->nchunks > 0 ? p->nchunks : 1;
  if (p->grid >= TRI_QUEUES) p->grid -= p->grid % TRI_QUEUES;    // every queue gets the same number of pullers
  // poll back-off cap (x 128 clocks).  With a row's pending values polled together (one round trip per round) P7(256), 344
  // workgroups: cap 1 2.66 ms, 2 2.67, 4 2.76, 8 2.83 per application (polled one after the other: 8.1 ms at cap 2, 2.9 at cap 8:
  // the polls flooded the L2); P7(128), 88 workgroups: cap 2 1.04 ms, cap 8 1.14
  { const char *e = getenv("MI355X_TRISOLVE_SLEEP"); p->sleep_cap = e ? atoi(e) : 2; if (p->sleep_cap < 1) p->sleep_cap = 1; }
  return 0;
}

extern "C" {

// Host analysis + upload.  n rows; lev[i] = dependency level of row i (0-based, every level non-empty); len(i) and
// the entries of row i come from (rp, cj, cv): row i's off-diagonal entries are cj/cv[rp[i] .. rp[i] + rl[i]).
// dinv_host != NULL marks the upper solve (inverted diagonals per row).
// by_level != 0: every row's entries are stored -- and therefore summed -- in the order of their dependencies' levels (oldest
// first, column order among equals) instead of column order, and a level of TRI_ALIGN_MIN rows or more starts on a slice
// boundary.  A row then waits only on its LAST entries, after everything older has been consumed, and a wavefront does not
// hold rows of two large levels.  For factors of matrices with inodes (3-dof FEM: ~38 entries per row, recent dependencies
// in the middle of the column order) the reference itself runs another routine with another order (MatSolve_SeqAIJ_Inode,
// inode.c); results then agree with the natural-ordering loop to rounding, not bit for bit.  Deterministic either way.
// every failure leaves through TRI_TRY / TRI_FAIL: pending copies out of the local vectors are drained first, and the caller
// (trisolve_plan_create_impl) hands the half-built plan to mi355x_trisolve_plan_destroy
#define TRI_TRY(expr) do { const int e__ = (int)(expr); if (e__) { (void)hipStreamSynchronize(h->stream); return e__; } } while (0)
#define TRI_FAIL() do { (void)hipStreamSynchronize(h->stream); return (int)hipErrorInvalidValue; } while (0)
static int trisolve_plan_fill(mi355x_handle_t h, mi355x_trisolve_plan_s *p, int n, int nlev, const int *lev, const int *rp, const int *rl, const int *cj,
                              const double *cv, const double *dinv_host, const double *rscale_host, int by_level) {
  p->n = n; p->upper = dinv_host != nullptr;
  const int W = MI355X_WAVE;
  const bool timing = getenv("MI355X_TRISOLVE_TIMING") != nullptr, upper = dinv_host != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tlast = now();
  auto tick = [&](const char *what) { if (timing) { const double t = now(); fprintf(stderr, "[mi355x trisolve plan %s] %-28s %.3f s\n", upper ? "U" : "L", what, t - tlast); tlast = t; } };
  // positions: by level, longer rows first inside a level (stable in the row number)
  std::vector<int> order((size_t)n), levptr((size_t)nlev + 1, 0);
  for (int i = 0; i < n; ++i) levptr[(size_t)lev[i] + 1]++;
  for (int l = 0; l < nlev; ++l) levptr[(size_t)l + 1] += levptr[(size_t)l];
  { std::vector<int> next(levptr.begin(), levptr.end() - 1);
    for (int i = 0; i < n; ++i) order[(size_t)next[(size_t)lev[i]]++] = i; }
  // (longer rows first inside a level, stable in the row number: a counting sort over the row lengths when they are few -- the
  // factors of a stencil operator have 0..3 entries per row -- instead of a comparison sort of 16.7 M rows)
  { int maxlen = 0;
    for (int i = 0; i < n; ++i) if (rl[i] > maxlen) maxlen = rl[i];
    if (maxlen <= 4096) {
      std::vector<int> cnt((size_t)maxlen + 2), tmp;
      for (int l = 0; l < nlev; ++l) {
        const int a = levptr[(size_t)l], b = levptr[(size_t)l + 1];
        if (b - a < 2) continue;
        std::fill(cnt.begin(), cnt.end(), 0);
        for (int t = a; t < b; ++t) cnt[(size_t)(maxlen - rl[order[(size_t)t]]) + 1]++;      // bucket 0 = the longest rows
        for (int q = 0; q <= maxlen; ++q) cnt[(size_t)q + 1] += cnt[(size_t)q];
        tmp.assign(order.begin() + a, order.begin() + b);
        for (int t = 0; t < b - a; ++t) order[(size_t)a + (size_t)cnt[(size_t)(maxlen - rl[tmp[(size_t)t]])]++] = tmp[(size_t)t];
      }
    } else {
      for (int l = 0; l < nlev; ++l)
        std::stable_sort(order.begin() + levptr[(size_t)l], order.begin() + levptr[(size_t)l + 1], [&](int a, int b) { return rl[a] > rl[b]; });
    }
  }
  tick("rows by level and length");
  std::vector<long> tpos((size_t)(n > 0 ? n : 1));
  long cur = 0;
  for (int l = 0; l < nlev; ++l) {
    const int sz = levptr[(size_t)l + 1] - levptr[(size_t)l];
    if (by_level && sz >= TRI_ALIGN_MIN && (cur % W)) cur += W - cur % W;     // padding positions: no row, never read
    for (int t = levptr[(size_t)l]; t < levptr[(size_t)l + 1]; ++t) tpos[(size_t)t] = cur++;
  }
  if (cur > 2147483000L) TRI_FAIL();
  p->nlev = nlev;
  p->levpos = (int *)malloc(sizeof(int) * 2 * (size_t)(nlev > 0 ? nlev : 1));
  if (!p->levpos) TRI_FAIL();
  for (int l = 0; l < nlev; ++l) {
    p->levpos[2 * l] = (int)tpos[(size_t)levptr[(size_t)l]];
    p->levpos[2 * l + 1] = (int)tpos[(size_t)levptr[(size_t)l + 1] - 1] + 1;
  }
  p->nslices = (int)((cur + W - 1) / W);
  p->nchunks = (p->nslices + 3) / 4;
  const size_t np = (size_t)p->nslices * W;
  // (host arrays without an initialising pass of their own: every entry is written below, by several threads)
  const size_t npa = np > 0 ? np : 1;
  HostBuf<int> pos((size_t)(n > 0 ? n : 1)), info(npa), rowof(npa), ptr((size_t)p->nslices + 1), slicemax((size_t)(p->nslices > 0 ? p->nslices : 1));
  HostBuf<unsigned char> nsub((size_t)(p->nslices > 0 ? p->nslices : 1));
  HostBuf<double> dinv(npa), rsc(npa);
  if (!pos.p || !info.p || !rowof.p || !ptr.p || !slicemax.p || !nsub.p || !dinv.p || !rsc.p) TRI_FAIL();
  host_parallel_for((long)npa, 1 << 20, [&](long a, long b) { for (long P = a; P < b; ++P) { rowof[(size_t)P] = -1; info[(size_t)P] = 0; dinv[(size_t)P] = 1.0; rsc[(size_t)P] = 1.0; } });
  host_parallel_for((long)n, 1 << 20, [&](long a, long b) { for (long t = a; t < b; ++t) { pos[(size_t)order[(size_t)t]] = (int)tpos[(size_t)t]; rowof[(size_t)tpos[(size_t)t]] = order[(size_t)t]; } });
  tick("positions");
  std::atomic<int> badslice(0);
  host_parallel_for((long)p->nslices, 1 << 14, [&](long s0, long s1) {
    for (long s = s0; s < s1; ++s) {
      int mx = 0, l0 = -1, ns = 1;
      for (int j = 0; j < W; ++j) {
        const size_t P = (size_t)s * W + j;
        const int i = rowof[P];
        if (i < 0) continue;
        if (l0 < 0) l0 = lev[i];                   // positions are in level order: the slice's first row has its lowest level
        const int sub = lev[i] - l0;
        if (sub < 0 || sub > 255) { badslice.store(1); return; }
        info[P] = (rl[i] << 8) | sub;
        if (dinv_host) dinv[P] = dinv_host[i];
        if (rscale_host) rsc[P] = rscale_host[i];
        if (rl[i] > mx) mx = rl[i];
        if (sub + 1 > ns) ns = sub + 1;
      }
      slicemax[(size_t)s] = mx; nsub[(size_t)s] = (unsigned char)ns;
    }
  });
  if (badslice.load()) TRI_FAIL();
  long total = 0;
  for (int s = 0; s < p->nslices; ++s) {
    ptr[(size_t)s] = (int)total;
    total += (long)slicemax[(size_t)s] * W;
    if (total > 2147483000L) TRI_FAIL();
  }
  ptr[(size_t)p->nslices] = (int)total;
  tick("slice layout");
  // (zero pages from the allocator for the sliced ELL arrays, first touched by the fill threads)
  const size_t ntot = (size_t)(total > 0 ? total : 1);
  HostBuf<int> col(ntot, true);
  HostBuf<double> val(ntot, true);
  if (!col.p || !val.p) TRI_FAIL();
  tick("allocation");
  // every row writes its own slots of the sliced ELL arrays: host threads take contiguous ranges of positions
  { std::atomic<int> bad(0);
    auto fill = [&](int t0, int t1) {
      std::vector<int> perm;
      for (int t = t0; t < t1 && !bad.load(std::memory_order_relaxed); ++t) {
        const int i = order[(size_t)t], P = (int)tpos[(size_t)t], s = P / W, lane = P % W;
        perm.resize((size_t)rl[i]);
        for (int q = 0; q < rl[i]; ++q) perm[(size_t)q] = q;
        if (by_level) std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return lev[cj[rp[i] + a]] < lev[cj[rp[i] + b]]; });
        for (int q = 0; q < rl[i]; ++q) {
          const int src_q = perm[(size_t)q];
          const int dep = cj[rp[i] + src_q];
          if (pos[(size_t)dep] >= P) { bad.store(1); break; }   // a dependency must come earlier
          col[(size_t)ptr[(size_t)s] + (size_t)q * W + lane] = pos[(size_t)dep];
          val[(size_t)ptr[(size_t)s] + (size_t)q * W + lane] = cv[rp[i] + src_q];
        }
      }
    };
    int nth = mi355x_host_threads(8);
    if (n < 200000) nth = 1;
    if (nth == 1) fill(0, n);
    else {
      std::vector<std::thread> th;
      for (int k = 0; k < nth; ++k) th.emplace_back(fill, (int)((long)n * k / nth), (int)((long)n * (k + 1) / nth));
      for (auto &t : th) t.join();
    }
    if (bad.load()) TRI_FAIL(); }
  tick("fill");
#define TRI_UP(dst, vec, T) do { TRI_TRY(hipMalloc((void **)&(dst), sizeof(T) * (vec).size())); \
    TRI_TRY(hipMemcpyAsync((dst), (vec).data(), sizeof(T) * (vec).size(), hipMemcpyHostToDevice, h->stream)); } while (0)
  TRI_UP(p->d_ptr, ptr, int); TRI_UP(p->d_info, info, int); TRI_UP(p->d_row, rowof, int); TRI_UP(p->d_col, col, int);
  TRI_UP(p->d_val, val, double); TRI_UP(p->d_nsub, nsub, unsigned char); TRI_UP(p->d_pos, pos, int);
  if (dinv_host) TRI_UP(p->d_dinv, dinv, double);
  if (dinv_host && rscale_host) TRI_UP(p->d_rscale, rsc, double);
#undef TRI_UP
  TRI_TRY(hipStreamSynchronize(h->stream));
  tick("upload");
  TRI_TRY(trisolve_plan_finish(h, p, nlev, by_level));
  TRI_TRY(hipStreamSynchronize(h->stream));
  return 0;
}
#undef TRI_TRY
#undef TRI_FAIL

static int trisolve_plan_fill_nodes(mi355x_handle_t h, mi355x_trisolve_plan_s *p, int n, int nnodes, const int *nstart_in, int nlev, const int *nodelev,
                                    const int *rp, const int *rl, const int *cj, const double *cv, const double *dinv_host, int by_level, int blk,
                                    const double *rscale_host, int singles);
static int trisolve_plan_create_impl(mi355x_handle_t h, int n, int nlev, const int *lev, const int *rp, const int *rl, const int *cj,
                                     const double *cv, const double *dinv_host, const double *rscale_host, int by_level,
                                     mi355x_trisolve_plan_t *out, bool singles = false) {
  mi355x_trisolve_plan_s *p = new mi355x_trisolve_plan_s();
  memset(p, 0, sizeof(*p));
  *out = nullptr;
  // Deep, narrow dependency graphs (the factor of an unstructured matrix: hundreds of rows per level, thousands of levels) are a
  // chain of hand-offs: they run through the split-role kernels of the node plans with every row a node of its own (a loader and a
  // solver wavefront per workgroup, lists end-aligned; same column order, one product after the other: the same bits).  Wide levels
  // (a stencil operator: tens of thousands of rows per level) are throughput-bound and keep the one-wavefront-per-slice kernel.
  // The choice is made for the two plans of a factor together (mi355x_trisolve_plan_create_pair).
  // Row plans in column order are laid out ON THE DEVICE (trisolve_build.hip: the factor's arrays go up as they are, a radix sort
  // orders the rows, kernels write the sliced-ELL arrays); MI355X_TRISOLVE_BUILD=host keeps the host threads' route, which also
  // serves the orders the device route does not build (dependencies oldest first, node plans).  Same plan either way, bit for bit.
  const char *bm = getenv("MI355X_TRISOLVE_BUILD");
  const bool on_device = !singles && !by_level && n > 0 && !(bm && !strcmp(bm, "host"));
  const int rc = singles ? trisolve_plan_fill_nodes(h, p, n, n, nullptr, nlev, lev, rp, rl, cj, cv, dinv_host, by_level, 0, rscale_host, 1)
                 : on_device ? trisolve_plan_fill_device(h, p, n, nlev, lev, rp, rl, cj, cv, dinv_host, rscale_host, by_level)
                             : trisolve_plan_fill(h, p, n, nlev, lev, rp, rl, cj, cv, dinv_host, rscale_host, by_level);
  if (rc) { mi355x_trisolve_plan_destroy(p); return rc; }     // one cleanup path: nothing allocated so far survives a failure
  *out = p;
  return 0;
}


// ---- node plans: host analysis ----
// n rows in nnodes nodes (nstart[u] .. nstart[u + 1] the rows of node u, sizes 1 .. 5); nodelev[u] its dependency level among the
// nodes (every level non-empty).  Row-level factor as for the row plans: row i's off-diagonal entries cj/cv[rp[i] .. rp[i] + rl[i])
// in the reference's stored order -- lower: the shared columns (those of the node's first row), then the node's own earlier rows;
// upper: the node's own later rows, then the shared columns (those of the node's last row).  A factor whose rows do not have that
// shape is refused (hipErrorInvalidValue): the caller keeps the row-granular plan.
#define TRI_TRY(expr) do { const int e__ = (int)(expr); if (e__) { (void)hipStreamSynchronize(h->stream); return e__; } } while (0)
#define TRI_FAIL() do { (void)hipStreamSynchronize(h->stream); return (int)hipErrorInvalidValue; } while (0)
// singles != 0: every row is a node of its own (nstart may be NULL) -- the row-granular solves (MatSolve_SeqAIJ_NaturalOrdering, one
// product after the other) through the same layout and kernels; rscale_host as in trisolve_plan_fill
static int trisolve_plan_fill_nodes(mi355x_handle_t h, mi355x_trisolve_plan_s *p, int n, int nnodes, const int *nstart_in, int nlev, const int *nodelev,
                                    const int *rp, const int *rl, const int *cj, const double *cv, const double *dinv_host, int by_level, int blk,
                                    const double *rscale_host, int singles) {
  std::vector<int> nstart_own;
  if (!nstart_in) {
    if (!singles || nnodes != n) return (int)hipErrorInvalidValue;
    nstart_own.resize((size_t)n + 1);
    for (int i = 0; i <= n; ++i) nstart_own[(size_t)i] = i;
  }
  const int *nstart = nstart_in ? nstart_in : nstart_own.data();
  const int W = MI355X_WAVE;
  const bool upper = dinv_host != nullptr;
  // MI355X_TRISOLVE_TIMING: where the set-up time of a plan goes (stderr)
  const bool timing = getenv("MI355X_TRISOLVE_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tlast = now();
  auto tick = [&](const char *what) { if (timing) { const double t = now(); fprintf(stderr, "[mi355x trisolve plan %s] %-28s %.3f s\n", upper ? "U" : "L", what, t - tlast); tlast = t; } };
  p->n = n; p->upper = upper; p->by_level = by_level; p->nlev = nlev;
  int NB = 1;
  for (int u = 0; u < nnodes; ++u) { const int z = nstart[u + 1] - nstart[u]; if (z < 1 || z > 5) TRI_FAIL(); if (z > NB) NB = z; }
  if (NB < 2 && !singles) TRI_FAIL();
  if (singles && (NB != 1 || blk)) TRI_FAIL();
  p->nb = NB; p->blkcols = blk ? 1 : 0;
  if (blk) {   // block columns: every node has NB rows (NB <= 4), every shared list is a run of WHOLE dependency nodes (checked below)
    if (NB > 4) TRI_FAIL();
    for (int u = 0; u < nnodes; ++u) if (nstart[u + 1] - nstart[u] != NB) TRI_FAIL();
  }
  std::vector<int> nodeof((size_t)(n > 0 ? n : 1));
  for (int u = 0; u < nnodes; ++u) for (int r = nstart[u]; r < nstart[u + 1]; ++r) nodeof[(size_t)r] = u;
  // shape check + shared column counts
  std::vector<int> nsh((size_t)nnodes);
  { std::atomic<int> bad(0);
    auto check = [&](int u0, int u1) {
      for (int u = u0; u < u1 && !bad.load(std::memory_order_relaxed); ++u) {
        const int r0 = nstart[u], z = nstart[u + 1] - r0, rL = r0 + z - 1;
        const int sh = upper ? rl[rL] : rl[r0];
        nsh[(size_t)u] = sh;
        const int *shared = upper ? cj + rp[rL] : cj + rp[r0];
        for (int k = 0; k < z; ++k) {
          const int r = upper ? rL - k : r0 + k;
          if (rl[r] != sh + k) { bad.store(1); return; }
          const int *mine = cj + rp[r] + (upper ? k : 0);              // where this row's copy of the shared list starts
          for (int q = 0; q < sh; ++q) if (mine[q] != shared[q]) { bad.store(1); return; }
          for (int l = 0; l < k; ++l) {                                // the k couplings inside the node
            const int c = upper ? cj[rp[r] + l] : cj[rp[r] + sh + l];
            if (c != (upper ? r + 1 + l : r0 + l)) { bad.store(1); return; }
          }
        }
        for (int q = 0; q < sh; ++q) if (nodeof[(size_t)shared[q]] == u) { bad.store(1); return; }
        if (blk) {
          if (sh % NB) { bad.store(1); return; }
          for (int q = 0; q < sh; ++q) if (shared[q] != nstart[nodeof[(size_t)shared[q - q % NB]]] + q % NB) { bad.store(1); return; }
        }
      }
    };
    int nth = mi355x_host_threads(8);
    if (nnodes < 100000) nth = 1;
    if (nth == 1) check(0, nnodes);
    else {
      std::vector<std::thread> th;
      for (int k = 0; k < nth; ++k) th.emplace_back(check, (int)((long)nnodes * k / nth), (int)((long)nnodes * (k + 1) / nth));
      for (auto &t : th) t.join();
    }
    if (bad.load()) TRI_FAIL(); }
  tick("shape check");
  // positions: nodes by level, more shared columns first inside a level (stable)
  std::vector<int> order((size_t)nnodes), levptr((size_t)nlev + 1, 0);
  for (int u = 0; u < nnodes; ++u) { if (nodelev[u] < 0 || nodelev[u] >= nlev) TRI_FAIL(); levptr[(size_t)nodelev[u] + 1]++; }
  for (int l = 0; l < nlev; ++l) levptr[(size_t)l + 1] += levptr[(size_t)l];
  { std::vector<int> next(levptr.begin(), levptr.end() - 1);
    for (int u = 0; u < nnodes; ++u) order[(size_t)next[(size_t)nodelev[u]]++] = u; }
  for (int l = 0; l < nlev; ++l)
    std::stable_sort(order.begin() + levptr[(size_t)l], order.begin() + levptr[(size_t)l + 1], [&](int a, int b) { return nsh[(size_t)a] > nsh[(size_t)b]; });
  std::vector<long> tpos((size_t)(nnodes > 0 ? nnodes : 1));
  long cur = 0;
  for (int l = 0; l < nlev; ++l) {
    const int sz = levptr[(size_t)l + 1] - levptr[(size_t)l];
    if (sz < 1) TRI_FAIL();
    // a level of TRI_ALIGN_MIN nodes or more starts on a slice boundary, whatever the order of the columns inside the lists
    // (positions do not enter the sums): such slices have no dependent sub-steps and take the batched path of the split-role kernels
    if (sz >= TRI_ALIGN_MIN && (cur % W)) cur += W - cur % W;
    for (int t = levptr[(size_t)l]; t < levptr[(size_t)l + 1]; ++t) tpos[(size_t)t] = cur++;
  }
  if (cur * NB > 2147483000L) TRI_FAIL();
  tick("positions");
  p->levpos = (int *)malloc(sizeof(int) * 2 * (size_t)(nlev > 0 ? nlev : 1));
  if (!p->levpos) TRI_FAIL();
  for (int l = 0; l < nlev; ++l) {
    p->levpos[2 * l] = (int)tpos[(size_t)levptr[(size_t)l]];
    p->levpos[2 * l + 1] = (int)tpos[(size_t)levptr[(size_t)l + 1] - 1] + 1;
  }
  p->nslices = (int)((cur + W - 1) / W);
  { const char *e = getenv("MI355X_TRISOLVE_NODE_WAVES"); p->spw = e ? atoi(e) : 4; if (p->spw != 1 && p->spw != 2 && p->spw != 4) p->spw = 4; }
  { const char *e = getenv("MI355X_TRISOLVE_SPLIT"); p->split = blk ? 0 : (e ? atoi(e) != 0 : 1); if (p->split) p->spw = 1; }
  // one-XCD form (trisolve_node_split_kernel): OFF unless MI355X_TRISOLVE_ONE_XCD=1 asks for it (development).  Its workgroups are
  // dealt to the queues by ticket, so a queue is served only if at least TRI_QUEUES workgroups land on the elected XCD; the launch
  // asks for 8 x 2 x TRI_QUEUES or more, but placement is the dispatcher's: a plan that relies on it would turn an unlucky placement
  // into bounded spins and an aborted solve.
  { const char *e = getenv("MI355X_TRISOLVE_ONE_XCD"); p->one_xcd = p->split && e && atoi(e) != 0; }
  p->nchunks = (p->nslices + p->spw - 1) / p->spw;
  const size_t np = (size_t)p->nslices * W;
  p->np = (int)np;
  const int NT = NB * (NB - 1) / 2, ND = NT + NB;
  std::vector<int> posn((size_t)(nnodes > 0 ? nnodes : 1)), slot((size_t)(n > 0 ? n : 1)), info(np > 0 ? np : 1, 0), rowof(np > 0 ? np : 1, -1),
      ptr((size_t)p->nslices + 1, 0);
  std::vector<unsigned char> nsub((size_t)(p->nslices > 0 ? p->nslices : 1), 1), nszv(np > 0 ? np : 1, 0);
  std::vector<double> din((np > 0 ? np : 1) * (size_t)ND, 0.0);
  for (int t = 0; t < nnodes; ++t) {
    const int u = order[(size_t)t]; const long P = tpos[(size_t)t];
    posn[(size_t)u] = (int)P; rowof[(size_t)P] = nstart[u]; nszv[(size_t)P] = (unsigned char)(nstart[u + 1] - nstart[u]);
    for (int r = nstart[u]; r < nstart[u + 1]; ++r) slot[(size_t)r] = blk ? (int)((size_t)P * NB + (size_t)(r - nstart[u])) : (int)((size_t)(r - nstart[u]) * np + (size_t)P);
  }
  long total = 0;
  for (int s = 0; s < p->nslices; ++s) {
    int mx = 0, l0 = -1;
    for (int j = 0; j < W; ++j) {
      const size_t P = (size_t)s * W + j;
      if (rowof[P] < 0) continue;
      const int u = nodeof[(size_t)rowof[P]];
      if (l0 < 0) l0 = nodelev[u];
      const int sub = nodelev[u] - l0;
      if (sub < 0 || sub > 255) TRI_FAIL();
      info[P] = (nsh[(size_t)u] << 8) | sub;
      if (nsh[(size_t)u] > mx) mx = nsh[(size_t)u];
      if (sub + 1 > nsub[(size_t)s]) nsub[(size_t)s] = (unsigned char)(sub + 1);
    }
    // without block columns the slice is an even number of slots wide and every lane's list ENDS at the slice's last pair of slots
    // (an even number of padding slots first: the pairs of the reference's summation stay pairs): the newest dependencies of all
    // lanes then sit in the slice's last batch, whatever the lengths of their lists
    if (!blk) mx = (mx + 1) & ~1;
    ptr[(size_t)s] = (int)total;
    total += (long)mx * W;
    if (mx > p->maxcol) p->maxcol = mx;
    if (total * NB > 2147483000L) TRI_FAIL();
  }
  ptr[(size_t)p->nslices] = (int)total;
  std::vector<int> col((size_t)(total > 0 ? total : 1) / (blk ? NB : 1) + 1, 0);     // blk: one entry per dependency node
  // (zero pages from the allocator, first touched by the fill threads: a memset of the ~0.5 GB here cost 0.13 s)
  const size_t nval = (size_t)(total > 0 ? total : 1) * NB;
  std::unique_ptr<double, decltype(&free)> val_mem((double *)calloc(nval, sizeof(double)), &free);
  if (!val_mem) TRI_FAIL();
  double *val = val_mem.get();
  tick("slice layout + allocation");
  { std::atomic<int> bad(0);
    auto fill = [&](int t0, int t1) {
      std::vector<int> perm;
      for (int t = t0; t < t1 && !bad.load(std::memory_order_relaxed); ++t) {
        const int u = order[(size_t)t], P = (int)tpos[(size_t)t], s = P / W, lane = P % W;
        const int r0 = nstart[u], z = nstart[u + 1] - r0, rL = r0 + z - 1, sh = nsh[(size_t)u];
        const int *shared = upper ? cj + rp[rL] : cj + rp[r0];
        const int off = blk ? 0 : ((ptr[(size_t)s + 1] - ptr[(size_t)s]) / W - sh) & ~1;
        perm.resize((size_t)sh);
        for (int q = 0; q < sh; ++q) perm[(size_t)q] = q;
        // (stable, and the columns of one dependency node share a level: whole nodes stay together and in their own order)
        if (by_level) std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return nodelev[nodeof[(size_t)shared[a]]] < nodelev[nodeof[(size_t)shared[b]]]; });
        for (int q = 0; q < sh; ++q) {
          const int sq = perm[(size_t)q], dep = shared[sq];
          if (posn[(size_t)nodeof[(size_t)dep]] >= P) { bad.store(1); break; }        // a dependency must come earlier
          if (blk) { if (q % NB == 0) col[(size_t)ptr[(size_t)s] / NB + (size_t)(q / NB) * W + lane] = posn[(size_t)nodeof[(size_t)dep]]; }
          else col[(size_t)ptr[(size_t)s] + (size_t)(off + q) * W + lane] = slot[(size_t)dep];
          for (int k = 0; k < z; ++k) {
            const int r = upper ? rL - k : r0 + k;
            val[(size_t)ptr[(size_t)s] * NB + ((size_t)(off + q) * NB + k) * W + lane] = cv[rp[r] + (upper ? k : 0) + sq];
          }
        }
        for (int k = 0; k < z; ++k) {
          const int r = upper ? rL - k : r0 + k;
          for (int l = 0; l < k; ++l) din[(size_t)(k * (k - 1) / 2 + l) * np + P] = cv[rp[r] + (upper ? l : sh + l)];
          if (upper) din[(size_t)(NT + k) * np + P] = dinv_host[r];
        }
      }
    };
    int nth = mi355x_host_threads(8);
    if (nnodes < 100000) nth = 1;
    if (nth == 1) fill(0, nnodes);
    else {
      std::vector<std::thread> th;
      for (int k = 0; k < nth; ++k) th.emplace_back(fill, (int)((long)nnodes * k / nth), (int)((long)nnodes * (k + 1) / nth));
      for (auto &t : th) t.join();
    }
    if (bad.load()) TRI_FAIL(); }
  tick("fill");
#define TRI_UP(dst, vec, T) do { TRI_TRY(hipMalloc((void **)&(dst), sizeof(T) * (vec).size())); \
    TRI_TRY(hipMemcpyAsync((dst), (vec).data(), sizeof(T) * (vec).size(), hipMemcpyHostToDevice, h->stream)); } while (0)
  TRI_UP(p->d_ptr, ptr, int); TRI_UP(p->d_info, info, int); TRI_UP(p->d_row, rowof, int); TRI_UP(p->d_col, col, int);
  TRI_TRY(hipMalloc((void **)&p->d_val, sizeof(double) * nval));
  TRI_TRY(hipMemcpyAsync(p->d_val, val, sizeof(double) * nval, hipMemcpyHostToDevice, h->stream));
  TRI_UP(p->d_nsub, nsub, unsigned char); TRI_UP(p->d_pos, slot, int);
  TRI_UP(p->d_nsz, nszv, unsigned char); TRI_UP(p->d_din, din, double);
  std::vector<double> rsc;
  if (rscale_host && singles && upper) {
    rsc.assign(np > 0 ? np : 1, 1.0);
    for (size_t P = 0; P < np; ++P) if (rowof[P] >= 0) rsc[P] = rscale_host[rowof[P]];
    TRI_UP(p->d_rscale, rsc, double);
  }
#undef TRI_UP
  const size_t nw = (np > 0 ? np : 1) * (size_t)NB;
  TRI_TRY(hipStreamSynchronize(h->stream));
  tick("upload");
  TRI_TRY(hipMalloc((void **)&p->d_w, sizeof(double) * nw));
  { std::vector<unsigned long long> sent(nw, TRI_SENTINEL);
    TRI_TRY(hipMemcpyAsync(p->d_w, sent.data(), sizeof(double) * sent.size(), hipMemcpyHostToDevice, h->stream));
    TRI_TRY(hipStreamSynchronize(h->stream)); }
  TRI_TRY(hipMalloc((void **)&p->d_queue, sizeof(unsigned int) * (TRI_QUEUES * TRI_QSTRIDE + 32)));
  TRI_TRY(hipMemsetAsync(p->d_queue, 0, sizeof(unsigned int) * (TRI_QUEUES * TRI_QSTRIDE + 32), h->stream));
  TRI_TRY(hipMemsetAsync(p->d_queue + TRI_XCD_WORD, 0xFF, sizeof(unsigned int), h->stream));
  TRI_TRY(hipHostMalloc((void **)&p->abort_flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
  *p->abort_flag = 0;
  int dev = 0;
  hipDeviceProp_t prop;
  TRI_TRY(hipGetDevice(&dev));
  TRI_TRY(hipGetDeviceProperties(&prop, dev));
  p->grid = prop.multiProcessorCount;                // register-heavy kernels (up to 254 VGPRs): one workgroup per CU is resident for every NB
  { const char *e = getenv("MI355X_TRISOLVE_AHEAD");
    const long ahead = e ? atol(e) : (p->split ? 16 : 4);      // (split-role kernels: profiles/r03_tri_variants.log, 21.9 ms at 4, 19.2 at 16, 19.1 at 64)
    const long per_level = ((long)p->nchunks + nlev - 1) / (nlev > 0 ? nlev : 1);
    long g = ahead * per_level;
    if (g < TRI_QUEUES) g = TRI_QUEUES;
    if (g < p->grid) p->grid = (int)g; }
  if (p->grid > p->nchunks) p->grid = p->nchunks > 0 ? p->nchunks : 1;
  if (p->grid >= TRI_QUEUES) p->grid -= p->grid % TRI_QUEUES;
  { const char *e = getenv("MI355X_TRISOLVE_SLEEP"); p->sleep_cap = e ? atoi(e) : (p->grid <= 128 ? 2 : 8); if (p->sleep_cap < 1) p->sleep_cap = 1; }
  TRI_TRY(hipStreamSynchronize(h->stream));
  tick("solution slots, queues");
  return 0;
}
#undef TRI_TRY
#undef TRI_FAIL

int mi355x_trisolve_plan_create_nodes(mi355x_handle_t h, int n, int nnodes, const int *nstart, int nlev, const int *nodelev, const int *rp, const int *rl,
                                      const int *cj, const double *cv, const double *dinv_host, int by_level, int block_columns, mi355x_trisolve_plan_t *out) {
  mi355x_trisolve_plan_s *p = new mi355x_trisolve_plan_s();
  memset(p, 0, sizeof(*p));
  *out = nullptr;
  const int rc = trisolve_plan_fill_nodes(h, p, n, nnodes, nstart, nlev, nodelev, rp, rl, cj, cv, dinv_host, by_level, block_columns, nullptr, 0);
  if (rc) { mi355x_trisolve_plan_destroy(p); return rc; }
  *out = p;
  return 0;
}

int mi355x_trisolve_plan_create_pair(mi355x_handle_t h, int n, int by_level,
                                     int nlev_lo, const int *lev_lo, const int *rp_lo, const int *rl_lo, const int *cj_lo, const double *cv_lo,
                                     int nlev_up, const int *lev_up, const int *rp_up, const int *rl_up, const int *cj_up, const double *cv_up,
                                     const double *dinv_up, const double *rscale_up, mi355x_trisolve_plan_t *lower, mi355x_trisolve_plan_t *upper) {
  *lower = *upper = nullptr;
  if (!dinv_up) return (int)hipErrorInvalidValue;
  int dev = 0;
  MI355X_TRY(hipGetDevice(&dev));
  mi355x_trisolve_plan_t lo = nullptr, up = nullptr;
  int rc_lo = 0, rc_up = 0;
  // MI355X_TRISOLVE_SPLIT=0 / MI355X_TRISOLVE_SPLIT_ROWS=<rows per level> move the switch between the two kernel families (default:
  // fewer than 4096 rows per dependency level -> split-role kernels, every row a node of its own)
  bool singles = false;
  { const char *e = getenv("MI355X_TRISOLVE_SPLIT"), *w = getenv("MI355X_TRISOLVE_SPLIT_ROWS");
    const long width = w ? atol(w) : 4096;
    const int nl = nlev_lo > nlev_up ? nlev_lo : nlev_up;
    singles = !(e && atoi(e) == 0) && nl > 0 && n >= 64 && (long)n / nl < width; }
  std::thread tl([&] { (void)hipSetDevice(dev); rc_lo = trisolve_plan_create_impl(h, n, nlev_lo, lev_lo, rp_lo, rl_lo, cj_lo, cv_lo, nullptr, nullptr, by_level, &lo, singles); });
  rc_up = trisolve_plan_create_impl(h, n, nlev_up, lev_up, rp_up, rl_up, cj_up, cv_up, dinv_up, rscale_up, rscale_up ? 0 : by_level, &up, singles);
  tl.join();
  if (rc_lo || rc_up) {
    if (lo) mi355x_trisolve_plan_destroy(lo);
    if (up) mi355x_trisolve_plan_destroy(up);
    return rc_lo ? rc_lo : rc_up;
  }
  *lower = lo; *upper = up;
  return 0;
}

int mi355x_trisolve_plan_create_nodes_pair(mi355x_handle_t h, int n, int nnodes, const int *nstart, int by_level, int block_columns,
                                           int nlev_lo, const int *nodelev_lo, const int *rp_lo, const int *rl_lo,
                                           int nlev_up, const int *nodelev_up, const int *rp_up, const int *rl_up,
                                           const int *cj, const double *cv, const double *dinv, mi355x_trisolve_plan_t *lower, mi355x_trisolve_plan_t *upper) {
  *lower = *upper = nullptr;
  if (!dinv) return (int)hipErrorInvalidValue;
  int dev = 0;
  MI355X_TRY(hipGetDevice(&dev));
  mi355x_trisolve_plan_t lo = nullptr, up = nullptr;
  int rc_lo = 0, rc_up = 0;
  std::thread tl([&] { (void)hipSetDevice(dev); rc_lo = mi355x_trisolve_plan_create_nodes(h, n, nnodes, nstart, nlev_lo, nodelev_lo, rp_lo, rl_lo, cj, cv, nullptr, by_level, block_columns, &lo); });
  rc_up = mi355x_trisolve_plan_create_nodes(h, n, nnodes, nstart, nlev_up, nodelev_up, rp_up, rl_up, cj, cv, dinv, by_level, block_columns, &up);
  tl.join();
  if (rc_lo || rc_up) {
    if (lo) mi355x_trisolve_plan_destroy(lo);
    if (up) mi355x_trisolve_plan_destroy(up);
    return rc_lo ? rc_lo : rc_up;
  }
  *lower = lo; *upper = up;
  return 0;
}

}  // extern "C"

template <int NB, bool BLK>
static int tri_node_go(mi355x_handle_t h, mi355x_trisolve_plan_t lo, mi355x_trisolve_plan_t up, const double *b, double *y, bool levels) {
  if (!levels && !BLK && lo->split && up->split) {
    // split-role kernels: the LDS ring holds a whole slice's batches (and a spare one) where that fits
    using G = TriSplitGeom<NB>;
    const int glo = lo->grid < TRI_QUEUES ? TRI_QUEUES : lo->grid, gup = up->grid < TRI_QUEUES ? TRI_QUEUES : up->grid;
    const int rmax = (int)((TRI_SPLIT_LDS_BYTES - 64 - 2 * G::HB) / G::SB);
    auto ring = [&](int maxcol) { int r = (maxcol + G::B - 1) / G::B + 1; if (r < TRI_LOOKAHEAD + 3) r = TRI_LOOKAHEAD + 3; if (r > rmax) r = rmax; return r; };
    const int rlo = ring(lo->maxcol), rup = ring(up->maxcol);
    const size_t blo = 64 + 2 * (size_t)G::HB + (size_t)rlo * G::SB, bup = 64 + 2 * (size_t)G::HB + (size_t)rup * G::SB;
    static int attr_dev = -2;         // the device the attribute was set on (per NB: this function is a template); -1: it refused
    {
      int dev = -1;
      MI355X_TRY(hipGetDevice(&dev));
      if (attr_dev != dev && attr_dev != -1) {
        if (hipFuncSetAttribute((const void *)trisolve_node_split_kernel<NB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, TRI_SPLIT_LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void *)trisolve_node_split_kernel<NB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, TRI_SPLIT_LDS_BYTES) != hipSuccess) {
          (void)hipGetLastError();
          attr_dev = -1;
        } else attr_dev = dev;
      }
      // a device that does not give a workgroup that much LDS: the same plans one launch per dependency level (same bits), not a failed MatSolve
      if (attr_dev == -1) return tri_node_go<NB, BLK>(h, lo, up, b, y, true);
    }
    // one-XCD form (see the kernel): 8 x the workgroups, those of seven XCDs leave at once
    const int xlo = lo->one_xcd && glo >= 2 * TRI_QUEUES, xup = up->one_xcd && gup >= 2 * TRI_QUEUES;
    hipLaunchKernelGGL((trisolve_node_split_kernel<NB, false>), dim3(xlo ? glo * MI355X_NXCD : glo), dim3(2 * MI355X_WAVE), blo, h->stream, lo->nslices, lo->np, rlo, lo->d_ptr, lo->d_info,
                       lo->d_row, lo->d_nsz, lo->d_col, lo->d_val, lo->d_din, lo->d_nsub, b, (const int *)nullptr, lo->d_w, (double *)nullptr, up->d_w,
                       up->np * NB, lo->d_queue, up->d_queue, lo->abort_flag, lo->sleep_cap, (const double *)nullptr, xlo);
    MI355X_LAUNCH_CHECK();
    hipLaunchKernelGGL((trisolve_node_split_kernel<NB, true>), dim3(xup ? gup * MI355X_NXCD : gup), dim3(2 * MI355X_WAVE), bup, h->stream, up->nslices, up->np, rup, up->d_ptr, up->d_info,
                       up->d_row, up->d_nsz, up->d_col, up->d_val, up->d_din, up->d_nsub, lo->d_w, lo->d_pos, up->d_w, y, lo->d_w, 0, up->d_queue,
                       lo->d_queue, up->abort_flag, up->sleep_cap, (const double *)up->d_rscale, xup);
    MI355X_LAUNCH_CHECK();
    return 0;
  }
  if (!levels) {
    const int glo = lo->grid < TRI_QUEUES ? TRI_QUEUES : lo->grid, gup = up->grid < TRI_QUEUES ? TRI_QUEUES : up->grid;
    hipLaunchKernelGGL((trisolve_node_kernel<NB, false, BLK>), dim3(glo), dim3(lo->spw * MI355X_WAVE), 0, h->stream, lo->nslices, lo->nchunks, lo->np, lo->d_ptr, lo->d_info,
                       lo->d_row, lo->d_nsz, lo->d_col, lo->d_val, lo->d_din, lo->d_nsub, b, (const int *)nullptr, lo->d_w, (double *)nullptr, up->d_w,
                       up->np * NB, lo->d_queue, up->d_queue, lo->abort_flag, lo->sleep_cap, (const double *)nullptr);
    MI355X_LAUNCH_CHECK();
    hipLaunchKernelGGL((trisolve_node_kernel<NB, true, BLK>), dim3(gup), dim3(up->spw * MI355X_WAVE), 0, h->stream, up->nslices, up->nchunks, up->np, up->d_ptr, up->d_info,
                       up->d_row, up->d_nsz, up->d_col, up->d_val, up->d_din, up->d_nsub, lo->d_w, lo->d_pos, up->d_w, y, lo->d_w, 0, up->d_queue,
                       lo->d_queue, up->abort_flag, up->sleep_cap, (const double *)up->d_rscale);
    MI355X_LAUNCH_CHECK();
    return 0;
  }
  for (int l = 0; l < lo->nlev; ++l) {
    const int p0 = lo->levpos[2 * l], p1 = lo->levpos[2 * l + 1];
    hipLaunchKernelGGL((trisolve_node_level_kernel<NB, false, BLK>), dim3((p1 - p0 + MI355X_BLOCK - 1) / MI355X_BLOCK), dim3(MI355X_BLOCK), 0, h->stream, p0, p1, lo->np,
                       lo->d_ptr, lo->d_info, lo->d_row, lo->d_nsz, lo->d_col, lo->d_val, lo->d_din, b, (const int *)nullptr, lo->d_w, (double *)nullptr, (const double *)nullptr);
    MI355X_LAUNCH_CHECK();
  }
  for (int l = 0; l < up->nlev; ++l) {
    const int p0 = up->levpos[2 * l], p1 = up->levpos[2 * l + 1];
    hipLaunchKernelGGL((trisolve_node_level_kernel<NB, true, BLK>), dim3((p1 - p0 + MI355X_BLOCK - 1) / MI355X_BLOCK), dim3(MI355X_BLOCK), 0, h->stream, p0, p1, up->np,
                       up->d_ptr, up->d_info, up->d_row, up->d_nsz, up->d_col, up->d_val, up->d_din, lo->d_w, lo->d_pos, up->d_w, y, (const double *)up->d_rscale);
    MI355X_LAUNCH_CHECK();
  }
  return 0;
}
static int tri_node_dispatch(mi355x_handle_t h, mi355x_trisolve_plan_t lo, mi355x_trisolve_plan_t up, const double *b, double *y, bool levels) {
  if (lo->nb != up->nb || lo->blkcols != up->blkcols) return (int)hipErrorInvalidValue;
  if (lo->blkcols) {
    switch (lo->nb) {
    case 2: return tri_node_go<2, true>(h, lo, up, b, y, levels);
    case 3: return tri_node_go<3, true>(h, lo, up, b, y, levels);
    case 4: return tri_node_go<4, true>(h, lo, up, b, y, levels);
    default: return (int)hipErrorInvalidValue;
    }
  }
  switch (lo->nb) {
  case 1: return tri_node_go<1, false>(h, lo, up, b, y, levels);
  case 2: return tri_node_go<2, false>(h, lo, up, b, y, levels);
  case 3: return tri_node_go<3, false>(h, lo, up, b, y, levels);
  case 4: return tri_node_go<4, false>(h, lo, up, b, y, levels);
  case 5: return tri_node_go<5, false>(h, lo, up, b, y, levels);
  default: return (int)hipErrorInvalidValue;
  }
}

extern "C" {

int mi355x_trisolve_plan_create_ordered(mi355x_handle_t h, int n, int nlev, const int *lev, const int *rp, const int *rl, const int *cj,
                                        const double *cv, const double *dinv_host, int by_level, mi355x_trisolve_plan_t *out) {
  return trisolve_plan_create_impl(h, n, nlev, lev, rp, rl, cj, cv, dinv_host, nullptr, by_level, out);
}

// upper solve whose right-hand side entry i is multiplied by rscale_host[i] before row i's sum starts: the D^-1 between the
// U^T and the U sweep of an incomplete Cholesky factor (MatSolve_SeqSBAIJ_1_NaturalOrdering, sbaijfact2.c:1977-2015)
int mi355x_trisolve_plan_create_scaled(mi355x_handle_t h, int n, int nlev, const int *lev, const int *rp, const int *rl, const int *cj,
                                       const double *cv, const double *dinv_host, const double *rscale_host, mi355x_trisolve_plan_t *out) {
  if (!dinv_host || !rscale_host) return (int)hipErrorInvalidValue;
  return trisolve_plan_create_impl(h, n, nlev, lev, rp, rl, cj, cv, dinv_host, rscale_host, 0, out);
}

int mi355x_trisolve_plan_create(mi355x_handle_t h, int n, int nlev, const int *lev, const int *rp, const int *rl, const int *cj,
                                const double *cv, const double *dinv_host, mi355x_trisolve_plan_t *out) {
  return mi355x_trisolve_plan_create_ordered(h, n, nlev, lev, rp, rl, cj, cv, dinv_host, 0, out);
}

int mi355x_trisolve_plan_destroy(mi355x_trisolve_plan_t p) {
  if (!p) return 0;
  (void)hipFree(p->d_ptr); (void)hipFree(p->d_info); (void)hipFree(p->d_row); (void)hipFree(p->d_col); (void)hipFree(p->d_val);
  (void)hipFree(p->d_nsub); (void)hipFree(p->d_pos); (void)hipFree(p->d_w); (void)hipFree(p->d_queue);
  if (p->d_dinv) (void)hipFree(p->d_dinv);
  if (p->d_rscale) (void)hipFree(p->d_rscale);
  if (p->d_nsz) (void)hipFree(p->d_nsz);
  if (p->d_din) (void)hipFree(p->d_din);
  if (p->abort_flag) (void)hipHostFree(p->abort_flag);
  free(p->levpos);
  delete p;
  return 0;
}

// One ILU(0) application y = U^-1 L^-1 b: two launches (lower, upper).  Returns hipErrorLaunchFailure if an EARLIER
// application on these plans gave up waiting (the flag lives in pinned memory; the caller then uses the level kernels).
int mi355x_trisolve_apply(mi355x_handle_t h, mi355x_trisolve_plan_t lo, mi355x_trisolve_plan_t up, const double *b, double *y) {
  if (!lo || !up || lo->n != up->n || lo->upper || !up->upper) return (int)hipErrorInvalidValue;
  if (*lo->abort_flag || *up->abort_flag) return (int)hipErrorLaunchFailure;
  if (lo->n == 0) return 0;
  if (lo->nb >= 1 || up->nb >= 1) return tri_node_dispatch(h, lo, up, b, y, false);
  // with fewer chunks than queues some queues have no puller: chunk c is then only served through queue c % 8 ...
  // so tiny systems use ONE workgroup per queue that exists (grid >= min(nchunks, 8) is guaranteed by plan_create)
  const int glo = lo->grid < TRI_QUEUES ? TRI_QUEUES : lo->grid, gup = up->grid < TRI_QUEUES ? TRI_QUEUES : up->grid;
#define TRI_GO()                                                                                                                 \
  do {                                                                                                                            \
    hipLaunchKernelGGL((trisolve_syncfree_kernel<false>), dim3(glo), dim3(MI355X_BLOCK), 0, h->stream, lo->nslices, lo->nchunks, \
                       lo->d_ptr, lo->d_info, lo->d_row, lo->d_col, lo->d_val, (const double *)nullptr, lo->d_nsub, b,          \
                       (const int *)nullptr, lo->d_w, (double *)nullptr, up->d_w, up->nslices * MI355X_WAVE, lo->d_queue, up->d_queue, lo->abort_flag, lo->sleep_cap, \
                       (const double *)nullptr); \
    MI355X_LAUNCH_CHECK();                                                                                                        \
    hipLaunchKernelGGL((trisolve_syncfree_kernel<true>), dim3(gup), dim3(MI355X_BLOCK), 0, h->stream, up->nslices, up->nchunks,  \
                       up->d_ptr, up->d_info, up->d_row, up->d_col, up->d_val, up->d_dinv, up->d_nsub, lo->d_w, lo->d_pos,       \
                       up->d_w, y, lo->d_w, 0, up->d_queue, lo->d_queue, up->abort_flag, up->sleep_cap, (const double *)up->d_rscale); \
    MI355X_LAUNCH_CHECK();                                                                                                        \
  } while (0)
  if (lo->by_level != up->by_level) return (int)hipErrorInvalidValue;
  TRI_GO();
#undef TRI_GO
  return 0;
}

// y = U^-1 L^-1 b over the same plans, one launch per dependency level (nlevL + nlevU launches), no hand-off between
// wavefronts: usable after an abort (the flags are not consulted, nothing is left armed or disarmed for the sync-free form)
int mi355x_trisolve_apply_levels(mi355x_handle_t h, mi355x_trisolve_plan_t lo, mi355x_trisolve_plan_t up, const double *b, double *y) {
  if (!lo || !up || lo->n != up->n || lo->upper || !up->upper) return (int)hipErrorInvalidValue;
  if (lo->n == 0) return 0;
  if (lo->nb >= 1 || up->nb >= 1) return tri_node_dispatch(h, lo, up, b, y, true);
  for (int l = 0; l < lo->nlev; ++l) {
    const int p0 = lo->levpos[2 * l], p1 = lo->levpos[2 * l + 1];
    hipLaunchKernelGGL((trisolve_level_kernel<false>), dim3((p1 - p0 + MI355X_BLOCK - 1) / MI355X_BLOCK), dim3(MI355X_BLOCK), 0, h->stream, p0, p1,
                       lo->d_ptr, lo->d_info, lo->d_row, lo->d_col, lo->d_val, (const double *)nullptr, b, (const int *)nullptr, lo->d_w,
                       (double *)nullptr, (const double *)nullptr);
    MI355X_LAUNCH_CHECK();
  }
  for (int l = 0; l < up->nlev; ++l) {
    const int p0 = up->levpos[2 * l], p1 = up->levpos[2 * l + 1];
    hipLaunchKernelGGL((trisolve_level_kernel<true>), dim3((p1 - p0 + MI355X_BLOCK - 1) / MI355X_BLOCK), dim3(MI355X_BLOCK), 0, h->stream, p0, p1,
                       up->d_ptr, up->d_info, up->d_row, up->d_col, up->d_val, up->d_dinv, lo->d_w, lo->d_pos, up->d_w, y,
                       (const double *)up->d_rscale);
    MI355X_LAUNCH_CHECK();
  }
  return 0;
}

// tests: one array of a ROW plan back on the host.  which: 0 slice offsets (nslices + 1 ints), 1 (length, sub-step) words, 2 position -> row,
// 3 sliced-ELL column positions, 4 sliced-ELL values, 5 inverted diagonals, 6 right-hand-side scales, 7 sub-steps per slice, 8 row ->
// position, 9 first / one-past-last position of every level (host array).  *bytes = the array's size; copied when it fits cap_bytes.
int mi355x_trisolve_debug_get(mi355x_trisolve_plan_t p, int which, void *out, size_t cap_bytes, size_t *bytes) {
  if (!p || p->nb >= 1) return (int)hipErrorInvalidValue;
  const size_t np = (size_t)p->nslices * MI355X_WAVE;
  int total = 0;
  if (p->nslices > 0) MI355X_TRY(hipMemcpy(&total, p->d_ptr + p->nslices, sizeof(int), hipMemcpyDeviceToHost));
  const void *src = nullptr; size_t nb = 0; bool host = false;
  switch (which) {
  case 0: src = p->d_ptr; nb = sizeof(int) * ((size_t)p->nslices + 1); break;
  case 1: src = p->d_info; nb = sizeof(int) * np; break;
  case 2: src = p->d_row; nb = sizeof(int) * np; break;
  case 3: src = p->d_col; nb = sizeof(int) * (size_t)total; break;
  case 4: src = p->d_val; nb = sizeof(double) * (size_t)total; break;
  case 5: src = p->d_dinv; nb = p->d_dinv ? sizeof(double) * np : 0; break;
  case 6: src = p->d_rscale; nb = p->d_rscale ? sizeof(double) * np : 0; break;
  case 7: src = p->d_nsub; nb = (size_t)p->nslices; break;
  case 8: src = p->d_pos; nb = sizeof(int) * (size_t)p->n; break;
  case 9: src = p->levpos; nb = sizeof(int) * 2 * (size_t)p->nlev; host = true; break;
  default: return (int)hipErrorInvalidValue;
  }
  *bytes = nb;
  if (nb && nb <= cap_bytes) {
    if (host) memcpy(out, src, nb);
    else MI355X_TRY(hipMemcpy(out, src, nb, hipMemcpyDeviceToHost));
  }
  return 0;
}

int mi355x_trisolve_debug_set_aborted(mi355x_trisolve_plan_t p, int value) {
  if (!p || !p->abort_flag) return (int)hipErrorInvalidValue;
  *p->abort_flag = value;
  return 0;
}

int mi355x_trisolve_aborted(mi355x_trisolve_plan_t p, int *aborted) {
  *aborted = p && p->abort_flag ? *p->abort_flag : 0;
  return 0;
}

}  // extern "C"
