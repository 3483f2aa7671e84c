// Column-tiled CSR SpMV for matrices whose x gathers miss the caches: y = A x and z = y + A x (MatMult_SeqAIJ / MatMultAdd_SeqAIJ,
// reference src/mat/impls/aij/seq/aij.c:1225-1358) with the x entries a row panel needs STAGED IN LDS.
//
// Why.  The row-block kernels of spmv_csr.hip gather x through L1 / L2.  On a matrix whose rows pick their columns from a wide window
// without neighbouring rows sharing them (the irregular stand-in of BASELINE configs[3]: 73 entries per row, +-50 000 band, a fifth of the
// entries anywhere) every gather is an L1 miss that moves a whole cache line out of L2 for 8 useful bytes: 120 M L1->L2 requests for 112 M
// nonzeros, 2.5x the algorithmic bytes on the fabric, 0.23 of the HBM roof (profiles/r02_cfg4_irr*).  LDS is the one on-chip memory with
// word-granular random access, so the product is re-cut so that the gathers go there.
//
// Layout (built once per nonzero pattern on host threads, values refreshed on the device through a permutation):
//   * rows in PANELS of TL_WAVES * 64 * TL_RPL rows, one workgroup per panel; columns in TILES of TL_TW (8192) entries of x = 64 KB of LDS;
//   * a (panel, tile) pair with at least `stage_min` entries is STAGED: the workgroup loads that tile of x into LDS once and all of the
//     panel's entries in it gather from there.  The panel walks its staged tiles in ascending order, so a row's products are added in
//     column order.  Entries of pairs too thin to stage (the long-range fifth) stay in a CSR remainder that the row-block kernel adds
//     afterwards (mi355x_spmv_csr_add) -- the split of MatMult_MPIAIJ's diagonal / off-diagonal blocks, inside one GPU;
//   * inside a workgroup every wavefront owns 64 * TL_RPL rows (lane j the rows j, j + 64, ..: the sums stay in registers for the whole
//     panel) and streams its own entries of the current tile in CHUNKS of <= 512: 16-byte non-temporal loads of the values, ONE 16-byte
//     load per lane of eight 2-byte in-tile column numbers, one 8-byte load per lane of its rows' end offsets; products parked in the
//     wavefront's 4 KB of LDS; each lane then adds its rows' products in order.  No workgroup barrier except at a tile switch;
//   * 10 bytes per entry + ~1 (offsets) instead of CSR's 12 + 4 per row.
// Arithmetic: a*x rounded, then added (-ffp-contract=off), staged products of a row in column order, remainder after them: agrees with
// the reference to rounding (tests: <= 1e-12 * sum |a_ij x_j|), run-to-run identical.
#include "common.hpp"
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>
#include <string.h>

#define TL_TW 8192            // columns of x per tile (64 KB of LDS)
#define TL_CH 512             // entries per chunk (the wavefront's LDS product stage: 4 KB)
#ifndef TL_WAVES
#define TL_WAVES 8            // wavefronts per workgroup
#endif
#ifndef TL_RPL
#define TL_RPL 4              // rows per lane
#endif
#define TL_SUB (64 * TL_RPL)              // rows per wavefront
#define TL_PANEL (TL_WAVES * TL_SUB)      // rows per workgroup

typedef double tl_v2d __attribute__((ext_vector_type(2)));
typedef unsigned short tl_us8 __attribute__((ext_vector_type(8)));
typedef unsigned short tl_us4 __attribute__((ext_vector_type(4)));

// what the builder leaves (host) and the plan holds (device); all offsets fit 32 bits (nnz < 2^31 as in the CSR arrays)
struct tl_host {
  int m = 0, n = 0, npanels = 0;
  long nnz = 0, nnz_near = 0, nnz_far = 0;
  std::vector<int> pt_ptr;        // [npanels + 1] -> staged (panel, tile) pairs
  std::vector<int> pt_tile;       // [npt] tile number
  std::vector<int> pt_chunk0;     // [npt * TL_WAVES + 1] first chunk of (pair, wavefront); the next entry ends it
  std::vector<int> chunk_e0;      // [nchunks + 1] first entry of a chunk in val / perm (even)
  std::vector<int> perm;          // [nval] entry -> position in the CSR value array (-1: padding, value 0)
  std::vector<unsigned short> lcol;   // [nchunks][64][8]   lane l holds entries 2 (l + 64 q) + h at [l][2 q + h]
  std::vector<unsigned short> cend;   // [nchunks][64][TL_RPL] end offset (exclusive, <= 512) of row a * 64 + j at [j][a]
  std::vector<int> far_i, far_j, far_perm;   // CSR remainder over all m rows, global columns
};

struct mi355x_spmv_tiled_s {
  tl_host *host;                  // kept until the upload (and for the debug getter)
  int m, n, npanels, nchunks, npt;
  long nnz_near, nnz_far, nval;
  int *d_pt_ptr, *d_pt_tile, *d_pt_chunk0, *d_chunk_e0, *d_perm;
  unsigned short *d_lcol, *d_cend;
  double *d_val;
  int *d_far_i, *d_far_j, *d_far_perm;
  double *d_far_a;
  mi355x_spmv_plan_t far_plan;
};

// ---------------------------------------------------------------------------------------------------------------------------------
// host: the layout
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {
struct PanelOut {
  std::vector<int> pt_tile, pt_chunk0, chunk_ne, perm;
  std::vector<unsigned short> lcol, cend;
  long near = 0;
};

// one panel: which tiles are staged, then the chunks of every (staged tile, wavefront) stream
static void build_panel(int p, int m, int n, const int *ai, const int *aj, int stage_min, std::vector<int> &cnt, std::vector<int> &touched,
                        std::vector<int> &cur, PanelOut &o) {
  const int r0 = p * TL_PANEL, r1 = std::min(m, r0 + TL_PANEL);
  touched.clear();
  for (int r = r0; r < r1; ++r)
    for (int k = ai[r]; k < ai[r + 1]; ++k) { const int t = aj[k] / TL_TW; if (cnt[t]++ == 0) touched.push_back(t); }
  std::sort(touched.begin(), touched.end());
  for (int r = r0; r < r1; ++r) cur[r - r0] = ai[r];
  for (int t : touched) {
    const bool staged = cnt[t] >= stage_min;
    cnt[t] = 0;
    if (!staged) continue;
    const int chi = (t + 1) * TL_TW;          // first column past the tile
    o.pt_tile.push_back(t);
    for (int w = 0; w < TL_WAVES; ++w) {
      o.pt_chunk0.push_back((int)o.chunk_ne.size());
      const int s0 = r0 + w * TL_SUB;
      // the chunk being filled
      int ne = 0;
      size_t lbase = 0, cbase = 0;
      auto open = [&]() {
        ne = 0;
        lbase = o.lcol.size(); o.lcol.resize(lbase + 64 * 8, 0);
        cbase = o.cend.size(); o.cend.resize(cbase + 64 * TL_RPL, 0);
      };
      auto close = [&](int last_row_local) {
        // rows behind the last one that received entries carry the final offset (their segments are empty)
        for (int rl = last_row_local + 1; rl < TL_SUB; ++rl) o.cend[cbase + (size_t)(rl & 63) * TL_RPL + (rl >> 6)] = (unsigned short)ne;
        if (ne & 1) { o.perm.push_back(-1); }   // chunks start on even entries (16-byte value loads)
        o.chunk_ne.push_back(ne);
      };
      bool is_open = false;
      int last_rl = -1;
      for (int rl = 0; rl < TL_SUB; ++rl) {
        const int r = s0 + rl;
        if (r >= r1) break;
        int k = cur[r - r0];
        const int kend = ai[r + 1];
        while (k < kend && aj[k] < t * TL_TW) ++k;                 // entries of thinner tiles in between: the remainder's
        while (k < kend && aj[k] < chi) {
          if (!is_open) { open(); is_open = true; last_rl = -1; }
          // rows skipped since the last entry end where the chunk stood
          for (int q = last_rl + 1; q < rl; ++q) o.cend[cbase + (size_t)(q & 63) * TL_RPL + (q >> 6)] = (unsigned short)ne;
          const int pi = ne >> 1, lane = pi & 63, qq = pi >> 6;
          o.lcol[lbase + (size_t)lane * 8 + 2 * qq + (ne & 1)] = (unsigned short)(aj[k] - t * TL_TW);
          o.perm.push_back(k);
          ++ne; ++k; ++o.near;
          last_rl = rl;
          o.cend[cbase + (size_t)(rl & 63) * TL_RPL + (rl >> 6)] = (unsigned short)ne;
          if (ne == TL_CH) { close(rl); is_open = false; }
        }
        cur[r - r0] = k;
      }
      if (is_open) close(last_rl);
    }
  }
  o.pt_chunk0.push_back((int)o.chunk_ne.size());    // end of the last (pair, wavefront)
}
}  // namespace

extern "C" {

// fraction of a sample of 32-row groups' gathers that touch a cache line (128 B of x) no other gather of the group touches: near 1 =
// every gather its own line (the row-block kernels then move a line per nonzero), small = neighbouring rows share their columns
int mi355x_spmv_tiled_probe(int m, const int *ai, const int *aj, double *lines_per_nonzero) {
  *lines_per_nonzero = 0.0;
  if (m <= 0 || ai[m] <= 0) return 0;
  const int G = 32, ngroups = (m + G - 1) / G, nsample = std::min(ngroups, 4096);
  long lines = 0, nz = 0;
  std::vector<int> buf;
  for (int s = 0; s < nsample; ++s) {
    const int g = (int)((long)s * ngroups / nsample);
    const int r0 = g * G, r1 = std::min(m, r0 + G);
    buf.clear();
    for (int k = ai[r0]; k < ai[r1]; ++k) buf.push_back(aj[k] >> 4);
    nz += (long)buf.size();
    std::sort(buf.begin(), buf.end());
    lines += (long)(std::unique(buf.begin(), buf.end()) - buf.begin());
  }
  *lines_per_nonzero = nz ? (double)lines / (double)nz : 0.0;
  return 0;
}

// Host part: cut the CSR matrix into staged (panel, tile) streams + remainder.  No device call.  stage_min <= 0: the default (1024).
int mi355x_spmv_tiled_build(int m, int n, const int *ai, const int *aj, int stage_min, mi355x_spmv_tiled_t *out) {
  *out = nullptr;
  if (m < 0 || n < 0) return (int)hipErrorInvalidValue;
  if (stage_min <= 0) stage_min = 1024;
  tl_host *H = new tl_host();
  H->m = m; H->n = n; H->nnz = m ? ai[m] : 0;
  H->npanels = (m + TL_PANEL - 1) / TL_PANEL;
  const int ntiles = (n + TL_TW - 1) / TL_TW;
  std::vector<PanelOut> po((size_t)H->npanels);
  {
    int nth = mi355x_host_threads(16);
    if (H->nnz < 2000000) nth = 1;
    if (nth > H->npanels) nth = H->npanels > 0 ? H->npanels : 1;
    std::atomic<int> next(0);
    auto work = [&]() {
      std::vector<int> cnt((size_t)ntiles + 1, 0), touched, cur((size_t)TL_PANEL, 0);
      for (int p = next.fetch_add(1); p < H->npanels; p = next.fetch_add(1)) build_panel(p, m, n, ai, aj, stage_min, cnt, touched, cur, po[(size_t)p]);
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nth; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
  }
  // concatenate in panel order
  size_t npt = 0, nch = 0, nval = 0;
  for (auto &o : po) { npt += o.pt_tile.size(); nch += o.chunk_ne.size(); nval += o.perm.size(); H->nnz_near += o.near; }
  H->pt_ptr.resize((size_t)H->npanels + 1);
  H->pt_tile.reserve(npt); H->pt_chunk0.reserve(npt * TL_WAVES + 1); H->chunk_e0.reserve(nch + 1); H->perm.reserve(nval + 2);
  H->lcol.reserve(nch * 64 * 8); H->cend.reserve(nch * 64 * TL_RPL);
  int e = 0;
  for (int p = 0; p < H->npanels; ++p) {
    PanelOut &o = po[(size_t)p];
    H->pt_ptr[(size_t)p] = (int)H->pt_tile.size();
    const int cbase = (int)H->chunk_e0.size();
    H->pt_tile.insert(H->pt_tile.end(), o.pt_tile.begin(), o.pt_tile.end());
    for (size_t i = 0; i + 1 < o.pt_chunk0.size(); ++i) H->pt_chunk0.push_back(cbase + o.pt_chunk0[i]);
    for (int ne : o.chunk_ne) { H->chunk_e0.push_back(e); e += ne + (ne & 1); }
    H->perm.insert(H->perm.end(), o.perm.begin(), o.perm.end());
    H->lcol.insert(H->lcol.end(), o.lcol.begin(), o.lcol.end());
    H->cend.insert(H->cend.end(), o.cend.begin(), o.cend.end());
    PanelOut().pt_tile.swap(o.pt_tile); std::vector<int>().swap(o.perm); std::vector<unsigned short>().swap(o.lcol); std::vector<unsigned short>().swap(o.cend);
  }
  H->pt_ptr[(size_t)H->npanels] = (int)H->pt_tile.size();
  H->pt_chunk0.push_back((int)H->chunk_e0.size());
  H->chunk_e0.push_back(e);
  // remainder: every entry no stream took, rows in order (a second walk with the same staging decisions)
  H->far_i.assign((size_t)m + 1, 0);
  H->nnz_far = H->nnz - H->nnz_near;
  H->far_j.reserve((size_t)H->nnz_far); H->far_perm.reserve((size_t)H->nnz_far);
  {
    std::vector<char> staged((size_t)ntiles + 1, 0);
    for (int p = 0; p < H->npanels; ++p) {
      for (int i = H->pt_ptr[(size_t)p]; i < H->pt_ptr[(size_t)p + 1]; ++i) staged[(size_t)H->pt_tile[(size_t)i]] = 1;
      const int r0 = p * TL_PANEL, r1 = std::min(m, r0 + TL_PANEL);
      for (int r = r0; r < r1; ++r) {
        for (int k = ai[r]; k < ai[r + 1]; ++k) if (!staged[(size_t)(aj[k] / TL_TW)]) { H->far_j.push_back(aj[k]); H->far_perm.push_back(k); }
        H->far_i[(size_t)r + 1] = (int)H->far_j.size();
      }
      for (int i = H->pt_ptr[(size_t)p]; i < H->pt_ptr[(size_t)p + 1]; ++i) staged[(size_t)H->pt_tile[(size_t)i]] = 0;
    }
  }
  if ((long)H->far_j.size() != H->nnz_far) { delete H; return (int)hipErrorUnknown; }
  mi355x_spmv_tiled_s *P = new mi355x_spmv_tiled_s();
  memset(P, 0, sizeof(*P));
  P->host = H;
  P->m = m; P->n = n; P->npanels = H->npanels; P->nchunks = (int)H->chunk_e0.size() - 1; P->npt = (int)H->pt_tile.size();
  P->nnz_near = H->nnz_near; P->nnz_far = H->nnz_far; P->nval = (long)H->perm.size();
  *out = P;
  return 0;
}

int mi355x_spmv_tiled_info(mi355x_spmv_tiled_t P, long *nnz_staged, long *nnz_remainder, int *npanels, int *npairs, int *nchunks) {
  if (nnz_staged) *nnz_staged = P->nnz_near;
  if (nnz_remainder) *nnz_remainder = P->nnz_far;
  if (npanels) *npanels = P->npanels;
  if (npairs) *npairs = P->npt;
  if (nchunks) *nchunks = P->nchunks;
  return 0;
}
int mi355x_spmv_tiled_geometry(int *panel_rows, int *tile_cols, int *waves, int *rows_per_lane, int *chunk) {
  *panel_rows = TL_PANEL; *tile_cols = TL_TW; *waves = TL_WAVES; *rows_per_lane = TL_RPL; *chunk = TL_CH;
  return 0;
}

// tests: one host array of the layout (which: 0 pt_ptr, 1 pt_tile, 2 pt_chunk0, 3 chunk_e0, 4 perm, 5 lcol, 6 cend, 7 far_i, 8 far_j, 9 far_perm);
// available until mi355x_spmv_tiled_drop_host
int mi355x_spmv_tiled_debug_get(mi355x_spmv_tiled_t P, int which, void *out, size_t cap_bytes, size_t *bytes) {
  if (!P->host) return (int)hipErrorInvalidValue;
  tl_host *H = P->host;
  const void *src = nullptr; size_t nb = 0;
  switch (which) {
    case 0: src = H->pt_ptr.data(); nb = H->pt_ptr.size() * 4; break;
    case 1: src = H->pt_tile.data(); nb = H->pt_tile.size() * 4; break;
    case 2: src = H->pt_chunk0.data(); nb = H->pt_chunk0.size() * 4; break;
    case 3: src = H->chunk_e0.data(); nb = H->chunk_e0.size() * 4; break;
    case 4: src = H->perm.data(); nb = H->perm.size() * 4; break;
    case 5: src = H->lcol.data(); nb = H->lcol.size() * 2; break;
    case 6: src = H->cend.data(); nb = H->cend.size() * 2; break;
    case 7: src = H->far_i.data(); nb = H->far_i.size() * 4; break;
    case 8: src = H->far_j.data(); nb = H->far_j.size() * 4; break;
    case 9: src = H->far_perm.data(); nb = H->far_perm.size() * 4; break;
    default: return (int)hipErrorInvalidValue;
  }
  *bytes = nb;
  if (out && nb <= cap_bytes && nb) memcpy(out, src, nb);
  return 0;
}
int mi355x_spmv_tiled_drop_host(mi355x_spmv_tiled_t P) { delete P->host; P->host = nullptr; return 0; }

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------------------------------------------
// val[k] = aa[perm[k]] (0 for padding): the layout's values out of the CSR array that is on the device anyway
__global__ __launch_bounds__(256) void tl_gather_values_kernel(const int *__restrict__ perm, const double *__restrict__ aa, double *__restrict__ val, long n) {
  const long stride = (long)gridDim.x * 256;
  for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < n; k += stride) { const int q = perm[k]; val[k] = q >= 0 ? aa[q] : 0.0; }
}

template <int ADD>
__global__ __launch_bounds__(TL_WAVES * 64) void spmv_tiled_kernel(
    int npanels, int chunkx, const int *__restrict__ pt_ptr, const int *__restrict__ pt_tile, const int *__restrict__ pt_chunk0,
    const int *__restrict__ chunk_e0, const double *__restrict__ val, const unsigned short *__restrict__ lcol,
    const unsigned short *__restrict__ cend, const double *__restrict__ x, const double *yin, double *yout, int m, int n) {
  extern __shared__ __attribute__((aligned(16))) double tl_lds[];
  double *xt = tl_lds;                                   // TL_TW doubles: the tile of x
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  double *stage = tl_lds + TL_TW + w * TL_CH;            // this wavefront's products

  // each XCD walks a contiguous eighth of the panels: neighbouring panels stage the same tiles, out of the same L2
  const int xcd = blockIdx.x % MI355X_NXCD, slot = blockIdx.x / MI355X_NXCD;
  const int p = xcd * chunkx + slot;
  if (slot >= chunkx || p >= npanels) return;

  const int row0 = p * TL_PANEL + w * TL_SUB;            // lane's rows: row0 + a * 64 + lane
  double acc[TL_RPL];
#pragma unroll
  for (int a = 0; a < TL_RPL; ++a) {
    const int r = row0 + a * 64 + lane;
    acc[a] = (ADD && r < m) ? yin[r] : 0.0;
  }

  const int pt0 = pt_ptr[p], pt1 = pt_ptr[p + 1];
  for (int pt = pt0; pt < pt1; ++pt) {
    const int t = pt_tile[pt];
    const int c0 = pt_chunk0[pt * TL_WAVES + w], c1 = pt_chunk0[pt * TL_WAVES + w + 1];
    __syncthreads();                                     // the previous tile is not read any more
    {
      const size_t base = (size_t)t * TL_TW;
      const int ncol = (n - (long)base) < TL_TW ? (int)(n - (long)base) : TL_TW;
      const tl_v2d *xs = reinterpret_cast<const tl_v2d *>(x + base);
      tl_v2d *xd = reinterpret_cast<tl_v2d *>(xt);
      for (int i = tid; i < (ncol >> 1); i += TL_WAVES * 64) xd[i] = xs[i];
      if ((ncol & 1) && tid == 0) xt[ncol - 1] = x[base + ncol - 1];
    }
    __syncthreads();
    for (int c = c0; c < c1; ++c) {
      const int e0 = chunk_e0[c], ne = chunk_e0[c + 1] - e0;        // ne even; the chunk's real entries: its last end offset
      const tl_us8 lc = __builtin_nontemporal_load(reinterpret_cast<const tl_us8 *>(lcol + (size_t)c * 512) + lane);
      const tl_us4 ce = __builtin_nontemporal_load(reinterpret_cast<const tl_us4 *>(cend + (size_t)c * (64 * TL_RPL)) + lane);
      tl_v2d v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = 2 * (lane + 64 * q);
        v[q] = __builtin_nontemporal_load(reinterpret_cast<const tl_v2d *>(val + e0 + (k < ne ? k : 0)));
      }
      // gather from the tile, multiply, park (entries past the chunk's end multiply padding by x[column 0]: never read back)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double xa = xt[lc[2 * q]], xb = xt[lc[2 * q + 1]];
        tl_v2d pr; pr.x = v[q].x * xa; pr.y = v[q].y * xb;
        *reinterpret_cast<tl_v2d *>(stage + 2 * (lane + 64 * q)) = pr;
      }
      // the wavefront's own LDS writes, read back by other lanes of the same wavefront: order them, no workgroup barrier
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // row a * 64 + lane of this wavefront: products [start, end) where start = the end of the row before it
      int en[TL_RPL], st[TL_RPL];
#pragma unroll
      for (int a = 0; a < TL_RPL; ++a) en[a] = ce[a];
#pragma unroll
      for (int a = 0; a < TL_RPL; ++a) {
        const int up = __shfl_up(en[a], 1, 64);
        const int prev_last = a ? __shfl(en[a - 1], 63, 64) : 0;
        st[a] = lane ? up : prev_last;
      }
      int longest = 0;
#pragma unroll
      for (int a = 0; a < TL_RPL; ++a) longest = max(longest, en[a] - st[a]);
      for (int off = 32; off > 0; off >>= 1) longest = max(longest, __shfl_xor(longest, off, 64));
      for (int i = 0; i < longest; ++i) {
        double pv[TL_RPL];
#pragma unroll
        for (int a = 0; a < TL_RPL; ++a) pv[a] = stage[(st[a] + i < en[a]) ? st[a] + i : 0];
#pragma unroll
        for (int a = 0; a < TL_RPL; ++a) { const double u = acc[a] + pv[a]; acc[a] = (st[a] + i < en[a]) ? u : acc[a]; }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the next chunk overwrites the stage
      __builtin_amdgcn_wave_barrier();
    }
  }
#pragma unroll
  for (int a = 0; a < TL_RPL; ++a) {
    const int r = row0 + a * 64 + lane;
    if (r < m) yout[r] = acc[a];
  }
}

extern "C" {

// device part: tables up, values gathered from the CSR value array on the device (aa_dev), remainder with a row-block plan of its own
int mi355x_spmv_tiled_upload(mi355x_handle_t h, mi355x_spmv_tiled_t P, const double *aa_dev) {
  tl_host *H = P->host;
  if (!H) return (int)hipErrorInvalidValue;
  auto up_i = [&](int **d, const std::vector<int> &v) -> int {
    MI355X_TRY(hipMalloc((void **)d, sizeof(int) * (v.size() ? v.size() : 1) + 16));
    if (v.size()) MI355X_TRY(hipMemcpyAsync(*d, v.data(), sizeof(int) * v.size(), hipMemcpyHostToDevice, h->stream));
    return 0;
  };
  auto up_s = [&](unsigned short **d, const std::vector<unsigned short> &v) -> int {
    MI355X_TRY(hipMalloc((void **)d, sizeof(unsigned short) * (v.size() ? v.size() : 1) + 16));
    if (v.size()) MI355X_TRY(hipMemcpyAsync(*d, v.data(), sizeof(unsigned short) * v.size(), hipMemcpyHostToDevice, h->stream));
    return 0;
  };
  int rc;
  if ((rc = up_i(&P->d_pt_ptr, H->pt_ptr)) || (rc = up_i(&P->d_pt_tile, H->pt_tile)) || (rc = up_i(&P->d_pt_chunk0, H->pt_chunk0)) ||
      (rc = up_i(&P->d_chunk_e0, H->chunk_e0)) || (rc = up_i(&P->d_perm, H->perm)) || (rc = up_s(&P->d_lcol, H->lcol)) ||
      (rc = up_s(&P->d_cend, H->cend)) || (rc = up_i(&P->d_far_i, H->far_i)) || (rc = up_i(&P->d_far_j, H->far_j)) ||
      (rc = up_i(&P->d_far_perm, H->far_perm)))
    return rc;
  MI355X_TRY(hipMalloc((void **)&P->d_val, sizeof(double) * (size_t)(P->nval > 0 ? P->nval : 1) + 16 * 4));
  MI355X_TRY(hipMalloc((void **)&P->d_far_a, sizeof(double) * (size_t)(P->nnz_far > 0 ? P->nnz_far : 1) + 16));
  if (P->nnz_far > 0) { rc = mi355x_spmv_plan_create(h, P->m, H->far_i.data(), nullptr, &P->far_plan); if (rc) return rc; }
  MI355X_TRY(hipStreamSynchronize(h->stream));
  return mi355x_spmv_tiled_refresh_values(h, P, aa_dev);
}

// the CSR values on the device changed (same pattern): one gather per part
int mi355x_spmv_tiled_refresh_values(mi355x_handle_t h, mi355x_spmv_tiled_t P, const double *aa_dev) {
  if (P->nval > 0) {
    hipLaunchKernelGGL(tl_gather_values_kernel, dim3(mi355x_grid_for((size_t)P->nval, 4)), dim3(256), 0, h->stream, P->d_perm, aa_dev, P->d_val, P->nval);
    MI355X_LAUNCH_CHECK();
  }
  if (P->nnz_far > 0) {
    hipLaunchKernelGGL(tl_gather_values_kernel, dim3(mi355x_grid_for((size_t)P->nnz_far, 4)), dim3(256), 0, h->stream, P->d_far_perm, aa_dev, P->d_far_a, P->nnz_far);
    MI355X_LAUNCH_CHECK();
  }
  return 0;
}

// y = A x (yin == NULL) or yout = yin + A x (yout may alias yin).  x must be 16-byte aligned (hipErrorNotSupported otherwise: the caller
// takes the row-block kernel).  which: 0 both parts, 1 the staged part only, 2 the remainder only (development: their separate cost)
int mi355x_spmv_tiled_parts(mi355x_handle_t h, mi355x_spmv_tiled_t P, const double *x, const double *yin, double *yout, int which) {
  if (!mi355x_aligned16(x)) return (int)hipErrorNotSupported;
  if (P->m == 0) return 0;
  const size_t lds = sizeof(double) * (TL_TW + TL_WAVES * TL_CH);
  static bool attr_set = false;
  if (!attr_set) {
    MI355X_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(spmv_tiled_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    MI355X_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(spmv_tiled_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int chunkx = (P->npanels + MI355X_NXCD - 1) / MI355X_NXCD;
  const int grid = chunkx * MI355X_NXCD;
  if (which != 2) {
    if (yin)
      hipLaunchKernelGGL((spmv_tiled_kernel<1>), dim3(grid), dim3(TL_WAVES * 64), lds, h->stream, P->npanels, chunkx, P->d_pt_ptr, P->d_pt_tile, P->d_pt_chunk0,
                         P->d_chunk_e0, P->d_val, P->d_lcol, P->d_cend, x, yin, yout, P->m, P->n);
    else
      hipLaunchKernelGGL((spmv_tiled_kernel<0>), dim3(grid), dim3(TL_WAVES * 64), lds, h->stream, P->npanels, chunkx, P->d_pt_ptr, P->d_pt_tile, P->d_pt_chunk0,
                         P->d_chunk_e0, P->d_val, P->d_lcol, P->d_cend, x, (const double *)nullptr, yout, P->m, P->n);
    MI355X_LAUNCH_CHECK();
  }
  if (which != 1 && P->nnz_far > 0) return mi355x_spmv_csr_add(h, P->far_plan, P->d_far_i, P->d_far_j, P->d_far_a, x, yout, yout);
  return 0;
}
int mi355x_spmv_tiled(mi355x_handle_t h, mi355x_spmv_tiled_t P, const double *x, const double *yin, double *yout) {
  return mi355x_spmv_tiled_parts(h, P, x, yin, yout, 0);
}

int mi355x_spmv_tiled_destroy(mi355x_spmv_tiled_t P) {
  if (!P) return 0;
  delete P->host;
  void *ptrs[] = {P->d_pt_ptr, P->d_pt_tile, P->d_pt_chunk0, P->d_chunk_e0, P->d_perm, P->d_lcol, P->d_cend, P->d_val, P->d_far_i, P->d_far_j, P->d_far_perm, P->d_far_a};
  for (void *q : ptrs) if (q) hipFree(q);
  if (P->far_plan) mi355x_spmv_plan_destroy(P->far_plan);
  delete P;
  return 0;
}

}  // extern "C"
