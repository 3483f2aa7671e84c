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
//   * rows in PANELS of TL_PANEL = 2048 rows, one workgroup per panel; columns in TILES of TL_TW (4096) entries of x = 32 KB of LDS, two of
//     them resident (the next tile arrives while the current one is gathered from);
//   * a (panel, tile) pair with at least `stage_min` entries is STAGED: the workgroup loads that tile of x into LDS once and all of the
//     panel's entries in it gather from there.  The panel walks its staged tiles in ascending order.  Entries of pairs too thin to
//     stage (the long-range fifth) stay in a CSR remainder that the row-block kernel adds afterwards (mi355x_spmv_csr_add) -- the split
//     of MatMult_MPIAIJ's diagonal / off-diagonal blocks, inside one GPU;
//   * per staged tile, ONE LANE PER ROW: the panel's rows with entries in the tile are sorted by their number of entries there
//     (descending) and cut into ROUNDS of 64 consecutive ranks -- rows of (nearly) equal length -- dealt to the workgroup's wavefronts in
//     turn (TL_RPL = 4 rounds each).  A round is stored as jagged diagonals: step j holds entry j of every row of the round that has more
//     than j entries -- the rows being sorted, those are lanes 0 .. n_j - 1 -- so a step is ONE coalesced load of n_j values and n_j
//     2-byte in-tile column numbers with no padding, n_j comes out of a ballot of the lanes' own counts, and a lane meets its row's
//     entries in column order: x from the LDS tile, multiply, add to the row's running sum.  No product stage, no cross-lane traffic,
//     no barrier except at a tile switch.  The running sums of the panel's rows live in LDS between tiles (the row <-> lane assignment
//     changes with the tile);
//   * per (wavefront, tile) one 16-byte word per lane says which rows it serves and how many entries each has: ~0.9 B per entry on the
//     stand-in; with 8 B values and 2 B columns 10.9 B per entry instead of CSR's 12 + 4 per row.
// Arithmetic: a*x rounded, then added (-ffp-contract=off); a row's staged products are added one after the other in column order
// starting from 0 (or y), the remainder after them (that part's rows of more than 16 entries by a tree): agrees with the reference to
// rounding (tests: <= 1e-12 * sum |a_ij x_j|), bit for bit when nothing is left to the remainder, and run-to-run identical.
#include "common.hpp"
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>

#ifndef TL_TW
#define TL_TW 4096            // columns of x per tile (32 KB of LDS; two buffers)
#endif
#ifndef TL_WAVES
#define TL_WAVES 8            // gathering wavefronts per workgroup
#endif
#define TL_RPL 4              // rounds per (wavefront, tile): rows per lane
#define TL_PANEL (TL_WAVES * 64 * TL_RPL)  // most rows a workgroup takes: every row of the panel is some lane's in some round, whatever the tile
#define TL_MAX_PASS 16                     // column ranges of the remainder
#define TL_PASS_BYTES (3u << 20)           // ... each covering <= 3 MiB of x: it stays in one XCD's 4 MiB L2 next to the streams passing through
#define TL_CNT_BITS 16                     // a descriptor word: (row of the panel << 16) | entries of the row in the tile (<= TL_TW)
static_assert(TL_TW < (1 << TL_CNT_BITS) && TL_PANEL <= (1 << (32 - TL_CNT_BITS)), "descriptor word");
#ifndef TL_U
#define TL_U 4                // steps per group: the loads a lane issues together
#endif
#ifndef TL_NG
#define TL_NG 3               // groups in flight per wavefront
#endif
#define TL_TRIPW (TL_NG * TL_U / 2)        // 32-bit words of step descriptors per trip of the kernel's loop (two steps per word)
static_assert((TL_U == 4 || TL_U == 8) && TL_TRIPW <= 64, "step words: a group is 2 or 4 whole 32-bit words, a trip's words one per lane");
#define TL_STEP_NEWTILE 0x8000u            // step word: first step of the wavefront's first group in the panel's next staged tile
#define TL_STEP_ROUNDEND 0x4000u           // ... last step of a round: the lanes' rows are complete for this tile, the next round's rows follow
#define TL_LDS_BYTES (8 * (2 * TL_TW + TL_PANEL))
#define TL_NCU 256
#define TL_WG_PER_CU ((160 * 1024 / TL_LDS_BYTES) < (32 / (TL_WAVES + 1)) ? (160 * 1024 / TL_LDS_BYTES) : (32 / (TL_WAVES + 1)))
static_assert(TL_WG_PER_CU >= 1, "one workgroup must fit a CU");

typedef double tl_v2d __attribute__((ext_vector_type(2)));
typedef unsigned int tl_u4 __attribute__((ext_vector_type(4)));

// what the builder leaves (host) and the plan holds (device); all offsets fit 32 bits (nnz < 2^31 as in the CSR arrays)
struct tl_host {
  int m = 0, n = 0, npanels = 0;
  long nnz = 0, nnz_near = 0, nnz_far = 0, nsteps = 0;
  std::vector<int> prow;          // [npanels + 1] first row of a panel (<= TL_PANEL rows each, cut so that the panels hold equal shares of the nonzeros)
  std::vector<int> pt_ptr;        // [npanels + 1] -> staged (panel, tile) pairs
  std::vector<int> pt_tile;       // [npt] tile number
  // a wavefront's stream runs through ALL staged tiles of its panel without a gap: (panel, wavefront)-major, tiles ascending inside
  std::vector<int> pw_e0;         // [npanels * TL_WAVES + 1] first entry of (panel, wavefront) in val / lcol / perm
  std::vector<int> pw_s0;         // [npanels * TL_WAVES + 1] first step word of (panel, wavefront) in `steps` (a multiple of TL_U)
  std::vector<unsigned int> desc; // [(pt_ptr[p] * TL_WAVES + w * ntiles(p) + i)][64][TL_RPL]: in tile i of the panel, in its round a, lane l of wavefront w serves row (word >> 16) of the panel, which has (word & 0xffff) entries in the tile
  std::vector<unsigned short> steps;   // one word per jagged diagonal: active lanes (1..64; 0 = padding), TL_STEP_ROUNDEND on a round's last
                                       // step (a tile's rounds follow each other, 0 .. 3, empty ones last and absent); a wavefront's steps of
                                       // one tile are padded to whole groups of TL_U (at least one group per tile) and the first word of the
                                       // tile's first group carries TL_STEP_NEWTILE
  std::vector<int> perm;          // [nnz_near] entry -> position in the CSR value array
  std::vector<unsigned short> lcol;   // [nnz_near] column - tile * TL_TW
  int npass = 1;                  // the remainder is cut into column ranges applied one after the other (x of one range stays in every XCD's L2)
  std::vector<int> far_i, far_j, far_perm;   // CSR remainder, pass-major: far_i[q * (m + 1) + r] .. [.. + r + 1] = row r's entries of pass q in far_j / far_perm (absolute positions), global columns
};

struct mi355x_spmv_tiled_s {
  tl_host *host;                  // kept until _drop_host (tests read it back)
  int m, n, npanels, npt;
  long nnz_near, nnz_far, nsteps;
  int *d_prow, *d_pt_ptr, *d_pt_tile, *d_pw_e0, *d_pw_s0, *d_perm;
  unsigned int *d_desc;
  unsigned short *d_lcol, *d_steps;
  double *d_val;
  int npass;
  long nfar_store;               // entries of the remainder's arrays (nnz_far + padding between passes)
  int *d_far_i, *d_far_j, *d_far_perm;
  double *d_far_a;
  mi355x_spmv_plan_t far_plan[TL_MAX_PASS];
};

// ---------------------------------------------------------------------------------------------------------------------------------
// host: the layout
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {
struct PanelOut {
  std::vector<int> pt_tile;
  std::vector<int> perm[TL_WAVES];
  std::vector<unsigned int> desc[TL_WAVES];
  std::vector<unsigned short> lcol[TL_WAVES], steps[TL_WAVES];
  long near = 0, nsteps = 0;
};
struct RowSeg { int rl, k0, cnt; };

// one panel (rows r0 .. r1 - 1): which tiles are staged, then per (wavefront, staged tile) the rounds of jagged diagonals
static void build_panel(int r0, int r1, int n, const int *ai, const int *aj, int stage_min, std::vector<int> &cnt, std::vector<int> &touched,
                        std::vector<int> &cur, PanelOut &o) {
  (void)n;
  touched.clear();
  for (int r = r0; r < r1; ++r)
    for (int k = ai[r]; k < ai[r + 1]; ++k) { const int t = aj[k] / TL_TW; if (cnt[t]++ == 0) touched.push_back(t); }
  std::sort(touched.begin(), touched.end());
  for (int r = r0; r < r1; ++r) cur[r - r0] = ai[r];
  std::vector<RowSeg> segs;
  segs.reserve(TL_PANEL);
  for (int t : touched) {
    const bool staged = cnt[t] >= stage_min;
    cnt[t] = 0;
    if (!staged) continue;
    const int clo = t * TL_TW, chi = clo + TL_TW;
    o.pt_tile.push_back(t);
    // the panel's rows with entries in this tile, longest first (ties: lower row first)
    segs.clear();
    for (int rl = 0; rl < r1 - r0; ++rl) {
      const int r = r0 + rl;
      int k = cur[rl];
      const int kend = ai[r + 1];
      while (k < kend && aj[k] < clo) ++k;                 // entries of thinner tiles in between: the remainder's
      const int kb = k;
      while (k < kend && aj[k] < chi) ++k;
      cur[rl] = k;
      if (k > kb) segs.push_back({rl, kb, k - kb});
    }
    std::stable_sort(segs.begin(), segs.end(), [](const RowSeg &a, const RowSeg &b) { return a.cnt > b.cnt; });
    // rounds of 64 consecutive ranks; round g goes to wavefront g % TL_WAVES as its round g / TL_WAVES
    for (int w = 0; w < TL_WAVES; ++w) {
      std::vector<unsigned int> &D = o.desc[w];
      std::vector<unsigned short> &S = o.steps[w];
      const size_t dbase = D.size(), sbase = S.size();
      D.resize(dbase + 64 * TL_RPL, 0u);
      int ns = 0;
      for (int a = 0; a < TL_RPL; ++a) {
        const size_t g = (size_t)a * TL_WAVES + w;
        const size_t lo = g * 64, hi = std::min(segs.size(), lo + 64);
        if (lo >= hi) break;
        for (size_t q = lo; q < hi; ++q) D[dbase + (q - lo) * TL_RPL + a] = ((unsigned int)segs[q].rl << TL_CNT_BITS) | (unsigned int)segs[q].cnt;
        const int maxc = segs[lo].cnt;
        for (int j = 0; j < maxc; ++j) {
          int nact = 0;
          for (size_t q = lo; q < hi && segs[q].cnt > j; ++q) {
            const int k = segs[q].k0 + j;
            o.perm[w].push_back(k);
            o.lcol[w].push_back((unsigned short)(aj[k] - clo));
            ++nact;
          }
          S.push_back((unsigned short)(nact | (j + 1 == maxc ? TL_STEP_ROUNDEND : 0u)));
          ++ns;
          o.near += nact;
        }
        o.nsteps += maxc;
      }
      if (ns == 0 || ns % TL_U) do { S.push_back(0); ++ns; } while (ns % TL_U);   // whole groups, and at least one per tile (the wavefront meets every tile switch)
      S[sbase] = (unsigned short)(S[sbase] | TL_STEP_NEWTILE);
    }
  }
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
  // panels: equal shares of the nonzeros (<= TL_PANEL rows each), and a whole number of rounds of the chip's workgroup slots when the
  // matrix is large enough to fill them more than once -- a last round that fills a fraction of the slots costs as much as a full one
  {
    const int slots = TL_NCU * TL_WG_PER_CU;
    long np = ((long)m + TL_PANEL - 1) / TL_PANEL;
    if (np > slots) np = (np + slots - 1) / slots * slots;
    const char *e_ = getenv("MI355X_TILED_PANELS");
    if (e_ && atol(e_) > 0) np = atol(e_);
    if (np < 1) np = 1;
    H->prow.push_back(0);
    int r = 0;
    for (long p = 0; r < m; ++p) {
      int end;
      if (H->nnz > 0 && p + 1 < np) {
        const long target = (long)(((__int128)H->nnz * (p + 1) + np - 1) / np);      // nonzeros the first p + 1 panels should hold
        end = (int)(std::lower_bound(ai + r, ai + m + 1, target, [](int v, long t) { return (long)v < t; }) - ai);
      } else end = (H->nnz > 0) ? m : (int)std::min<long>(m, ((long)m * (p + 1) + np - 1) / np);
      if (end <= r) end = r + 1;
      if (end > r + TL_PANEL) end = r + TL_PANEL;
      if (end > m) end = m;
      H->prow.push_back(end);
      r = end;
    }
    H->npanels = (int)H->prow.size() - 1;
  }
  const int ntiles = (n + TL_TW - 1) / TL_TW;
  std::vector<PanelOut> po((size_t)H->npanels);
  {
    int nth = mi355x_host_threads(16);
    if (H->nnz < 2000000) nth = 1;
    if (nth > H->npanels) nth = H->npanels > 0 ? H->npanels : 1;
    std::atomic<int> next(0);
    auto work = [&]() {
      std::vector<int> cnt((size_t)ntiles + 1, 0), touched, cur((size_t)TL_PANEL, 0);
      for (int p = next.fetch_add(1); p < H->npanels; p = next.fetch_add(1))
        build_panel(H->prow[(size_t)p], H->prow[(size_t)p + 1], n, ai, aj, stage_min, cnt, touched, cur, po[(size_t)p]);
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nth; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
  }
  // concatenate: panel after panel, inside a panel wavefront after wavefront
  size_t npt = 0, nval = 0, nsw = 0;
  for (auto &o : po) {
    npt += o.pt_tile.size(); H->nnz_near += o.near; H->nsteps += o.nsteps;
    for (int w = 0; w < TL_WAVES; ++w) { nval += o.perm[w].size(); nsw += o.steps[w].size(); }
  }
  H->pt_ptr.resize((size_t)H->npanels + 1);
  H->pt_tile.reserve(npt); H->pw_e0.reserve((size_t)H->npanels * TL_WAVES + 1); H->pw_s0.reserve((size_t)H->npanels * TL_WAVES + 1);
  H->desc.reserve(npt * TL_WAVES * 64 * TL_RPL); H->perm.reserve(nval); H->lcol.reserve(nval + 8); H->steps.reserve(nsw + 8);
  long e = 0, sw = 0;
  for (int p = 0; p < H->npanels; ++p) {
    PanelOut &o = po[(size_t)p];
    H->pt_ptr[(size_t)p] = (int)H->pt_tile.size();
    H->pt_tile.insert(H->pt_tile.end(), o.pt_tile.begin(), o.pt_tile.end());
    for (int w = 0; w < TL_WAVES; ++w) {
      H->pw_e0.push_back((int)e); e += (long)o.perm[w].size();
      while (o.steps[w].size() % (TL_NG * TL_U)) o.steps[w].push_back(0);       // whole trips of the kernel's loop
      H->pw_s0.push_back((int)sw); sw += (long)o.steps[w].size();
      H->steps.insert(H->steps.end(), o.steps[w].begin(), o.steps[w].end());
      H->desc.insert(H->desc.end(), o.desc[w].begin(), o.desc[w].end());
      H->perm.insert(H->perm.end(), o.perm[w].begin(), o.perm[w].end());
      H->lcol.insert(H->lcol.end(), o.lcol[w].begin(), o.lcol[w].end());
      std::vector<unsigned short>().swap(o.steps[w]); std::vector<int>().swap(o.perm[w]);
      std::vector<unsigned short>().swap(o.lcol[w]); std::vector<unsigned int>().swap(o.desc[w]);
    }
  }
  H->pt_ptr[(size_t)H->npanels] = (int)H->pt_tile.size();
  H->pw_e0.push_back((int)e);
  H->pw_s0.push_back((int)sw);
  // remainder: every entry no stream took (a second walk with the same staging decisions), cut into column ranges of <= 3 MiB of x that
  // are applied one after the other: the gathers of one pass then hit the L2 of whichever XCD issues them instead of going out to the
  // Infinity Cache for every one (the remainder is what is scattered over all of x).  Within a pass rows in order, columns ascending;
  // a row's remainder products are still added in column order, pass after pass.
  {
    int np = (int)(((size_t)n * sizeof(double) + TL_PASS_BYTES - 1) / TL_PASS_BYTES);
    const char *e_ = getenv("MI355X_TILED_FAR_PASSES");
    if (e_ && atoi(e_) > 0) np = atoi(e_);
    if (np < 1) np = 1;
    if (np > TL_MAX_PASS) np = TL_MAX_PASS;
    H->npass = np;
  }
  const int colsper = (n + H->npass - 1) / H->npass > 0 ? (n + H->npass - 1) / H->npass : 1;
  H->nnz_far = H->nnz - H->nnz_near;
  H->far_i.assign((size_t)H->npass * ((size_t)m + 1), 0);
  H->far_j.assign((size_t)H->nnz_far, 0); H->far_perm.assign((size_t)H->nnz_far, 0);
  {
    // count per (pass, row), prefix over pass-major order, fill
    std::vector<char> staged((size_t)ntiles + 1, 0);
    std::vector<int> cntpr((size_t)H->npass * (size_t)(m > 0 ? m : 1), 0);
    auto for_far = [&](auto &&fn) {
      for (int p = 0; p < H->npanels; ++p) {
        for (int i = H->pt_ptr[(size_t)p]; i < H->pt_ptr[(size_t)p + 1]; ++i) staged[(size_t)H->pt_tile[(size_t)i]] = 1;
        const int r0 = H->prow[(size_t)p], r1 = H->prow[(size_t)p + 1];
        for (int r = r0; r < r1; ++r)
          for (int k = ai[r]; k < ai[r + 1]; ++k) if (!staged[(size_t)(aj[k] / TL_TW)]) fn(r, k, aj[k] / colsper);
        for (int i = H->pt_ptr[(size_t)p]; i < H->pt_ptr[(size_t)p + 1]; ++i) staged[(size_t)H->pt_tile[(size_t)i]] = 0;
      }
    };
    long nfar = 0;
    for_far([&](int r, int, int q) { cntpr[(size_t)q * m + r]++; ++nfar; });
    if (nfar != H->nnz_far) { delete H; return (int)hipErrorUnknown; }
    long run = 0;
    for (int q = 0; q < H->npass; ++q) {
      for (int r = 0; r < m; ++r) { H->far_i[(size_t)q * (m + 1) + r] = (int)run; run += cntpr[(size_t)q * m + r]; }
      H->far_i[(size_t)q * (m + 1) + m] = (int)run;
      if (run & 1) ++run;                          // every pass starts on an even entry (the row-block kernel's 16-byte value loads)
    }
    H->far_j.assign((size_t)run, 0); H->far_perm.assign((size_t)run, -1);
    std::vector<int> nextpr((size_t)H->npass * (size_t)(m > 0 ? m : 1));
    for (int q = 0; q < H->npass; ++q) for (int r = 0; r < m; ++r) nextpr[(size_t)q * m + r] = H->far_i[(size_t)q * (m + 1) + r];
    for_far([&](int r, int k, int q) { const int pos = nextpr[(size_t)q * m + r]++; H->far_j[(size_t)pos] = aj[k]; H->far_perm[(size_t)pos] = k; });
  }
  if (e != H->nnz_near) { delete H; return (int)hipErrorUnknown; }
  mi355x_spmv_tiled_s *P = new mi355x_spmv_tiled_s();
  memset(P, 0, sizeof(*P));
  P->host = H;
  P->m = m; P->n = n; P->npanels = H->npanels; P->npt = (int)H->pt_tile.size();
  P->nnz_near = H->nnz_near; P->nnz_far = H->nnz_far; P->nsteps = H->nsteps; P->npass = H->npass;
  *out = P;
  return 0;
}

// nsteps: jagged diagonals over all wavefronts (nnz_staged / nsteps = lanes busy per load, of 64)
int mi355x_spmv_tiled_info(mi355x_spmv_tiled_t P, long *nnz_staged, long *nnz_remainder, int *npanels, int *npairs, long *nsteps) {
  if (nnz_staged) *nnz_staged = P->nnz_near;
  if (nnz_remainder) *nnz_remainder = P->nnz_far;
  if (npanels) *npanels = P->npanels;
  if (npairs) *npairs = P->npt;
  if (nsteps) *nsteps = P->nsteps;
  return 0;
}
int mi355x_spmv_tiled_geometry(int *panel_rows, int *tile_cols, int *waves, int *rounds, int *group_steps, int *trip_steps) {
  *panel_rows = TL_PANEL; *tile_cols = TL_TW; *waves = TL_WAVES; *rounds = TL_RPL; *group_steps = TL_U; *trip_steps = TL_NG * TL_U;
  return 0;
}

// tests: one host array of the layout (which: 0 pt_ptr, 1 pt_tile, 2 pw_e0, 3 desc, 4 perm, 5 lcol, 6 steps, 7 far_i, 8 far_j, 9 far_perm, 10 pw_s0, 11 prow);
// available until mi355x_spmv_tiled_drop_host
int mi355x_spmv_tiled_debug_get(mi355x_spmv_tiled_t P, int which, void *out, size_t cap_bytes, size_t *bytes) {
  if (!P->host) return (int)hipErrorInvalidValue;
  tl_host *H = P->host;
  const void *src = nullptr; size_t nb = 0;
  switch (which) {
    case 0: src = H->pt_ptr.data(); nb = H->pt_ptr.size() * 4; break;
    case 1: src = H->pt_tile.data(); nb = H->pt_tile.size() * 4; break;
    case 2: src = H->pw_e0.data(); nb = H->pw_e0.size() * 4; break;
    case 3: src = H->desc.data(); nb = H->desc.size() * 4; break;
    case 4: src = H->perm.data(); nb = H->perm.size() * 4; break;
    case 5: src = H->lcol.data(); nb = H->lcol.size() * 2; break;
    case 6: src = H->steps.data(); nb = H->steps.size() * 2; break;
    case 10: src = H->pw_s0.data(); nb = H->pw_s0.size() * 4; break;
    case 11: src = H->prow.data(); nb = H->prow.size() * 4; break;
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
// val[k] = aa[perm[k]]: the layout's values out of the CSR array that is on the device anyway
__global__ __launch_bounds__(256) void tl_gather_values_kernel(const int *__restrict__ perm, const double *__restrict__ aa, double *__restrict__ val, long n) {
  const long stride = (long)gridDim.x * 256;
  for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < n; k += stride) { const int q = perm[k]; val[k] = q >= 0 ? aa[q] : 0.0; }   // (-1: padding between the remainder's passes)
}

// One workgroup = TL_WAVES gathering wavefronts + ONE loader wavefront.  The loader brings the panel's next staged tile of x into the
// idle one of two LDS buffers while the others gather from the current one: a tile switch is one barrier, no load on anybody's path.
// A gathering wavefront's stream runs through all the panel's tiles without a gap and it keeps TL_NG groups of TL_U steps' loads in
// flight ACROSS the tile switches (the value / column loads do not depend on the tile of x, only their use does): the barrier costs the
// skew between the wavefronts, not a drained pipeline.  The step words (round and active-lane count of each step, wave-uniform) come
// through a vector load one trip ahead, one 32-bit word per lane, and are read out with readlane: they return in order with the value
// loads and never sit between an LDS read and its use.  A lane's four running sums (one per round) live in registers during a tile and
// in LDS between tiles.
struct tl_group { double v[TL_U]; unsigned short c[TL_U]; unsigned int w[TL_U / 2]; tl_u4 d; };   // d: the row words of the tile the group opens (if it opens one)

__device__ __forceinline__ unsigned int tl_stepword(const tl_group &g, int u) { return (g.w[u >> 1] >> ((u & 1) * 16)) & 0xffffu; }

// group k of the trip whose words are in wv (lane j: word j of the trip)
// (the row words of a tile travel with the group that opens it: a value loaded long before its use and carried through the loop in a
//  register of its own makes the compiler wait for everything issued since, whenever that register is copied)
__device__ __forceinline__ void tl_issue(tl_group &g, const unsigned int wv, const int k, int &off, const int lane,
                                         const double *__restrict__ val, const unsigned short *__restrict__ lcol, const tl_u4 *&dq) {
#pragma unroll
  for (int i = 0; i < TL_U / 2; ++i) g.w[i] = (unsigned int)__builtin_amdgcn_readlane((int)wv, k * (TL_U / 2) + i);
  if (g.w[0] & TL_STEP_NEWTILE) { g.d = __builtin_nontemporal_load(dq); dq += 64; }
  else g.d = tl_u4{0u, 0u, 0u, 0u};
#pragma unroll
  for (int u = 0; u < TL_U; ++u) {
    const int nact = (int)(tl_stepword(g, u) & 0x7fu);               // lanes 0 .. nact - 1 take part in this step (0: padding)
    const int idx = off + (lane < nact ? lane : 0);                  // every load unconditional (idle lanes re-read the step's first entry; the arrays carry slack)
    g.v[u] = __builtin_nontemporal_load(val + idx);
    g.c[u] = __builtin_nontemporal_load(lcol + idx);
    off += nact;
  }
}
// One group: the LDS reads go out together, then step after step  s += a * x  for the lanes the step has (the others add -0.0, the one
// addend that leaves every double -- both zeros included -- as it is: no branch, the compiler's wait counts stay exact).  At a round's
// last step the lanes' rows are complete for this tile: the sum goes back to LDS and the next round's row, sum and count move up.
struct tl_rows { unsigned int d[TL_RPL]; double q[TL_RPL]; };    // d[0] / q[0]: the current round's row word and running sum
__device__ __forceinline__ void tl_consume(const tl_group &g, const double *xc, double *acc, const int lane, tl_rows &r) {
  double xv[TL_U];
#pragma unroll
  for (int u = 0; u < TL_U; ++u) {
#ifdef TL_EXP_NOGATHER
    xv[u] = (double)g.c[u];
#else
    xv[u] = xc[g.c[u]];
#endif
  }
#pragma unroll
  for (int u = 0; u < TL_U; ++u) {
    const unsigned int sw = tl_stepword(g, u);
    const int nact = (int)(sw & 0x7fu);                              // wave-uniform
    const double p = g.v[u] * xv[u];
    r.q[0] = r.q[0] + (lane < nact ? p : -0.0);
    if (sw & TL_STEP_ROUNDEND) {                                     // wave-uniform
      if (r.d[0] & ((1u << TL_CNT_BITS) - 1)) acc[r.d[0] >> TL_CNT_BITS] = r.q[0];
#pragma unroll
      for (int a = 0; a + 1 < TL_RPL; ++a) { r.d[a] = r.d[a + 1]; r.q[a] = r.q[a + 1]; }
      r.d[TL_RPL - 1] = 0u;
    }
  }
}
static_assert(TL_RPL == 4, "the row words of a (wavefront, tile) are one 16-byte load per lane");

#ifdef TL_PROFILE
// development build: per panel {start, end (100 MHz wall clock), hardware id (XCC << 32 | HW_ID), then core cycles of wavefront 0: in tile-switch barriers,
// in tl_consume, in tl_issue, whole gather loop; of the loader: loading tiles}
__device__ unsigned long long tl_prof_buf[8 * 8192];
#endif

template <int ADD>
__global__ __launch_bounds__((TL_WAVES + 1) * 64) void spmv_tiled_kernel(
    int npanels, int chunkx, const int *__restrict__ prow, const int *__restrict__ pt_ptr, const int *__restrict__ pt_tile,
    const int *__restrict__ pw_e0, const int *__restrict__ pw_s0, const unsigned short *__restrict__ steps, const unsigned int *__restrict__ desc,
    const double *__restrict__ val, const unsigned short *__restrict__ lcol,
    const double *__restrict__ x, const double *yin, double *yout, int n) {
  extern __shared__ __attribute__((aligned(16))) double tl_lds[];
  double *xt = tl_lds;                                   // 2 x TL_TW doubles: two tiles of x
  double *acc = tl_lds + 2 * TL_TW;                      // running sums of the panel's rows
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NT = (TL_WAVES + 1) * 64;

  // each XCD walks a contiguous eighth of the panels: neighbouring panels stage the same tiles, out of the same L2
  const int xcd = blockIdx.x % MI355X_NXCD, slot = blockIdx.x / MI355X_NXCD;
  const int p = xcd * chunkx + slot;
  if (slot >= chunkx || p >= npanels) return;

#ifdef TL_PROFILE
  unsigned long long prof_t0 = __builtin_amdgcn_s_memrealtime(), prof_bar = 0, prof_con = 0, prof_iss = 0, prof_loop = 0, prof_ld = 0;
#endif
  const int row0 = prow[p], nrow = prow[p + 1] - row0;
  for (int rl = tid; rl < nrow; rl += NT) acc[rl] = ADD ? yin[row0 + rl] : 0.0;

  const int pt0 = pt_ptr[p], ntp = pt_ptr[p + 1] - pt0;
  // a tile into a buffer: whole double2's inside x, then the last column of an odd-sized last tile
  auto tile_tail = [&](int t, double *buf) {
    const size_t base = (size_t)t * TL_TW;
    const long left = (long)n - (long)base;
    if (left < TL_TW && (left & 1) && lane == 0) buf[left - 1] = x[base + left - 1];
  };
  if (ntp > 0) {                                         // the first tile: everybody
    const int t = pt_tile[pt0];
    const size_t base = (size_t)t * TL_TW;
    const int ncol = (n - (long)base) < TL_TW ? (int)(n - (long)base) : TL_TW;
    const tl_v2d *xs = reinterpret_cast<const tl_v2d *>(x + base);
    for (int i = tid; i < (ncol >> 1); i += NT) reinterpret_cast<tl_v2d *>(xt)[i] = xs[i];
    if (w == 0) tile_tail(t, xt);
  }

  if (w == TL_WAVES) {
    __syncthreads();                                     // first tile in place, the sums' first stores done
    // ---- the loader: tile i + 1 into the other buffer while the others gather from tile i ----
    constexpr int LB = 16;                               // 16-byte loads in flight: 1 KB each, two batches of 16 KB per 4096-column tile
    for (int i = 0; i + 1 < ntp; ++i) {
#ifdef TL_EXP_NOLOAD
      __syncthreads();
      continue;
#endif
#ifdef TL_PROFILE
      const unsigned long long pq0 = __builtin_readcyclecounter();
#endif
      const int t = pt_tile[pt0 + i + 1];
      double *buf = xt + ((i & 1) ^ 1) * TL_TW;
      const size_t base = (size_t)t * TL_TW;
      const int ncol = (n - (long)base) < TL_TW ? (int)(n - (long)base) : TL_TW;
      const int n2 = ncol >> 1;
      const tl_v2d *xs = reinterpret_cast<const tl_v2d *>(x + base);
      for (int i0 = 0; i0 < n2; i0 += LB * 64) {
        tl_v2d r[LB];
#pragma unroll
        for (int k = 0; k < LB; ++k) { const int j = i0 + k * 64 + lane; r[k] = xs[j < n2 ? j : 0]; }
#pragma unroll
        for (int k = 0; k < LB; ++k) { const int j = i0 + k * 64 + lane; if (j < n2) reinterpret_cast<tl_v2d *>(buf)[j] = r[k]; }
      }
      tile_tail(t, buf);
#ifdef TL_PROFILE
      __builtin_amdgcn_s_waitcnt(0);
      prof_ld += __builtin_readcyclecounter() - pq0;
#endif
#ifndef TL_EXP_NOBARRIER
      __syncthreads();                                   // (the gathering wavefronts' switch to tile i + 1)
#endif
    }
  } else {
    // ---- a gathering wavefront ----
    // (the stream is padded to whole trips of TL_NG groups and every load below is issued on every path -- past the stream's end into
    //  padding words that say "no lanes" -- so that the loads in flight are the same number wherever the code is: a conditional issue
    //  makes the compiler wait for the younger groups' loads too, and the pipeline is one group deep whatever the source says)
    const int ipw = p * TL_WAVES + w;
    int off = pw_e0[ipw];
    const int s0 = pw_s0[ipw];
    const int ntrips = (pw_s0[ipw + 1] - s0) / (TL_NG * TL_U);
    const unsigned int *sw32 = reinterpret_cast<const unsigned int *>(steps + s0);       // (s0 a multiple of TL_U: 8- or 16-byte aligned)
    const int wl = lane < TL_TRIPW ? lane : TL_TRIPW - 1;
    const tl_u4 *dq = reinterpret_cast<const tl_u4 *>(desc + ((size_t)pt0 * TL_WAVES + (size_t)w * ntp) * (64 * TL_RPL)) + lane;   // the next tile's row words: + 64 per tile
    if (ntrips == 0) {                                   // no staged tile in this panel (true for all its wavefronts alike)
      __syncthreads();
    } else {
      tl_group grp[TL_NG];
      unsigned int wv = sw32[wl], wvn = sw32[TL_TRIPW + wl];
      if (ntrips < 2) wvn = 0u;                          // (what was read lies in the next stream or the array's slack)
#pragma unroll
      for (int k = 0; k < TL_NG; ++k) tl_issue(grp[k], wv, k, off, lane, val, lcol, dq);
      __syncthreads();                                   // first tile in place, the sums' first stores done

      int it = -1;                                       // the tile being gathered from (index among the panel's staged tiles)
      const double *xc = xt;
      tl_rows r;
#pragma unroll
      for (int a = 0; a < TL_RPL; ++a) { r.d[a] = 0u; r.q[a] = 0.0; }
      auto tile_switch = [&](const tl_u4 d) {
#ifdef TL_PROFILE
        const unsigned long long pb0 = __builtin_readcyclecounter();
#endif
#ifndef TL_EXP_NOBARRIER
        if (it >= 0) __syncthreads();                    // nobody reads tile `it` any more, every row's sum is back in LDS; the loader has completed the next tile
#endif
#ifdef TL_PROFILE
        prof_bar += __builtin_readcyclecounter() - pb0;
#endif
        ++it;
        xc = xt + (it & 1) * TL_TW;
#pragma unroll
        for (int a = 0; a < TL_RPL; ++a) {
          r.d[a] = d[a];
          r.q[a] = (d[a] & ((1u << TL_CNT_BITS) - 1)) ? acc[d[a] >> TL_CNT_BITS] : 0.0;
        }
      };
#ifdef TL_PROFILE
      const unsigned long long pl0 = __builtin_readcyclecounter();
#endif
      for (int t = 0; t < ntrips; ++t) {
        unsigned int wvnn = sw32[(t + 2) * TL_TRIPW + wl];   // the words of the trip after next
        if (t + 2 >= ntrips) wvnn = 0u;
#pragma unroll
        for (int k = 0; k < TL_NG; ++k) {
          if (grp[k].w[0] & TL_STEP_NEWTILE) tile_switch(grp[k].d);
#ifdef TL_PROFILE
          const unsigned long long pc0 = __builtin_readcyclecounter();
#endif
          tl_consume(grp[k], xc, acc, lane, r);
#ifdef TL_PROFILE
          const unsigned long long pc1 = __builtin_readcyclecounter();
#endif
          tl_issue(grp[k], wvn, k, off, lane, val, lcol, dq);   // (the last trip: words of zeros, loads nobody uses)
#ifdef TL_PROFILE
          prof_con += pc1 - pc0; prof_iss += __builtin_readcyclecounter() - pc1;
#endif
        }
        wvn = wvnn;
      }
#ifdef TL_PROFILE
      prof_loop = __builtin_readcyclecounter() - pl0;
#endif
    }
  }
  __syncthreads();
  for (int rl = tid; rl < nrow; rl += NT) yout[row0 + rl] = acc[rl];
#ifdef TL_PROFILE
  if (tid == 0 && p < 8192) {
    tl_prof_buf[8 * p + 0] = prof_t0;
    tl_prof_buf[8 * p + 1] = __builtin_amdgcn_s_memrealtime();
    tl_prof_buf[8 * p + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | (unsigned int)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    tl_prof_buf[8 * p + 3] = prof_bar; tl_prof_buf[8 * p + 4] = prof_con; tl_prof_buf[8 * p + 5] = prof_iss; tl_prof_buf[8 * p + 6] = prof_loop;
  }
  if (tid == TL_WAVES * 64 && p < 8192) tl_prof_buf[8 * p + 7] = prof_ld;
#endif
}

extern "C" {

// device part: tables up, values gathered from the CSR value array on the device (aa_dev), remainder with a row-block plan of its own
int mi355x_spmv_tiled_upload(mi355x_handle_t h, mi355x_spmv_tiled_t P, const double *aa_dev) {
  tl_host *H = P->host;
  if (!H) return (int)hipErrorInvalidValue;
  auto up = [&](void **d, const void *src, size_t nbytes) -> int {
    MI355X_TRY(hipMalloc(d, (nbytes ? nbytes : 1) + 1024));          // slack: the kernel's unconditional loads run past a stream's end (idle lanes, step words two trips ahead)
    if (nbytes) MI355X_TRY(hipMemcpyAsync(*d, src, nbytes, hipMemcpyHostToDevice, h->stream));
    return 0;
  };
  int rc;
  if ((rc = up((void **)&P->d_pt_ptr, H->pt_ptr.data(), H->pt_ptr.size() * 4)) || (rc = up((void **)&P->d_pt_tile, H->pt_tile.data(), H->pt_tile.size() * 4)) ||
      (rc = up((void **)&P->d_prow, H->prow.data(), H->prow.size() * 4)) ||
      (rc = up((void **)&P->d_pw_e0, H->pw_e0.data(), H->pw_e0.size() * 4)) || (rc = up((void **)&P->d_desc, H->desc.data(), H->desc.size() * 4)) ||
      (rc = up((void **)&P->d_pw_s0, H->pw_s0.data(), H->pw_s0.size() * 4)) || (rc = up((void **)&P->d_steps, H->steps.data(), H->steps.size() * 2)) ||
      (rc = up((void **)&P->d_perm, H->perm.data(), H->perm.size() * 4)) || (rc = up((void **)&P->d_lcol, H->lcol.data(), H->lcol.size() * 2)) ||
      (rc = up((void **)&P->d_far_i, H->far_i.data(), H->far_i.size() * 4)) || (rc = up((void **)&P->d_far_j, H->far_j.data(), H->far_j.size() * 4)) ||
      (rc = up((void **)&P->d_far_perm, H->far_perm.data(), H->far_perm.size() * 4)))
    return rc;
  MI355X_TRY(hipMalloc((void **)&P->d_val, sizeof(double) * (size_t)(P->nnz_near > 0 ? P->nnz_near : 1) + 64));
  P->nfar_store = (long)H->far_perm.size();
  MI355X_TRY(hipMalloc((void **)&P->d_far_a, sizeof(double) * (size_t)(P->nfar_store > 0 ? P->nfar_store : 1) + 64));
  if (P->nnz_far > 0)
    for (int q = 0; q < P->npass; ++q) {
      const int *fi = H->far_i.data() + (size_t)q * ((size_t)P->m + 1);
      if (fi[P->m] == fi[0]) continue;                                    // nothing in this column range
      rc = mi355x_spmv_plan_create(h, P->m, fi, nullptr, &P->far_plan[q]);
      if (rc) return rc;
    }
  MI355X_TRY(hipStreamSynchronize(h->stream));
  return mi355x_spmv_tiled_refresh_values(h, P, aa_dev);
}

// the CSR values on the device changed (same pattern): one gather per part
int mi355x_spmv_tiled_refresh_values(mi355x_handle_t h, mi355x_spmv_tiled_t P, const double *aa_dev) {
  if (P->nnz_near > 0) {
    hipLaunchKernelGGL(tl_gather_values_kernel, dim3(mi355x_grid_for((size_t)P->nnz_near, 4)), dim3(256), 0, h->stream, P->d_perm, aa_dev, P->d_val, P->nnz_near);
    MI355X_LAUNCH_CHECK();
  }
  if (P->nnz_far > 0) {
    hipLaunchKernelGGL(tl_gather_values_kernel, dim3(mi355x_grid_for((size_t)P->nfar_store, 4)), dim3(256), 0, h->stream, P->d_far_perm, aa_dev, P->d_far_a, P->nfar_store);
    MI355X_LAUNCH_CHECK();
  }
  return 0;
}

// y = A x (yin == NULL) or yout = yin + A x (yout may alias yin).  x must be 16-byte aligned (hipErrorNotSupported otherwise: the caller
// takes the row-block kernel).  which: 0 both parts, 1 the staged part only, 2 the remainder only (development: their separate cost)
int mi355x_spmv_tiled_parts(mi355x_handle_t h, mi355x_spmv_tiled_t P, const double *x, const double *yin, double *yout, int which) {
  if (!mi355x_aligned16(x)) return (int)hipErrorNotSupported;
  if (P->m == 0) return 0;
  const size_t lds = TL_LDS_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    if (getenv("MI355X_TILED_DEBUG")) {
      int nb = 0;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, spmv_tiled_kernel<0>, (TL_WAVES + 1) * 64, lds);
      fprintf(stderr, "[mi355x tiled] tile %d columns, %d + 1 wavefronts per workgroup, %d groups of %d steps in flight, %zu B of LDS: %d workgroups per CU (layout cut for %d)\n", TL_TW, TL_WAVES, TL_NG, TL_U, lds, nb, TL_WG_PER_CU);
    }
    MI355X_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(spmv_tiled_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    MI355X_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(spmv_tiled_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int chunkx = (P->npanels + MI355X_NXCD - 1) / MI355X_NXCD;
  const int grid = chunkx * MI355X_NXCD;
  if (which != 2) {
#define TL_GO(A_, YIN) hipLaunchKernelGGL((spmv_tiled_kernel<A_>), dim3(grid), dim3((TL_WAVES + 1) * 64), lds, h->stream, P->npanels, chunkx, P->d_prow, P->d_pt_ptr, \
                                          P->d_pt_tile, P->d_pw_e0, P->d_pw_s0, P->d_steps, P->d_desc, P->d_val, P->d_lcol, x, YIN, yout, P->n)
    if (yin) TL_GO(1, yin); else TL_GO(0, (const double *)nullptr);
#undef TL_GO
    MI355X_LAUNCH_CHECK();
#ifdef TL_PROFILE
    if (const char *pf = getenv("MI355X_TILED_PROF")) {
      std::vector<unsigned long long> hb(8 * 8192);
      MI355X_TRY(hipStreamSynchronize(h->stream));
      MI355X_TRY(hipMemcpyFromSymbol(hb.data(), HIP_SYMBOL(tl_prof_buf), hb.size() * 8));
      if (FILE *f = fopen(pf, "wb")) { fwrite(hb.data(), 8, (size_t)8 * (size_t)(P->npanels < 8192 ? P->npanels : 8192), f); fclose(f); }
    }
#endif
  }
  if (which != 1 && P->nnz_far > 0)
    for (int q = 0; q < P->npass; ++q) {
      if (!P->far_plan[q]) continue;
      const int rc = mi355x_spmv_csr_add(h, P->far_plan[q], P->d_far_i + (size_t)q * ((size_t)P->m + 1), P->d_far_j, P->d_far_a, x, yout, yout);
      if (rc) return rc;
    }
  return 0;
}
int mi355x_spmv_tiled(mi355x_handle_t h, mi355x_spmv_tiled_t P, const double *x, const double *yin, double *yout) {
  return mi355x_spmv_tiled_parts(h, P, x, yin, yout, 0);
}

int mi355x_spmv_tiled_destroy(mi355x_spmv_tiled_t P) {
  if (!P) return 0;
  delete P->host;
  void *ptrs[] = {P->d_prow, P->d_pt_ptr, P->d_pt_tile, P->d_pw_e0, P->d_pw_s0, P->d_steps, P->d_desc, P->d_perm, P->d_lcol, P->d_val, P->d_far_i, P->d_far_j, P->d_far_perm, P->d_far_a};
  for (void *q : ptrs) if (q) hipFree(q);
  for (int q = 0; q < TL_MAX_PASS; ++q) if (P->far_plan[q]) mi355x_spmv_plan_destroy(P->far_plan[q]);
  delete P;
  return 0;
}

}  // extern "C"
