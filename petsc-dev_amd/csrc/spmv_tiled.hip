// Column-tiled CSR SpMV for matrices whose x gathers miss the caches: y = A x and z = y + A x (MatMult_SeqAIJ / MatMultAdd_SeqAIJ,
// reference src/mat/impls/aij/seq/aij.c:1225-1358) with the x entries a row panel needs STAGED IN LDS.
//
// Why.  The row-block kernels of spmv_csr.hip gather x through L1 / L2.  On a matrix whose rows pick their columns from a wide window
// without neighbouring rows sharing them (the irregular stand-in of BASELINE configs[3]: 73 entries per row, +-50 000 band, a fifth of the
// entries anywhere) every gather is an L1 miss that moves a whole cache line out of L2 for 8 useful bytes: 120 M L1->L2 requests for 112 M
// nonzeros, 2.5x the algorithmic bytes on the fabric, 0.23 of the HBM roof (profiles/r02_cfg4_irr*).  LDS is the one on-chip memory with
// word-granular random access, so the product is re-cut so that the gathers -- and the row sums -- live there.
//
// Layout (built once per nonzero pattern on host threads, values refreshed on the device through a permutation):
//   * rows in PANELS of <= TL_PANEL = 6143 rows holding equal shares of the nonzeros, one workgroup per panel; columns in TILES of TL_TW
//     (2048) entries of x = 16 KB of LDS, two of them resident (a loader wavefront brings the next tile while the current one is
//     gathered from); the panel's row sums are 6144 doubles of LDS for the whole kernel;
//   * a (panel, tile) pair with at least `stage_min` entries is STAGED: all of the panel's entries in that tile gather x from LDS.  The
//     panel walks its staged tiles in ascending order.  Entries of pairs too thin to stage (the long-range fifth: the REMAINDER) follow
//     in the same streams, grouped by WINDOWS of 2^18 columns (2 MiB of x) in ascending order: their x comes from global memory, one
//     L2 request per entry, and all workgroups reach the same window at about the same time, so those requests hit the XCD's L2 --
//     the split of MatMult_MPIAIJ's diagonal / off-diagonal blocks, inside one GPU and inside one kernel;
//   * the panel's rows are dealt to the workgroup's TL_WAVES (4) gathering wavefronts in contiguous ranges of equal nonzeros: a row's sum
//     is only ever touched by its wavefront.  A wavefront's entries of one staged tile, in CSR order (rows ascending, columns
//     ascending), are cut into BLOCKS of 128 padded with zeros that go to a spare accumulator; its blocks of all the panel's tiles
//     follow each other without a gap, then its blocks of the remainder, window after window.  An entry is 12 bytes: the value and
//     one 32-bit word (staged: row of the panel << 16 | column in the tile, bit 15 of a block's first word: the block opens the next
//     tile; remainder: row << 18 | column in the window, bit 31: the block opens the wavefront's next window);
//   * the kernel keeps TL_NG blocks' loads in flight per wavefront, across tile switches.  A block is TWO full-width loads (16 bytes of
//     values, 8 bytes of words per lane: two entries) and per entry one LDS read of x, one multiply and one ds_add_f64 into the row's
//     sum: no sorting, no descriptors, no per-row bookkeeping -- 10 instructions per entry and lane where the jagged-diagonal forms of
//     this file's history took 50 and were bound by the wavefronts' own instruction streams (profiles/r04_tiled_*: 0.30 ms whatever
//     the pipelining), while the memory system gives this access pattern 6.8 TB/s (profiles/r04_dense_atomic_probe.log).
//     At 12 bytes the staged part moves 1.2 GB in 0.19 ms: HBM's rate.  10-byte entries (a 16-bit word: column, 3-bit row step, rows
//     rebuilt by two 64-lane prefix sums per block) were built, pass the same tests and are slower, 0.207 ms: the prefix sums'
//     instructions cost more than the bytes save (profiles/r04_tiled_10byte_entries.log).
// What the form pays for: lanes of one ds_add_f64 that meet on one accumulator are serialised, so a matrix whose rows have MANY entries
// in one tile runs badly here (the FEM stand-in, 77 entries per row in a handful of tiles, forced through it: 0.62 ms against 0.25 for
// the grouped-row kernel, profiles/r04_fem_tiled_try.log; with a run ordered "entry j of every row, j = 0, 1, .." instead of row after
// row, so that the lanes meet on different sums, 0.49 ms -- and the value refresh twice as slow, its gather then jumping rows with
// every entry: r04_tiled_jorder.log; not kept).  The Mat type's selection rule does not admit such matrices: more than half
// a line of x per nonzero in 32-row groups means fewer than 8 entries per row and tile, and matrices with inodes or an offset
// dictionary keep their kernels (host/aijhip.c).
// Arithmetic: a*x rounded, then added (-ffp-contract=off).  Entry q of a block is stored at position 2 (q mod 64) + q div 64: lane l's
// pair is entries l and 64 + l, so the block's first ds_add_f64 instruction adds entries 0..63 and the second 64..127, and lanes of one
// instruction that meet on one accumulator are added in ascending lane order (measured: dense_atomic_probe; asserted by the GPU tests'
// bitwise comparison with the layout's order): a row's staged products are added one after the other in column order starting from 0
// (or y), its remainder products after them, again in column order: agrees with the reference to rounding (tests: <= 1e-12 * sum
// |a_ij x_j|), bit for bit when nothing (or everything) is left to the remainder, and the whole product is bit for bit what the layout's
// order gives on the host (run-to-run identical).
#include "common.hpp"
#include <algorithm>
#include <atomic>
#include <memory>
#include <new>
#include <thread>
#include <vector>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>

#ifndef TL_TW
#define TL_TW 2048            // columns of x per tile (16 KB of LDS; two buffers)
#endif
#ifndef TL_WAVES
#define TL_WAVES 4            // gathering wavefronts per workgroup.  (2 / 3 / 4 / 5 / 6 / 8 / 12 / 15 on the stand-in: the staged part wants >= 4 to keep
#endif                        // HBM busy -- 0.28 / 0.22 / 0.19 ms, then flat --, the remainder is bound by the CUs' address units (77 % busy, ~3 cycles per
                              // gathered line) and pays every further wavefront with padding of its runs: 0.140 ms at 4, 0.152 at 8, 0.166 at 15:
                              // profiles/r04_tiled_sweep10.log, r04_tiled_sweep11.log, r04_tiled_pmc_far.csv)
#ifndef TL_NG
#define TL_NG 8               // staged blocks in flight per wavefront (with 4 wavefronts: 2 / 4 / 6 / 8 -> 0.267 / 0.191 / 0.184 / 0.185 ms for the staged part, profiles/r04_tiled_sweep13.log)
#endif
#ifndef TL_FG
#define TL_FG 2               // remainder: blocks whose gathers are in flight per wavefront (their stream loads run another TL_FG blocks ahead; 2 / 4 / 6: the same time)
#endif
#ifndef TL_SLOTS
#define TL_SLOTS 6144                      // row sums in LDS (48 KB): the panel's rows and one spare for the padding.  With the two tiles 80 KB: two
#endif                                     // workgroups fit a CU.  (Entries per (panel, tile) pair -- what a tile switch and a block's padding are paid
                                           // from -- go with rows x columns: 6144 x 2048 holds 1.5x those of 2048 x 4096 in the same LDS and moves
                                           // a third of the x tiles: 0.19 ms against 0.26 on the stand-in, profiles/r04_tiled_sweep5.log)
#define TL_PANEL (TL_SLOTS - 1)            // most rows a workgroup takes
#define TL_BLOCK 128                       // entries per block: two per lane
#define TL_FW_SHIFT 18                     // a window of the remainder: 2^18 columns = 2 MiB of x, what stays in an XCD's 4 MiB L2 next to the streams passing through
#define TL_WORD_NEWWIN 0x80000000u         // remainder: a block's first word: the block opens the wavefront's next window
#define TL_WORD_NEWTILE 0x8000u            // a block's first word: the block opens the panel's next staged tile
static_assert(TL_TW <= 0x8000 && TL_SLOTS <= 0x2000 && TL_TW % 128 == 0, "entry words: row << 16 | flag << 15 | column in the tile; flag << 31 | row << 18 | column in the window; a tile is whole 1 KB loads");
#define TL_LDS_BYTES (8 * (2 * TL_TW + TL_SLOTS))
#define TL_NCU 256
static_assert(TL_LDS_BYTES <= 160 * 1024 && (TL_WAVES + 1) * 64 <= 1024, "one workgroup must fit a CU");

typedef double tl_v2d __attribute__((ext_vector_type(2)));
typedef unsigned int tl_v2u __attribute__((ext_vector_type(2)));

// what the builder leaves (host) and the plan holds (device); all offsets fit 32 bits (nnz < 2^31 as in the CSR arrays)
struct tl_host {
  int m = 0, n = 0, npanels = 0;
  long nnz = 0, nnz_near = 0, nnz_far = 0, nstore = 0;
  std::vector<int> prow;          // [npanels + 1] first row of a panel
  std::vector<int> wrow;          // [npanels * (TL_WAVES + 1)] first row (of the matrix) of a wavefront's range inside its panel; the last one = the panel's end
  std::vector<int> pt_ptr;        // [npanels + 1] -> staged (panel, tile) pairs
  std::vector<int> pt_tile;       // [npt] tile number
  std::vector<int> pw_e0;         // [npanels * TL_WAVES + 1] first stored entry of (panel, wavefront): a multiple of TL_BLOCK
  std::vector<int> perm;          // [nstore] stored entry -> position in the CSR value array, -1: padding (value 0)
  std::vector<unsigned int> word; // [nstore] staged: row of the panel << 16 | column - tile * TL_TW (padding: TL_PANEL << 16), TL_WORD_NEWTILE on the first stored word
                                  // of a tile's first block; remainder: row << 18 | column - (window << 18) (padding: TL_PANEL << 18), TL_WORD_NEWWIN likewise
  std::vector<int> pw_f0;         // [npanels * TL_WAVES] first stored entry of the wavefront's remainder blocks (its staged blocks come before)
  std::vector<int> fw_ptr;        // [npanels * TL_WAVES + 1] -> the windows of a wavefront's remainder, ascending (only windows it has entries in)
  std::vector<int> fw_win;        // window numbers (column >> TL_FW_SHIFT)
};

struct mi355x_spmv_tiled_s {
  tl_host *host;                  // kept until _drop_host (tests read it back)
  int m, n, npanels, npt;
  long nnz_near, nnz_far, nstore;
  int *d_prow, *d_pt_ptr, *d_pt_tile, *d_pw_e0, *d_perm;
  unsigned int *d_word;
  double *d_val;
  int *d_pw_f0, *d_fw_ptr, *d_fw_win;
};

// ---------------------------------------------------------------------------------------------------------------------------------
// host: the layout
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {
struct PanelOut {
  std::vector<int> pt_tile;
  std::vector<int> perm[TL_WAVES];
  std::vector<unsigned int> word[TL_WAVES];
  int wrow[TL_WAVES + 1];
  int nstaged[TL_WAVES];          // stored entries of the wavefront's staged blocks
  std::vector<int> fwin[TL_WAVES];
  long near = 0, far = 0;
};

// position of a block's entry q (CSR order) in storage: lane q mod 64 holds entries q and q + 64 side by side
static inline int tl_slot(int q) { return 2 * (q & 63) + (q >> 6); }

// one panel (rows r0 .. r1 - 1): its wavefronts' row ranges, which tiles are staged, then per (wavefront, staged tile) the blocks
static void build_panel(int r0, int r1, const int *ai, const int *aj, int stage_min, std::vector<int> &cnt, std::vector<int> &touched,
                        std::vector<int> &cur, std::vector<char> &stagedflag, PanelOut &o) {
  // wavefront ranges: equal shares of the panel's nonzeros
  {
    const long base = ai[r0], tot = (long)ai[r1] - base;
    o.wrow[0] = r0;
    for (int w = 1; w < TL_WAVES; ++w) {
      int r;
      if (tot > 0) r = (int)(std::lower_bound(ai + r0, ai + r1 + 1, base + (tot * w + TL_WAVES - 1) / TL_WAVES, [](int v, long t) { return (long)v < t; }) - ai);
      else r = r0 + (int)(((long)(r1 - r0) * w) / TL_WAVES);
      if (r < o.wrow[w - 1]) r = o.wrow[w - 1];
      if (r > r1) r = r1;
      o.wrow[w] = r;
    }
    o.wrow[TL_WAVES] = r1;
  }
  touched.clear();
  for (int r = r0; r < r1; ++r)
    for (int k = ai[r]; k < ai[r + 1]; ++k) { const int t = aj[k] / TL_TW; if (cnt[t]++ == 0) touched.push_back(t); }
  std::sort(touched.begin(), touched.end());
  for (int r = r0; r < r1; ++r) cur[r - r0] = ai[r];
  std::vector<int> kk;                                     // one (wavefront, tile)'s entries in CSR order
  for (int t : touched) {
    const bool staged = cnt[t] >= stage_min;
    cnt[t] = 0;
    if (!staged) continue;
    stagedflag[(size_t)t] = 1;
    const int clo = t * TL_TW, chi = clo + TL_TW;
    o.pt_tile.push_back(t);
    for (int w = 0; w < TL_WAVES; ++w) {
      kk.clear();
      for (int r = o.wrow[w]; r < o.wrow[w + 1]; ++r) {
        int k = cur[r - r0];
        const int kend = ai[r + 1];
        while (k < kend && aj[k] < clo) ++k;               // entries of thinner tiles in between: the remainder's
        while (k < kend && aj[k] < chi) kk.push_back(k++);
        cur[r - r0] = k;
      }
      const int nreal = (int)kk.size();
      const int nb = nreal ? (nreal + TL_BLOCK - 1) / TL_BLOCK : 1;     // at least one block per tile: the wavefront meets every tile switch
      const size_t base = o.perm[w].size();
      o.perm[w].resize(base + (size_t)nb * TL_BLOCK, -1);
      o.word[w].resize(base + (size_t)nb * TL_BLOCK, (unsigned int)TL_PANEL << 16);
      int r = o.wrow[w];
      for (int q = 0; q < nreal; ++q) {
        const int k = kk[(size_t)q];
        while (ai[r + 1] <= k) ++r;
        const size_t pos = base + (size_t)(q / TL_BLOCK) * TL_BLOCK + (size_t)tl_slot(q % TL_BLOCK);
        o.perm[w][pos] = k;
        o.word[w][pos] = ((unsigned int)(r - r0) << 16) | (unsigned int)(aj[k] - clo);
      }
      o.word[w][base] |= TL_WORD_NEWTILE;
      o.near += nreal;
    }
  }
  // the remainder: per wavefront what no staged tile took, window after window, CSR order inside a window
  std::vector<std::pair<int, int>> fe;
  for (int w = 0; w < TL_WAVES; ++w) {
    o.nstaged[w] = (int)o.perm[w].size();
    fe.clear();
    for (int r = o.wrow[w]; r < o.wrow[w + 1]; ++r)
      for (int k = ai[r]; k < ai[r + 1]; ++k)
        if (!stagedflag[(size_t)(aj[k] / TL_TW)]) fe.push_back({aj[k] >> TL_FW_SHIFT, k});
    std::stable_sort(fe.begin(), fe.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first < b.first; });
    size_t i = 0;
    int r = o.wrow[w];
    while (i < fe.size()) {
      size_t j = i;
      while (j < fe.size() && fe[j].first == fe[i].first) ++j;
      const int win = fe[i].first, nreal = (int)(j - i), nb = (nreal + TL_BLOCK - 1) / TL_BLOCK;
      const size_t base = o.perm[w].size();
      o.fwin[w].push_back(win);
      o.perm[w].resize(base + (size_t)nb * TL_BLOCK, -1);
      o.word[w].resize(base + (size_t)nb * TL_BLOCK, (unsigned int)TL_PANEL << TL_FW_SHIFT);
      r = o.wrow[w];
      for (int q = 0; q < nreal; ++q) {
        const int k = fe[i + (size_t)q].second;
        while (ai[r + 1] <= k) ++r;
        const size_t pos = base + (size_t)(q / TL_BLOCK) * TL_BLOCK + (size_t)tl_slot(q % TL_BLOCK);
        o.perm[w][pos] = k;
        o.word[w][pos] = ((unsigned int)(r - r0) << TL_FW_SHIFT) | (unsigned int)(aj[k] - (win << TL_FW_SHIFT));
      }
      o.word[w][base] |= TL_WORD_NEWWIN;
      o.far += nreal;
      i = j;
    }
  }
  for (int t : o.pt_tile) stagedflag[(size_t)t] = 0;
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
static int tl_build_impl(int m, int n, const int *ai, const int *aj, int stage_min, mi355x_spmv_tiled_t *out);
int mi355x_spmv_tiled_build(int m, int n, const int *ai, const int *aj, int stage_min, mi355x_spmv_tiled_t *out) {
  *out = nullptr;
  if (m < 0 || n < 0) return (int)hipErrorInvalidValue;
  try { return tl_build_impl(m, n, ai, aj, stage_min, out); }       // (the layout is a few times the CSR arrays in host memory: an allocation that fails must come back as an error code)
  catch (const std::bad_alloc &) { *out = nullptr; return (int)hipErrorOutOfMemory; }
  catch (...) { *out = nullptr; return (int)hipErrorUnknown; }
}
static int tl_build_impl(int m, int n, const int *ai, const int *aj, int stage_min, mi355x_spmv_tiled_t *out) {
  if (stage_min <= 0) stage_min = 1024;
  std::unique_ptr<tl_host> Hown(new tl_host());
  tl_host *H = Hown.get();
  H->m = m; H->n = n; H->nnz = m ? ai[m] : 0;
  // panels: equal shares of the nonzeros (<= TL_PANEL rows each), and the same number on every CU when there are more than CUs -- the
  // product runs at the rate of the CU with the most panels (306 panels on 256 CUs cost what 512 do: profiles/r04_tiled_sweep5.log)
  {
    const int slots = TL_NCU;
    long np = ((long)m + TL_PANEL - 1) / TL_PANEL;
    if (np > slots) np = (np + slots - 1) / slots * slots;
    const char *e_ = getenv("MI355X_TILED_PANELS");
    const bool forced = e_ && atol(e_) > 0;
    if (forced) np = atol(e_);
    if (np < 1) np = 1;
    // the smallest bound on a panel's nonzeros with which the rows fit np panels of <= TL_PANEL rows (greedy cuts, bisection on the
    // bound); rows too sparse for that (the row cap binds): another panel per CU
    auto cut = [&](long bound, std::vector<int> *outv) -> long {
      long cnt = 0;
      int r = 0;
      if (outv) { outv->clear(); outv->push_back(0); }
      while (r < m) {
        int end = (int)(std::upper_bound(ai + r, ai + m + 1, (long)ai[r] + bound, [](long t, int v) { return t < (long)v; }) - ai) - 1;   // last end with ai[end] - ai[r] <= bound
        if (end <= r) end = r + 1;                         // (a row heavier than the bound: a panel of its own)
        if (end > r + TL_PANEL) end = r + TL_PANEL;
        if (end > m) end = m;
        if (outv) outv->push_back(end);
        r = end;
        ++cnt;
      }
      return cnt;
    };
    if (m > 0) {
      while (!forced && cut(H->nnz + 1, nullptr) > np) np += (np >= slots) ? slots : 1;
      long lo = H->nnz / np, hi = H->nnz + 1;              // cut(hi) <= np holds (or the count is forced: the best the row cap allows)
      if (lo < 1) lo = 1;
      while (lo < hi) {
        const long mid = lo + (hi - lo) / 2;
        if (cut(mid, nullptr) <= np) hi = mid; else lo = mid + 1;
      }
      cut(hi, &H->prow);
    } else H->prow.push_back(0);
    H->npanels = (int)H->prow.size() - 1;
  }
  const int ntiles = (n + TL_TW - 1) / TL_TW;
  std::vector<PanelOut> po((size_t)H->npanels);
  {
    int nth = mi355x_host_threads(16);
    if (H->nnz < 2000000) nth = 1;
    if (nth > H->npanels) nth = H->npanels > 0 ? H->npanels : 1;
    std::atomic<int> next(0), failed(0);
    auto work = [&]() {
      try {
        std::vector<int> cnt((size_t)ntiles + 1, 0), touched, cur((size_t)TL_PANEL, 0);
        std::vector<char> stagedflag((size_t)ntiles + 1, 0);
        for (int p = next.fetch_add(1); p < H->npanels && !failed.load(); p = next.fetch_add(1))
          build_panel(H->prow[(size_t)p], H->prow[(size_t)p + 1], ai, aj, stage_min, cnt, touched, cur, stagedflag, po[(size_t)p]);
      } catch (...) { failed.store(1); }                  // (an exception must not leave a thread)
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nth; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (failed.load()) return (int)hipErrorOutOfMemory;
  }
  // concatenate: panel after panel, inside a panel wavefront after wavefront
  size_t npt = 0, nst = 0;
  for (auto &o : po) {
    npt += o.pt_tile.size(); H->nnz_near += o.near; H->nnz_far += o.far;
    for (int w = 0; w < TL_WAVES; ++w) nst += o.perm[w].size();
  }
  if (nst >= (size_t)1 << 31) return (int)hipErrorInvalidValue;
  H->pt_ptr.resize((size_t)H->npanels + 1);
  H->pt_tile.reserve(npt); H->pw_e0.reserve((size_t)H->npanels * TL_WAVES + 1); H->wrow.reserve((size_t)H->npanels * (TL_WAVES + 1));
  H->perm.reserve(nst); H->word.reserve(nst);
  long e = 0;
  for (int p = 0; p < H->npanels; ++p) {
    PanelOut &o = po[(size_t)p];
    H->pt_ptr[(size_t)p] = (int)H->pt_tile.size();
    H->pt_tile.insert(H->pt_tile.end(), o.pt_tile.begin(), o.pt_tile.end());
    H->wrow.insert(H->wrow.end(), o.wrow, o.wrow + TL_WAVES + 1);
    for (int w = 0; w < TL_WAVES; ++w) {
      H->pw_e0.push_back((int)e); H->pw_f0.push_back((int)(e + o.nstaged[w])); e += (long)o.perm[w].size();
      H->fw_ptr.push_back((int)H->fw_win.size());
      H->fw_win.insert(H->fw_win.end(), o.fwin[w].begin(), o.fwin[w].end());
      H->perm.insert(H->perm.end(), o.perm[w].begin(), o.perm[w].end());
      H->word.insert(H->word.end(), o.word[w].begin(), o.word[w].end());
      std::vector<int>().swap(o.perm[w]); std::vector<unsigned int>().swap(o.word[w]);
    }
  }
  H->pt_ptr[(size_t)H->npanels] = (int)H->pt_tile.size();
  H->pw_e0.push_back((int)e);
  H->fw_ptr.push_back((int)H->fw_win.size());
  H->nstore = e;
  if (H->nnz_near + H->nnz_far != H->nnz) return (int)hipErrorUnknown;
  if (e != H->nstore) return (int)hipErrorUnknown;
  mi355x_spmv_tiled_s *P = new mi355x_spmv_tiled_s();
  memset(P, 0, sizeof(*P));
  P->host = Hown.release();
  P->m = m; P->n = n; P->npanels = H->npanels; P->npt = (int)H->pt_tile.size();
  P->nnz_near = H->nnz_near; P->nnz_far = H->nnz_far; P->nstore = H->nstore;
  *out = P;
  return 0;
}

// nblocks: 128-entry blocks over all wavefronts (nnz_staged / (128 nblocks) = the share of the stored entries that are not padding)
int mi355x_spmv_tiled_info(mi355x_spmv_tiled_t P, long *nnz_staged, long *nnz_remainder, int *npanels, int *npairs, long *nblocks) {
  if (nnz_staged) *nnz_staged = P->nnz_near;
  if (nnz_remainder) *nnz_remainder = P->nnz_far;
  if (npanels) *npanels = P->npanels;
  if (npairs) *npairs = P->npt;
  if (nblocks) *nblocks = P->nstore / TL_BLOCK;
  return 0;
}
int mi355x_spmv_tiled_geometry(int *panel_rows, int *tile_cols, int *waves, int *block_entries) {
  *panel_rows = TL_PANEL; *tile_cols = TL_TW; *waves = TL_WAVES; *block_entries = TL_BLOCK;
  return 0;
}

// tests: one host array of the layout (which: 0 pt_ptr, 1 pt_tile, 2 pw_e0, 3 word, 4 perm, 7 pw_f0, 8 fw_ptr, 9 fw_win, 10 wrow, 11 prow);
// available until mi355x_spmv_tiled_drop_host
int mi355x_spmv_tiled_debug_get(mi355x_spmv_tiled_t P, int which, void *out, size_t cap_bytes, size_t *bytes) {
  if (!P->host) return (int)hipErrorInvalidValue;
  tl_host *H = P->host;
  const void *src = nullptr; size_t nb = 0;
  switch (which) {
    case 0: src = H->pt_ptr.data(); nb = H->pt_ptr.size() * 4; break;
    case 1: src = H->pt_tile.data(); nb = H->pt_tile.size() * 4; break;
    case 2: src = H->pw_e0.data(); nb = H->pw_e0.size() * 4; break;
    case 3: src = H->word.data(); nb = H->word.size() * 4; break;
    case 4: src = H->perm.data(); nb = H->perm.size() * 4; break;
    case 7: src = H->pw_f0.data(); nb = H->pw_f0.size() * 4; break;
    case 8: src = H->fw_ptr.data(); nb = H->fw_ptr.size() * 4; break;
    case 9: src = H->fw_win.data(); nb = H->fw_win.size() * 4; break;
    case 10: src = H->wrow.data(); nb = H->wrow.size() * 4; break;
    case 11: src = H->prow.data(); nb = H->prow.size() * 4; break;
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
// idle one of two LDS buffers while the others gather from the current one: a tile switch is one barrier, no load on anybody's path
// (the row sums need no barrier: a row belongs to one wavefront).  A gathering wavefront keeps TL_NG blocks' loads in flight, across
// the tile switches -- the loads do not depend on the tile of x, only their use does.  Every load is issued on every path (past the
// stream's end: the last block again), so that the number of loads in flight is the same wherever the code is: a conditional issue
// makes the compiler wait for the younger blocks' loads too, and the pipeline is one block deep whatever the source says.
// After its staged blocks a gathering wavefront walks its remainder blocks: the same 12-byte entries, x gathered from global memory
// (window base + 18-bit column).  2 TL_FG blocks in a ring: a block's two stream loads go out 2 TL_FG blocks ahead, its two gathers
// TL_FG blocks ahead (their addresses are the words just arrived), both on every path; no barrier -- a row sum is still its
// wavefront's own.  (Gathers of 2 / 4 / 8 blocks in flight: 153 / 207 / 187 G gathers/s in profiles/r04_dense_atomic_probe2.log.)
struct tl_blk { tl_v2d v; tl_v2u c; };
struct tl_fblk { tl_v2d v; tl_v2u c; double x0, x1; };

__device__ __forceinline__ void tl_lds_add(double *p, double v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }   // ds_add_f64

#ifdef TL_PROFILE
// development build: per panel {start, end (100 MHz wall clock), hardware id (XCC << 32 | HW_ID), core cycles wavefront 0 spent in tile-switch barriers,
// the loader: cycles from a tile's first load to its arrival, cycles at the barriers, wavefront 0: cycles of its whole gather loop}
__device__ unsigned long long tl_prof_buf[8 * 8192];
#endif

template <int ADD>
__global__ __launch_bounds__((TL_WAVES + 1) * 64) void spmv_tiled_kernel(
    int npanels, int chunkx, const int *__restrict__ prow, const int *__restrict__ pt_ptr, const int *__restrict__ pt_tile,
    const int *__restrict__ pw_e0, const int *__restrict__ pw_f0, const int *__restrict__ fw_ptr, const int *__restrict__ fw_win,
    const tl_v2d *__restrict__ val, const tl_v2u *__restrict__ word,
    const double *__restrict__ x, const double *yin, double *yout, int n, int phases) {
  extern __shared__ __attribute__((aligned(16))) double tl_lds[];
  double *xt = tl_lds;                                   // 2 x TL_TW doubles: two tiles of x
  double *acc = tl_lds + 2 * TL_TW;                      // the panel's row sums, and the padding's
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NT = (TL_WAVES + 1) * 64;

  // each XCD walks a contiguous eighth of the panels: neighbouring panels stage the same tiles, out of the same L2
  const int xcd = blockIdx.x % MI355X_NXCD, slot = blockIdx.x / MI355X_NXCD;
  const int p = xcd * chunkx + slot;
  if (slot >= chunkx || p >= npanels) return;
#ifdef TL_PROFILE
  unsigned long long prof_t0 = __builtin_amdgcn_s_memrealtime(), prof_bar = 0, prof_ld = 0, prof_lbar = 0, prof_loop = 0;
#endif

  const int row0 = prow[p], nrow = prow[p + 1] - row0;
  for (int rl = tid; rl < TL_SLOTS; rl += NT) acc[rl] = (ADD && rl < nrow) ? yin[row0 + rl] : 0.0;

  const int pt0 = pt_ptr[p], ntp = (phases & 1) ? pt_ptr[p + 1] - pt0 : 0;     // (phases: 1 the staged tiles, 2 the remainder; development: their separate cost)
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
    // (straight into LDS -- global_load_lds_dwordx4: lane l's 16 bytes land at the instruction's LDS base + 16 l -- so the whole tile is
    //  in flight at once without a register: one memory round trip per tile, and the kernel needs 40 VGPRs instead of 78.  Through
    //  registers, 16 loads at a time, a 4096-column tile took two round trips behind the gathering wavefronts' loads in the CU's queue
    //  and the gathering wavefronts waited at the tile switch for most of the kernel; timeline of the present form, from the
    //  TL_PROFILE build: profiles/r04_tiled_prof_v7.log)
    for (int i = 0; i + 1 < ntp; ++i) {
#ifdef TL_PROFILE
      const unsigned long long pq0 = __builtin_readcyclecounter();
#endif
      const int t = pt_tile[pt0 + i + 1];
      double *buf = xt + ((i & 1) ^ 1) * TL_TW;
      const size_t base = (size_t)t * TL_TW;
      const int ncol = (n - (long)base) < TL_TW ? (int)(n - (long)base) : TL_TW;
      const int n2 = ncol >> 1;
      const tl_v2d *xs = reinterpret_cast<const tl_v2d *>(x + base);
#pragma unroll
      for (int k = 0; k < TL_TW / 128; ++k) {
        if (k * 64 < n2) {                               // (a short last tile: whole instructions past its end are skipped, lanes past it re-read its first pair into columns nobody gathers)
          const int j = k * 64 + lane;
          __builtin_amdgcn_global_load_lds(xs + (j < n2 ? j : 0), (__attribute__((address_space(3))) void *)(buf + k * 128), 16, 0, 0);
        }
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): the tile is in LDS
      tile_tail(t, buf);
#ifdef TL_PROFILE
      const unsigned long long pq1 = __builtin_readcyclecounter();
      prof_ld += pq1 - pq0;
#endif
      __syncthreads();                                   // (the gathering wavefronts' switch to tile i + 1)
#ifdef TL_PROFILE
      prof_lbar += __builtin_readcyclecounter() - pq1;
#endif
    }
  } else {
    // ---- a gathering wavefront ----
    const int ipw = p * TL_WAVES + w;
    const int e0 = pw_e0[ipw], f0 = pw_f0[ipw];
    const int nblocks = (phases & 1) ? (f0 - e0) / TL_BLOCK : 0;
    const int nfb = (phases & 2) ? (pw_e0[ipw + 1] - f0) / TL_BLOCK : 0;
    const tl_v2d *vq = val + (e0 >> 1) + lane;           // + 64 per block
    const tl_v2u *cq = word + (e0 >> 1) + lane;
    if (nblocks == 0) {                                  // no staged tile in this panel (true for all its wavefronts alike)
      __syncthreads();
    } else {
      tl_blk blk[TL_NG];
#pragma unroll
      for (int k = 0; k < TL_NG; ++k) {
        const size_t o = (size_t)(k < nblocks ? k : nblocks - 1) * 64;
        blk[k].v = __builtin_nontemporal_load(vq + o); blk[k].c = __builtin_nontemporal_load(cq + o);
      }
      __syncthreads();                                   // first tile in place, the sums' first stores done
      int it = -1;                                       // the tile being gathered from (index among the panel's staged tiles)
      const double *xc = xt;
#ifdef TL_PROFILE
      const unsigned long long pl0 = __builtin_readcyclecounter();
#endif
      for (int b = 0; b < nblocks; b += TL_NG) {
#pragma unroll
        for (int k = 0; k < TL_NG; ++k) {
          if (b + k < nblocks) {
            const unsigned int c0 = blk[k].c.x, c1 = blk[k].c.y;
            if ((unsigned int)__builtin_amdgcn_readfirstlane((int)c0) & TL_WORD_NEWTILE) {
#ifdef TL_PROFILE
              const unsigned long long pb0 = __builtin_readcyclecounter();
#endif
              if (it >= 0) __syncthreads();              // nobody reads tile `it` any more; the loader has completed the next one
#ifdef TL_PROFILE
              prof_bar += __builtin_readcyclecounter() - pb0;
#endif
              ++it;
              xc = xt + (it & 1) * TL_TW;
            }
            const double x0 = xc[c0 & (TL_WORD_NEWTILE - 1u)], x1 = xc[c1 & (TL_WORD_NEWTILE - 1u)];
            const double p0 = blk[k].v.x * x0, p1 = blk[k].v.y * x1;
            tl_lds_add(acc + (c0 >> 16), p0);            // entries 0 .. 63 of the block, ascending with the lane
            tl_lds_add(acc + (c1 >> 16), p1);            // entries 64 .. 127
          }
          const int nb = b + k + TL_NG;
          const size_t o = (size_t)(nb < nblocks ? nb : nblocks - 1) * 64;
          blk[k].v = __builtin_nontemporal_load(vq + o); blk[k].c = __builtin_nontemporal_load(cq + o);
        }
      }
#ifdef TL_PROFILE
      prof_loop = __builtin_readcyclecounter() - pl0;
#endif
    }
    if (nfb > 0) {
      // ---- the remainder: x from global memory, window after window ----
      const tl_v2d *vf = val + (f0 >> 1) + lane;
      const tl_v2u *cf = word + (f0 >> 1) + lane;
      const int *wl = fw_win + fw_ptr[ipw];              // the wavefront's windows in order
      int iw = -1;                                       // (the gathers' side: they run two blocks ahead of the sums)
      const double *xw = x;
      constexpr int FR = 2 * TL_FG;
      tl_fblk s[FR];
      auto stream = [&](tl_fblk &q, int blk) {
        const size_t o = (size_t)(blk < nfb ? blk : nfb - 1) * 64;
        q.v = __builtin_nontemporal_load(vf + o); q.c = __builtin_nontemporal_load(cf + o);
      };
      auto gather = [&](tl_fblk &q, int blk) {
        if (blk < nfb && ((unsigned int)__builtin_amdgcn_readfirstlane((int)q.c.x) & TL_WORD_NEWWIN)) { ++iw; xw = x + ((size_t)wl[iw] << TL_FW_SHIFT); }
        q.x0 = xw[q.c.x & ((1u << TL_FW_SHIFT) - 1u)]; q.x1 = xw[q.c.y & ((1u << TL_FW_SHIFT) - 1u)];
      };
#pragma unroll
      for (int k = 0; k < FR; ++k) stream(s[k], k);
#pragma unroll
      for (int k = 0; k < TL_FG; ++k) gather(s[k], k);
      for (int b = 0; b < nfb; b += FR) {
#pragma unroll
        for (int k = 0; k < FR; ++k) {
          gather(s[(k + TL_FG) % FR], b + k + TL_FG);
          if (b + k < nfb) {
            tl_lds_add(acc + ((s[k].c.x >> TL_FW_SHIFT) & (unsigned int)(0x1fff)), s[k].v.x * s[k].x0);
            tl_lds_add(acc + ((s[k].c.y >> TL_FW_SHIFT) & (unsigned int)(0x1fff)), s[k].v.y * s[k].x1);
          }
          stream(s[k], b + k + FR);
        }
      }
    }
  }
  __syncthreads();
  for (int rl = tid; rl < nrow; rl += NT) yout[row0 + rl] = acc[rl];
#ifdef TL_PROFILE
  if (tid == 0 && p < 8192) {
    tl_prof_buf[8 * p + 0] = prof_t0;
    tl_prof_buf[8 * p + 1] = __builtin_amdgcn_s_memrealtime();
    tl_prof_buf[8 * p + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | (unsigned int)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    tl_prof_buf[8 * p + 3] = prof_bar; tl_prof_buf[8 * p + 6] = prof_loop;
  }
  if (tid == TL_WAVES * 64 && p < 8192) { tl_prof_buf[8 * p + 4] = prof_ld; tl_prof_buf[8 * p + 5] = prof_lbar; }
#endif
}

extern "C" {

// device part: tables up, values gathered from the CSR value array on the device (aa_dev)
int mi355x_spmv_tiled_upload(mi355x_handle_t h, mi355x_spmv_tiled_t P, const double *aa_dev) {
  tl_host *H = P->host;
  if (!H) return (int)hipErrorInvalidValue;
  auto up = [&](void **d, const void *src, size_t nbytes) -> int {
    MI355X_TRY(hipMalloc(d, (nbytes ? nbytes : 1) + 1024));
    if (nbytes) MI355X_TRY(hipMemcpyAsync(*d, src, nbytes, hipMemcpyHostToDevice, h->stream));
    return 0;
  };
  int rc;
  if ((rc = up((void **)&P->d_pt_ptr, H->pt_ptr.data(), H->pt_ptr.size() * 4)) || (rc = up((void **)&P->d_pt_tile, H->pt_tile.data(), H->pt_tile.size() * 4)) ||
      (rc = up((void **)&P->d_prow, H->prow.data(), H->prow.size() * 4)) || (rc = up((void **)&P->d_pw_e0, H->pw_e0.data(), H->pw_e0.size() * 4)) ||
      (rc = up((void **)&P->d_perm, H->perm.data(), H->perm.size() * 4)) || (rc = up((void **)&P->d_word, H->word.data(), H->word.size() * 4)) ||
      (rc = up((void **)&P->d_pw_f0, H->pw_f0.data(), H->pw_f0.size() * 4)) || (rc = up((void **)&P->d_fw_ptr, H->fw_ptr.data(), H->fw_ptr.size() * 4)) ||
      (rc = up((void **)&P->d_fw_win, H->fw_win.data(), H->fw_win.size() * 4)))
    return rc;
  MI355X_TRY(hipMalloc((void **)&P->d_val, sizeof(double) * (size_t)(P->nstore > 0 ? P->nstore : 1) + 1024));
  MI355X_TRY(hipStreamSynchronize(h->stream));
  return mi355x_spmv_tiled_refresh_values(h, P, aa_dev);
}

// the CSR values on the device changed (same pattern): one gather
int mi355x_spmv_tiled_refresh_values(mi355x_handle_t h, mi355x_spmv_tiled_t P, const double *aa_dev) {
  if (P->nstore > 0) {
    hipLaunchKernelGGL(tl_gather_values_kernel, dim3(mi355x_grid_for((size_t)P->nstore, 4)), dim3(256), 0, h->stream, P->d_perm, aa_dev, P->d_val, P->nstore);
    MI355X_LAUNCH_CHECK();
  }
  return 0;
}

// y = A x (yin == NULL) or yout = yin + A x (yout may alias yin).  x must be 16-byte aligned (hipErrorNotSupported otherwise: the caller
// takes the row-block kernel).  which: 0 both parts, 1 the staged part only, 2 the remainder only (development: their separate cost).
// (History of the remainder: a CSR of its own added by the row-block kernel in column ranges of 3 MiB of x, four launches and four
//  read-modify-writes of y, 0.17 ms on the stand-in; on a stream of its own beside the staged kernel: no gain; inside the kernel, as
//  it is now, 0.14 ms: profiles/r04_tiled_sweep10.log, r04_tiled_sweep11.log; its counters r04_tiled_pmc_far.csv.  Both kinds of
//  blocks in ONE loop, one remainder block per four staged ones -- the staged part bound by HBM, the remainder by the address units --
//  was built too and passes the same bitwise tests against its own interleaved order: 0.32-0.33 ms against 0.34, whatever the depth of
//  the rings (profiles/r04_tiled_sweep12.log); not kept: 5 % for a row's products no longer being added in column order.)
int mi355x_spmv_tiled_parts(mi355x_handle_t h, mi355x_spmv_tiled_t P, const double *x, const double *yin, double *yout, int which) {
  if (!mi355x_aligned16(x)) return (int)hipErrorNotSupported;
  if (P->m == 0) return 0;
  const size_t lds = TL_LDS_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    if (getenv("MI355X_TILED_DEBUG")) {
      int nb = 0;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, spmv_tiled_kernel<0>, (TL_WAVES + 1) * 64, lds);
      fprintf(stderr, "[mi355x tiled] tile %d columns, panel <= %d rows, %d + 1 wavefronts per workgroup, %d blocks of %d entries in flight, %zu B of LDS: %d workgroups per CU\n", TL_TW, TL_PANEL, TL_WAVES, TL_NG, TL_BLOCK, lds, nb);
    }
    MI355X_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(spmv_tiled_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    MI355X_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(spmv_tiled_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int chunkx = (P->npanels + MI355X_NXCD - 1) / MI355X_NXCD;
  const int grid = chunkx * MI355X_NXCD;
  {
#define TL_GO(A_, YIN) hipLaunchKernelGGL((spmv_tiled_kernel<A_>), dim3(grid), dim3((TL_WAVES + 1) * 64), lds, h->stream, P->npanels, chunkx, P->d_prow, P->d_pt_ptr, \
                                          P->d_pt_tile, P->d_pw_e0, P->d_pw_f0, P->d_fw_ptr, P->d_fw_win, reinterpret_cast<const tl_v2d *>(P->d_val), reinterpret_cast<const tl_v2u *>(P->d_word), x, YIN, yout, P->n, \
                                          which == 1 ? 1 : which == 2 ? 2 : 3)
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
  return 0;
}
int mi355x_spmv_tiled(mi355x_handle_t h, mi355x_spmv_tiled_t P, const double *x, const double *yin, double *yout) {
  return mi355x_spmv_tiled_parts(h, P, x, yin, yout, 0);
}

int mi355x_spmv_tiled_destroy(mi355x_spmv_tiled_t P) {
  if (!P) return 0;
  delete P->host;
  void *ptrs[] = {P->d_prow, P->d_pt_ptr, P->d_pt_tile, P->d_pw_e0, P->d_pw_f0, P->d_fw_ptr, P->d_fw_win, P->d_perm, P->d_word, P->d_val};
  for (void *q : ptrs) if (q) (void)hipFree(q);
  delete P;
  return 0;
}

}  // extern "C"
