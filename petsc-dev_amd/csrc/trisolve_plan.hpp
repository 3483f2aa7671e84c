// Plan of a sync-free triangular solve: shared by the kernels + host analysis (trisolve.hip) and the device-side construction
// of the row plans (trisolve_build.hip).
#pragma once
#include "common.hpp"

#define TRI_SENTINEL 0xFFF8DEADBEEFCAFEull
#define TRI_QUEUES 8
#define TRI_QSTRIDE 16          // queue counters 64 bytes apart
#define TRI_ALIGN_MIN 32        // by_level plans: levels of at least this many rows start on a slice boundary
#define TRI_SPIN_LIMIT (1 << 19)   // x ~1 us per poll once backed off: gives up after ~0.5 s
#define TRI_XCD_WORD (TRI_QUEUES * TRI_QSTRIDE)   // behind the queue counters: [0] the XCD that solves (0xffffffff: not chosen yet), [1] tickets


struct mi355x_trisolve_plan_s {
  int n, nslices, nchunks, upper;
  int *d_ptr;        // nslices + 1 entry offsets (multiples of 64)
  int *d_info;       // per position: (row length << 8) | sub-step inside the slice; padding positions: 0
  int *d_row;        // per position: the row it holds, -1 for padding
  int *d_col;        // sliced-ELL column POSITIONS
  double *d_val;     // sliced-ELL values
  double *d_dinv;    // upper: inverted diagonal per position
  double *d_rscale;  // upper, optional: factor applied to the right-hand side entry before the row's sum starts (ICC: D^-1 between the two solves)
  unsigned char *d_nsub;   // sub-steps per slice
  int *d_pos;        // per row: its position (for the other solve's gather of this solve's result)
  double *d_w;       // solution in position order, 64 * nslices doubles
  unsigned int *d_queue;   // TRI_QUEUES counters
  int *abort_flag;   // pinned + mapped
  int grid, sleep_cap;
  int by_level;      // rows in dependency-level order, levels on slice boundaries
  int nlev;
  int *levpos;       // host, 2 * nlev: first and one-past-last position of every dependency level (level-by-level fall-back)
  // node plans (nb > 1): a position holds a NODE -- up to nb consecutive rows with one shared column list (the inodes of the
  // reference, Mat_CheckInode); d_row = the node's first row, d_col = SLOTS (k * np + position) of the shared columns, d_val = nb
  // values per shared column, d_din = the couplings inside the node and (upper) the inverted diagonals, d_w = nb * np slots
  int nb, np;
  int spw;                 // node plans: slices (waves) per workgroup
  int split, ring, maxcol; // split-role kernel (loader + solver wavefront per workgroup): on, batches in the LDS ring, widest slice
  int one_xcd;             // split-role kernels: all participating workgroups on one XCD (hand-offs through its L2)
  int blkcols;             // the shared lists hold whole dependency nodes: one list entry per node, solution stored node by node
  unsigned char *d_nsz;    // per position: rows in the node
  double *d_din;           // [nb (nb - 1) / 2 + nb][np]
};

// the part of a plan that does not depend on who laid out the sliced-ELL arrays: solution vector armed with the sentinel, queue
// counters, abort flag, launch geometry (trisolve.hip)
int trisolve_plan_finish(mi355x_handle_t h, mi355x_trisolve_plan_s *p, int nlev, int by_level);
// construction of a row plan on the device from the host's factor arrays (trisolve_build.hip); same arrays as the host route's
int trisolve_plan_fill_device(mi355x_handle_t h, mi355x_trisolve_plan_s *p, int n, int nlev, const int *lev, const int *rp, const int *rl, const int *cj,
                              const double *cv, const double *dinv_host, const double *rscale_host, int by_level);
