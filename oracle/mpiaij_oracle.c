/*
 * mpiaij_oracle.c -- TEST INFRASTRUCTURE (see oracle.h).  Sequential restatement of the integer
 * set-up work behind MatMult_MPIAIJ: the diagonal/off-diagonal split, garray, the compaction of
 * the off-diagonal block's column indices, and the VecScatter index lists.  All of it must
 * match the product bit for bit.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

static int cmp_int(const void *a, const void *b) {
  int x = *(const int *)a, y = *(const int *)b;
  return (x > y) - (x < y);
}

/* Column test of MatSetValues_MPIAIJ (src/mat/impls/aij/mpi/mpiaij.c:517-560): a column in
 * [cstart,cend) goes to the diagonal block with index col-cstart, anything else to the
 * off-diagonal block with its global index.  Then MatSetUpMultiply_MPIAIJ
 * (src/mat/impls/aij/mpi/mmaij.c:27-66): garray = sorted distinct global columns of B,
 * B.j[] replaced by the position in garray. */
int orc_mpiaij_split(int rstart, int rend, int cstart, int cend, const int *ai, const int *aj, const double *aa,
                     int *ad_i, int *ad_j, double *ad_a, int *bo_i, int *bo_j, double *bo_a, int *garray) {
  int na = 0, nb = 0, ec = 0;
  ad_i[0] = 0; bo_i[0] = 0;
  for (int r = rstart; r < rend; r++) {
    for (int k = ai[r]; k < ai[r + 1]; k++) {
      int c = aj[k];
      if (c >= cstart && c < cend) { ad_j[na] = c - cstart; ad_a[na++] = aa[k]; }
      else { bo_j[nb] = c; bo_a[nb++] = aa[k]; }
    }
    ad_i[r - rstart + 1] = na;
    bo_i[r - rstart + 1] = nb;
  }
  if (nb) {
    int *tmp = (int *)malloc(sizeof(int) * (size_t)nb);
    memcpy(tmp, bo_j, sizeof(int) * (size_t)nb);
    qsort(tmp, (size_t)nb, sizeof(int), cmp_int);
    for (int k = 0; k < nb; k++) if (k == 0 || tmp[k] != tmp[k - 1]) garray[ec++] = tmp[k];
    free(tmp);
    for (int k = 0; k < nb; k++) {
      int *p = (int *)bsearch(&bo_j[k], garray, (size_t)ec, sizeof(int), cmp_int);
      bo_j[k] = (int)(p - garray);
    }
  }
  return ec;
}

/* Owner of global index idx under ranges[] -- the search of vpscat.c:1762-1772. */
static int owner_of(int size, const int *ranges, int idx) {
  for (int j = 0; j < size; j++) if (idx < ranges[j + 1]) return j;
  return -1;
}

/* VecScatterCreate_PtoS, src/vec/vec/utils/vpscat.c:1730-1924, for inidx = garray (general IS) and
 * inidy = 0..ec-1 (stride IS), as MatSetUpMultiply_MPIAIJ sets it up (mmaij.c:131-145):
 *  - "from" (receive) side: one entry per owning rank with a non-empty request, ranks ascending
 *    (:1871-1878); indices = destination slots inidy[i] in order of appearance (:1880-1885);
 *  - "to" (send) side: requesting ranks sorted ascending (PetscSortMPIIntWithArray :1782);
 *    indices = requested global index - owners[rank], in the requester's order (:1846-1856);
 *  - local part (:1897-1910). */
void orc_scatter_create(int size, int rank, const int *ranges, const int *const *garrays, const int *ecs,
                        int *nrecv, int *rprocs, int *rstarts, int *rindices,
                        int *nsend, int *sprocs, int *sstarts, int *sindices,
                        int *nlocal, int *lto, int *lfrom) {
  /* receive side of `rank` */
  int nr = 0, cnt = 0;
  rstarts[0] = 0;
  for (int p = 0; p < size; p++) {
    if (p == rank) continue;
    int have = 0;
    for (int i = 0; i < ecs[rank]; i++) {
      if (owner_of(size, ranges, garrays[rank][i]) == p) { rindices[cnt++] = i; have = 1; }
    }
    if (have) { rprocs[nr++] = p; rstarts[nr] = cnt; }
  }
  *nrecv = nr;
  /* local part */
  int nl = 0;
  for (int i = 0; i < ecs[rank]; i++) {
    int g = garrays[rank][i];
    if (g >= ranges[rank] && g < ranges[rank + 1]) { lto[nl] = g - ranges[rank]; lfrom[nl++] = i; }
  }
  *nlocal = nl;
  /* send side of `rank`: what every other rank q requests from it */
  int ns = 0; cnt = 0;
  sstarts[0] = 0;
  for (int q = 0; q < size; q++) {
    if (q == rank) continue;
    int have = 0;
    for (int i = 0; i < ecs[q]; i++) {
      int g = garrays[q][i];
      if (g >= ranges[rank] && g < ranges[rank + 1]) { sindices[cnt++] = g - ranges[rank]; have = 1; }
    }
    if (have) { sprocs[ns++] = q; sstarts[ns] = cnt; }
  }
  *nsend = ns;
}
