/* MATSEQAIJHIPMI355X (and MATSEQBAIJHIPMI355X): host CSR container for assembly (the part of
 * HipAIJ the path needs: src/mat/impls/aij/seq/aij.h:10-39,99-115; MatSetValues_SeqAIJ aij.c:~330,
 * MatAssemblyEnd_SeqAIJ aij.c:~860) plus the device mirror and the ops the reference's GPU subclass
 * overrides (MatCreate_SeqAIJCUSP, src/mat/impls/aij/seq/seqcusp/aijcusp.cu:657-681): mult, multadd,
 * multtranspose[add], getdiagonal, assemblyend, getvecs, destroy. */
#include "hipmi355ximpl.h"
#include <time.h>

/* the host CSR container: this file's own on the harness, a view of the parent MATSEQAIJ's arrays inside a PETSc tree */
#define SA(A) HipAIJGet(A)
#define SD(A) ((Mat_SeqAIJHIP *)(A)->spptr)
PetscErrorCode MatSeqAIJGetArrays(Mat A, PetscInt *m, const PetscInt **i, const PetscInt **j, const PetscScalar **a);
static PetscErrorCode device_free(Mat A);
static PetscBool device_values_current(Mat A);

#if !defined(PETSCHIPMI355X_WITH_PETSC)   /* inside a PETSc tree the parent type MATSEQAIJ owns the container and its assembly (aij.c) */
/* ---------------------------------------------------------------- host container */
#define CHUNKSIZE 15   /* aij.h: rows grow by this many slots when preallocation is exceeded */
static PetscErrorCode seqaij_prealloc(Mat A, PetscInt nz, const PetscInt *nnz) {
  PetscErrorCode ierr;
  HipAIJ *a = SA(A);
  PetscInt m = a->m;
  if (nz == PETSC_DEFAULT || nz == PETSC_DECIDE) nz = 5;   /* aij.c MatSeqAIJSetPreallocation_SeqAIJ */
  if (nz < 0) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_OUTOFRANGE, "nz cannot be less than 0: value %d", nz);
  device_free(A);   /* a new pattern is coming: the mirror, plan, index dictionary, transpose and batch map go with the old one */
  HipFree(a->i); HipFree(a->j); HipFree(a->a); HipFree(a->ilen); HipFree(a->imax);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(m + 1), &a->i);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(m, 1), &a->ilen);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(m, 1), &a->imax);CHKERRQ(ierr);
  a->i[0] = 0;
  for (PetscInt r = 0; r < m; r++) {
    PetscInt c = nnz ? nnz[r] : nz;
    if (c < 0) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_OUTOFRANGE, "nnz cannot be less than 0: local row %d value %d", r, c);
    a->imax[r] = c; a->ilen[r] = 0; a->i[r + 1] = a->i[r] + c;
  }
  a->maxnz = a->i[m];
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(a->maxnz, 1), &a->j);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)PetscMax(a->maxnz, 1), &a->a);CHKERRQ(ierr);
  a->nz = 0; a->compact = PETSC_FALSE;
  A->preallocated = PETSC_TRUE;
  return 0;
}

/* grow row r by CHUNKSIZE slots (MatSeqXAIJReallocateAIJ, aij.h) */
static PetscErrorCode seqaij_grow(HipAIJ *a, PetscInt r) {
  PetscErrorCode ierr;
  PetscInt m = a->m, add = CHUNKSIZE, newmax = a->i[m] + add;
  PetscInt *nj; PetscScalar *na;
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)newmax, &nj);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)newmax, &na);CHKERRQ(ierr);
  PetscInt upto = a->i[r] + a->ilen[r];
  memcpy(nj, a->j, sizeof(PetscInt) * (size_t)upto);
  memcpy(na, a->a, sizeof(PetscScalar) * (size_t)upto);
  PetscInt tail = a->i[m] - a->i[r + 1];
  memcpy(nj + a->i[r + 1] + add, a->j + a->i[r + 1], sizeof(PetscInt) * (size_t)tail);
  memcpy(na + a->i[r + 1] + add, a->a + a->i[r + 1], sizeof(PetscScalar) * (size_t)tail);
  for (PetscInt q = r + 1; q <= m; q++) a->i[q] += add;
  a->imax[r] += add;
  HipFree(a->j); HipFree(a->a);
  a->j = nj; a->a = na; a->maxnz = newmax;
  return 0;
}

/* one entry of MatSetValues_SeqAIJ: sorted insertion into the row, INSERT or ADD on a hit */
static PetscErrorCode seqaij_set(HipAIJ *a, PetscInt r, PetscInt c, PetscScalar v, InsertMode mode, PetscBool *newnz) {
  PetscErrorCode ierr;
  PetscInt *rp = a->j + a->i[r], n = a->ilen[r], lo = 0, hi = n;
  PetscScalar *ap = a->a + a->i[r];
  while (hi - lo > 5) { PetscInt t = (lo + hi) / 2; if (rp[t] > c) hi = t; else lo = t; }
  PetscInt k;
  for (k = lo; k < n; k++) {
    if (rp[k] > c) break;
    if (rp[k] == c) { if (mode == ADD_VALUES) ap[k] += v; else ap[k] = v; return 0; }
  }
  if (n >= a->imax[r]) {
    ierr = seqaij_grow(a, r);CHKERRQ(ierr);
    rp = a->j + a->i[r]; ap = a->a + a->i[r];
  }
  for (PetscInt q = n - 1; q >= k; q--) { rp[q + 1] = rp[q]; ap[q + 1] = ap[q]; }
  rp[k] = c; ap[k] = v;
  a->ilen[r] = n + 1;
  a->nz++;
  if (newnz) *newnz = PETSC_TRUE;
  return 0;
}

/* MatAssemblyEnd_SeqAIJ (aij.c:~860-930): squeeze out the unused slots of every row */
static PetscErrorCode seqaij_compact(HipAIJ *a) {
  PetscInt m = a->m, shift = 0;
  a->nonzerorows = 0;
  for (PetscInt r = 0; r < m; r++) {
    PetscInt start = a->i[r], n = a->ilen[r];
    if (shift) {
      memmove(a->j + start - shift, a->j + start, sizeof(PetscInt) * (size_t)n);
      memmove(a->a + start - shift, a->a + start, sizeof(PetscScalar) * (size_t)n);
    }
    PetscInt slack = a->imax[r] - n;
    a->i[r] = start - shift;
    shift += slack;
    a->imax[r] = n;
    a->nonzerorows += (n > 0);
  }
  a->i[m] -= shift;
  a->nz = a->i[m];
  a->compact = PETSC_TRUE;
  return 0;
}

#else
static PetscErrorCode device_free(Mat A);
#endif

/* Mat_CheckInode (src/mat/impls/aij/seq/inode.c:3964-4034): consecutive rows with identical column lists form a node of
 * at most `limit` rows (-mat_inode_limit, default 5, inode2.c:85-99); with more than 0.8 m nodes -- or -mat_no_inode -- the
 * matrix keeps the plain routines (inode_count = 0). */
typedef struct { const PetscInt *ii, *jj; unsigned char *same; PetscInt m; } InodeCmp;
static void inode_compare_rows(void *c_, PetscInt lo, PetscInt hi) {     /* same[r]: row r + 1 has row r's column list */
  InodeCmp *c = (InodeCmp *)c_;
  for (PetscInt r = lo; r < hi; r++) {
    const PetscInt nzx = c->ii[r + 1] - c->ii[r];
    c->same[r] = (unsigned char)(r + 1 < c->m && c->ii[r + 2] - c->ii[r + 1] == nzx && !memcmp(c->jj + c->ii[r], c->jj + c->ii[r + 1], sizeof(PetscInt) * (size_t)nzx));
  }
}
static PetscErrorCode seqaij_check_inode(Mat A) {
  PetscErrorCode ierr;
  HipAIJ *a = SA(A);
  PetscInt m = a->m, limit = 5, i = 0, node_count = 0, *ns;
  PetscBool set; char buf[16];
  HipFree(a->inode_size); a->inode_size = NULL; a->inode_count = 0;
  ierr = PetscOptionsGetString(NULL, "-mat_no_inode", buf, sizeof(buf), &set);CHKERRQ(ierr);
  if (set || !m) return 0;
  ierr = PetscOptionsGetInt(NULL, "-mat_inode_limit", &limit, &set);CHKERRQ(ierr);
  if (limit < 1) limit = 1;
  if (limit > 5) limit = 5;
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(m + 1), &ns);CHKERRQ(ierr);
  /* the comparisons (row r + 1 against row r: "the same list as the node's first row" is transitive) are the pass over the
   * column indices and run on host threads; the greedy grouping of the reference's loop then reads one byte per row */
  unsigned char *same;
  ierr = PetscMalloc((size_t)m + 1, &same);CHKERRQ(ierr);
  { InodeCmp ic = {a->i, a->j, same, m};
    HipParallelRanges(m, inode_compare_rows, &ic); }
  while (i < m) {
    PetscInt j, blk_size;
    for (j = i + 1, blk_size = 1; j < m && blk_size < limit; ++j, ++blk_size) if (!same[j - 1]) break;
    ns[node_count++] = blk_size;
    i = j;
  }
  HipFree(same);
  if (node_count > .8 * m) { HipFree(ns); return 0; }
  a->inode_size = ns; a->inode_count = node_count;
  return 0;
}

/* ---------------------------------------------------------------- the type's options
 * -mat_hipmi355x_index_compression, _row_patterns, _value_patterns, _tiled, _tiled_stage_min.  MatSetFromOptions (ops->setfromoptions,
 * slot 76, matimpl.h:110; gcreate.c:201-203) reads them under the matrix's own options prefix and keeps them with the matrix; a matrix
 * that was never asked falls back to the global database when its device copy is built (blocks of an MPIAIJ matrix, matrices created
 * by MatCreateSeqAIJWithArrays and used at once). */
enum { HOPT_IC = 0, HOPT_RP, HOPT_VP, HOPT_TILED, HOPT_TILED_SMIN, HOPT_BLOCKED, HOPT_N };
static const char *const hopt_name[HOPT_N] = {"-mat_hipmi355x_index_compression", "-mat_hipmi355x_row_patterns", "-mat_hipmi355x_value_patterns",
                                              "-mat_hipmi355x_tiled", "-mat_hipmi355x_tiled_stage_min", "-mat_hipmi355x_blocked"};
static PetscErrorCode hip_mat_option(Mat A, int which, PetscInt *val) {
  Mat_SeqAIJHIP *d = SD(A);
  PetscBool set;
  if (d->opt_set[which]) { *val = d->opt[which]; return 0; }
  return PetscOptionsGetInt(NULL, hopt_name[which], val, &set);
}
static PetscErrorCode MatSetFromOptions_SeqAIJHIP(Mat A) {
  PetscErrorCode ierr;
  Mat_SeqAIJHIP *d = SD(A);
  for (int k = 0; k < HOPT_N; k++) {
    PetscInt v = 0; PetscBool set = PETSC_FALSE;
    ierr = PetscOptionsGetInt(HipObjPrefix(A), hopt_name[k], &v, &set);CHKERRQ(ierr);
    if (set && (!d->opt_set[k] || d->opt[k] != v)) {
      d->opt[k] = v; d->opt_set[k] = PETSC_TRUE;
      d->uploaded_state = -1; d->pattern_nz = -1;            /* the analyses run again with the new choice */
    }
  }
  return 0;
}

/* ---------------------------------------------------------------- device mirror */
static PetscErrorCode device_free(Mat A) {
  Mat_SeqAIJHIP *d = SD(A);
  if (!d) return 0;
  { PetscErrorCode ierr = VecHIPProductMatrixChanges(A);CHKERRQ(ierr); }   /* a noted product of this matrix runs while its arrays exist */
  if (d->d_i) mi355x_free(d->d_i);
  if (d->d_j) mi355x_free(d->d_j);
  if (d->d_a) mi355x_free(d->d_a);
  if (d->plan) mi355x_spmv_plan_destroy(d->plan);
  if (d->t_i) mi355x_free(d->t_i);
  if (d->t_j) mi355x_free(d->t_j);
  if (d->t_a) mi355x_free(d->t_a);
  if (d->t_plan) mi355x_spmv_plan_destroy(d->t_plan);
  if (d->t_tiled) { mi355x_spmv_tiled_destroy(d->t_tiled); d->t_tiled = NULL; }
  if (d->t_perm) mi355x_free(d->t_perm);
  if (d->tiled) mi355x_spmv_tiled_destroy(d->tiled);
  if (d->b_i) mi355x_free(d->b_i);
  if (d->b_j) mi355x_free(d->b_j);
  if (d->b_a) mi355x_free(d->b_a);
  if (d->b_perm) mi355x_free(d->b_perm);
  if (d->b_plan) mi355x_spmv_plan_destroy(d->b_plan);
  if (d->tb_i) mi355x_free(d->tb_i);
  if (d->tb_j) mi355x_free(d->tb_j);
  if (d->tb_a) mi355x_free(d->tb_a);
  if (d->tb_perm) mi355x_free(d->tb_perm);
  if (d->tb_plan) mi355x_spmv_plan_destroy(d->tb_plan);
  if (d->bm_order) mi355x_free(d->bm_order);
  if (d->bm_segptr) mi355x_free(d->bm_segptr);
  if (d->bm_segslot) mi355x_free(d->bm_segslot);
  if (d->bm_v) mi355x_free(d->bm_v);
  const PetscInt nup = d->n_uploads, tb = d->t_builds, tr = d->t_refreshes;
#if !defined(PETSCHIPMI355X_WITH_PETSC)
  HipTriFactors *tri = d->tri;
#endif
  const PetscBool cprow = d->cprow, timing = d->timing;
  const PetscInt tn = d->time_n, tcap = d->time_cap; mi355x_event_t *tev = d->time_ev;
  PetscInt opt[8]; PetscBool opt_set[8];
  memcpy(opt, d->opt, sizeof(opt)); memcpy(opt_set, d->opt_set, sizeof(opt_set));
  memset(d, 0, sizeof(*d));
  memcpy(d->opt, opt, sizeof(opt)); memcpy(d->opt_set, opt_set, sizeof(opt_set));   /* what MatSetFromOptions was told outlives the arrays */
  d->uploaded_state = -1; d->t_state = -1; d->pattern_nz = -1;
  d->n_uploads = nup; d->cprow = cprow;   /* a count and a request: they outlive the arrays */
  d->t_builds = tb; d->t_refreshes = tr;
#if !defined(PETSCHIPMI355X_WITH_PETSC)
  d->tri = tri;
#endif
  d->timing = timing; d->time_n = tn; d->time_cap = tcap; d->time_ev = tev;
  return 0;
}

/* MatCUSPCopyToGPU (aijcusp.cu:126-253): H2D of i, j, a when the host copy is newer.  Unlike the
 * reference, a value-only change (same pattern) re-sends only `a`. */
#if defined(PETSCHIPMI355X_WITH_PETSC)
static PetscErrorCode hipaij_refresh_view_if_stale(Mat A);   /* integration/petsc-3.3/aijhipmi355x_ctor.h */
#endif
/* the blocked companion (see MatSeqAIJHIPUpload): BCSR arrays of an AIJ matrix whose nodes are complete bs x bs blocks; leaves d->b_plan
 * NULL when the matrix is not of that shape */
static PetscErrorCode blocked_companion_build(Mat A, PetscDeviceCtx *dc) {
  PetscErrorCode ierr;
  HipAIJ *a = SA(A); Mat_SeqAIJHIP *d = SD(A);
  const PetscInt m = a->m, n = a->n, nn = a->inode_count;
  if (nn <= 0 || !a->inode_size) return 0;
  const PetscInt bs = a->inode_size[0];
  if (bs < 2 || bs > 5 || nn * bs != m || n % bs) return 0;
  for (PetscInt i = 0; i < nn; i++) if (a->inode_size[i] != bs) return 0;
  /* every node's column list: whole aligned groups of bs consecutive columns (the rows of a node share the list: check its first row) */
  PetscInt nblk = 0;
  for (PetscInt i = 0; i < nn; i++) {
    const PetscInt r = i * bs, k0 = a->i[r], len = a->i[r + 1] - k0;
    if (len % bs) return 0;
    for (PetscInt g = 0; g < len; g += bs) {
      const PetscInt c0 = a->j[k0 + g];
      if (c0 % bs) return 0;
      for (PetscInt q = 1; q < bs; q++) if (a->j[k0 + g + q] != c0 + q) return 0;
    }
    for (PetscInt q = 1; q < bs; q++) if (a->i[r + q + 1] - a->i[r + q] != len) return 0;     /* (Mat_CheckInode compared the lists themselves) */
    nblk += len / bs;
  }
  if ((double)nblk * bs * bs > 2147483000.0) return 0;
  PetscInt *bi, *bj, *perm, *sc;
  const PetscInt bs2 = bs * bs;
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(nn + 1), &bi);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(nblk, 1), &bj);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(nblk, 1) * (size_t)bs2, &perm);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(nn + 1), &sc);CHKERRQ(ierr);
  bi[0] = 0;
  for (PetscInt i = 0, b = 0; i < nn; i++) {
    const PetscInt r = i * bs, len = a->i[r + 1] - a->i[r];
    for (PetscInt g = 0; g < len; g += bs, b++) {
      bj[b] = a->j[a->i[r] + g] / bs;
      for (PetscInt c = 0; c < bs; c++) for (PetscInt q = 0; q < bs; q++) perm[(size_t)b * bs2 + c * bs + q] = a->i[r + q] + g + c;   /* value (row q, column c) of the block */
    }
    bi[i + 1] = bi[i] + len / bs;
  }
  for (PetscInt i = 0; i <= nn; i++) sc[i] = bi[i] * bs2;
  CHKHIP(mi355x_malloc((void **)&d->b_i, sizeof(PetscInt) * (size_t)(nn + 1)));
  CHKHIP(mi355x_malloc((void **)&d->b_j, sizeof(PetscInt) * (size_t)PetscMax(nblk, 1) + 16));
  CHKHIP(mi355x_malloc((void **)&d->b_perm, sizeof(PetscInt) * (size_t)PetscMax(nblk, 1) * (size_t)bs2));
  CHKHIP(mi355x_malloc((void **)&d->b_a, sizeof(PetscScalar) * (size_t)PetscMax(nblk, 1) * (size_t)bs2 + 16));
  CHKHIP(mi355x_memcpy_h2d(dc->h, d->b_i, bi, sizeof(PetscInt) * (size_t)(nn + 1)));
  CHKHIP(mi355x_memcpy_h2d(dc->h, d->b_j, bj, sizeof(PetscInt) * (size_t)nblk));
  CHKHIP(mi355x_memcpy_h2d(dc->h, d->b_perm, perm, sizeof(PetscInt) * (size_t)nblk * (size_t)bs2));
  CHKHIP(mi355x_spmv_plan_create(dc->h, nn, sc, NULL, &d->b_plan));
  CHKHIP(mi355x_handle_synchronize(dc->h));
  HipFree(bi); HipFree(bj); HipFree(perm); HipFree(sc);
  d->b_bs = bs; d->b_nblocks = nblk; d->b_fresh = PETSC_FALSE;
  return 0;
}
static PetscErrorCode blocked_values_current(Mat A, PetscDeviceCtx *dc) {      /* after a device-side change of d_a: one gather, when next used */
  Mat_SeqAIJHIP *d = SD(A);
  if (d->b_plan && !d->b_fresh) { CHKHIP(mi355x_pack(dc->h, (size_t)SA(A)->nz, d->b_perm, d->d_a, d->b_a)); d->b_fresh = PETSC_TRUE; }
  return 0;
}
PetscErrorCode MatSeqAIJHIPUpload(Mat A) {
  PetscErrorCode ierr;
  HipAIJ *a = SA(A);
  Mat_SeqAIJHIP *d = SD(A);
  PetscDeviceCtx *dc;
#if defined(PETSCHIPMI355X_WITH_PETSC)
  /* the parent class fills or replaces its arrays on paths that never pass this type's MatAssemblyEnd (MatDuplicate_SeqAIJ,
   * MatCopy, MatConvert set assembled = TRUE themselves): the view of them is checked before every use */
  ierr = hipaij_refresh_view_if_stale(A);CHKERRQ(ierr);
#endif
  if (d->uploaded_state == HipObjState(A) && d->d_a) return 0;
  ierr = VecHIPProductMatrixChanges(A);CHKERRQ(ierr);      /* (the device copy still holds the values the noted product was asked with) */
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  const int up_timing = getenv("PETSC_HIPMI355X_SETUP_TIMING") != NULL;
  double up_t0 = 0.0;
#define UP_TICK(what) do { if (up_timing) { struct timespec ts__; (void)mi355x_handle_synchronize(dc->h); clock_gettime(CLOCK_MONOTONIC, &ts__); const double t__ = (double)ts__.tv_sec + 1e-9 * (double)ts__.tv_nsec; \
    if (up_t0 > 0.0) { fprintf(stderr, "[hipmi355x]   upload: %-30s %.3f s\n", what, t__ - up_t0); } \
    up_t0 = t__; } } while (0)
  UP_TICK("");
  if (!a->compact) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_WRONGSTATE, "matrix must be assembled before it is sent to the GPU");
  PetscBool same_pattern = (PetscBool)(d->d_a && d->plan && d->pattern_nz == a->nz);   /* entries are never removed: same nz == same pattern */
  if (!same_pattern) {
    device_free(A);
    PetscInt m = a->m, nrows = m;
    const PetscInt *ip = a->i; PetscInt *ci = NULL, *ridx = NULL;
    /* compressed rows when >= 60% of the rows are empty (Mat_CheckCompressedRow ratio, compressedrow.c:28;
     * the reference forces it off for B, mpiaij.c:705, because its CPU loop gains little -- on the GPU the
     * off-diagonal block is >99% empty rows and visiting them costs a full pass over y) */
    PetscBool use_cprow = PETSC_FALSE;
    if (d->cprow && m > 0 && (double)(m - a->nonzerorows) > 0.6 * m) {
      use_cprow = PETSC_TRUE;
      nrows = a->nonzerorows;
      ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(nrows + 1), &ci);CHKERRQ(ierr);
      ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(nrows, 1), &ridx);CHKERRQ(ierr);
      PetscInt k = 0; ci[0] = 0;
      for (PetscInt r = 0; r < m; r++) if (a->i[r + 1] > a->i[r]) { ridx[k] = r; ci[++k] = a->i[r + 1]; }
      ip = ci;
    }
    CHKHIP(mi355x_malloc((void **)&d->d_i, sizeof(PetscInt) * (size_t)(nrows + 1)));
    /* +16 B: the SpMV kernels read aligned pairs and the pair holding the last element may extend past it (mi355x_kernels.h) */
    CHKHIP(mi355x_malloc((void **)&d->d_j, sizeof(PetscInt) * (size_t)PetscMax(a->nz, 1) + 16));
    CHKHIP(mi355x_malloc((void **)&d->d_a, sizeof(PetscScalar) * (size_t)PetscMax(a->nz, 1) + 16));
    CHKHIP(mi355x_memcpy_h2d(dc->h, d->d_i, ip, sizeof(PetscInt) * (size_t)(nrows + 1)));
    CHKHIP(mi355x_memcpy_h2d(dc->h, d->d_j, a->j, sizeof(PetscInt) * (size_t)a->nz));
    UP_TICK("row pointer and columns up");
    if (a->bs <= 1) {
      PetscInt ic = 1;
      CHKHIP(mi355x_spmv_plan_create(dc->h, nrows, ip, use_cprow ? ridx : NULL, &d->plan));
      UP_TICK("row-block plan");
      /* -mat_hipmi355x_index_compression <0|1> (default 1): one byte per nonzero instead of a 4-byte column index
       * when the matrix uses <= 256 distinct (col - row) offsets; plain CSR otherwise */
      ierr = hip_mat_option(A, HOPT_IC, &ic);CHKERRQ(ierr);
      if (ic && !use_cprow) {
        PetscInt rp = 1;
        CHKHIP(mi355x_spmv_plan_compress_indices(dc->h, d->plan, a->i, a->j));
        /* -mat_hipmi355x_row_patterns <0|1> (default 1): stencil matrices whose rows' offset lists come from a small dictionary
         * stream 4 bytes per ROW instead of 1 byte per nonzero + the row pointer (spmv_csr_rowblock_pat_kernel); same bits */
        ierr = hip_mat_option(A, HOPT_RP, &rp);CHKERRQ(ierr);
        CHKHIP(mi355x_spmv_plan_use_patterns(d->plan, rp ? 1 : 0, NULL));
        UP_TICK("offset / row-pattern dictionaries");
      }
      /* inodes: when the reference's Mat_CheckInode would switch this matrix to MatMult_SeqAIJ_Inode, the row sums take
       * that routine's two-at-a-time order (same bits), and -- unless the 1-byte index dictionary already applies --
       * the rows of a node share one stored column list (mi355x_spmv_plan_group_rows) */
      ierr = seqaij_check_inode(A);CHKERRQ(ierr);
      UP_TICK("inode check");
      if (a->inode_count) {
        int ntab = 0;
        CHKHIP(mi355x_spmv_plan_set_pairsum(d->plan, 1));
        CHKHIP(mi355x_spmv_plan_is_compressed(d->plan, &ntab));
        if (!ntab && !use_cprow) CHKHIP(mi355x_spmv_plan_group_rows(dc->h, d->plan, a->i, a->j, a->inode_count, a->inode_size));
      }
      /* -mat_hipmi355x_blocked <-1|0|1> (default -1 = decide): the blocked companion.  When every node Mat_CheckInode found has the same
       * size bs (2..5) and its shared column list is made of whole aligned groups of bs columns -- the 3-dof matrices of FEM codes
       * assembled into AIJ: every coupling a complete bs x bs block -- the matrix IS a BAIJ matrix, and MatMult_SeqBAIJ_bs's kernel
       * reads 8 bs^2 + 4 bytes per block where the grouped-row kernel reads 8 bs^2 + 4 bs: BCSR arrays are laid out beside the CSR ones
       * (block values column-major, baij.h:13-30, as a permutation of d_a kept on the device) and the products take the BCSR row-block
       * kernel: FEM stand-in 0.245 -> 0.197 ms, the same sums bit for bit (profiles/r04_fem_as_baij.log).  Decided here only for rows
       * long enough that the grouped-row kernel does not carry the reference's bits anyway (more than 16 nonzeros per row). */
      if (a->inode_count && !use_cprow) {
        PetscInt bl = -1;
        ierr = hip_mat_option(A, HOPT_BLOCKED, &bl);CHKERRQ(ierr);
        if (bl != 0 && (bl > 0 || (double)a->nz > 16.0 * (double)a->m)) { ierr = blocked_companion_build(A, dc);CHKERRQ(ierr); }
        UP_TICK("blocked companion");
      }
      /* -mat_hipmi355x_tiled <-1|0|1> (default -1 = decide): the column-tiled product (csrc/spmv_tiled.hip).  A matrix that got neither
       * an offset dictionary nor grouped rows gathers x once per nonzero; when a sample of its 32-row groups shows those gathers
       * landing on lines of x of their own (> 0.5 line per nonzero: rows that share no columns with their neighbours -- the
       * irregular matrices of BASELINE configs[3]) and it is large enough for x to leave the L2 (>= 2^17 columns), the product is
       * re-cut into row panels x column tiles with the tiles of x staged in LDS; kept only if at least half of the nonzeros fall into
       * pairs worth staging.  -mat_hipmi355x_tiled_stage_min <n> (default 1024): entries a (panel, tile) pair needs to be staged. */
      {
        PetscInt tl = -1, smin = 0; int ntab = 0, ng = 0; long ngj = 0;
        ierr = hip_mat_option(A, HOPT_TILED, &tl);CHKERRQ(ierr);
        ierr = hip_mat_option(A, HOPT_TILED_SMIN, &smin);CHKERRQ(ierr);
        CHKHIP(mi355x_spmv_plan_is_compressed(d->plan, &ntab));
        CHKHIP(mi355x_spmv_plan_group_info(d->plan, &ng, &ngj, NULL));
        if (tl != 0 && !use_cprow && !ntab && !ng && a->nz > 0) {
          PetscBool want = (PetscBool)(tl > 0);
          if (tl < 0 && a->n >= (1 << 17) && a->nz >= (1 << 22)) {
            double lpn = 0.0;
            CHKHIP(mi355x_spmv_tiled_probe(a->m, a->i, a->j, &lpn));
            want = (PetscBool)(lpn > 0.5);
          }
          if (want) {
            long staged = 0, rest = 0;
            int rcb = mi355x_spmv_tiled_build(a->m, a->n, a->i, a->j, (int)smin, &d->tiled);
            if (rcb && tl > 0) CHKHIP(rcb);                                  /* asked for: report; decided here: the row-block kernels serve (e.g. no host memory for the layout) */
            if (rcb) d->tiled = NULL;
            else {
              CHKHIP(mi355x_spmv_tiled_info(d->tiled, &staged, &rest, NULL, NULL, NULL));
              if (tl < 0 && 2 * staged < (long)a->nz) { mi355x_spmv_tiled_destroy(d->tiled); d->tiled = NULL; }
            }
          }
          UP_TICK("column-tiled layout");
        }
      }
    }
    else {   /* BAIJ: the plan partitions the VALUE stream, i.e. the block-row pointer scaled by bs*bs */
      PetscInt *sc, bs2 = a->bs * a->bs;
      if (a->bs == 4) {   /* -mat_hipmi355x_baij4 <mfma|fma>: the matrix cores (v_mfma_f64_4x4x4, 16-byte loads: 1.22-1.28 ms at 128^3 nodes, 27
                           * blocks per row; the default -- BASELINE configs[4]'s "MFMA 4x4 tile path") or the row-block FMA kernel with x staged
                           * in LDS (1.25-1.32 ms in the same processes, three boxes: profiles/r03_cfg5.log) */
        char kind[16] = "mfma"; PetscBool set;
        ierr = PetscOptionsGetString(NULL, "-mat_hipmi355x_baij4", kind, sizeof(kind), &set);CHKERRQ(ierr);
        if (strcmp(kind, "mfma") && strcmp(kind, "fma")) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_WRONG, "-mat_hipmi355x_baij4 <mfma|fma>, got %s", kind);
        d->baij4_mfma = (PetscBool)!strcmp(kind, "mfma");
      }
      if ((double)a->nz * bs2 > 2147483000.0) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_OUTOFRANGE, "BAIJ matrix too large for 32-bit value offsets");
      ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(nrows + 1), &sc);CHKERRQ(ierr);
      for (PetscInt r = 0; r <= nrows; r++) sc[r] = a->i[r] * bs2;
      CHKHIP(mi355x_spmv_plan_create(dc->h, nrows, sc, NULL, &d->plan));
      HipFree(sc);
    }
    CHKHIP(mi355x_handle_synchronize(dc->h));
    HipFree(ci); HipFree(ridx);
    d->pattern_nz = a->nz;
    if (!use_cprow) d->cprow = PETSC_FALSE;
  }
  size_t vals = (size_t)a->nz * (size_t)(a->bs > 1 ? a->bs * a->bs : 1);
  if (a->bs > 1 && !same_pattern) { mi355x_free(d->d_a); CHKHIP(mi355x_malloc((void **)&d->d_a, sizeof(PetscScalar) * PetscMax(vals, 1) + 16)); }
  UP_TICK("(grouped rows, bookkeeping)");
  CHKHIP(mi355x_memcpy_h2d(dc->h, d->d_a, a->a, sizeof(PetscScalar) * vals));
  UP_TICK("values up");
  if (d->b_plan) {
    CHKHIP(mi355x_pack(dc->h, (size_t)a->nz, d->b_perm, d->d_a, d->b_a));
    d->b_fresh = PETSC_TRUE; d->tb_fresh = PETSC_FALSE;
    UP_TICK("blocked companion's values");
  }
  if (d->tiled) {
    if (!same_pattern) { CHKHIP(mi355x_spmv_tiled_upload(dc->h, d->tiled, d->d_a)); CHKHIP(mi355x_spmv_tiled_drop_host(d->tiled)); }
    else CHKHIP(mi355x_spmv_tiled_refresh_values(dc->h, d->tiled, d->d_a));
    d->tiled_fresh = PETSC_TRUE;
    UP_TICK("column-tiled values");
  }
  if (a->bs <= 1 && d->plan) {
    /* -mat_hipmi355x_value_patterns <0|1> (default 1): constant-coefficient operators -- whole rows, offsets and values,
     * from a dictionary of <= 512 entries -- run a kernel that reads 2 bytes per row and no values (spmv_csr_valpat_kernel);
     * same bits.  The dictionary belongs to THESE values: derived again on every upload, dropped by every device-side change. */
    PetscInt vp = 1;
    ierr = hip_mat_option(A, HOPT_VP, &vp);CHKERRQ(ierr);
    CHKHIP(mi355x_spmv_plan_use_value_patterns(d->plan, vp ? 1 : 0, NULL));
    if (vp) CHKHIP(mi355x_spmv_plan_value_patterns(dc->h, d->plan, a->i, a->j, a->a, NULL));
  }
  CHKHIP(mi355x_handle_synchronize(dc->h));
  UP_TICK("value-pattern analysis");
#undef UP_TICK
  d->n_uploads++;
  d->uploaded_state = HipObjState(A);
  return 0;
}

PetscErrorCode MatSeqAIJHIPSetCompressedRow(Mat A, PetscBool flg) { SD(A)->cprow = flg; SD(A)->uploaded_state = -1; SD(A)->pattern_nz = -1; return 0; }

/* explicit transpose, contributions of each output row in increasing original-row order (the order
 * MatMultTransposeAdd_SeqAIJ's scatter loop adds them in, aij.c:1100-1112) */
static PetscErrorCode upload_transpose(Mat A) {
  PetscErrorCode ierr;
  HipAIJ *a = SA(A);
  Mat_SeqAIJHIP *d = SD(A);
  PetscDeviceCtx *dc;
  if (d->t_state == HipObjState(A) && d->t_a) return 0;
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  /* the device copy of A first: building it discards everything that belonged to an older pattern, a cached transpose included
   * (device_free), so it must not happen between the check below and the use of the cached arrays */
  ierr = MatSeqAIJHIPUpload(A);CHKERRQ(ierr);
  if (a->bs > 1) {
    /* BAIJ (MatMultTranspose_SeqBAIJ / MatMultTransposeAdd_SeqBAIJ, baij2.c:1579, 1740): the block transpose -- block rows and columns
     * exchanged by the same stable counting sort (an output row's contributions in increasing original block row, the order the
     * reference's scatter loop adds them in), every bs x bs block (column-major, baij.h:13-30) transposed -- built on the host whenever
     * the matrix moved, then A^T x is the row-block BCSR kernel over it. */
    PetscInt mbs = a->m, bs = a->bs, bs2 = bs * bs, nbs = a->n / bs, nzb = a->nz;      /* (a->m counts block rows, a->n scalar columns) */
    PetscInt *ti, *tj, *next, *sc; PetscScalar *ta;
    ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(nbs + 1), &ti);CHKERRQ(ierr);
    ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(nzb, 1), &tj);CHKERRQ(ierr);
    ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)PetscMax(nzb, 1) * (size_t)bs2, &ta);CHKERRQ(ierr);
    ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(nbs, 1), &next);CHKERRQ(ierr);
    memset(ti, 0, sizeof(PetscInt) * (size_t)(nbs + 1));
    for (PetscInt k = 0; k < nzb; k++) ti[a->j[k] + 1]++;
    for (PetscInt c = 0; c < nbs; c++) ti[c + 1] += ti[c];
    for (PetscInt c = 0; c < nbs; c++) next[c] = ti[c];
    for (PetscInt r = 0; r < mbs; r++)
      for (PetscInt k = a->i[r]; k < a->i[r + 1]; k++) {
        PetscInt p = next[a->j[k]]++;
        const PetscScalar *blk = a->a + (size_t)k * bs2; PetscScalar *tb = ta + (size_t)p * bs2;
        tj[p] = r;
        for (PetscInt c = 0; c < bs; c++) for (PetscInt q = 0; q < bs; q++) tb[c * bs + q] = blk[q * bs + c];
      }
    if (d->t_i) { mi355x_free(d->t_i); mi355x_free(d->t_j); mi355x_free(d->t_a); mi355x_spmv_plan_destroy(d->t_plan); d->t_plan = NULL; d->t_i = NULL; }
    CHKHIP(mi355x_malloc((void **)&d->t_i, sizeof(PetscInt) * (size_t)(nbs + 1)));
    CHKHIP(mi355x_malloc((void **)&d->t_j, sizeof(PetscInt) * (size_t)PetscMax(nzb, 1) + 16));
    CHKHIP(mi355x_malloc((void **)&d->t_a, sizeof(PetscScalar) * (size_t)PetscMax(nzb, 1) * (size_t)bs2 + 16));
    CHKHIP(mi355x_memcpy_h2d(dc->h, d->t_i, ti, sizeof(PetscInt) * (size_t)(nbs + 1)));
    CHKHIP(mi355x_memcpy_h2d(dc->h, d->t_j, tj, sizeof(PetscInt) * (size_t)nzb));
    CHKHIP(mi355x_memcpy_h2d(dc->h, d->t_a, ta, sizeof(PetscScalar) * (size_t)nzb * (size_t)bs2));
    ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(nbs + 1), &sc);CHKERRQ(ierr);
    for (PetscInt c = 0; c <= nbs; c++) sc[c] = ti[c] * bs2;            /* the plan partitions the VALUE stream (as the matrix's own) */
    CHKHIP(mi355x_spmv_plan_create(dc->h, nbs, sc, NULL, &d->t_plan));
    CHKHIP(mi355x_handle_synchronize(dc->h));
    HipFree(ti); HipFree(tj); HipFree(ta); HipFree(next); HipFree(sc);
    d->t_state = HipObjState(A);
    d->t_pattern_nz = nzb;
    d->t_builds++;
    return 0;
  }
  if (d->t_a && d->t_perm && d->t_pattern_nz == a->nz && d->pattern_nz == a->nz) {
    /* only the VALUES changed since the transpose was built (a time step, a Newton iteration, MatScale / MatDiagonalScale /
     * MatSetValuesBatch on the device copy): A^T's values are the matrix's values in another order, and that order -- the
     * permutation of the counting sort below -- is on the device.  One gather kernel over the current device values; nothing
     * is rebuilt on the host, nothing crosses PCIe beyond what MatSeqAIJHIPUpload needed for the matrix itself. */
    CHKHIP(mi355x_pack(dc->h, (size_t)a->nz, d->t_perm, d->d_a, d->t_a));
    if (d->t_tiled) CHKHIP(mi355x_spmv_tiled_refresh_values(dc->h, d->t_tiled, d->t_a));
    d->t_state = HipObjState(A);
    d->t_refreshes++;
    return 0;
  }
  PetscInt m = a->m, n = a->n, nz = a->nz;
  PetscInt *ti, *tj, *next, *perm; PetscScalar *ta;
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(n + 1), &ti);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(nz, 1), &tj);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)PetscMax(nz, 1), &ta);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(n, 1), &next);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(nz, 1), &perm);CHKERRQ(ierr);
  memset(ti, 0, sizeof(PetscInt) * (size_t)(n + 1));
  for (PetscInt k = 0; k < nz; k++) ti[a->j[k] + 1]++;
  for (PetscInt c = 0; c < n; c++) ti[c + 1] += ti[c];
  for (PetscInt c = 0; c < n; c++) next[c] = ti[c];
  for (PetscInt r = 0; r < m; r++)
    for (PetscInt k = a->i[r]; k < a->i[r + 1]; k++) { PetscInt p = next[a->j[k]]++; tj[p] = r; ta[p] = a->a[k]; perm[p] = k; }
  if (d->t_i) { mi355x_free(d->t_i); mi355x_free(d->t_j); mi355x_free(d->t_a); mi355x_spmv_plan_destroy(d->t_plan); d->t_plan = NULL; }
  if (d->t_tiled) { mi355x_spmv_tiled_destroy(d->t_tiled); d->t_tiled = NULL; }
  if (d->t_perm) { mi355x_free(d->t_perm); d->t_perm = NULL; }
  CHKHIP(mi355x_malloc((void **)&d->t_perm, sizeof(PetscInt) * (size_t)PetscMax(nz, 1)));
  CHKHIP(mi355x_memcpy_h2d(dc->h, d->t_perm, perm, sizeof(PetscInt) * (size_t)nz));
  CHKHIP(mi355x_malloc((void **)&d->t_i, sizeof(PetscInt) * (size_t)(n + 1)));
  CHKHIP(mi355x_malloc((void **)&d->t_j, sizeof(PetscInt) * (size_t)PetscMax(nz, 1) + 16));
  CHKHIP(mi355x_malloc((void **)&d->t_a, sizeof(PetscScalar) * (size_t)PetscMax(nz, 1) + 16));
  CHKHIP(mi355x_memcpy_h2d(dc->h, d->t_i, ti, sizeof(PetscInt) * (size_t)(n + 1)));
  CHKHIP(mi355x_memcpy_h2d(dc->h, d->t_j, tj, sizeof(PetscInt) * (size_t)nz));
  CHKHIP(mi355x_memcpy_h2d(dc->h, d->t_a, ta, sizeof(PetscScalar) * (size_t)nz));
  CHKHIP(mi355x_spmv_plan_create(dc->h, n, ti, NULL, &d->t_plan));
  { /* the transpose of a stencil matrix is a stencil matrix: same index compression / row patterns as the matrix itself */
    PetscInt ic = 1, rp = 1;
    ierr = hip_mat_option(A, HOPT_IC, &ic);CHKERRQ(ierr);
    ierr = hip_mat_option(A, HOPT_RP, &rp);CHKERRQ(ierr);
    if (ic) {
      CHKHIP(mi355x_spmv_plan_compress_indices(dc->h, d->t_plan, ti, tj));
      CHKHIP(mi355x_spmv_plan_use_patterns(d->t_plan, rp ? 1 : 0, NULL));
    }
  }
  CHKHIP(mi355x_handle_synchronize(dc->h));
  if (d->tiled) {
    /* the matrix took the column-tiled product (its gathers miss the caches): so do its transpose's, whose rows pick their columns
     * from the same wide windows.  Same rule: kept if at least half of the nonzeros fall into pairs worth staging. */
    PetscInt tl = -1, smin = 0; long staged = 0, rest = 0;
    ierr = hip_mat_option(A, HOPT_TILED, &tl);CHKERRQ(ierr);
    ierr = hip_mat_option(A, HOPT_TILED_SMIN, &smin);CHKERRQ(ierr);
    if (mi355x_spmv_tiled_build(n, m, ti, tj, (int)smin, &d->t_tiled)) d->t_tiled = NULL;   /* (the row-block kernel over the cached transpose serves) */
    else {
      CHKHIP(mi355x_spmv_tiled_info(d->t_tiled, &staged, &rest, NULL, NULL, NULL));
      if (tl < 0 && 2 * staged < (long)nz) { mi355x_spmv_tiled_destroy(d->t_tiled); d->t_tiled = NULL; }
      else { CHKHIP(mi355x_spmv_tiled_upload(dc->h, d->t_tiled, d->t_a)); CHKHIP(mi355x_spmv_tiled_drop_host(d->t_tiled)); }
    }
  }
  HipFree(ti); HipFree(tj); HipFree(ta); HipFree(next); HipFree(perm);
  d->t_state = HipObjState(A);
  d->t_pattern_nz = nz;
  d->t_builds++;
  return 0;
}

/* ---------------------------------------------------------------- ops */
#if !defined(PETSCHIPMI355X_WITH_PETSC)   /* the parent MATSEQAIJ's job inside a PETSc tree */
static PetscErrorCode MatSetUp_SeqAIJHIP(Mat A) { return seqaij_prealloc(A, PETSC_DEFAULT, NULL); }

static PetscErrorCode MatSetValues_SeqAIJHIP(Mat A, PetscInt m, const PetscInt im[], PetscInt n, const PetscInt in[], const PetscScalar v[], InsertMode is) {
  PetscErrorCode ierr;
  HipAIJ *a = SA(A);
  /* a host-side insertion: whatever the device-side updates stamped (MatSetValuesBatch, MatScale, ... look one state
   * bump ahead), the device copy is stale from here on */
  SD(A)->uploaded_state = -1;
  for (PetscInt k = 0; k < m; k++) {
    PetscInt row = im[k];
    if (row < 0) continue;
    if (row >= a->m) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_OUTOFRANGE, "Row too large: row %d max %d", row, a->m - 1);
    for (PetscInt l = 0; l < n; l++) {
      if (in[l] < 0) continue;
      if (in[l] >= a->n) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_OUTOFRANGE, "Column too large: col %d max %d", in[l], a->n - 1);
      ierr = seqaij_set(a, row, in[l], v[k * n + l], is, NULL);CHKERRQ(ierr);   /* row-oriented values, aij.c roworiented */
    }
  }
  return 0;
}

#endif
/* MatSetValuesBatch (matrix.c:1698; the reference's GPU version aijAssemble.cu:157 sorts and reduces a COO list on every
 * call): nb square blocks of bs x bs values, rows[] = their row = column indices, ADD_VALUES.  With an assembled matrix
 * whose pattern already holds every (row, col) pair -- the re-assembly of a time step or Newton iteration -- the values
 * are assembled ON THE DEVICE through a map built once per connectivity: contributions grouped by nonzero, kept in call
 * order, one lane per nonzero adds them one after the other (mi355x_csr_assemble), so the result carries the bits of the
 * reference's loop of MatSetValues.  The host copy is refreshed from the device afterwards.  Anything else (first
 * assembly, new nonzeros, BAIJ) takes that loop itself. */
static unsigned long long fnv1a(const void *p, size_t nbytes) {
  const unsigned char *c = (const unsigned char *)p; unsigned long long h = 1469598103934665603ULL;
  for (size_t k = 0; k < nbytes; k++) { h ^= c[k]; h *= 1099511628211ULL; }
  return h;
}
static PetscErrorCode batch_map_build(Mat A, PetscInt nb, PetscInt bs, const PetscInt rows[], PetscBool *ok) {
  PetscErrorCode ierr;
  HipAIJ *a = SA(A); Mat_SeqAIJHIP *d = SD(A);
  PetscDeviceCtx *dc;
  const size_t T = (size_t)nb * (size_t)bs * (size_t)bs;
  PetscInt *slot = NULL, *count = NULL, *order = NULL, *segptr = NULL, *segslot = NULL;
  *ok = PETSC_FALSE;
  if (T == 0 || T > 2147483000UL) return 0;
  ierr = PetscMalloc(sizeof(PetscInt) * T, &slot);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(a->nz + 1), &count);CHKERRQ(ierr);
  memset(count, 0, sizeof(PetscInt) * (size_t)(a->nz + 1));
  size_t used = 0;
  for (PetscInt b = 0; b < nb; b++) {
    const PetscInt *rb = rows + (size_t)b * bs;
    for (PetscInt i = 0; i < bs; i++) {
      const PetscInt row = rb[i];
      if (row >= a->m) { HipFree(slot); HipFree(count); SETERRQ(HipObjComm(A), PETSC_ERR_ARG_OUTOFRANGE, "Row too large: row %d max %d", row, a->m - 1); }
      for (PetscInt j = 0; j < bs; j++) {
        const size_t t = ((size_t)b * bs + i) * bs + j;
        const PetscInt col = rb[j];
        slot[t] = -1;
        if (row < 0 || col < 0) continue;                     /* MatSetValues ignores negative indices */
        if (col >= a->n) { HipFree(slot); HipFree(count); SETERRQ(HipObjComm(A), PETSC_ERR_ARG_OUTOFRANGE, "Column too large: col %d max %d", col, a->n - 1); }
        PetscInt k = a->i[row];
        for (; k < a->i[row + 1]; k++) if (a->j[k] == col) break;
        if (k == a->i[row + 1]) { HipFree(slot); HipFree(count); return 0; }   /* a new nonzero: not a pure value re-assembly */
        slot[t] = k; count[k + 1]++; used++;
      }
    }
  }
  /* counting sort by nonzero, stable in t: the order of the reference's loop */
  PetscInt nseg = 0;
  for (PetscInt k = 0; k < a->nz; k++) if (count[k + 1]) nseg++;
  ierr = PetscMalloc(sizeof(PetscInt) * PetscMax(used, 1), &order);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(nseg + 1), &segptr);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(nseg, 1), &segslot);CHKERRQ(ierr);
  { PetscInt s = 0, run = 0;
    for (PetscInt k = 0; k < a->nz; k++) { const PetscInt c = count[k + 1]; count[k + 1] = run; if (c) { segptr[s] = run; segslot[s] = k; s++; } run += c; }
    segptr[nseg] = run; }
  for (size_t t = 0; t < T; t++) if (slot[t] >= 0) order[count[slot[t] + 1]++] = (PetscInt)t;
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  if (d->bm_order) mi355x_free(d->bm_order);
  if (d->bm_segptr) mi355x_free(d->bm_segptr);
  if (d->bm_segslot) mi355x_free(d->bm_segslot);
  d->bm_order = d->bm_segptr = d->bm_segslot = NULL;
  CHKHIP(mi355x_malloc((void **)&d->bm_order, sizeof(PetscInt) * PetscMax(used, 1)));
  CHKHIP(mi355x_malloc((void **)&d->bm_segptr, sizeof(PetscInt) * (size_t)(nseg + 1)));
  CHKHIP(mi355x_malloc((void **)&d->bm_segslot, sizeof(PetscInt) * (size_t)PetscMax(nseg, 1)));
  CHKHIP(mi355x_memcpy_h2d(dc->h, d->bm_order, order, sizeof(PetscInt) * used));
  CHKHIP(mi355x_memcpy_h2d(dc->h, d->bm_segptr, segptr, sizeof(PetscInt) * (size_t)(nseg + 1)));
  CHKHIP(mi355x_memcpy_h2d(dc->h, d->bm_segslot, segslot, sizeof(PetscInt) * (size_t)nseg));
  CHKHIP(mi355x_handle_synchronize(dc->h));
  HipFree(slot); HipFree(count); HipFree(order); HipFree(segptr); HipFree(segslot);
  d->bm_nb = nb; d->bm_bs = bs; d->bm_nseg = nseg; d->bm_T = T;
  d->bm_hash = fnv1a(rows, sizeof(PetscInt) * (size_t)nb * (size_t)bs);
  *ok = PETSC_TRUE;
  return 0;
}
static PetscErrorCode MatSetValuesBatch_SeqAIJHIP(Mat A, PetscInt nb, PetscInt bs, PetscInt rows[], const PetscScalar v[]) {
  PetscErrorCode ierr;
  HipAIJ *a = SA(A); Mat_SeqAIJHIP *d = SD(A);
  PetscBool ok = PETSC_FALSE;
  if (a->bs <= 1 && a->compact && A->assembled && nb > 0 && bs > 0 && !d->cprow) {
    ierr = MatSeqAIJHIPUpload(A);CHKERRQ(ierr);               /* device values current before they are added to */
    ok = (PetscBool)(d->bm_order && d->bm_nb == nb && d->bm_bs == bs && d->pattern_nz == a->nz &&
                     d->bm_hash == fnv1a(rows, sizeof(PetscInt) * (size_t)nb * (size_t)bs));
    if (!ok) { ierr = batch_map_build(A, nb, bs, rows, &ok);CHKERRQ(ierr); }
  }
  if (!ok) {                                                   /* the reference's default (matrix.c:1715-1718) */
    for (PetscInt b = 0; b < nb; b++) { ierr = MatSetValues(A, bs, &rows[(size_t)b * bs], bs, &rows[(size_t)b * bs], &v[(size_t)b * bs * bs], ADD_VALUES);CHKERRQ(ierr); }
    return 0;
  }
  PetscDeviceCtx *dc;
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  if (d->bm_vcap < d->bm_T) {
    if (d->bm_v) mi355x_free(d->bm_v);
    d->bm_v = NULL;
    CHKHIP(mi355x_malloc((void **)&d->bm_v, sizeof(PetscScalar) * d->bm_T));
    d->bm_vcap = d->bm_T;
  }
  CHKHIP(mi355x_memcpy_h2d(dc->h, d->bm_v, v, sizeof(PetscScalar) * d->bm_T));
  { PetscErrorCode e__ = VecHIPProductMatrixChanges(A);CHKERRQ(e__); }
  CHKHIP(mi355x_csr_assemble(dc->h, d->bm_nseg, d->bm_segptr, d->bm_segslot, d->bm_order, d->bm_v, d->d_a));
  CHKHIP(mi355x_memcpy_d2h(dc->h, a->a, d->d_a, sizeof(PetscScalar) * (size_t)a->nz));   /* host mirror follows */
  CHKHIP(mi355x_handle_synchronize(dc->h));                    /* v and a->a are pageable host memory */
  /* MatSetValuesBatch's wrapper leaves the state alone and the MatAssemblyEnd that has to follow bumps it once: the
   * device copy is stamped with that state, so the assembly does not trigger an upload */
  d->uploaded_state = HipObjState(A) + 1;
  d->t_state = -1; d->tiled_fresh = PETSC_FALSE; d->b_fresh = PETSC_FALSE; d->tb_fresh = PETSC_FALSE;
  CHKHIP(mi355x_spmv_plan_drop_value_patterns(d->plan));
  ierr = PetscLogFlops((PetscLogDouble)d->bm_T);CHKERRQ(ierr);
  return 0;
}

#if !defined(PETSCHIPMI355X_WITH_PETSC)   /* the parent MATSEQAIJ's job inside a PETSc tree */
static PetscErrorCode MatAssemblyEnd_SeqAIJHIP(Mat A, MatAssemblyType mode) {
  if (mode == MAT_FLUSH_ASSEMBLY) return 0;
  /* (the reference re-installs ops->mult here because the inode check may have replaced it,
   *  aijcusp.cu:462-466; this container has no inode variant) */
  return seqaij_compact(SA(A));
}

#endif
static PetscErrorCode MatMult_SeqAIJHIP_device(Mat A, Vec xx, Vec yy);
PetscErrorCode MatMultDiagonalScale_HIPMI355X(Mat A, Vec dd, Vec xx, Vec yy, PetscBool *ok);
/* the column-tiled form keeps its own copy of the values in its own order: after a device-side change of d_a (MatScale,
 * MatDiagonalScale, MatZeroEntries, MatSetValuesBatch on the device copy) one gather brings it up to date, when it is next used */
static PetscErrorCode tiled_values_current(Mat A, PetscDeviceCtx *dc) {
  Mat_SeqAIJHIP *d = SD(A);
  if (d->tiled && !d->tiled_fresh) { CHKHIP(mi355x_spmv_tiled_refresh_values(dc->h, d->tiled, d->d_a)); d->tiled_fresh = PETSC_TRUE; }
  return 0;
}
static PetscErrorCode MatMult_SeqAIJHIP(Mat A, Vec xx, Vec yy) {   /* MatMult_SeqAIJCUSP aijcusp.cu:349 */
  PetscErrorCode ierr = MatSeqAIJHIPUpload(A);CHKERRQ(ierr);       /* from here on the device copy holds the values of THIS call */
  /* a point-wise AIJ matrix on one rank: the product is noted with the Vec type (host/vechip.c, "a noted product"): if PCApply_Jacobi
   * follows, the two become one kernel and the work vector in between is never written; anything else runs the product as it is.
   * Whatever changes the device copy afterwards (a new upload, MatScale / MatDiagonalScale / assembly on the device, the matrix going
   * away) first lets a noted product of this matrix run (VecHIPProductMatrixChanges). */
  if (SA(A)->bs <= 1 && !SD(A)->cprow) {
    PetscBool noted = PETSC_FALSE;
    ierr = VecHIPNoteProduct(A, xx, yy, MatMult_SeqAIJHIP_device, MatMultDiagonalScale_HIPMI355X, &noted);CHKERRQ(ierr);
    if (noted) return 0;
  }
  return MatMult_SeqAIJHIP_device(A, xx, yy);
}
static PetscErrorCode MatMult_SeqAIJHIP_device(Mat A, Vec xx, Vec yy) {   /* y = A x with the device copy as it is */
  PetscErrorCode ierr;
  HipAIJ *a = SA(A); Mat_SeqAIJHIP *d = SD(A);
  const PetscScalar *x; PetscScalar *y; PetscDeviceCtx *dc;
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  ierr = VecHIPGetRead(xx, &x);CHKERRQ(ierr);
  ierr = VecHIPGetWrite(yy, &y);CHKERRQ(ierr);
  ierr = MatTimingBegin(A, dc->h);CHKERRQ(ierr);
  if (a->bs == 4 && d->baij4_mfma && !(((size_t)y) & 15)) CHKHIP(mi355x_spmv_bsr4_mfma(dc->h, a->m, 0, d->d_i, d->d_j, d->d_a, x, y));   /* matrix cores: MatMult_SeqBAIJ_4 (16-byte stores of y; a vector borrowing storage at an odd offset takes the FMA kernel) */
  else if (a->bs > 1) CHKHIP(mi355x_spmv_bsr_planned(dc->h, d->plan, a->bs, d->d_i, d->d_j, d->d_a, x, y));
  else if (d->b_plan) { ierr = blocked_values_current(A, dc);CHKERRQ(ierr); CHKHIP(mi355x_spmv_bsr_planned(dc->h, d->b_plan, (int)d->b_bs, d->b_i, d->b_j, d->b_a, x, y)); }   /* the blocked companion */
  else {
    int rc = 801;
    if (d->tiled) { ierr = tiled_values_current(A, dc);CHKERRQ(ierr); rc = mi355x_spmv_tiled(dc->h, d->tiled, x, NULL, y); if (rc && rc != 801) CHKHIP(rc); }
    if (rc) {                                                            /* no tiled form, or an x it cannot take (storage borrowed at an odd offset) */
      if (d->cprow) CHKHIP(mi355x_vec_set(dc->h, (size_t)a->m, 0.0, y));   /* rows without entries */
      CHKHIP(mi355x_spmv_csr(dc->h, d->plan, d->d_i, d->d_j, d->d_a, x, y));
    }
  }
  ierr = MatTimingEnd(A, dc->h);CHKERRQ(ierr);
  ierr = VecHIPRestoreWrite(yy);CHKERRQ(ierr);
  if (a->bs > 1) { ierr = PetscLogFlops(2.0 * a->bs * a->bs * a->nz - (double)a->bs * a->nonzerorows);CHKERRQ(ierr); }
  else { ierr = PetscLogFlops(2.0 * a->nz - a->nonzerorows);CHKERRQ(ierr); }   /* aij.c:1281 */
  return 0;
}

/* how many times the values of a sequential matrix of this type have crossed to the device (tests: value updates with an
 * unchanged pattern must not add to it) */
PetscErrorCode MatHIPMI355XGetUploadCount(Mat A, PetscInt *n) {
  *n = 0;
  if (!A || A->ops->mult != MatMult_SeqAIJHIP) SETERRQ(A ? HipObjComm(A) : 0, PETSC_ERR_ARG_WRONG, "sequential HIPMI355X matrix expected");
  *n = SD(A)->n_uploads;
  return 0;
}
/* number of distinct (col - row) offsets of the index-compressed SpMV plan, 0 when the matrix streams plain 4-byte
 * column indices (bench.py labels its roofline kernel with it; an MPIAIJ matrix answers for its diagonal block) */
PetscErrorCode MatHIPMI355XGetIndexCompression(Mat A, PetscInt *noffsets) {
  PetscErrorCode ierr; int ntab = 0;
  *noffsets = 0;
  if (!A) return 0;
  if (A->ops->mult != MatMult_SeqAIJHIP) {
    Mat Ad = NULL;
    if (!strcmp(HipObjTypeName(A), MATMPIAIJHIPMI355X)) { ierr = MatMPIAIJGetSeqAIJ(A, &Ad, NULL, NULL);CHKERRQ(ierr); }
    if (!Ad || Ad->ops->mult != MatMult_SeqAIJHIP) return 0;
    A = Ad;
  }
  ierr = MatSeqAIJHIPUpload(A);CHKERRQ(ierr);
  if (SD(A)->plan && SA(A)->bs <= 1) CHKHIP(mi355x_spmv_plan_is_compressed(SD(A)->plan, &ntab));
  *noffsets = ntab;
  return 0;
}

/* the block transpose of the blocked companion (MatMultTranspose / MatMultTransposeAdd of such a matrix): built from the host CSR arrays
 * at the first transpose product after the pattern changed -- block rows and columns exchanged by a stable counting sort (an output
 * row's contributions in increasing original block row, the order aij.c:1100-1112 adds them in), every block transposed --, its
 * values a permutation gather of d_a like the companion's own */
static PetscErrorCode blocked_transpose_current(Mat A, PetscDeviceCtx *dc) {
  PetscErrorCode ierr;
  HipAIJ *a = SA(A); Mat_SeqAIJHIP *d = SD(A);
  if (!d->b_plan) return 0;
  if (!d->tb_plan) {
    const PetscInt bs = d->b_bs, bs2 = bs * bs, nn = a->m / bs, nbs = a->n / bs, nblk = d->b_nblocks;
    PetscInt *ti, *tj, *next, *perm, *sc;
    ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(nbs + 1), &ti);CHKERRQ(ierr);
    ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(nblk, 1), &tj);CHKERRQ(ierr);
    ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(nbs, 1), &next);CHKERRQ(ierr);
    ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(nblk, 1) * (size_t)bs2, &perm);CHKERRQ(ierr);
    ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(nbs + 1), &sc);CHKERRQ(ierr);
    memset(ti, 0, sizeof(PetscInt) * (size_t)(nbs + 1));
    for (PetscInt i = 0; i < nn; i++) { const PetscInt r = i * bs; for (PetscInt g = 0; g < a->i[r + 1] - a->i[r]; g += bs) ti[a->j[a->i[r] + g] / bs + 1]++; }
    for (PetscInt c = 0; c < nbs; c++) ti[c + 1] += ti[c];
    for (PetscInt c = 0; c < nbs; c++) next[c] = ti[c];
    for (PetscInt i = 0; i < nn; i++) {
      const PetscInt r = i * bs, len = a->i[r + 1] - a->i[r];
      for (PetscInt g = 0; g < len; g += bs) {
        const PetscInt p = next[a->j[a->i[r] + g] / bs]++;
        tj[p] = i;
        /* transposed block, column-major: entry (row qq, column cc) of it is entry (row cc, column qq) of the block */
        for (PetscInt cc = 0; cc < bs; cc++) for (PetscInt qq = 0; qq < bs; qq++) perm[(size_t)p * bs2 + cc * bs + qq] = a->i[r + cc] + g + qq;
      }
    }
    for (PetscInt c = 0; c <= nbs; c++) sc[c] = ti[c] * bs2;
    CHKHIP(mi355x_malloc((void **)&d->tb_i, sizeof(PetscInt) * (size_t)(nbs + 1)));
    CHKHIP(mi355x_malloc((void **)&d->tb_j, sizeof(PetscInt) * (size_t)PetscMax(nblk, 1) + 16));
    CHKHIP(mi355x_malloc((void **)&d->tb_perm, sizeof(PetscInt) * (size_t)PetscMax(nblk, 1) * (size_t)bs2));
    CHKHIP(mi355x_malloc((void **)&d->tb_a, sizeof(PetscScalar) * (size_t)PetscMax(nblk, 1) * (size_t)bs2 + 16));
    CHKHIP(mi355x_memcpy_h2d(dc->h, d->tb_i, ti, sizeof(PetscInt) * (size_t)(nbs + 1)));
    CHKHIP(mi355x_memcpy_h2d(dc->h, d->tb_j, tj, sizeof(PetscInt) * (size_t)nblk));
    CHKHIP(mi355x_memcpy_h2d(dc->h, d->tb_perm, perm, sizeof(PetscInt) * (size_t)nblk * (size_t)bs2));
    CHKHIP(mi355x_spmv_plan_create(dc->h, nbs, sc, NULL, &d->tb_plan));
    CHKHIP(mi355x_handle_synchronize(dc->h));
    HipFree(ti); HipFree(tj); HipFree(next); HipFree(perm); HipFree(sc);
    d->tb_fresh = PETSC_FALSE;
  }
  if (!d->tb_fresh) { CHKHIP(mi355x_pack(dc->h, (size_t)a->nz, d->tb_perm, d->d_a, d->tb_a)); d->tb_fresh = PETSC_TRUE; }
  return 0;
}
/* the blocked companion of a sequential matrix, if the analysis chose it: block size and number of blocks (0, 0: none) */
PetscErrorCode MatHIPMI355XGetBlockedInfo(Mat A, PetscInt *bs, PetscInt *nblocks) {
  *bs = 0; *nblocks = 0;
  if (!A || A->ops->mult != MatMult_SeqAIJHIP) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_WRONG, "not a sequential HIPMI355X AIJ matrix");
  if (SD(A)->b_plan) { *bs = SD(A)->b_bs; *nblocks = SD(A)->b_nblocks; }
  return 0;
}

/* the column-tiled form of a sequential matrix's product, if the analysis chose it: nonzeros that gather from LDS tiles / that stay in
 * the CSR remainder (both 0: the row-block kernels run) */
PetscErrorCode MatHIPMI355XGetTiledInfo(Mat A, PetscInt *staged, PetscInt *remainder) {
  PetscErrorCode ierr; long s_ = 0, r_ = 0;
  *staged = 0; *remainder = 0;
  if (!A || A->ops->mult != MatMult_SeqAIJHIP) return 0;
  ierr = MatSeqAIJHIPUpload(A);CHKERRQ(ierr);
  if (SD(A)->tiled) CHKHIP(mi355x_spmv_tiled_info(SD(A)->tiled, &s_, &r_, NULL, NULL, NULL));
  *staged = (PetscInt)s_; *remainder = (PetscInt)r_;
  return 0;
}

/* size of the row-pattern dictionary the SpMV plan runs with (0: none, or switched off) */
PetscErrorCode MatHIPMI355XGetRowPatterns(Mat A, PetscInt *npat) {
  PetscErrorCode ierr; int np_ = 0;
  *npat = 0;
  if (!A) return 0;
  if (A->ops->mult != MatMult_SeqAIJHIP) {
    Mat Ad = NULL;
    if (!strcmp(HipObjTypeName(A), MATMPIAIJHIPMI355X)) { ierr = MatMPIAIJGetSeqAIJ(A, &Ad, NULL, NULL);CHKERRQ(ierr); }
    if (!Ad || Ad->ops->mult != MatMult_SeqAIJHIP) return 0;
    A = Ad;
  }
  ierr = MatSeqAIJHIPUpload(A);CHKERRQ(ierr);
  if (SD(A)->plan && SA(A)->bs <= 1) {
    PetscInt rp = 1;
    CHKHIP(mi355x_spmv_plan_use_patterns(SD(A)->plan, -1, &np_));
    ierr = hip_mat_option(A, HOPT_RP, &rp);CHKERRQ(ierr);
    if (!rp) np_ = 0;
  }
  *npat = np_;
  return 0;
}

/* size of the value-pattern dictionary the SpMV runs with right now (0: none -- varying coefficients, switched off, or
 * dropped by a device-side change of the values since the last upload) */
PetscErrorCode MatHIPMI355XGetValuePatterns(Mat A, PetscInt *nvpat) {
  PetscErrorCode ierr; int nv = 0;
  *nvpat = 0;
  if (!A) return 0;
  if (A->ops->mult != MatMult_SeqAIJHIP) {
    Mat Ad = NULL;
    if (!strcmp(HipObjTypeName(A), MATMPIAIJHIPMI355X)) { ierr = MatMPIAIJGetSeqAIJ(A, &Ad, NULL, NULL);CHKERRQ(ierr); }
    if (!Ad || Ad->ops->mult != MatMult_SeqAIJHIP) return 0;
    A = Ad;
  }
  ierr = MatSeqAIJHIPUpload(A);CHKERRQ(ierr);
  if (SD(A)->plan && SA(A)->bs <= 1) CHKHIP(mi355x_spmv_plan_use_value_patterns(SD(A)->plan, -1, &nv));
  *nvpat = nv;
  return 0;
}

/* A/B switch for one matrix (the option -mat_hipmi355x_value_patterns is read at every upload; this overrides it until
 * the next upload): off -> the SpMV streams the value array again; on -> the dictionary is derived from the host copy now */
PetscErrorCode MatHIPMI355XSetValuePatterns(Mat A, PetscBool on) {
  PetscErrorCode ierr; PetscDeviceCtx *dc;
  if (!A) return 0;
  if (A->ops->mult != MatMult_SeqAIJHIP) {
    Mat Ad = NULL;
    if (!strcmp(HipObjTypeName(A), MATMPIAIJHIPMI355X)) { ierr = MatMPIAIJGetSeqAIJ(A, &Ad, NULL, NULL);CHKERRQ(ierr); }
    if (!Ad || Ad->ops->mult != MatMult_SeqAIJHIP) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_WRONG, "HIPMI355X AIJ matrix expected");
    A = Ad;
  }
  ierr = MatSeqAIJHIPUpload(A);CHKERRQ(ierr);
  if (!SD(A)->plan || SA(A)->bs > 1) return 0;
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  CHKHIP(mi355x_spmv_plan_use_value_patterns(SD(A)->plan, on ? 1 : 0, NULL));
  if (on && device_values_current(A)) CHKHIP(mi355x_spmv_plan_value_patterns(dc->h, SD(A)->plan, SA(A)->i, SA(A)->j, SA(A)->a, NULL));
  return 0;
}

/* row grouping of the SpMV plan: number of nodes Mat_CheckInode found (0: plain routines), groups the device plan stores
 * one column list for (0: the plan streams per-nonzero indices), and the shared indices stored */
/* host builds of the explicit transpose / device-side value refreshes of it so far (tests: a time-stepping caller of
 * MatMultTranspose must not go back to the host for A^T) */
PetscErrorCode MatHIPMI355XGetTransposeCounts(Mat A, PetscInt *builds, PetscInt *refreshes) {
  if (!A || A->ops->mult != MatMult_SeqAIJHIP) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_WRONG, "not a sequential HIPMI355X AIJ matrix");
  if (builds) *builds = SD(A)->t_builds;
  if (refreshes) *refreshes = SD(A)->t_refreshes;
  return 0;
}

/* the nodes Mat_CheckInode finds (consecutive rows with identical column lists): count and sizes, NULL / 0 when the matrix keeps the
 * plain routines.  For the factorisations (host/ilu.c): the reference solves the factor of such a matrix node by node */
PetscErrorCode MatSeqAIJHIPGetInodes(Mat A, PetscInt *count, const PetscInt **sizes) {
  PetscErrorCode ierr;
  *count = 0; *sizes = NULL;
  if (!A || A->ops->mult != MatMult_SeqAIJHIP) return 0;
  ierr = MatSeqAIJHIPUpload(A);CHKERRQ(ierr);          /* runs Mat_CheckInode's restatement on the current pattern */
  *count = SA(A)->inode_count; *sizes = SA(A)->inode_size;
  return 0;
}

PetscErrorCode MatHIPMI355XGetInodeInfo(Mat A, PetscInt *nodes, PetscInt *groups, PetscInt *shared_indices) {
  PetscErrorCode ierr; int ng = 0; long ngj = 0;
  if (nodes) *nodes = 0;
  if (groups) *groups = 0;
  if (shared_indices) *shared_indices = 0;
  if (!A) return 0;
  if (A->ops->mult != MatMult_SeqAIJHIP) {
    Mat Ad = NULL;
    if (!strcmp(HipObjTypeName(A), MATMPIAIJHIPMI355X)) { ierr = MatMPIAIJGetSeqAIJ(A, &Ad, NULL, NULL);CHKERRQ(ierr); }
    if (!Ad || Ad->ops->mult != MatMult_SeqAIJHIP) return 0;
    A = Ad;
  }
  ierr = MatSeqAIJHIPUpload(A);CHKERRQ(ierr);
  if (SD(A)->plan && SA(A)->bs <= 1) CHKHIP(mi355x_spmv_plan_group_info(SD(A)->plan, &ng, &ngj, NULL));
  if (nodes) *nodes = SA(A)->inode_count;
  if (groups) *groups = ng;
  if (shared_indices) *shared_indices = (PetscInt)ngj;
  return 0;
}

/* w = A p with dpi = p'w as a by-product of the same pass (KSPSolve_CG cg.c:190-191); dpi is left in the device
 * scratch slot the fused CG update reads (all-reduced there when the vectors' communicator has an RCCL communicator).
 * *ok = PETSC_FALSE and nothing done unless A is a square, sequential AIJ matrix of this type whose product kernel is one of the
 * row-block kernels that also leave the per-block sums (value patterns, row patterns, 8-bit column offsets). */
PetscErrorCode MatMultTDotBegin_HIPMI355X(Mat A, Vec xx, Vec yy, PetscBool *ok) {
  PetscErrorCode ierr;
  *ok = PETSC_FALSE;
  if (!A || A->ops->mult != MatMult_SeqAIJHIP) return 0;
  HipAIJ *a = SA(A); Mat_SeqAIJHIP *d = SD(A);
  const PetscScalar *x; PetscScalar *y; PetscDeviceCtx *dc; int ntab = 0;
  if (a->bs > 1 || d->cprow || a->m != a->n || xx == yy) return 0;
  if (xx->map->n != a->n || yy->map->n != a->m || (HipCommSize(HipObjComm(xx)) > 1 && !HipCommDevice(HipObjComm(xx)))) return 0;
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  ierr = MatSeqAIJHIPUpload(A);CHKERRQ(ierr);
  if (!d->plan) return 0;
  CHKHIP(mi355x_spmv_plan_dot_available(d->plan, d->d_a, &ntab));   /* a plan whose kernel also leaves the per-block sums: patterns or 8-bit offsets */
  if (!ntab) return 0;
  ierr = VecHIPGetRead(xx, &x);CHKERRQ(ierr);
  ierr = VecHIPGetWrite(yy, &y);CHKERRQ(ierr);
  ierr = MatTimingBegin(A, dc->h);CHKERRQ(ierr);
  CHKHIP(mi355x_spmv_csr_dot(dc->h, d->plan, d->d_i, d->d_j, d->d_a, x, y));
  ierr = MatTimingEnd(A, dc->h);CHKERRQ(ierr);
  double *slot = mi355x_handle_device_scratch(dc->h) + PETSC_HIP_DPI_SLOT;
  CHKHIP(mi355x_spmv_dot_finish(dc->h, d->plan, slot));
  if (HipCommDevice(HipObjComm(xx))) CHKHIP(mi355x_comm_allreduce_sum(HipCommDevice(HipObjComm(xx)), dc->h, slot, 1));
  ierr = VecHIPRestoreWrite(yy);CHKERRQ(ierr);
  HipStateIncrease(yy);
  ierr = PetscLogFlops(2.0 * a->nz - a->nonzerorows + 2.0 * a->m - 1);CHKERRQ(ierr);
  *ok = PETSC_TRUE;
  return 0;
}

/* "MatMultDiagonalScale_C": y = d .* (A x), i.e. MatMult followed by PCApply_Jacobi's VecPointwiseMult(y, w, d) (jacobi.c:266)
 * with the scaling in the product's epilogue: the intermediate vector is neither written nor read.  Same bits as the two calls. */
PetscErrorCode MatMultDiagonalScale_HIPMI355X(Mat A, Vec dd, Vec xx, Vec yy, PetscBool *ok) {
  PetscErrorCode ierr;
  *ok = PETSC_FALSE;
  if (!A || A->ops->mult != MatMult_SeqAIJHIP) return 0;
  HipAIJ *a = SA(A); Mat_SeqAIJHIP *d = SD(A);
  const PetscScalar *x, *dg; PetscScalar *y; PetscDeviceCtx *dc;
  if (a->bs > 1 || xx == yy || dd == yy || xx->map->n != a->n || yy->map->n != a->m || dd->map->n != a->m) return 0;
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  ierr = MatSeqAIJHIPUpload(A);CHKERRQ(ierr);
  if (!d->plan || d->cprow || d->tiled || d->b_plan) return 0;         /* (the column-tiled and the blocked product have no scaling epilogue: the two calls stay two) */
  ierr = VecHIPGetRead(xx, &x);CHKERRQ(ierr);
  ierr = VecHIPGetRead(dd, &dg);CHKERRQ(ierr);
  ierr = VecHIPGetWrite(yy, &y);CHKERRQ(ierr);
  ierr = MatTimingBegin(A, dc->h);CHKERRQ(ierr);
  CHKHIP(mi355x_spmv_csr_scaled(dc->h, d->plan, d->d_i, d->d_j, d->d_a, x, dg, y));
  ierr = MatTimingEnd(A, dc->h);CHKERRQ(ierr);
  ierr = VecHIPRestoreWrite(yy);CHKERRQ(ierr);
  HipStateIncrease(yy);
  ierr = PetscLogFlops(2.0 * a->nz - a->nonzerorows + a->m);CHKERRQ(ierr);
  *ok = PETSC_TRUE;
  return 0;
}

static PetscErrorCode MatMultAdd_SeqAIJHIP(Mat A, Vec xx, Vec yy, Vec zz) {   /* MatMultAdd_SeqAIJCUSP aijcusp.cu:405 */
  PetscErrorCode ierr;
  HipAIJ *a = SA(A); Mat_SeqAIJHIP *d = SD(A);
  const PetscScalar *x, *y; PetscScalar *z; PetscDeviceCtx *dc;
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  ierr = MatSeqAIJHIPUpload(A);CHKERRQ(ierr);
  if (a->bs > 1) {   /* MatMultAdd_SeqBAIJ_3/_4/_N (baij2.c:1168-1480): the row-block kernel with y as the sums' start */
    ierr = VecHIPGetRead(xx, &x);CHKERRQ(ierr);
    if (zz == yy) { ierr = VecHIPGetReadWrite(zz, &z);CHKERRQ(ierr); y = z; }
    else { ierr = VecHIPGetRead(yy, &y);CHKERRQ(ierr); ierr = VecHIPGetWrite(zz, &z);CHKERRQ(ierr); }
    ierr = MatTimingBegin(A, dc->h);CHKERRQ(ierr);
    CHKHIP(mi355x_spmv_bsr_planned_add(dc->h, d->plan, a->bs, d->d_i, d->d_j, d->d_a, x, y, z));
    ierr = MatTimingEnd(A, dc->h);CHKERRQ(ierr);
    ierr = VecHIPRestoreWrite(zz);CHKERRQ(ierr);
    return PetscLogFlops(2.0 * a->bs * a->bs * a->nz);                   /* baij2.c: 2 bs^2 nz */
  }
  ierr = VecHIPGetRead(xx, &x);CHKERRQ(ierr);
  if (zz == yy) { ierr = VecHIPGetReadWrite(zz, &z);CHKERRQ(ierr); y = z; }
  else {
    ierr = VecHIPGetRead(yy, &y);CHKERRQ(ierr);
    ierr = VecHIPGetWrite(zz, &z);CHKERRQ(ierr);
    if (d->cprow) { CHKHIP(mi355x_vec_copy(dc->h, (size_t)a->m, y, z)); y = z; }   /* aij.c:1314-1316 */
  }
  ierr = MatTimingBegin(A, dc->h);CHKERRQ(ierr);
  if (d->b_plan && !d->cprow) { ierr = blocked_values_current(A, dc);CHKERRQ(ierr); CHKHIP(mi355x_spmv_bsr_planned_add(dc->h, d->b_plan, (int)d->b_bs, d->b_i, d->b_j, d->b_a, x, y, z)); }
  else { int rc = 801;
    if (d->tiled) { ierr = tiled_values_current(A, dc);CHKERRQ(ierr); rc = mi355x_spmv_tiled(dc->h, d->tiled, x, y, z); if (rc && rc != 801) CHKHIP(rc); }
    if (rc) CHKHIP(mi355x_spmv_csr_add(dc->h, d->plan, d->d_i, d->d_j, d->d_a, x, y, z)); }
  ierr = MatTimingEnd(A, dc->h);CHKERRQ(ierr);
  ierr = VecHIPRestoreWrite(zz);CHKERRQ(ierr);
  ierr = PetscLogFlops(2.0 * a->nz);CHKERRQ(ierr);
  return 0;
}

static PetscErrorCode MatMultTransposeAdd_SeqAIJHIP(Mat A, Vec xx, Vec zz, Vec yy) {   /* aij.c:1078: yy = zz + A^T xx */
  PetscErrorCode ierr;
  HipAIJ *a = SA(A); Mat_SeqAIJHIP *d = SD(A);
  const PetscScalar *x, *z; PetscScalar *y; PetscDeviceCtx *dc;
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  if (a->bs <= 1) { ierr = MatSeqAIJHIPUpload(A);CHKERRQ(ierr); }
  if (a->bs <= 1 && d->b_plan) {                                      /* the blocked companion's block transpose: no scalar transpose is built */
    ierr = blocked_transpose_current(A, dc);CHKERRQ(ierr);
    ierr = VecHIPGetRead(xx, &x);CHKERRQ(ierr);
    if (zz == yy) { ierr = VecHIPGetReadWrite(yy, &y);CHKERRQ(ierr); z = y; }
    else { ierr = VecHIPGetRead(zz, &z);CHKERRQ(ierr); ierr = VecHIPGetWrite(yy, &y);CHKERRQ(ierr); }
    CHKHIP(mi355x_spmv_bsr_planned_add(dc->h, d->tb_plan, (int)d->b_bs, d->tb_i, d->tb_j, d->tb_a, x, z, y));
    ierr = VecHIPRestoreWrite(yy);CHKERRQ(ierr);
    return PetscLogFlops(2.0 * a->nz);
  }
  ierr = upload_transpose(A);CHKERRQ(ierr);
  ierr = VecHIPGetRead(xx, &x);CHKERRQ(ierr);
  if (zz == yy) { ierr = VecHIPGetReadWrite(yy, &y);CHKERRQ(ierr); z = y; }
  else { ierr = VecHIPGetRead(zz, &z);CHKERRQ(ierr); ierr = VecHIPGetWrite(yy, &y);CHKERRQ(ierr); }
  if (a->bs > 1) CHKHIP(mi355x_spmv_bsr_planned_add(dc->h, d->t_plan, a->bs, d->t_i, d->t_j, d->t_a, x, z, y));   /* MatMultTransposeAdd_SeqBAIJ, baij2.c:1740 */
  else {
    int rc = 801;
    if (d->t_tiled) { rc = mi355x_spmv_tiled(dc->h, d->t_tiled, x, z, y); if (rc && rc != 801) CHKHIP(rc); }
    if (rc) CHKHIP(mi355x_spmv_csr_add(dc->h, d->t_plan, d->t_i, d->t_j, d->t_a, x, z, y));
  }
  ierr = VecHIPRestoreWrite(yy);CHKERRQ(ierr);
  ierr = PetscLogFlops(2.0 * a->nz);CHKERRQ(ierr);
  return 0;
}
static PetscErrorCode MatMultTranspose_SeqAIJHIP(Mat A, Vec xx, Vec yy) {   /* aij.c:1124: VecSet(yy,0); Add */
  PetscErrorCode ierr;
  HipAIJ *a = SA(A); Mat_SeqAIJHIP *d = SD(A);
  const PetscScalar *x; PetscScalar *y; PetscDeviceCtx *dc;
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  if (a->bs <= 1) { ierr = MatSeqAIJHIPUpload(A);CHKERRQ(ierr); }
  if (a->bs <= 1 && d->b_plan) {
    ierr = blocked_transpose_current(A, dc);CHKERRQ(ierr);
    ierr = VecHIPGetRead(xx, &x);CHKERRQ(ierr);
    ierr = VecHIPGetWrite(yy, &y);CHKERRQ(ierr);
    CHKHIP(mi355x_spmv_bsr_planned(dc->h, d->tb_plan, (int)d->b_bs, d->tb_i, d->tb_j, d->tb_a, x, y));
    ierr = VecHIPRestoreWrite(yy);CHKERRQ(ierr);
    return PetscLogFlops(2.0 * a->nz);
  }
  ierr = upload_transpose(A);CHKERRQ(ierr);
  ierr = VecHIPGetRead(xx, &x);CHKERRQ(ierr);
  ierr = VecHIPGetWrite(yy, &y);CHKERRQ(ierr);
  /* 0 + p1 + p2 ... == p1 + p2 ... bit for bit, so the plain product kernel serves */
  if (a->bs > 1) CHKHIP(mi355x_spmv_bsr_planned(dc->h, d->t_plan, a->bs, d->t_i, d->t_j, d->t_a, x, y));          /* MatMultTranspose_SeqBAIJ, baij2.c:1579 */
  else {
    int rc = 801;
    if (d->t_tiled) { rc = mi355x_spmv_tiled(dc->h, d->t_tiled, x, NULL, y); if (rc && rc != 801) CHKHIP(rc); }
    if (rc) CHKHIP(mi355x_spmv_csr(dc->h, d->t_plan, d->t_i, d->t_j, d->t_a, x, y));
  }
  ierr = VecHIPRestoreWrite(yy);CHKERRQ(ierr);
  ierr = PetscLogFlops(2.0 * a->nz);CHKERRQ(ierr);
  return 0;
}

static PetscErrorCode MatGetDiagonal_SeqAIJHIP(Mat A, Vec v) {   /* aij.c:1040 */
  PetscErrorCode ierr;
  HipAIJ *a = SA(A); Mat_SeqAIJHIP *d = SD(A);
  PetscScalar *dv; PetscDeviceCtx *dc;
  if (v->map->n != A->rmap->n) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_SIZ, "Nonconforming matrix and vector");
  if (a->bs > 1 || d->cprow) {   /* host route for the rarely used shapes */
    PetscScalar *h;
    ierr = VecGetArray(v, &h);CHKERRQ(ierr);
    if (a->bs > 1) {
      PetscInt bs = a->bs, mbs = a->m;                 /* BAIJ: a->m counts block rows */
      for (PetscInt r = 0; r < mbs * bs; r++) h[r] = 0.0;
      for (PetscInt br = 0; br < mbs; br++) for (PetscInt k = a->i[br]; k < a->i[br + 1]; k++) if (a->j[k] == br)
        for (PetscInt q = 0; q < bs; q++) h[br * bs + q] = a->a[(size_t)k * bs * bs + q * bs + q];
    } else {
      for (PetscInt r = 0; r < a->m; r++) { h[r] = 0.0; for (PetscInt k = a->i[r]; k < a->i[r + 1]; k++) if (a->j[k] == r) { h[r] = a->a[k]; break; } }
    }
    return VecRestoreArray(v, &h);
  }
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  ierr = MatSeqAIJHIPUpload(A);CHKERRQ(ierr);
  ierr = VecHIPGetWrite(v, &dv);CHKERRQ(ierr);
  CHKHIP(mi355x_csr_get_diagonal(dc->h, a->m, d->d_i, d->d_j, d->d_a, dv));
  return VecHIPRestoreWrite(v);
}

/* Value updates with an unchanged pattern: the host copy and the device copy are updated side by side, so
 * the next MatMult finds the device values current and nothing crosses PCIe (the reference's GPU back end re-sent the
 * whole matrix after every such call, aijcusp.cu:138-152).  The wrappers in mat.c bump the object state AFTER the op:
 * the device copy is stamped with that future state.  The cached transpose is dropped. */
static PetscBool device_values_current(Mat A) {
  Mat_SeqAIJHIP *d = SD(A);
  return (PetscBool)(d->d_a && d->uploaded_state == HipObjState(A) && SA(A)->bs <= 1);
}
static PetscErrorCode MatScale_SeqAIJHIP(Mat A, PetscScalar alpha) {   /* MatScale_SeqAIJ: dscal on a->a */
  PetscErrorCode ierr;
  HipAIJ *a = SA(A); Mat_SeqAIJHIP *d = SD(A);
  size_t vals = (size_t)a->nz * (size_t)(a->bs > 1 ? a->bs * a->bs : 1);
  const PetscBool on_device = (PetscBool)(device_values_current(A) && alpha != 0.0);   /* alpha == 0: signs of zero, take the upload */
  for (size_t k = 0; k < vals; k++) a->a[k] = alpha * a->a[k];
  if (on_device) {
    PetscDeviceCtx *dc;
    ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
    { PetscErrorCode e__ = VecHIPProductMatrixChanges(A);CHKERRQ(e__); }
    CHKHIP(mi355x_vec_scale(dc->h, vals, alpha, d->d_a));
    d->uploaded_state = HipObjState(A) + 1;
    d->t_state = -1; d->tiled_fresh = PETSC_FALSE; d->b_fresh = PETSC_FALSE; d->tb_fresh = PETSC_FALSE;
    CHKHIP(mi355x_spmv_plan_drop_value_patterns(d->plan));
  }
  return PetscLogFlops((PetscLogDouble)vals);
}
static PetscErrorCode MatZeroEntries_SeqAIJHIP(Mat A) {
  PetscErrorCode ierr;
  HipAIJ *a = SA(A); Mat_SeqAIJHIP *d = SD(A);
  size_t vals = (size_t)(a->compact ? a->nz : a->maxnz) * (size_t)(a->bs > 1 ? a->bs * a->bs : 1);
  const PetscBool on_device = (PetscBool)(device_values_current(A) && a->compact);
  memset(a->a, 0, sizeof(PetscScalar) * vals);
  if (on_device) {
    PetscDeviceCtx *dc;
    ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
    { PetscErrorCode e__ = VecHIPProductMatrixChanges(A);CHKERRQ(e__); }
    CHKHIP(mi355x_memset(dc->h, d->d_a, 0, sizeof(PetscScalar) * vals));
    d->uploaded_state = HipObjState(A) + 1;
    d->t_state = -1; d->tiled_fresh = PETSC_FALSE; d->b_fresh = PETSC_FALSE; d->tb_fresh = PETSC_FALSE;
    CHKHIP(mi355x_spmv_plan_drop_value_patterns(d->plan));
  }
  return 0;
}
/* MatDiagonalScale_SeqAIJ, aij.c:2055-2092: left scaling pass, then right scaling pass ((a*l)*r) */
static PetscErrorCode MatDiagonalScale_SeqAIJHIP(Mat A, Vec ll, Vec rr) {
  PetscErrorCode ierr;
  HipAIJ *a = SA(A); Mat_SeqAIJHIP *d = SD(A);
  const PetscScalar *l = NULL, *r = NULL;
  if (!a->compact) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_WRONGSTATE, "matrix must be assembled");
  if (a->bs > 1) {   /* MatDiagonalScale_SeqBAIJ, baij2.c:2026-2084: blocks column-major, v[r + c bs] *= l[row bs + r], then *= r[col bs + c]; on the
                      * host copy (a set-up operation of this type); the wrapper's state bump sends the values to the device at the next use */
    const PetscInt bs = a->bs, bs2 = bs * bs, mbs = a->m;
    if (ll && ll->map->n != mbs * bs) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_SIZ, "Left scaling vector wrong length");
    if (rr && rr->map->n != a->n) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_SIZ, "Right scaling vector wrong length");
    if (ll) {
      ierr = VecGetArrayRead(ll, &l);CHKERRQ(ierr);
      for (PetscInt i = 0; i < mbs; i++) {
        const PetscScalar *li = l + (size_t)i * bs;
        PetscScalar *v = a->a + (size_t)bs2 * a->i[i];
        for (PetscInt j = a->i[i]; j < a->i[i + 1]; j++) for (PetscInt k = 0; k < bs2; k++) (*v++) *= li[k % bs];
      }
      ierr = VecRestoreArrayRead(ll, &l);CHKERRQ(ierr);
      ierr = PetscLogFlops((PetscLogDouble)a->nz);CHKERRQ(ierr);       /* (the reference logs the block count, baij2.c:2060) */
    }
    if (rr) {
      ierr = VecGetArrayRead(rr, &r);CHKERRQ(ierr);
      for (PetscInt i = 0; i < mbs; i++) {
        PetscScalar *v = a->a + (size_t)bs2 * a->i[i];
        for (PetscInt j = a->i[i]; j < a->i[i + 1]; j++) {
          const PetscScalar *ri = r + (size_t)bs * a->j[j];
          for (PetscInt k = 0; k < bs; k++) { const PetscScalar x = ri[k]; for (PetscInt t = 0; t < bs; t++) v[t] *= x; v += bs; }
        }
      }
      ierr = VecRestoreArrayRead(rr, &r);CHKERRQ(ierr);
      ierr = PetscLogFlops((PetscLogDouble)a->nz);CHKERRQ(ierr);
    }
    d->uploaded_state = -1;
    return 0;
  }
  if (ll && ll->map->n != a->m) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_SIZ, "Left scaling vector wrong length");
  if (rr && rr->map->n != a->n) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_SIZ, "Right scaling vector wrong length");
  const PetscBool on_device = (PetscBool)(device_values_current(A) && !d->cprow);
  if (on_device) {   /* device pointers first: fetching the host arrays below must not be what makes them stale */
    const PetscScalar *dl = NULL, *dr = NULL; PetscDeviceCtx *dc;
    ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
    if (ll) { ierr = VecHIPGetRead(ll, &dl);CHKERRQ(ierr); }
    if (rr) { ierr = VecHIPGetRead(rr, &dr);CHKERRQ(ierr); }
    { PetscErrorCode e__ = VecHIPProductMatrixChanges(A);CHKERRQ(e__); }
    CHKHIP(mi355x_csr_diagonal_scale(dc->h, a->m, d->d_i, d->d_j, d->d_a, dl, dr));
    d->uploaded_state = HipObjState(A) + 1;
    d->t_state = -1; d->tiled_fresh = PETSC_FALSE; d->b_fresh = PETSC_FALSE; d->tb_fresh = PETSC_FALSE;
    CHKHIP(mi355x_spmv_plan_drop_value_patterns(d->plan));
  }
  if (ll) {
    ierr = VecGetArrayRead(ll, &l);CHKERRQ(ierr);
    for (PetscInt i = 0; i < a->m; i++) for (PetscInt k = a->i[i]; k < a->i[i + 1]; k++) a->a[k] *= l[i];
    ierr = VecRestoreArrayRead(ll, &l);CHKERRQ(ierr);
    ierr = PetscLogFlops((PetscLogDouble)a->nz);CHKERRQ(ierr);
  }
  if (rr) {
    ierr = VecGetArrayRead(rr, &r);CHKERRQ(ierr);
    for (PetscInt k = 0; k < a->nz; k++) a->a[k] *= r[a->j[k]];
    ierr = VecRestoreArrayRead(rr, &r);CHKERRQ(ierr);
    ierr = PetscLogFlops((PetscLogDouble)a->nz);CHKERRQ(ierr);
  }
  return 0;
}

static PetscErrorCode MatGetVecs_HIP(Mat A, Vec *right, Vec *left) {   /* MatGetVecs_SeqAIJCUSP aijcusp.cu:324-345 */
  PetscErrorCode ierr;
  if (right) {
    ierr = VecCreate(HipObjComm(A), right);CHKERRQ(ierr);
    ierr = VecSetSizes(*right, A->cmap->n, A->cmap->N);CHKERRQ(ierr);
    ierr = VecSetType(*right, VECHIPMI355X);CHKERRQ(ierr);
  }
  if (left) {
    ierr = VecCreate(HipObjComm(A), left);CHKERRQ(ierr);
    ierr = VecSetSizes(*left, A->rmap->n, A->rmap->N);CHKERRQ(ierr);
    ierr = VecSetType(*left, VECHIPMI355X);CHKERRQ(ierr);
  }
  return 0;
}
PetscErrorCode MatGetVecs_HIPMI355X(Mat A, Vec *right, Vec *left) { return MatGetVecs_HIP(A, right, left); }

#if !defined(PETSCHIPMI355X_WITH_PETSC)   /* the parent MATSEQAIJ's job inside a PETSc tree */
static PetscErrorCode MatDestroy_SeqAIJHIP(Mat A) {   /* free the mirror and zero spptr first, aijcusp.cu:584-586 */
  HipAIJ *a = SA(A);
  if (SD(A)) {
    Mat_SeqAIJHIP *d = SD(A);
    (void)HipTriFactorsDestroy(&d->tri);    /* a factored matrix: its triangular factors */
    device_free(A);
    if (d->time_ev) { for (PetscInt k = 0; k < 2 * d->time_cap; k++) mi355x_event_destroy(d->time_ev[k]); HipFree(d->time_ev); }
    HipFree(A->spptr); A->spptr = NULL;
  }
  if (a) { HipFree(a->i); HipFree(a->j); HipFree(a->a); HipFree(a->ilen); HipFree(a->imax); HipFree(a->inode_size); HipFree(a); A->data = NULL; }
  return 0;
}

#endif
#if defined(PETSCHIPMI355X_WITH_PETSC)
#include "aijhipmi355x_ctor.h"    /* integration/petsc-3.3/: the constructor as a subclass of the reference's MATSEQAIJ */
#include "baijhipmi355x_ctor.h"   /* ... and MATSEQBAIJHIPMI355X as a subclass of MATSEQBAIJ */
#else
static PetscErrorCode MatSeqAIJSetPreallocation_SeqAIJHIP(Mat A, PetscInt nz, const PetscInt nnz[]) { return seqaij_prealloc(A, nz, nnz); }
static PetscErrorCode MatSeqAIJSetPreallocationCSR_SeqAIJHIP(Mat B, const PetscInt *i, const PetscInt *j, const PetscScalar *a);
static PetscErrorCode MatSeqBAIJSetPreallocationCSR_SeqBAIJHIP(Mat B, PetscInt bs, const PetscInt *i, const PetscInt *j, const PetscScalar *a);
static PetscErrorCode MatDuplicate_SeqAIJHIP(Mat A, MatDuplicateOption op, Mat *M);

/* MatCreate_SeqAIJCUSP (aijcusp.cu:657-681) fills, after the parent constructor, the slots mult, multadd, multtranspose,
 * multtransposeadd, assemblyend, destroy, getvecs, setvaluesbatch; the container and its assembly are this file's too
 * (host/aijhip.c) on the harness, the parent MATSEQAIJ's inside a PETSc tree. */
static PetscErrorCode create_common(Mat B, const char *tname, PetscInt bs) {
  PetscErrorCode ierr;
  HipAIJ *a; Mat_SeqAIJHIP *d;
  if (HipCommSize(HipObjComm(B)) > 1) SETERRQ(HipObjComm(B), PETSC_ERR_ARG_WRONG, "Comm must be of size 1");
  ierr = PetscMalloc(sizeof(*a), &a);CHKERRQ(ierr);
  memset(a, 0, sizeof(*a));
  ierr = PetscMalloc(sizeof(*d), &d);CHKERRQ(ierr);
  memset(d, 0, sizeof(*d));
  d->uploaded_state = -1; d->t_state = -1; d->pattern_nz = -1;
  a->m = B->rmap->n; a->n = B->cmap->n; a->bs = bs;
  B->data = a; B->spptr = d;
  ierr = PetscObjectChangeTypeName((PetscObject)B, tname);CHKERRQ(ierr);
  B->ops->setvalues = MatSetValues_SeqAIJHIP;
  B->ops->mult = MatMult_SeqAIJHIP;
  B->ops->multadd = MatMultAdd_SeqAIJHIP;
  B->ops->multtranspose = MatMultTranspose_SeqAIJHIP;
  B->ops->multtransposeadd = MatMultTransposeAdd_SeqAIJHIP;
  B->ops->getdiagonal = MatGetDiagonal_SeqAIJHIP;
  B->ops->assemblyend = MatAssemblyEnd_SeqAIJHIP;
  B->ops->zeroentries = MatZeroEntries_SeqAIJHIP;
  B->ops->setup = MatSetUp_SeqAIJHIP;
  B->ops->scale = MatScale_SeqAIJHIP;
  B->ops->diagonalscale = MatDiagonalScale_SeqAIJHIP;
  B->ops->setvaluesbatch = MatSetValuesBatch_SeqAIJHIP;
  B->ops->duplicate = MatDuplicate_SeqAIJHIP;
  B->ops->setfromoptions = MatSetFromOptions_SeqAIJHIP;
  B->ops->destroy = MatDestroy_SeqAIJHIP;
  B->ops->getvecs = MatGetVecs_HIP;
  ierr = PetscObjectComposeFunction((PetscObject)B, "MatSeqAIJGetArrays_C", "MatSeqAIJGetArrays", (PetscVoidFunction)MatSeqAIJGetArrays);CHKERRQ(ierr);
  ierr = PetscObjectComposeFunction((PetscObject)B, "MatMultTDotBegin_C", "MatMultTDotBegin_HIPMI355X", (PetscVoidFunction)MatMultTDotBegin_HIPMI355X);CHKERRQ(ierr);
  ierr = PetscObjectComposeFunction((PetscObject)B, "MatMultDiagonalScale_C", "MatMultDiagonalScale_HIPMI355X", (PetscVoidFunction)MatMultDiagonalScale_HIPMI355X);CHKERRQ(ierr);
  if (bs == 1) {
    /* ILU(0) / ICC(0) of this type: the factored matrix carries the device triangular solves behind ops->solve (host/ilu.c), as
     * MatCreate_SeqAIJCUSPARSE overloads "MatGetFactor_petsc_C" (aijcusparse.cu:837-840) */
    ierr = PetscObjectComposeFunction((PetscObject)B, "MatGetFactor_petsc_C", "MatGetFactor_seqaijhipmi355x_petsc", (PetscVoidFunction)MatGetFactor_seqaijhipmi355x_petsc);CHKERRQ(ierr);
    ierr = PetscObjectComposeFunction((PetscObject)B, "MatGetFactorAvailable_petsc_C", "MatGetFactorAvailable_seqaijhipmi355x_petsc", (PetscVoidFunction)MatGetFactorAvailable_seqaijhipmi355x_petsc);CHKERRQ(ierr);
    ierr = PetscObjectComposeFunction((PetscObject)B, "MatSeqAIJSetPreallocation_C", "MatSeqAIJSetPreallocation_SeqAIJHIP", (PetscVoidFunction)MatSeqAIJSetPreallocation_SeqAIJHIP);CHKERRQ(ierr);
    ierr = PetscObjectComposeFunction((PetscObject)B, "MatSeqAIJSetPreallocationCSR_C", "MatSeqAIJSetPreallocationCSR_SeqAIJHIP", (PetscVoidFunction)MatSeqAIJSetPreallocationCSR_SeqAIJHIP);CHKERRQ(ierr);
  } else {
    ierr = PetscObjectComposeFunction((PetscObject)B, "MatSeqBAIJSetPreallocationCSR_C", "MatSeqBAIJSetPreallocationCSR_SeqBAIJHIP", (PetscVoidFunction)MatSeqBAIJSetPreallocationCSR_SeqBAIJHIP);CHKERRQ(ierr);
  }
  return 0;
}
/* MatDuplicate_SeqAIJ (aij.c:3964) / MatDuplicate_SeqBAIJ (baij.c:2874): same type, same layouts, the pattern copied, the values copied
 * (MAT_COPY_VALUES) or zero; MAT_SHARE_NONZERO_PATTERN copies the pattern as well (the harness container has no shared arrays).  The
 * type's options go along. */
static PetscErrorCode adopt_csr(Mat B, PetscInt nrows, PetscInt bs, const PetscInt *i, const PetscInt *j, const PetscScalar *a);
static PetscErrorCode MatDuplicate_SeqAIJHIP(Mat A, MatDuplicateOption op, Mat *M) {
  PetscErrorCode ierr;
  HipAIJ *a = SA(A);
  Mat B;
  if (!a->compact) SETERRQ(HipObjComm(A), PETSC_ERR_ARG_WRONGSTATE, "Not for unassembled matrix");
  ierr = MatCreate(HipObjComm(A), &B);CHKERRQ(ierr);
  ierr = MatSetSizes(B, A->rmap->n, A->cmap->n, A->rmap->n, A->cmap->n);CHKERRQ(ierr);
  ierr = MatSetType(B, HipObjTypeName(A));CHKERRQ(ierr);
  ierr = adopt_csr(B, a->m, a->bs > 1 ? a->bs : 1, a->i, a->j, a->a);CHKERRQ(ierr);
  if (op != MAT_COPY_VALUES) memset(SA(B)->a, 0, sizeof(PetscScalar) * (size_t)a->nz * (size_t)(a->bs > 1 ? a->bs * a->bs : 1));
  memcpy(SD(B)->opt, SD(A)->opt, sizeof(SD(A)->opt)); memcpy(SD(B)->opt_set, SD(A)->opt_set, sizeof(SD(A)->opt_set));
  SD(B)->cprow = SD(A)->cprow;
  *M = B;
  return 0;
}
PetscErrorCode MatCreate_SeqAIJHIPMI355X(Mat B) { return create_common(B, MATSEQAIJHIPMI355X, 1); }
PetscErrorCode MatCreate_SeqBAIJHIPMI355X(Mat B) { return create_common(B, MATSEQBAIJHIPMI355X, 0); }

/* MatCreateSeqAIJWithArrays (aij.c): the arrays are copied (the reference aliases them); i, j, a may be the matrix's own (MatDuplicate) */
static PetscErrorCode adopt_csr(Mat B, PetscInt nrows, PetscInt bs, const PetscInt *i, const PetscInt *j, const PetscScalar *a) {
  PetscErrorCode ierr;
  HipAIJ *s = SA(B);
  PetscInt nz = i[nrows];
  size_t vals = (size_t)nz * (size_t)(bs > 1 ? bs * bs : 1);
  if (i[0] != 0) SETERRQ(HipObjComm(B), PETSC_ERR_ARG_OUTOFRANGE, "i (row indices) must start with 0");
  for (PetscInt r = 0; r < nrows; r++) if (i[r + 1] < i[r]) SETERRQ(HipObjComm(B), PETSC_ERR_ARG_OUTOFRANGE, "Negative row length in i (row indices) row = %d length = %d", r, i[r + 1] - i[r]);
  {   /* column indices in range and ascending within each row (the kernels gather x[col] unchecked) */
    const PetscInt ncols = bs > 1 ? s->n / bs : s->n;
    for (PetscInt r = 0; r < nrows; r++) for (PetscInt k = i[r]; k < i[r + 1]; k++) {
      if (j[k] < 0 || j[k] >= ncols) SETERRQ(HipObjComm(B), PETSC_ERR_ARG_OUTOFRANGE, "Column index %d out of range [0,%d) in row %d", j[k], ncols, r);
      if (k > i[r] && j[k] <= j[k - 1]) SETERRQ(HipObjComm(B), PETSC_ERR_ARG_WRONG, "Column indices of row %d are not sorted and unique", r);
    }
  }
  device_free(B);   /* the matrix may have been used before (MatLoad into a used Mat): nothing of the old pattern survives */
  HipFree(s->i); HipFree(s->j); HipFree(s->a); HipFree(s->ilen); HipFree(s->imax);
  s->i = s->j = s->ilen = s->imax = NULL; s->a = NULL;
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)(nrows + 1), &s->i);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(nz, 1), &s->j);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscScalar) * PetscMax(vals, 1), &s->a);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(nrows, 1), &s->ilen);CHKERRQ(ierr);
  ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(nrows, 1), &s->imax);CHKERRQ(ierr);
  memcpy(s->i, i, sizeof(PetscInt) * (size_t)(nrows + 1));
  memcpy(s->j, j, sizeof(PetscInt) * (size_t)nz);
  memcpy(s->a, a, sizeof(PetscScalar) * vals);
  s->nonzerorows = 0;
  for (PetscInt r = 0; r < nrows; r++) {
    s->ilen[r] = s->imax[r] = i[r + 1] - i[r];
    s->nonzerorows += (s->ilen[r] > 0);
  }
  s->nz = s->maxnz = nz; s->compact = PETSC_TRUE;
  if (bs > 1) { s->m = nrows; s->bs = bs; }
  B->preallocated = PETSC_TRUE; B->assembled = PETSC_TRUE; B->was_assembled = PETSC_TRUE; HipStateIncrease(B);
  return 0;
}
/* "MatSeqAIJSetPreallocationCSR_C" (aij.c:3795) and its BAIJ analogue (baij.c): m, n are point sizes; i, j index blocks */
static PetscErrorCode MatSeqAIJSetPreallocationCSR_SeqAIJHIP(Mat B, const PetscInt *i, const PetscInt *j, const PetscScalar *a) { return adopt_csr(B, B->rmap->n, 1, i, j, a); }
static PetscErrorCode MatSeqBAIJSetPreallocationCSR_SeqBAIJHIP(Mat B, PetscInt bs, const PetscInt *i, const PetscInt *j, const PetscScalar *a) {
  PetscErrorCode ierr;
  if (bs < 1 || B->rmap->n % bs || B->cmap->n % bs) SETERRQ(HipObjComm(B), PETSC_ERR_ARG_SIZ, "block size %d must divide the local sizes %d, %d", bs, B->rmap->n, B->cmap->n);
  SA(B)->bs = bs;                       /* the column check of adopt_csr counts block columns */
  ierr = adopt_csr(B, B->rmap->n / bs, bs, i, j, a);CHKERRQ(ierr);
  if (bs == 1) SA(B)->bs = 1;
  return 0;
}
#endif

PetscErrorCode MatSeqAIJGetArrays(Mat A, PetscInt *m, const PetscInt **i, const PetscInt **j, const PetscScalar **a) {
  if (!A || !A->data || (strcmp(HipObjTypeName(A), MATSEQAIJHIPMI355X) && strcmp(HipObjTypeName(A), MATSEQBAIJHIPMI355X))) SETERRQ(0, PETSC_ERR_ARG_WRONG, "not a SeqAIJHIPMI355X matrix");
  HipAIJ *s = SA(A);
  if (m) *m = s->m;
  if (i) *i = s->i;
  if (j) *j = s->j;
  if (a) *a = s->a;
  return 0;
}

/* ---- per-launch device timing used by bench.py (hipEvent pairs on the compute stream) ---- */
PetscErrorCode MatHIPMI355XSetTiming(Mat A, PetscBool on) {
  if (!A || A->ops->mult != MatMult_SeqAIJHIP) SETERRQ(A ? HipObjComm(A) : 0, PETSC_ERR_ARG_WRONG, "sequential HIPMI355X matrix expected");
  Mat_SeqAIJHIP *d = SD(A);
  d->timing = on; d->time_n = 0;
  if (on && !d->time_ev) {
    d->time_cap = 4096;
    PetscErrorCode ierr = PetscMalloc(sizeof(mi355x_event_t) * 2 * (size_t)d->time_cap, &d->time_ev);CHKERRQ(ierr);
    for (PetscInt k = 0; k < 2 * d->time_cap; k++) CHKHIP(mi355x_event_create(&d->time_ev[k]));
  }
  return 0;
}
PetscErrorCode MatTimingBegin(Mat A, mi355x_handle_t h) {
  Mat_SeqAIJHIP *d = SD(A);
  if (d->timing && d->time_n < d->time_cap) CHKHIP(mi355x_event_record(d->time_ev[2 * d->time_n], h));
  return 0;
}
PetscErrorCode MatTimingEnd(Mat A, mi355x_handle_t h) {
  Mat_SeqAIJHIP *d = SD(A);
  if (d->timing && d->time_n < d->time_cap) { CHKHIP(mi355x_event_record(d->time_ev[2 * d->time_n + 1], h)); d->time_n++; }
  return 0;
}
PetscErrorCode MatHIPMI355XGetTiming(Mat A, PetscInt *nlaunches, PetscLogDouble *total_ms) {
  if (!A || A->ops->mult != MatMult_SeqAIJHIP) SETERRQ(A ? HipObjComm(A) : 0, PETSC_ERR_ARG_WRONG, "sequential HIPMI355X matrix expected");
  Mat_SeqAIJHIP *d = SD(A);
  double tot = 0.0;
  for (PetscInt k = 0; k < d->time_n; k++) {
    float ms = 0.f;
    CHKHIP(mi355x_event_synchronize(d->time_ev[2 * k + 1]));
    CHKHIP(mi355x_event_elapsed_ms(d->time_ev[2 * k], d->time_ev[2 * k + 1], &ms));
    tot += ms;
  }
  if (nlaunches) *nlaunches = d->time_n;
  if (total_ms) *total_ms = tot;
  return 0;
}
