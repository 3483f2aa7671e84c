/*
 * mi355x_kernels.h -- C ABI of the MI355X (gfx950 / CDNA4) kernel library
 * (libmi355x_kernels.so) behind the PETSc HIPMI355X Vec/Mat implementations.
 *
 * Plain C: raw device pointers, sizes, an opaque handle that owns one HIP
 * stream.  No PETSc types, no torch types.  Every function returns 0 on
 * success or a non-zero hipError_t value (mi355x_error_string() decodes it);
 * nothing throws, nothing exits.  PetscScalar = double, PetscInt = int32
 * (reference: include/petscsys.h:188, include/petscmath.h:198).
 *
 * Each entry cites the reference CPU routine it replaces (path relative to
 * the PETSc tree, erdc/petsc-dev 3.3.0-dev).  Floating-point contract: all
 * kernels are compiled with -ffp-contract=off, i.e. a*b+c is a rounded
 * multiply followed by a rounded add, exactly what the reference's C loops
 * do when built without FMA.  Element-wise kernels and the row-sequential
 * SpMV paths therefore reproduce the reference bit for bit; reductions use
 * a fixed tree (run-to-run reproducible) and agree to a stated tolerance.
 */
#ifndef MI355X_KERNELS_H
#define MI355X_KERNELS_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi355x_handle_s    *mi355x_handle_t;    /* one HIP stream + reduction workspace */
typedef struct mi355x_event_s     *mi355x_event_t;     /* hipEvent_t */
typedef struct mi355x_spmv_plan_s *mi355x_spmv_plan_t; /* row-block partition of one CSR matrix */

/* ---- runtime --------------------------------------------------------- */
const char *mi355x_error_string(int err);
int  mi355x_device_count(int *count);
int  mi355x_set_device(int dev);
int  mi355x_get_device(int *dev);
int  mi355x_device_name(char *buf, size_t len);
int  mi355x_device_synchronize(void);
int  mi355x_mem_info(size_t *free_bytes, size_t *total_bytes);   /* hipMemGetInfo of the current device */
/* host threads THIS process may use for the set-up passes (pattern analyses, factorisation, plan construction), at most cap:
 * the CPUs of its affinity mask, cut to the cgroup's CPU quota, shared among the ranks torchrun started on this node
 * (LOCAL_WORLD_SIZE); MI355X_HOST_THREADS=<n> in the environment overrides.  Never less than 1. */
int  mi355x_host_threads(int cap);

int  mi355x_handle_create(mi355x_handle_t *h);
int  mi355x_handle_destroy(mi355x_handle_t h);
int  mi355x_handle_synchronize(mi355x_handle_t h);
/* wait for the last reduction written to the handle's pinned scratch (polls a completion word the kernel stores
 * after the result; bounded, falls back to a stream synchronise) */
int  mi355x_handle_wait_result(mi355x_handle_t h);
/* queue a copy of count (<= 64) doubles from device memory to the pinned scratch followed by the completion number:
 * mi355x_handle_wait_result then returns as soon as THAT point of the stream is reached, whatever was queued behind it */
int  mi355x_handle_publish(mi355x_handle_t h, const double *src_dev, int count);
int  mi355x_handle_publish_at(mi355x_handle_t h, const double *src_dev, int count, int dst_offset);   /* into host_scratch[dst_offset ..) */
void *mi355x_handle_stream(mi355x_handle_t h);           /* the raw hipStream_t */
/* pinned, device-visible scratch of >= 64 doubles owned by the handle
 * (reduction results are written here by the device, read by the host
 * after mi355x_handle_synchronize) */
double *mi355x_handle_host_scratch(mi355x_handle_t h);
double *mi355x_handle_device_scratch(mi355x_handle_t h);  /* >= 64 doubles in HBM */

int  mi355x_malloc(void **dptr, size_t bytes);
int  mi355x_free(void *dptr);
int  mi355x_host_malloc(void **hptr, size_t bytes);      /* pinned + device-mapped */
int  mi355x_host_free(void *hptr);
int  mi355x_memcpy_h2d(mi355x_handle_t h, void *dst, const void *src, size_t bytes); /* async on h */
int  mi355x_memcpy_d2h(mi355x_handle_t h, void *dst, const void *src, size_t bytes); /* async on h */
int  mi355x_memcpy_d2d(mi355x_handle_t h, void *dst, const void *src, size_t bytes); /* async on h */
int  mi355x_memset(mi355x_handle_t h, void *dst, int byte, size_t bytes);

int  mi355x_event_create(mi355x_event_t *e);
int  mi355x_event_destroy(mi355x_event_t e);
int  mi355x_event_record(mi355x_event_t e, mi355x_handle_t h);
int  mi355x_event_synchronize(mi355x_event_t e);
int  mi355x_event_elapsed_ms(mi355x_event_t start, mi355x_event_t stop, float *ms);
int  mi355x_handle_wait_event(mi355x_handle_t h, mi355x_event_t e);  /* hipStreamWaitEvent */

/* hipGraph capture of whatever is enqueued on the handle between begin and end; replay with one call */
int  mi355x_graph_capture_begin(mi355x_handle_t h);
int  mi355x_graph_capture_end(mi355x_handle_t h, void **graph_exec);
int  mi355x_graph_launch(mi355x_handle_t h, void *graph_exec);
int  mi355x_graph_destroy(void *graph_exec);

/* ---- Vec element-wise kernels (24-32 B/element, HBM-bound) ------------ */
/* VecSet_Seq          src/vec/vec/impls/seq/dvec2.c:722      x[i] = alpha */
int mi355x_vec_set(mi355x_handle_t h, size_t n, double alpha, double *x);
/* VecCopy_Seq         src/vec/vec/impls/seq/bvec2.c:464      y = x */
int mi355x_vec_copy(mi355x_handle_t h, size_t n, const double *x, double *y);
/* VecScale_Seq        src/vec/vec/impls/seq/bvec1.c:183      x *= alpha (dscal) */
int mi355x_vec_scale(mi355x_handle_t h, size_t n, double alpha, double *x);
/* VecSwap_Seq         src/vec/vec/impls/seq/bvec2.c:519 */
int mi355x_vec_swap(mi355x_handle_t h, size_t n, double *x, double *y);
/* VecAXPY_Seq         src/vec/vec/impls/seq/bvec1.c:244      y += alpha x (daxpy) */
int mi355x_vec_axpy(mi355x_handle_t h, size_t n, double alpha, const double *x, double *y);
/* VecAYPX_Seq         src/vec/vec/impls/seq/dvec2.c:971      y = x + alpha y */
int mi355x_vec_aypx(mi355x_handle_t h, size_t n, double alpha, const double *x, double *y);
/* VecAXPBY_Seq        src/vec/vec/impls/seq/bvec1.c:320      y = alpha x + beta y */
int mi355x_vec_axpby(mi355x_handle_t h, size_t n, double alpha, double beta, const double *x, double *y);
/* VecWAXPY_Seq        src/vec/vec/impls/seq/dvec2.c:1082     w = y + alpha x */
int mi355x_vec_waxpy(mi355x_handle_t h, size_t n, double alpha, const double *x, const double *y, double *w);
/* VecAXPBYPCZ_Seq     src/vec/vec/impls/seq/bvec1.c:418      z = alpha x + beta y + gamma z */
int mi355x_vec_axpbypcz(mi355x_handle_t h, size_t n, double alpha, double beta, double gamma,
                        const double *x, const double *y, double *z);
/* VecPointwiseMult_Seq   src/vec/vec/impls/seq/bvec2.c:234   w = x .* y (w may alias x or y) */
int mi355x_vec_pointwise_mult(mi355x_handle_t h, size_t n, const double *x, const double *y, double *w);
/* VecPointwiseDivide_Seq src/vec/vec/impls/seq/bvec2.c:298   w = x ./ y */
int mi355x_vec_pointwise_divide(mi355x_handle_t h, size_t n, const double *x, const double *y, double *w);
/* VecReciprocal_Default  src/vec/vec/utils/vinv.c            x[i] = 1/x[i] where x[i] != 0 */
int mi355x_vec_reciprocal(mi355x_handle_t h, size_t n, double *x);
/* PCSetUp_Jacobi host loop  src/ksp/pc/impls/jacobi/jacobi.c:182-190  d = (d==0) ? 1 : 1/d, done on device */
int mi355x_vec_jacobi_invert(mi355x_handle_t h, size_t n, double *d, int *nzero_dev);
/* VecMAXPY_Seq        src/vec/vec/impls/seq/dvec2.c:836      x += sum_j alpha[j] y_j ; grouping
 * (nv%4 first, then fours) and left-to-right product sums follow petscaxpy.h:101-110.
 * y: host array of nv device pointers; alpha: host array. */
int mi355x_vec_maxpy(mi355x_handle_t h, size_t n, int nv, const double *alpha, const double *const *y, double *x);

/* ---- Vec reductions (8-16 B/element) ---------------------------------- */
/* Results are written to `out` (device-accessible: HBM or pinned host memory) when the
 * kernel completes on the handle's stream; the caller synchronises.  Two-level
 * fixed-shape tree: per-lane strided partial -> wavefront shuffle -> LDS -> per-block
 * partial -> last-arriving block sums the per-block partials in block order. */
/* VecDot_Seq/VecTDot_Seq  src/vec/vec/impls/seq/bvec1.c:57,122   out[0] = sum x_i y_i */
int mi355x_vec_dot(mi355x_handle_t h, size_t n, const double *x, const double *y, double *out);
/* VecNorm_Seq         src/vec/vec/impls/seq/bvec2.c:605
 * type 0: NORM_1 -> out[0]=sum|x|; 1: NORM_2 -> out[0]=sum x^2 (caller takes sqrt, as pvec2.c:62 does);
 * 3: NORM_INFINITY -> out[0]=max|x| (NaN propagates, bvec2.c:628-630); 4: NORM_1_AND_2 -> out[0]=sum|x|, out[1]=sum x^2 */
int mi355x_vec_norm(mi355x_handle_t h, size_t n, int type, const double *x, double *out);
/* VecDotNorm2 (BiCGStab)  src/vec/vec/utils/vinv.c:1200   out[0] = sum s_i t_i, out[1] = sum t_i^2 */
int mi355x_vec_dotnorm2(mi355x_handle_t h, size_t n, const double *s, const double *t, double *out);
/* VecAYPX with the scalar's numerator on the device: y = x + (*num_dev / den) y (alpha == 0 -> copy, dvec2.c:980).
 * KSPSolve_CG's p = z + (beta_new/beta_old) p when beta_new has not reached the host yet. */
int mi355x_vec_aypx_dev(mi355x_handle_t h, size_t n, const double *num_dev, double den, const double *x, double *y);
/* Fused CG update, one sweep for cg.c:206-232 + PCApply_Jacobi (jacobi.c:266-277):
 * x += a p; r += (-a) w; z = r .* d (d == NULL: z = r); out[0] = sum z*z, out[1] = sum z*r, out[2] = sum r*r.  Same
 * bits as mi355x_vec_axpy x2, mi355x_vec_pointwise_mult, mi355x_vec_norm(2) of z / of r, mi355x_vec_dot in sequence. */
int mi355x_vec_cg_update(mi355x_handle_t h, size_t n, double a, const double *p, const double *w, const double *d,
                         double *x, double *r, double *z, double *out);
/* The same sweep with the step length formed on the device: a = beta / *dpi_dev (dpi = p'w left in device memory by
 * mi355x_vec_dot, all-reduced there on several ranks).  The break-down tests of cg.c:196-199 (dpi NaN/Inf, dpi == 0,
 * check_sign && dpi*dpiold <= 0) are evaluated in the kernel: if one fires, x, r, z are left untouched.
 * out[0] = sum z*z, out[1] = sum z*r, out[2] = sum r*r, out[3] = dpi (for the host's own copy of those tests).
 * also_to_host != 0 with a device `out`: the four values are stored to the first pinned scratch slots as well, followed by the completion
 * number (mi355x_handle_wait_result) -- the result serves a later kernel and the host without a second launch. */
int mi355x_vec_cg_update_dev(mi355x_handle_t h, size_t n, double beta, const double *dpi_dev, double dpiold, int check_sign,
                             const double *p, const double *w, const double *d, double *x, double *r, double *z, double *out,
                             int also_to_host);
/* Fused forms for KSPSolve_BCGS (src/ksp/ksp/impls/bcgs/bcgs.c:43-160); same bits as the calls they replace.
 * pmult_dot:      w = x .* d (PCApply_Jacobi jacobi.c:266; d == NULL: identity), out[0] = sum w*y       (bcgs.c:108-109)
 * pmult_dotnorm2: w = x .* d,                       out[0] = sum s*w, out[1] = sum w*w               (bcgs.c:116-117)
 * bcgs_update:    x = alpha p + omega s + x ; r = s - omega t ; out[0] = sum r*r, out[1] = sum r*rp  (bcgs.c:134-136,103) */
int mi355x_vec_pmult_dot(mi355x_handle_t h, size_t n, const double *x, const double *d, const double *y, double *w, double *out);
int mi355x_vec_pmult_dotnorm2(mi355x_handle_t h, size_t n, const double *x, const double *d, const double *s, double *w, double *out);
int mi355x_vec_bcgs_update(mi355x_handle_t h, size_t n, double alpha, double omega, const double *p, const double *s, const double *t,
                           const double *rp, double *x, double *r, double *out);
/* VecMDot_Seq         src/vec/vec/impls/seq/dvec2.c:146     out[j] = sum_i x_i y_j,i , j<nv ; x read once per 16 y's */
int mi355x_vec_mdot(mi355x_handle_t h, size_t n, int nv, const double *x, const double *const *y, double *out);
/* KSPGMRESCycle fusions (gmres.c:118-209, borthog2.c:60-66): x += sum_j sign*coef_dev[j]*y_j in VecMAXPY_Seq's grouping with
 * sum x_new^2 from the same sweep (bits of VecMAXPY + VecNorm); x *= 1/sqrt(*norm2_dev) with VecNormalize's special cases */
int mi355x_vec_maxpy_dev_norm2(mi355x_handle_t h, size_t n, int nv, const double *coef_dev, double sign, const double *const *y,
                               double *x, double *out);
int mi355x_vec_scale_rnorm_dev(mi355x_handle_t h, size_t n, const double *norm2_dev, double *x);
/* VecSum / VecMax helpers are not on the Krylov path and are not provided. */

/* ---- CSR SpMV (MatMult_SeqAIJ family) --------------------------------- */
/* Analysis: partitions rows into row blocks (<= MI355X_SPMV_BLOCK_NNZ nonzeros staged
 * through LDS per workgroup).  ai_host is the host copy of the row pointer (m+1 ints).
 * If rows_host != NULL the matrix is in compressed-row form (reference:
 * src/mat/utils/compressedrow.c:28): ai_host has nrows+1 entries and rows_host[k]
 * is the output row of compressed row k (device copy kept in the plan). */
int mi355x_spmv_plan_create(mi355x_handle_t h, int nrows, const int *ai_host, const int *rows_host,
                            mi355x_spmv_plan_t *plan);
int mi355x_spmv_plan_destroy(mi355x_spmv_plan_t plan);
/* Optional second analysis step (not for compressed-row plans): if the matrix uses <= 256 distinct offsets
 * (col - row) -- stencil operators -- one byte per nonzero is stored next to the CSR arrays and the SpMV
 * kernels stream val + 1 B instead of val + 4 B col.  Same arithmetic, same bits; no-op otherwise. */
int mi355x_spmv_plan_compress_indices(mi355x_handle_t h, mi355x_spmv_plan_t plan, const int *ai_host, const int *aj_host);
/* row-pattern kernel (stencil matrices: 4 bytes per row instead of 1 byte per nonzero + the row pointer; found by mi355x_spmv_plan_compress_indices):
 * on = 0/1 switches it, on < 0 only asks; *npat = size of the dictionary, 0 when the plan has none */
int mi355x_spmv_plan_use_patterns(mi355x_spmv_plan_t p, int on, int *npat);
/* value patterns (constant-coefficient operators: whole rows -- offsets AND values, bit for bit -- taken from a table of
 * <= 512 entries; 2 bytes per row, the value array is not read).  _value_patterns derives the table from the host copy of
 * the values that are (about to be) on the device and must be called again after every upload; _drop_ after any change of
 * the device values that did not go through it; _use_: on = 0/1 switches, on < 0 only asks.  *nvpat = distinct rows, 0 =
 * not in use (varying coefficients, compressed-row plan, switched off).  Replaces nothing in the reference: the reference
 * streams MatMult_SeqAIJ's a[] (aij.c:1225); the result carries the same bits. */
int mi355x_spmv_plan_value_patterns(mi355x_handle_t h, mi355x_spmv_plan_t p, const int *ai_host, const int *aj_host,
                                    const double *aa_host, int *nvpat);
int mi355x_spmv_plan_drop_value_patterns(mi355x_spmv_plan_t p);
int mi355x_spmv_plan_use_value_patterns(mi355x_spmv_plan_t p, int on, int *nvpat);
int mi355x_spmv_plan_is_compressed(mi355x_spmv_plan_t plan, int *ntab);
/* 1 when mi355x_spmv_csr_dot would run on this plan with this value array (hipErrorNotSupported otherwise) */
int mi355x_spmv_plan_dot_available(mi355x_spmv_plan_t plan, const double *a, int *yes);
/* Optional analysis step for matrices with repeated row patterns (finite elements with several dof per node): the
 * MI355X form of the reference's inodes.  ns[nnodes] are the node sizes Mat_CheckInode finds (src/mat/impls/aij/seq/
 * inode.c:3981-3998: consecutive rows with identical column lists, <= limit rows per node).  Each group's column list is
 * stored once and the SpMV streams val + (4 / rows-per-group) B instead of val + 4 B per nonzero; `a` stays the CSR
 * array.  No-op (returns 0, plan unchanged) when it would not pay or the plan is compressed-row / index-compressed. */
int mi355x_spmv_plan_group_rows(mi355x_handle_t h, mi355x_spmv_plan_t plan, const int *ai_host, const int *aj_host,
                                int nnodes, const int *ns);
/* Row sums two products at a time, sum += a0 x0 + a1 x1 (MatMult_SeqAIJ_Inode / MatMultAdd_SeqAIJ_Inode, inode.c:430-440,
 * 619-631), instead of one at a time (MatMult_SeqAIJ, aij.h:383-386): set when the reference would run its inode
 * routines on this matrix, so that the result carries ITS rounding.  Applies to the one-lane-per-row paths. */
int mi355x_spmv_plan_set_pairsum(mi355x_spmv_plan_t plan, int on);
int mi355x_spmv_plan_group_info(mi355x_spmv_plan_t plan, int *ngroups, long *nshared_indices, int *pairsum);
int mi355x_spmv_plan_info(mi355x_spmv_plan_t plan, int *nblocks, int *nlong, size_t *workspace_bytes);
/* MatMult_SeqAIJ      src/mat/impls/aij/seq/aij.c:1225 (loop 1269-1277, macro aij.h:383-386)
 *   y[r] = sum_k a[k] x[j[k]], products summed in k order starting from 0.0
 * aa and aj must be allocated with 16 bytes of slack past their last element (mi355x_malloc of nz elements + 16 B):
 * the kernels read aligned pairs, and the pair holding the last element may extend one element past it. */
int mi355x_spmv_csr(mi355x_handle_t h, mi355x_spmv_plan_t plan, const int *ai, const int *aj,
                    const double *aa, const double *x, double *y);
/* y = A x and sum_r x_r y_r from ONE pass over the matrix (KSPSolve_CG: w = A p, dpi = p'w; cg.c:190-191): the SpMV
 * leaves one value per row block in the plan, mi355x_spmv_dot_finish adds them in block order into out[0] (device-
 * accessible).  Square matrices whose plan runs one of the row-block kernels with per-block sums (value patterns, row patterns,
 * 8-bit column offsets: mi355x_spmv_plan_dot_available): mi355x_spmv_csr_dot returns hipErrorNotSupported (801) otherwise and does
 * nothing.  y carries the bits of mi355x_spmv_csr; the dot uses a fixed tree. */
int mi355x_spmv_csr_dot(mi355x_handle_t h, mi355x_spmv_plan_t plan, const int *ai, const int *aj, const double *aa,
                        const double *x, double *y);
int mi355x_spmv_dot_finish(mi355x_handle_t h, mi355x_spmv_plan_t plan, double *out);
/* MatMultAdd_SeqAIJ   src/mat/impls/aij/seq/aij.c:1291   z[r] = y[r] + sum_k ... (sum starts from y[r]);
 * z may alias y.  With a compressed-row plan only the listed rows are touched (z must alias y
 * or already hold y, as aij.c:1314-1316 arranges). */
/* y = d .* (A x): MatMult + PCApply_Jacobi's VecPointwiseMult (jacobi.c:266) in one kernel, bits of the two calls */
int mi355x_spmv_csr_scaled(mi355x_handle_t h, mi355x_spmv_plan_t plan, const int *ai, const int *aj, const double *aa,
                           const double *x, const double *d, double *y);
int mi355x_spmv_csr_add(mi355x_handle_t h, mi355x_spmv_plan_t plan, const int *ai, const int *aj,
                        const double *aa, const double *x, const double *y, double *z);
/* MatMultTranspose[Add]_SeqAIJ  src/mat/impls/aij/seq/aij.c:1078-1135 is served by the same two
 * kernels applied to an explicit transpose whose rows list contributions in increasing
 * original-row order (the order the reference's scatter loop adds them in). */

/* MatSetValuesBatch with an unchanged pattern (src/mat/interface/matrix.c:1698; GPU analogue aijAssemble.cu:157):
 * aa[segslot[s]] += v[order[k]] for k in [segptr[s], segptr[s+1]) in that order, s < nseg.  The map is built on the host
 * once per connectivity; contributions are added in the order of the reference's loop of MatSetValues(ADD_VALUES). */
int mi355x_csr_assemble(mi355x_handle_t h, int nseg, const int *segptr, const int *segslot, const int *order, const double *v, double *aa);
/* MatDiagonalScale_SeqAIJ  src/mat/impls/aij/seq/aij.c:2055   a[k] = (a[k] * l[row]) * r[col]; l or r may be NULL */
int mi355x_csr_diagonal_scale(mi355x_handle_t h, int m, const int *ai, const int *aj, double *aa, const double *l, const double *r);
/* MatGetDiagonal_SeqAIJ  src/mat/impls/aij/seq/aij.c:1040   d[r] = A[r,r] or 0 */
int mi355x_csr_get_diagonal(mi355x_handle_t h, int m, const int *ai, const int *aj, const double *aa, double *d);

/* ---- column-tiled CSR SpMV: x staged in LDS (csrc/spmv_tiled.hip) ------ */
/* For matrices whose x gathers miss the caches (rows that pick columns from a wide window without neighbouring rows sharing them):
 * MatMult_SeqAIJ / MatMultAdd_SeqAIJ (aij.c:1225, 1291) re-cut into row panels (<= 6143 rows, equal shares of the nonzeros) x column
 * tiles of 2048 entries of x; a (panel, tile) pair with >= stage_min entries gathers from a copy of that tile in LDS and adds into row
 * sums that live in LDS too (12-byte entries in dense blocks of 128, two per lane, ds_add_f64); the thin pairs' entries (the remainder)
 * follow in the same streams, grouped by windows of 2^18 columns, their x gathered from global memory -- one kernel, y written once.
 * Products of a row's staged entries are added in column order, then the remainder's in column order: agrees with the reference to
 * rounding (<= 1e-12 * sum |a_ij x_j|; bit for bit when nothing or everything is left to the remainder), reproducible.
 *   _probe    fraction of sampled gathers that touch a 128-byte line of x nothing else in their 32-row group touches (host only)
 *   _build    the layout from the host CSR pattern (host only: no device call); stage_min <= 0: 1024
 *   _upload   tables to the device; values gathered from the CSR value array that is on the device (aa_dev)
 *   _refresh_values   after the device values changed under the same pattern
 *   mi355x_spmv_tiled  yout = (yin ? yin : 0) + A x; x 16-byte aligned, else hipErrorNotSupported (801) and nothing is launched */
typedef struct mi355x_spmv_tiled_s *mi355x_spmv_tiled_t;
int mi355x_spmv_tiled_probe(int m, const int *ai_host, const int *aj_host, double *lines_per_nonzero);
int mi355x_spmv_tiled_build(int m, int n, const int *ai_host, const int *aj_host, int stage_min, mi355x_spmv_tiled_t *plan);
/* nblocks: 128-entry blocks stored over all wavefronts (nnz_staged / (128 nblocks): the share that is not padding) */
int mi355x_spmv_tiled_info(mi355x_spmv_tiled_t plan, long *nnz_staged, long *nnz_remainder, int *npanels, int *npairs, long *nblocks);
int mi355x_spmv_tiled_geometry(int *panel_rows, int *tile_cols, int *waves, int *block_entries);
int mi355x_spmv_tiled_upload(mi355x_handle_t h, mi355x_spmv_tiled_t plan, const double *aa_dev);
int mi355x_spmv_tiled_refresh_values(mi355x_handle_t h, mi355x_spmv_tiled_t plan, const double *aa_dev);
int mi355x_spmv_tiled(mi355x_handle_t h, mi355x_spmv_tiled_t plan, const double *x, const double *yin, double *yout);
/* development: which = 1 the staged part only, 2 the remainder only (their separate cost), 0 both */
int mi355x_spmv_tiled_parts(mi355x_handle_t h, mi355x_spmv_tiled_t plan, const double *x, const double *yin, double *yout, int which);
/* tests: one host array of the layout (0 pt_ptr, 1 pt_tile, 2 pw_e0, 3 word, 4 perm, 7 pw_f0, 8 fw_ptr, 9 fw_win, 10 wrow, 11 prow)
 * until _drop_host releases the host copy */
int mi355x_spmv_tiled_debug_get(mi355x_spmv_tiled_t plan, int which, void *out, size_t cap_bytes, size_t *bytes);
int mi355x_spmv_tiled_drop_host(mi355x_spmv_tiled_t plan);
int mi355x_spmv_tiled_destroy(mi355x_spmv_tiled_t plan);

/* ---- BCSR SpMV (MatMult_SeqBAIJ_3/_4/_N) ------------------------------- */
/* src/mat/impls/baij/seq/baij2.c:331 (bs=3), :387 (bs=4), :981 (N); blocks column-major */
int mi355x_spmv_bsr(mi355x_handle_t h, int mbs, int bs, const int *ai, const int *aj,
                    const double *aa, const double *x, double *y);          /* one wavefront per block row, no analysis */
/* row-block streaming variant: `plan` = mi355x_spmv_plan_create over the block-row pointer multiplied by bs*bs
 * (it then partitions by VALUES); aligned 16-byte loads need aa 16-byte aligned */
int mi355x_spmv_bsr_planned(mi355x_handle_t h, mi355x_spmv_plan_t plan, int bs, const int *ai, const int *aj,
                            const double *aa, const double *x, double *y);
/* MatMultAdd_SeqBAIJ_3/_4/_N  src/mat/impls/baij/seq/baij2.c:1168-1480   z = y + A x with the same kernel; z may alias y */
int mi355x_spmv_bsr_planned_add(mi355x_handle_t h, mi355x_spmv_plan_t plan, int bs, const int *ai, const int *aj,
                                const double *aa, const double *x, const double *y, double *z);
/* the same product with the row-block kernel's form chosen by the caller (development / A-B runs): x_in_lds != 0 stages the x
 * entries of the row block's block columns in LDS once per block instead of gathering them once per value */
int mi355x_spmv_bsr_planned_form(mi355x_handle_t h, mi355x_spmv_plan_t plan, int bs, int x_in_lds, const int *ai, const int *aj,
                                 const double *aa, const double *x, double *y);

/* bs = 4 on the matrix cores (v_mfma_f64_4x4x4_4b_f64 as a fused multiply + 4-lane reduction; BASELINE configs[4]); one
 * wavefront per block row, no analysis.  variant 0: 16-byte loads (8 blocks per step), 1: 8-byte loads (4 per step).
 * FMA arithmetic and a tree over the four block columns: agrees with MatMult_SeqBAIJ_4 (baij2.c:387) to rounding. */
int mi355x_spmv_bsr4_mfma(mi355x_handle_t h, int mbs, int variant, const int *ai, const int *aj, const double *aa,
                          const double *x, double *y);

/* PCApply_PBJacobi_N  src/ksp/pc/impls/pbjacobi/pbjacobi.c:20-200   y_i = D_i^-1 x_i, idiag = mbs inverted bs x bs blocks
 * (column-major, as MatInvertBlockDiagonal_SeqBAIJ baij.c:13 leaves them); the reference's left-to-right row sums */
int mi355x_pbjacobi_apply(mi355x_handle_t h, int mbs, int bs, const double *idiag, const double *x, double *y);

/* ---- ILU(0) triangular solves ---------------------------- */
/* MatSolve_SeqAIJ_NaturalOrdering  src/mat/impls/aij/seq/aijfact.c:3126-3172 on the factor layout of :1628-1700,
 * one launch per dependency level; `rows` lists the rows of the level (device array).  Lower: x[i] = b[i] - L(i,:)x
 * (b may alias x); upper: x[i] = (x[i] - U(i,:)x) * inverted diagonal. */
int mi355x_ilu0_lower_level(mi355x_handle_t h, int nrows, const int *rows, const int *bi, const int *bj,
                            const double *ba, const double *b, double *x);
int mi355x_ilu0_upper_level(mi355x_handle_t h, int nrows, const int *rows, const int *bj, const double *ba,
                            const int *bdiag, double *x);

/* The same two solves WITHOUT a kernel boundary per level: one launch per triangular solve, rows sorted by level and
 * stored as sliced ELL (one wavefront per 64 rows), dependencies handed over through the solution values themselves
 * (a sentinel bit pattern means "not computed yet"; write-through stores, polling loads).  Same one-lane-per-row,
 * column-order arithmetic: same bits.  plan_create: lev[i] = dependency level of row i (every level non-empty), row i's
 * off-diagonal entries are cj/cv[rp[i] .. rp[i]+rl[i]), dinv != NULL (inverted diagonal per row) marks the upper solve.
 * apply: y = U^-1 L^-1 b; returns non-zero without launching if an earlier application timed out on a dependency. */
typedef struct mi355x_trisolve_plan_s *mi355x_trisolve_plan_t;
int mi355x_trisolve_plan_create(mi355x_handle_t h, int n, int nlev, const int *lev, const int *rp, const int *rl, const int *cj,
                                const double *cv, const double *dinv, mi355x_trisolve_plan_t *plan);
/* by_level != 0: rows summed in the order of their dependencies' levels instead of column order (factors of inode matrices;
 * tolerance instead of bit-exactness against MatSolve_SeqAIJ_NaturalOrdering), large levels on slice boundaries */
int mi355x_trisolve_plan_create_ordered(mi355x_handle_t h, int n, int nlev, const int *lev, const int *rp, const int *rl, const int *cj,
                                        const double *cv, const double *dinv, int by_level, mi355x_trisolve_plan_t *plan);
/* upper solve of an incomplete Cholesky factor U^T D U: right-hand side entry i is multiplied by rscale[i] (= 1/D(i)) before row
 * i's sum starts -- the x[i] = xi * (1/D(i)) between the two sweeps of MatSolve_SeqSBAIJ_1_NaturalOrdering (sbaijfact2.c:1977-2015);
 * entries in the caller's order */
int mi355x_trisolve_plan_create_scaled(mi355x_handle_t h, int n, int nlev, const int *lev, const int *rp, const int *rl, const int *cj,
                                       const double *cv, const double *dinv, const double *rscale, mi355x_trisolve_plan_t *plan);
/* node-blocked plans for the factor of a matrix with inodes (MatSolve_SeqAIJ_Inode, src/mat/impls/aij/seq/inode.c:2327-2760): the
 * n rows form nnodes nodes of 1..5 consecutive rows (nstart[nnodes + 1]) whose factor rows share one column list and are coupled by a
 * dense triangle; nodelev[u] = dependency level of node u among the nodes.  Row-level arrays as above, in the reference's stored order.
 * One lane per node; the reference routine's summation order (shared columns two at a time, then the couplings inside the node):
 * bit for bit MatSolve_SeqAIJ_Inode with by_level == 0.  Returns hipErrorInvalidValue for a factor without that shape.  Lower and
 * upper plan of an application must both be node plans of the same partition and the same block_columns.
 * block_columns != 0: additionally every node has the same number of rows (<= 4) and every shared column list is a run of WHOLE
 * dependency nodes (a FEM matrix with a fixed number of dofs per node): one list entry and one contiguous gather per dependency node
 * -- fewer, larger batches; same column sequence, same bits.  hipErrorInvalidValue when the factor is not of that kind. */
int mi355x_trisolve_plan_create_nodes(mi355x_handle_t h, int n, int nnodes, const int *nstart, int nlev, const int *nodelev, const int *rp, const int *rl,
                                      const int *cj, const double *cv, const double *dinv, int by_level, int block_columns, mi355x_trisolve_plan_t *plan);
/* the lower and the upper row-granular plan of one factor, built side by side (two host threads): the arguments of
 * mi355x_trisolve_plan_create_ordered for the lower plan, of ..._create_ordered / ..._create_scaled (rscale_up != NULL) for the upper
 * one; on failure neither plan is returned */
int mi355x_trisolve_plan_create_pair(mi355x_handle_t h, int n, int by_level,
                                     int nlev_lo, const int *lev_lo, const int *rp_lo, const int *rl_lo, const int *cj_lo, const double *cv_lo,
                                     int nlev_up, const int *lev_up, const int *rp_up, const int *rl_up, const int *cj_up, const double *cv_up,
                                     const double *dinv_up, const double *rscale_up, mi355x_trisolve_plan_t *lower, mi355x_trisolve_plan_t *upper);
/* the lower and the upper node plan of one factor, built side by side (two host threads: the analyses are independent): arguments as
 * above, once per factor; on failure neither plan is returned */
int mi355x_trisolve_plan_create_nodes_pair(mi355x_handle_t h, int n, int nnodes, const int *nstart, int by_level, int block_columns,
                                           int nlev_lo, const int *nodelev_lo, const int *rp_lo, const int *rl_lo,
                                           int nlev_up, const int *nodelev_up, const int *rp_up, const int *rl_up,
                                           const int *cj, const double *cv, const double *dinv, mi355x_trisolve_plan_t *lower, mi355x_trisolve_plan_t *upper);
int mi355x_trisolve_plan_destroy(mi355x_trisolve_plan_t plan);
int mi355x_trisolve_apply(mi355x_handle_t h, mi355x_trisolve_plan_t lower, mi355x_trisolve_plan_t upper, const double *b, double *y);
int mi355x_trisolve_aborted(mi355x_trisolve_plan_t plan, int *aborted);
/* tests: one array of a row plan copied back (which: 0 slice offsets, 1 (length, sub-step) words, 2 position -> row, 3 / 4 sliced-ELL
 * column positions / values, 5 inverted diagonals, 6 right-hand-side scales, 7 sub-steps per slice, 8 row -> position, 9 level
 * extents); *bytes = its size, copied when it fits cap_bytes.  Row plans in column order are laid out on the device
 * (csrc/trisolve_build.hip); MI355X_TRISOLVE_BUILD=host in the environment keeps the host threads' route: the same arrays. */
int mi355x_trisolve_debug_get(mi355x_trisolve_plan_t plan, int which, void *out, size_t cap_bytes, size_t *bytes);
/* the same application over the same plans with one launch per dependency level and no hand-off between wavefronts (same
 * per-row order of the products: same bits): what a caller falls back to after a sync-free application gave up, whatever the
 * factor (ILU(0), ICC(0)); does not consult or change the abort flags */
int mi355x_trisolve_apply_levels(mi355x_handle_t h, mi355x_trisolve_plan_t lower, mi355x_trisolve_plan_t upper, const double *b, double *y);
/* development / tests: raise a plan's abort flag as a dependency wait that gave up would */
int mi355x_trisolve_debug_set_aborted(mi355x_trisolve_plan_t plan, int value);

/* ---- halo pack / unpack (VecScatter) ---------------------------------- */
/* Pack_1    src/vec/vec/utils/vpscat.c:493   buf[k] = x[idx[k]] */
int mi355x_pack(mi355x_handle_t h, size_t n, const int *idx, const double *x, double *buf);
/* UnPack_1  src/vec/vec/utils/vpscat.c:503-534   INSERT: y[idx[k]] = buf[k]; ADD: y[idx[k]] += buf[k]; MAX: y[idx[k]] =
 * PetscMax(y[idx[k]], buf[k]) (idx == NULL means contiguous: y[k]).  ADD / MAX require idx entries to be distinct within one call. */
int mi355x_unpack_insert(mi355x_handle_t h, size_t n, const int *idx, const double *buf, double *y);
int mi355x_unpack_add(mi355x_handle_t h, size_t n, const int *idx, const double *buf, double *y);
int mi355x_unpack_max(mi355x_handle_t h, size_t n, const int *idx, const double *buf, double *y);

/* ---- measurement ------------------------------------------------------- */
/* STREAM-style copy (pattern: src/benchmarks/streams/CUDAVersion.cu) for the achievable-bandwidth line */
int mi355x_stream_triad(mi355x_handle_t h, size_t n, double alpha, const double *b, const double *c, double *a);

#ifdef __cplusplus
}
#endif
#endif
