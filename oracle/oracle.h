/*
 * oracle.h -- CPU restatement of the reference's Krylov hot path (erdc/petsc-dev, PETSc 3.3.0-dev).
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the checker: it may be imported, linked or
 * executed by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, and by
 * nothing else.  The product (petsc-dev_amd/) never calls into it and has no CPU fallback.
 *
 * Every function is a plain-C, single-thread restatement of one reference routine (cited
 * file:line, relative to the PETSc tree) operating on raw arrays: same loop order, same
 * special cases, products and sums in the same order.  Built with -O2 -ffp-contract=off:
 * a*b+c is a rounded multiply followed by a rounded add, as in a reference build for
 * baseline x86-64 (no FMA).  BLAS-1 calls of the reference (ddot_/daxpy_/dscal_/dasum_,
 * include/petscblaslapack_uscore.h) are restated with netlib reference-BLAS semantics
 * (strictly left-to-right accumulation).
 *
 * Pinning: the reference cannot be built under this round's rules (its headers need the
 * configure-generated petscconf.h; config/BuildSystem is an un-vendored submodule), so the
 * oracle is pinned by the reference's own golden outputs, committed under tests/golden/
 * (tests/test_oracle_golden.py): src/mat/examples/tests/output/ex5_{11_A,11_B,21,23}.out,
 * src/ksp/ksp/examples/{tests/output/ex3_1,ex3_2,ex4_1, tutorials/output/ex2_bjacobi*,ex2f_1,
 * ex5_1,ex5_2,ex9_1}.out -- see DESIGN.md section 3.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- Vec (src/vec/vec/impls/seq) ---- */
void   orc_vec_set(size_t n, double alpha, double *x);                                   /* dvec2.c:722 */
void   orc_vec_copy(size_t n, const double *x, double *y);                               /* bvec2.c:464 */
void   orc_vec_scale(size_t n, double alpha, double *x);                                 /* bvec1.c:183 */
void   orc_vec_swap(size_t n, double *x, double *y);                                     /* bvec2.c:519 */
void   orc_vec_axpy(size_t n, double alpha, const double *x, double *y);                 /* bvec1.c:244 */
void   orc_vec_aypx(size_t n, double alpha, const double *x, double *y);                 /* dvec2.c:971 */
void   orc_vec_axpby(size_t n, double alpha, double beta, const double *x, double *y);   /* bvec1.c:320 */
void   orc_vec_waxpy(size_t n, double alpha, const double *x, const double *y, double *w); /* dvec2.c:1082 */
void   orc_vec_axpbypcz(size_t n, double alpha, double beta, double gamma, const double *x, const double *y, double *z); /* bvec1.c:418 */
void   orc_vec_pointwise_mult(size_t n, const double *x, const double *y, double *w);    /* bvec2.c:234 */
void   orc_vec_pointwise_divide(size_t n, const double *x, const double *y, double *w);  /* bvec2.c:298 */
void   orc_vec_reciprocal(size_t n, double *x);                                          /* vinv.c VecReciprocal_Default */
void   orc_vec_maxpy(size_t n, int nv, const double *alpha, const double *const *y, double *x); /* dvec2.c:836 */
void   orc_set_device_reduction_order(int on);   /* test aid: reductions in the HIP kernels' summation tree instead of the reference's loop */
double orc_vec_dot(size_t n, const double *x, const double *y);                          /* bvec1.c:57,122 */
void   orc_vec_mdot(size_t n, int nv, const double *x, const double *const *y, double *z); /* dvec2.c:146 */
/* type: 0 NORM_1, 1 NORM_2, 2 FROBENIUS, 3 INFINITY, 4 NORM_1_AND_2 (out[0],out[1]) */
void   orc_vec_norm(size_t n, int type, const double *x, double *out);                   /* bvec2.c:605 */
void   orc_vec_dotnorm2(size_t n, const double *s, const double *t, double *dp, double *nm); /* vinv.c:1200 */

/* ---- SeqAIJ (src/mat/impls/aij/seq/aij.c) ---- */
void orc_spmv_csr(int m, const int *ai, const int *aj, const double *aa, const double *x, double *y);           /* :1225 */
void orc_spmv_csr_add(int m, const int *ai, const int *aj, const double *aa, const double *x, const double *y, double *z); /* :1291 */
void orc_spmv_csr_transpose(int m, int n, const int *ai, const int *aj, const double *aa, const double *x, double *y);     /* :1124 */
void orc_spmv_csr_transpose_add(int m, int n, const int *ai, const int *aj, const double *aa, const double *x, const double *z, double *y); /* :1078 */
void orc_csr_diagonal_scale(int m, const int *ai, const int *aj, double *aa, const double *l, const double *r);                 /* :2055 */
void orc_csr_get_diagonal(int m, const int *ai, const int *aj, const double *aa, double *d);                    /* :1040 */
/* inode variant (src/mat/impls/aij/seq/inode.c:392-578 mult, :3964-4034 detection) */
int  orc_check_inode(int m, const int *ai, const int *aj, int limit, int *ns);
void orc_spmv_csr_inode(int m, const int *ai, const int *aj, const double *aa, const double *x, double *y);
void orc_spmv_csr_inode_add(int m, const int *ai, const int *aj, const double *aa, const double *x, const double *z, double *y);
/* point-block Jacobi: pbjacobi.c, baij.c:13-160, dgefa*.c / dgedi.c */
int  orc_block_inverse(int n, double *a);
int  orc_bsr_invert_block_diagonal(int mbs, int bs, const int *ai, const int *aj, const double *aa, double *idiag);
void orc_pbjacobi_apply(int mbs, int bs, const double *idiag, const double *x, double *y);
int  orc_matmult_seqaij(int m, const int *ai, const int *aj, const double *aa, const double *x, const double *z, double *y, int *ns_work);
/* explicit transpose, rows of A^T listing contributions in increasing original-row order */
void orc_csr_transpose(int m, int n, const int *ai, const int *aj, const double *aa, int *ti, int *tj, double *ta);
/* ---- SeqBAIJ (src/mat/impls/baij/seq/baij2.c:331,387,981) ---- */
void orc_spmv_bsr(int mbs, int bs, const int *ai, const int *aj, const double *aa, const double *x, double *y);

/* ---- MPIAIJ set-up, integer work (bit-exact) ---- */
/* Split rank `rank`'s rows [rstart,rend) of a global CSR (sorted columns) into the diagonal block
 * (local column indices) and off-diagonal block (global columns, then compacted through garray):
 * MatSetValues_MPIAIJ column test src/mat/impls/aij/mpi/mpiaij.c:517-560, MatSetUpMultiply_MPIAIJ
 * mmaij.c:9-161.  Caller allocates: ad_i,bo_i [mloc+1]; ad_j,ad_a,bo_j,bo_a [nnz of the row range];
 * garray [nnz bound].  Returns ec (number of ghost columns). */
int orc_mpiaij_split(int rstart, int rend, int cstart, int cend, const int *ai, const int *aj, const double *aa,
                     int *ad_i, int *ad_j, double *ad_a, int *bo_i, int *bo_j, double *bo_a, int *garray);
/* VecScatterCreate_PtoS (src/vec/vec/utils/vpscat.c:1730-1924) for the MPIAIJ pattern
 * (from = garray, to = stride 0..ec-1), evaluated for rank `rank` given every rank's garray.
 * ranges[size+1] = column ownership.  Outputs (caller allocates generously):
 *  recv side ("from"): nrecv procs, rprocs[], rstarts[nrecv+1], rindices[] (slots in lvec)
 *  send side ("to"):   nsend procs, sprocs[], sstarts[nsend+1], sindices[] (local x indices)
 *  local part: nlocal, lto[] (local x idx), lfrom[] (lvec slot). */
void orc_scatter_create(int size, int rank, const int *ranges, const int *const *garrays, const int *ecs,
                        int *nrecv, int *rprocs, int *rstarts, int *rindices,
                        int *nsend, int *sprocs, int *sstarts, int *sindices,
                        int *nlocal, int *lto, int *lfrom);

/* ---- KSP (src/ksp/ksp/impls/{cg/cg.c:92, gmres/gmres.c:118-409 + borthog2.c:35, bcgs/bcgs.c:43}) ---- */
enum { ORC_KSP_CG = 0, ORC_KSP_GMRES = 1, ORC_KSP_BCGS = 2, ORC_KSP_PREONLY = 3, ORC_KSP_GROPPCG = 4, ORC_KSP_PIPECG = 5 };
enum { ORC_PC_NONE = 0, ORC_PC_JACOBI = 1, ORC_PC_BJACOBI = 2, ORC_PC_ILU = 3, ORC_PC_PBJACOBI = 4, ORC_PC_ICC = 5 };
/* ILU(0), natural ordering (src/mat/impls/aij/seq/aijfact.c:1628 symbolic, :461 numeric, :3126 solve); bi[n+1], bj/ba[nz+1], bdiag[n+1] */
int  orc_ilu0_factor(int n, const int *ai, const int *aj, const double *aa, int *bi, int *bj, int *bdiag, double *ba);
int  orc_ilu0_factor_shift(int n, const int *ai, const int *aj, const double *aa, int *bi, int *bj, int *bdiag, double *ba, int *nshift);   /* same; *nshift = restarts MatPivotCheck_nz asked for */
void orc_ilu0_solve(int n, const int *bi, const int *bj, const int *bdiag, const double *ba, const double *b, double *x);
/* MatSolve_SeqAIJ_Inode (inode.c:2327-2760): the factor of a matrix with inodes; ns[] node sizes (orc_check_inode of A) */
void orc_ilu0_solve_inode(int n, int nnodes, const int *ns, const int *bi, const int *bj, const int *bdiag, const double *ba, const double *b, double *x);
/* ICC(0), natural ordering, of the upper triangle of a sequential AIJ matrix (MatICCFactorSymbolic_SeqAIJ with levels 0 +
 * MatCholeskyFactorNumeric_SeqAIJ, aijfact.c:2076-2230,2405-2600): ui[n+1], uj/ua[nnz of the upper triangle incl. diagonal]; row k holds
 * its off-diagonal entries (stored NEGATED and scaled, as the reference leaves them) in column order and then 1/D(k).  Returns the number
 * of positive-definite shifts the factorisation needed (MatPivotCheck_pd), < 0 on failure.  orc_icc0_count: entries of the factor. */
int  orc_icc0_count(int n, const int *ai, const int *aj);
int  orc_icc0_factor(int n, const int *ai, const int *aj, const double *aa, int *ui, int *uj, double *ua);
/* MatSolve_SeqSBAIJ_1_NaturalOrdering (sbaijfact2.c:1977-2015): U^T D sweep forward, U sweep backward */
void orc_icc0_solve(int n, const int *ui, const int *uj, const double *ua, const double *b, double *x);
typedef struct {
  int ksp_type, pc_type;
  double rtol, abstol, dtol;
  int max_it;
  int restart;            /* GMRES(m) */
  int refine_always;      /* -ksp_gmres_cgs_refinement_type refine_always */
  int guess_nonzero;
  /* block Jacobi: nblocks contiguous blocks with boundaries blk[0..nblocks]; sub-solver */
  int nblocks;
  const int *blk;
  int sub_ksp_type, sub_pc_type;
  double sub_rtol, sub_abstol, sub_dtol;
  int sub_max_it;
  int cg_single;          /* -ksp_cg_single_reduction (cg.c:116-122,200-203,263-270) */
  int norm_type;          /* KSPNormType for CG (cg.c:136-161): 0 none, 1 preconditioned (default), 2 unpreconditioned, 3 natural */
  int pb_bs;              /* PCPBJACOBI: the matrix's block size (pbjacobi.c:240: taken from the Mat); n % pb_bs == 0 */
  int pc_right;           /* -ksp_pc_side right (GMRES): KSPInitialResidual itres.c:55-64, PCApplyBAorAB precon.c:617, gmres.c:343-346 */
  /* block Jacobi with sub-solvers set block by block, what a program does through PCBJacobiGetSubKSP (tutorials/ex7.c:173-195): when not
   * NULL, nblocks entries each, replacing sub_ksp_type / sub_pc_type / sub_rtol for that block */
  const int *blk_ksp_type, *blk_pc_type;
  const double *blk_rtol;
} orc_ksp_opts;
void orc_ksp_default_opts(orc_ksp_opts *o);
/* Solves A x = b.  hist[0..] receives the residual norms the monitor would print (hist_cap entries
 * at most); returns iteration count in *its, KSPConvergedReason in *reason. */
int orc_ksp_solve(const orc_ksp_opts *o, int n, const int *ai, const int *aj, const double *aa, const double *b,
                  double *x, double *hist, int hist_cap, int *nhist, int *its, int *reason);

#ifdef __cplusplus
}
#endif
/* cpu_baseline_mt.c: CG + Jacobi with one thread per "rank" (row blocks, per-thread partial sums added in rank order),
 * the arrangement of the reference's MPI run inside one process; `its` iterations from x = 0; returns the seconds they
 * took, x and the last preconditioned residual norm.  Timing aid for bench.py's cpu_baseline. */
double orc_cg_jacobi_mt(int n, const int *ai, const int *aj, const double *aa, const double *b, int its, int nthreads,
                        double *x, double *rnorm_out);

#endif
