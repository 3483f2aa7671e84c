/* Private header of the PLUGIN (libpetschipmi355x.so): the HIPMI355X Vec/Mat/PC implementations.
 *
 * Object model: in a PETSc tree (PETSCHIPMI355X_WITH_PETSC, integration/petsc-3.3/) PETSc's own private headers; on a box
 * without PETSc the harness's stand-ins (petsc-dev_amd/harness/petscimpl.h), which keep the reference's names for the
 * header fields (hdr.comm, hdr.type_name, hdr.state), the layouts (map->n, rmap, cmap), data / spptr, the flags
 * (petscnative, assembled, preallocated) and every function-table slot the types fill, so that the sources in this
 * directory compile against either.  What differs between the two is collected in the HIPOBJ_* / HipComm* names below. */
#ifndef HIPMI355XIMPL_H
#define HIPMI355XIMPL_H
#include "petschipmi355x.h"
#if defined(PETSCHIPMI355X_WITH_PETSC)
#include "hipmi355x_petsc33.h"          /* integration/petsc-3.3/: real petsc-private headers + the names below over MPI */
#else
#include "../harness/petscimpl.h"
#define MPI_Comm PetscComm               /* the plugin's sources use PETSc's spelling */
/* ---- what the plugin needs from "Sys" beyond the public API (all of it exists in libpetsc under these names) ---- */
#define HipCommSize(comm) ((comm)->size)
#define HipCommRank(comm) ((comm)->rank)
/* host collectives of the reference's set-up phase (MPI_Allgather / MPI_Allreduce / persistent send-recv pairs) */
#define HipCommAllgather(comm, sbuf, nbytes, rbuf) ((comm)->allgather((comm)->ctx, (sbuf), (nbytes), (rbuf)))
#define HipCommAllreduce(comm, buf, count, is_double, op) ((comm)->allreduce((comm)->ctx, (buf), (count), (is_double), (op)))
#define HipCommHasExchange(comm) ((comm)->exchange != NULL)
#define HipCommExchange(comm, ns, sp, sb, sbytes, nr, rp, rb, rbytes) ((comm)->exchange((comm)->ctx, (ns), (sp), (sb), (sbytes), (nr), (rp), (rb), (rbytes)))
/* the RCCL communicators hung on the communicator: slot 0 reductions (compute stream), slot 1 halo (halo stream) */
#define HipCommDevice(comm) ((mi355x_comm_t)(comm)->plugin[0])
#define HipCommDeviceHalo(comm) ((mi355x_comm_t)(comm)->plugin[1])
#define HipObjComm(obj) (((PetscObject)(obj))->comm)
#define HipObjTypeName(obj) (((PetscObject)(obj))->type_name)
#define HipObjPrefix(obj) (((PetscObject)(obj))->prefix)
#define HipObjState(obj) (((PetscObject)(obj))->state)
#define HipStateIncrease(obj) PetscObjectStateIncrease(obj)
#define HipFree(p) free(p)
#endif
#include "mi355x_kernels.h"
#include "mi355x_comm.h"

/* map a kernel-library (hipError_t) failure to PETSC_ERR_LIB like CHKERRCUSP, cuspvecimpl.h:79 */
#define CHKHIP(e) do { int e__ = (e); if (e__) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_LIB, "HIP/RCCL error %d: %s", e__, mi355x_comm_error_string(e__)); } while (0)

/* ---- device context of this process (one GPU, two streams) ---- */
typedef struct {
  int initialized, device;
  mi355x_handle_t h;        /* compute stream */
  mi355x_handle_t hcomm;    /* halo stream ("second HIP stream" of the north star) */
} PetscDeviceCtx;
PetscErrorCode PetscDeviceGet(PetscDeviceCtx **ctx);   /* lazily creates handles; PETSC_ERR_LIB without a GPU */

/* ---- off-process entries set with VecSetValues / MatSetValues (harness flavour; inside a PETSc tree the parent classes stash:
 * src/vec/vec/utils/vecstash.c, src/mat/utils/matstash.c).  Entries wait here until the assembly, which hands every rank the
 * entries of all ranks in RANK order (the reference processes messages in arrival order). ---- */
typedef struct { PetscInt n, cap, *i, *j; PetscScalar *v; int mode; /* 0 unset, else InsertMode */ } HipStash;
PetscErrorCode HipStashAdd(HipStash *s, PetscInt i, PetscInt j, PetscScalar v, int mode);
PetscErrorCode HipStashExchange(MPI_Comm comm, HipStash *s, PetscInt *nrecv, PetscInt **ri, PetscInt **rj, PetscScalar **rv, int *mode);
void HipStashFree(HipStash *s);

/* ---- Vec ---- */
/* coherence flags, as PETSC_CUSP_UNALLOCATED/CPU/GPU/BOTH (include/petsc-private/vecimpl.h) */
enum { VALID_NONE = 0, VALID_HOST = 1, VALID_DEVICE = 2, VALID_BOTH = 3 };
typedef struct {
  PetscScalar *host;        /* host mirror, allocated on first host access; FIRST member, as VECHEADER (vecimpl.h:446-449) */
  PetscScalar *dev;         /* HBM */
  int valid;
  PetscScalar *placed_save; /* VecPlaceArray */
  int host_owned;
  PetscScalar *alias_save; int alias_valid;   /* "VecShareArrayBegin_C": the storage this vector owns while it borrows another's */
  HipStash stash;           /* off-process VecSetValues (harness flavour) */
} Vec_HIPMI355X;

PetscErrorCode VecCreate_SeqHIPMI355X(Vec v);
PetscErrorCode VecCreate_MPIHIPMI355X(Vec v);
PetscErrorCode VecCreate_HIPMI355X(Vec v);
PetscErrorCode VecCreateSeqHIPMI355X(MPI_Comm comm, PetscInt n, Vec *v);
/* device pointers with coherence (analogue of VecCUSPGetArrayRead/Write, cuspvecimpl.h:95-150) */
PetscErrorCode VecHIPGetRead(Vec v, const PetscScalar **d);
PetscErrorCode VecHIPGetWrite(Vec v, PetscScalar **d);       /* contents will be overwritten */
PetscErrorCode VecHIPGetReadWrite(Vec v, PetscScalar **d);
PetscErrorCode VecHIPRestoreWrite(Vec v);                    /* device newer; state++ */
#define PETSC_HIP_DPI_SLOT 8   /* device scratch slot holding p'w between the dot (or the SpMV by-product) and the CG update */
PetscErrorCode MatMultTDotBegin_HIPMI355X(Mat A, Vec x, Vec y, PetscBool *ok);   /* y = A x, x'y left on the device */
PetscErrorCode MatMultDiagonalScale_HIPMI355X(Mat A, Vec d, Vec x, Vec y, PetscBool *ok);   /* y = d .* (A x) */

/* ---- VecScatter (VecScatter_MPI_General, include/petsc-private/vecimpl.h:509-555) ---- */
typedef struct {
  PetscInt n;                 /* number of neighbours */
  PetscInt *procs, *starts, *indices;
  PetscInt *d_indices;        /* device copy */
  PetscScalar *d_values;      /* device message buffer */
  PetscBool contiq;           /* indices contiguous: exchange in place (vpscat.c:1951-1960) */
  PetscInt local_n;
  PetscInt *local_slots, *d_local_slots;
} VecScatterSide;
/* the plugin's own scatter context: device index lists, message buffers, events.  On the harness it IS the VecScatter
 * (there is no other); inside a PETSc tree it lives beside the reference's Mvctx, which MatMult no longer uses. */
#if !defined(PETSCHIPMI355X_WITH_PETSC)
typedef struct _p_VecScatter *HipScatter;
struct _p_VecScatter {
#else
typedef struct _p_HipScatter *HipScatter;
struct _p_HipScatter {
#endif
  MPI_Comm comm;
  VecScatterSide to, from;
  PetscBool inuse;            /* guard, vscat.c:1637 */
  mi355x_event_t ev_packed, ev_done;
  int device_ready;
  int ready_marked;           /* ev_packed already recorded by VecScatterMarkReady */
  PetscScalar *h_send, *h_recv;   /* host staging buffers of the host-staged transport */
  void *nbr_work;                 /* per-neighbour work arrays of an exchange (pointers, counts, ranks, byte counts): sized for max(to.n, from.n) */
  PetscScalar *d_local_tmp;       /* device staging of the local (self) part, local_n doubles */
  /* per-exchange device timing for bench.py: event pairs on the HALO stream around each forward exchange (first operation after the
   * wait on "x is final" .. last operation before "halo done") */
  PetscBool timing; PetscInt time_n, time_cap; mi355x_event_t *time_ev;
};
PetscErrorCode HipScatterCreate_PtoS_MPIAIJ(MPI_Comm comm, PetscLayout xmap, PetscInt ec, const PetscInt *garray, HipScatter *ctx);
PetscErrorCode HipScatterMarkReady(HipScatter ctx, Vec x);
PetscErrorCode HipScatterBegin(HipScatter ctx, Vec x, Vec y, InsertMode addv, ScatterMode mode);
PetscErrorCode HipScatterEnd(HipScatter ctx, Vec x, Vec y, InsertMode addv, ScatterMode mode);
PetscErrorCode HipScatterDestroy(HipScatter *ctx);
PetscErrorCode HipScatterSetTiming(HipScatter ctx, PetscBool on);
PetscErrorCode HipScatterGetLists(HipScatter ctx, PetscInt *nrecv, const PetscInt **rprocs, const PetscInt **rstarts, const PetscInt **rindices,
                                  PetscInt *nsend, const PetscInt **sprocs, const PetscInt **sstarts, const PetscInt **sindices,
                                  PetscInt *nlocal, const PetscInt **lto, const PetscInt **lfrom);

/* ---- Mat ---- */
/* host CSR container: the part of Mat_SeqAIJ the path needs (src/mat/impls/aij/seq/aij.h:10-39,99-115), same member
 * names.  On the harness it IS the container (A->data) and aijhip.c assembles into it; inside a PETSc tree the parent
 * type MATSEQAIJ owns the container (aij.h) and this struct is a VIEW of its arrays, refreshed at MatAssemblyEnd and
 * kept in the device mirror (integration/petsc-3.3/aijhipmi355x.c) -- the kernels' callers read the same fields. */
typedef struct {
  PetscInt m, n;            /* local rows / columns */
  PetscInt *i, *j;          /* row pointer / column index */
  PetscScalar *a;
  PetscInt *ilen, *imax;    /* used / allocated per row during assembly */
  PetscInt nz, maxnz;
  PetscInt bs;              /* block size (BAIJ reuse: i,j index blocks, a holds bs*bs per block) */
  PetscBool compact;        /* rows are packed (after assembly) */
  PetscInt nonzerorows;
  PetscInt inode_count, *inode_size;   /* Mat_SeqAIJ_Inode node_count / size (aij.h:99-115); 0 / NULL: plain routines */
} HipAIJ;

/* ---- factored matrices: what MatGetFactor(A, "petsc", MAT_FACTOR_ILU | MAT_FACTOR_ICC, &F) hangs on F (host/ilu.c, host/icc.c;
 * the role of Mat_SeqAIJCUSPARSETriFactors, src/mat/impls/aij/seq/seqcusparse/cusparsematimpl.h) ---- */
typedef struct {
  MatFactorType kind;                                      /* MAT_FACTOR_ILU or MAT_FACTOR_ICC */
  PetscInt n, nz;
  PetscInt *bi, *bj, *bdiag; PetscScalar *ba;              /* ILU(0): host factor in the reference's L / reversed-U layout (aijfact.c:1628-1700) */
  PetscBool owns_host;                                     /* harness: ours; inside PETSc: the arrays of F's own Mat_SeqAIJ */
  PetscInt *d_bi, *d_bj, *d_bdiag; PetscScalar *d_ba;      /* device copies for the level-scheduled ILU kernels */
  PetscInt nlevL, nlevU, *levptrL, *levptrU;               /* dependency levels (host) */
  PetscInt *rlevL, *rlevU;                                 /* level of every row of L / U, left by the host factorisation for the solves' analysis (else NULL) */
  PetscInt *d_rowsL, *d_rowsU;                             /* rows ordered by level (device) */
  PetscScalar *d_work; void *graph; int graph_tried;       /* hipGraph of the level launches working in place on d_work */
  mi355x_trisolve_plan_t tri_lo, tri_up;                   /* sync-free solves (NULL: level launches) */
  int by_level;                                            /* rows summed in dependency-level order (inode matrices) instead of column order */
  int block_columns;                                       /* node plans whose columns are whole dependency nodes */
  PetscInt nodes, nlevL_nodes, nlevU_nodes;                /* node-blocked plans (the factor of a matrix with inodes): nodes and their levels; 0: row-granular */
  int use_levels, aborted;                                 /* a sync-free application gave up: the same plans run level by level from now on */
  PetscInt nshift;                                         /* restarts / shifts the factorisation took (largest count over the blocks) */
  int factored_state; void *factored_of;                   /* operator and operator state of the last numeric factorisation */
  PetscInt nblk, *blk;                                     /* "MatFactorSetIndependentBlocks_C": the matrix stands for that many separate matrices (row ranges) */
  PetscErrorCode (*parent_destroy)(Mat);                   /* inside PETSc: the parent factor matrix's destroy */
} HipTriFactors;
PetscErrorCode HipTriFactorsDestroy(HipTriFactors **f);
PetscErrorCode HipTriFactorsApply(Mat F, HipTriFactors *f, Vec b, Vec x, PetscLogDouble flops);
typedef void (*HipRangeFn)(void *ctx, PetscInt lo, PetscInt hi);
void HipParallelRanges(PetscInt n, HipRangeFn fn, void *ctx);   /* fn over contiguous parts of [0, n) on up to 16 host threads (hipsys.c); one thread below 200 000 */
int HipHostThreads(int cap);                                    /* host threads this RANK may use for set-up passes: affinity mask, cgroup quota, ranks on the node (hipsys.c) */
typedef PetscErrorCode (*HipProductNowFn)(Mat A, Vec x, Vec t);                           /* t = A x, launched now */
typedef PetscErrorCode (*HipProductScaledFn)(Mat A, Vec d, Vec x, Vec w, PetscBool *ok);   /* w = d .* (A x) in one kernel */
PetscErrorCode VecHIPNoteProduct(Mat A, Vec x, Vec t, HipProductNowFn now, HipProductScaledFn scaled, PetscBool *noted);   /* host/vechip.c, "a noted product" */
PetscErrorCode VecHIPProductMatrixChanges(Mat A);              /* to be called before A's device values change or go away */
PetscErrorCode VecHIPMI355XFlushDeferred(void);                /* host/vechip.c: run the noted element-wise operations */
PetscErrorCode VecHIPMI355XSetDeferral(PetscInt on);
void HipFactorJoinHelpers(void);                              /* host/ilu.c: the thread that returns the factorisation's work arrays */
PetscErrorCode HipTriWatchCheck(void);                     /* at every host wait: did a sync-free solve queued earlier give up? */
void HipTriWatchAdd(HipTriFactors *f);
PetscErrorCode MatICCFactorSymbolic_SeqAIJHIP(Mat F, Mat A, IS perm, const MatFactorInfo *info);
PetscErrorCode MatGetFactor_seqaijhipmi355x_petsc(Mat A, MatFactorType ftype, Mat *B);
PetscErrorCode MatGetFactorAvailable_seqaijhipmi355x_petsc(Mat A, MatFactorType ftype, PetscBool *flg);

/* device mirror */
typedef struct {
  PetscInt *d_i, *d_j;
  PetscScalar *d_a;
  mi355x_spmv_plan_t plan;
  int uploaded_state;        /* Mat state at last upload (compare state instead of valid_GPU_matrix) */
  /* compressed-row form for mostly-empty blocks (src/mat/utils/compressedrow.c:28) */
  PetscBool cprow;            /* compressed-row form requested (off-diagonal block) */
  PetscBool baij4_mfma;       /* BAIJ bs = 4: MatMult on the matrix cores (mi355x_spmv_bsr4_mfma) */
  PetscInt pattern_nz;        /* nz of the pattern the mirror was built for (-1: none) */
  /* cached explicit transpose for MatMultTranspose */
  PetscInt *t_i, *t_j; PetscScalar *t_a; mi355x_spmv_plan_t t_plan; int t_state;
  PetscInt *t_perm, t_pattern_nz;   /* device: position in A^T -> position in A (values follow by one gather when only values change) */
  PetscInt t_builds, t_refreshes;   /* host builds / device refreshes so far */
  PetscInt n_uploads;        /* value uploads so far */
  /* MatSetValuesBatch map for one connectivity (rows array): contributions grouped by nonzero, in call order */
  PetscInt bm_nb, bm_bs, bm_nseg; unsigned long long bm_hash; size_t bm_T, bm_vcap;
  PetscInt *bm_order, *bm_segptr, *bm_segslot;   /* device */
  PetscScalar *bm_v;                             /* device staging of the element values */
  /* column-tiled form of the product (csrc/spmv_tiled.hip: x staged in LDS) for matrices whose gathers miss the caches; NULL: the
   * row-block kernels.  tiled_fresh: its values are those of d_a */
  mi355x_spmv_tiled_t tiled; PetscBool tiled_fresh;
  mi355x_spmv_tiled_t t_tiled;        /* the column-tiled form of the cached transpose (built when the matrix itself took that form) */
  /* the blocked companion of an AIJ matrix whose nodes are complete bs x bs blocks (3-dof FEM matrices): BCSR arrays on the device, values
   * a permutation gather of d_a, multiplied by the BAIJ row-block kernel (host/aijhip.c, "blocked companion") */
  PetscInt *b_i, *b_j, *b_perm, b_bs, b_nblocks; PetscScalar *b_a; mi355x_spmv_plan_t b_plan; PetscBool b_fresh;
  PetscInt *tb_i, *tb_j, *tb_perm; PetscScalar *tb_a; mi355x_spmv_plan_t tb_plan; PetscBool tb_fresh;   /* ... and its block transpose, for the transpose products */
  PetscInt opt[8]; PetscBool opt_set[8];   /* the type's options as MatSetFromOptions read them under the matrix's prefix (host/aijhip.c) */
  /* per-launch device timing for bench.py (hipEvent pairs on the compute stream) */
  PetscBool timing; PetscInt time_n, time_cap; mi355x_event_t *time_ev;
#if defined(PETSCHIPMI355X_WITH_PETSC)
  HipAIJ view;               /* of the parent's Mat_SeqAIJ (or Mat_SeqBAIJ: baij_parent) */
  PetscBool baij_parent;
#else
  HipTriFactors *tri;        /* a factored matrix (A->factortype != MAT_FACTOR_NONE): its triangular factors on the device */
#endif
} Mat_SeqAIJHIP;
/* where a factored matrix keeps its device-side factors: inside PETSc the factor is the PARENT's matrix (MATSEQAIJ for ILU, MATSEQSBAIJ
 * for ICC, from MatGetFactor_seqaij_petsc) whose spptr is free; on the harness it is a matrix of this type */
#if defined(PETSCHIPMI355X_WITH_PETSC)
#define HipTriGet(F) ((HipTriFactors *)(F)->spptr)
#else
#define HipTriGet(F) (((Mat_SeqAIJHIP *)(F)->spptr)->tri)
#endif
#if defined(PETSCHIPMI355X_WITH_PETSC)
#define HipAIJGet(A) (&((Mat_SeqAIJHIP *)(A)->spptr)->view)
#else
#define HipAIJGet(A) ((HipAIJ *)(A)->data)
#endif

/* the members of Mat_MPIAIJ the path needs (src/mat/impls/aij/mpi/mpiaij.h:35-77), same names, with the plugin's scatter
 * in place of Mvctx.  On the harness it IS the container (A->data); inside a PETSc tree the parent MATMPIAIJ owns the
 * container and this struct hangs off A->spptr with A, B, garray aliasing the parent's after each assembly. */
typedef struct {
  Mat A, B;                 /* diagonal / off-diagonal blocks (SeqAIJHIPMI355X) */
  PetscInt *garray, ec;
  Vec lvec;
  HipScatter hscat;
  PetscInt rstart, rend, cstart, cend;
  HipStash stash;           /* off-process MatSetValues (harness flavour) */
} HipMPIAIJ;
#if defined(PETSCHIPMI355X_WITH_PETSC)
#define HipMPIAIJGet(A) ((HipMPIAIJ *)(A)->spptr)
#else
#define HipMPIAIJGet(A) ((HipMPIAIJ *)(A)->data)
#endif

PetscErrorCode MatCreate_SeqAIJHIPMI355X(Mat);
PetscErrorCode MatCreate_MPIAIJHIPMI355X(Mat);
PetscErrorCode MatCreate_AIJHIPMI355X(Mat);
PetscErrorCode MatCreate_SeqBAIJHIPMI355X(Mat);
PetscErrorCode MatSeqAIJHIPUpload(Mat A);
PetscErrorCode MatSeqAIJHIPGetInodes(Mat A, PetscInt *count, const PetscInt **sizes);
PetscErrorCode MatSeqAIJHIPSetCompressedRow(Mat A, PetscBool flg);
PetscErrorCode MatSetUpMultiply_MPIAIJ(Mat mat);
PetscErrorCode MatTimingBegin(Mat A, mi355x_handle_t h);
PetscErrorCode MatTimingEnd(Mat A, mi355x_handle_t h);
PetscErrorCode MatGetVecs_HIPMI355X(Mat A, Vec *right, Vec *left);
PetscErrorCode PCCreate_PBJacobi_HIPMI355X(PC);
PetscErrorCode KSPCreate_CGHIPMI355X(KSP);
PetscErrorCode KSPCreate_GMRESHIPMI355X(KSP);
PetscErrorCode KSPCreate_BCGSHIPMI355X(KSP);

#endif
