/*
 * petschipmi355x.h -- C host side of the MI355X-native Krylov hot path.
 *
 * It mirrors, name for name and argument for argument, the slice of PETSc's public interface
 * (erdc/petsc-dev 3.3.0-dev: include/petscvec.h, petscmat.h, petscksp.h, petscpc.h) that the path
 * BASELINE.json names runs through, and provides the new Vec/Mat implementations
 *     VECSEQHIPMI355X / VECMPIHIPMI355X / VECHIPMI355X
 *     MATSEQAIJHIPMI355X / MATMPIAIJHIPMI355X / MATAIJHIPMI355X , MATSEQBAIJHIPMI355X
 * behind the same per-object function tables (struct _VecOps include/petsc-private/vecimpl.h:221-294,
 * struct _MatOps include/petsc-private/matimpl.h:17-188) and string-keyed type registry
 * (VecSetType src/vec/vec/interface/vecreg.c, MatSetType src/mat/interface/matreg.c:42-82) the
 * reference uses, so that KSPSolve_CG / _GMRES / _BCGS and PCJACOBI / PCBJACOBI run over them
 * unchanged.  Inside a real PETSc tree the same Create functions are registered with
 * VecRegister()/MatRegister(); here the minimal object model those functions need
 * is supplied by this library so that the path is self-contained on a box without PETSc.
 *
 * All compute goes to the HIP kernels of mi355x_kernels.h.  There is no CPU compute path:
 * every op fails with PETSC_ERR_LIB if no gfx950 device is available.
 */
#ifndef PETSCHIPMI355X_H
#define PETSCHIPMI355X_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef int    PetscErrorCode;   /* include/petscsys.h:123 */
typedef int    PetscInt;         /* 32-bit indices, include/petscsys.h:188 */
typedef double PetscScalar;      /* real double, include/petscmath.h:198 */
typedef double PetscReal;
typedef int    PetscBool;
typedef int    PetscMPIInt;
typedef double PetscLogDouble;
#define PETSC_TRUE 1
#define PETSC_FALSE 0
#define PETSC_DECIDE (-1)
#define PETSC_DETERMINE PETSC_DECIDE
#define PETSC_DEFAULT (-2)
#define PETSC_NULL 0

/* error codes: include/petscerror.h:41-84 */
#define PETSC_ERR_MEM 55
#define PETSC_ERR_SUP 56
#define PETSC_ERR_ORDER 58
#define PETSC_ERR_ARG_SIZ 60
#define PETSC_ERR_ARG_IDN 61
#define PETSC_ERR_ARG_WRONG 62
#define PETSC_ERR_ARG_OUTOFRANGE 63
#define PETSC_ERR_ARG_CORRUPT 64
#define PETSC_ERR_ARG_NOTSAMETYPE 69
#define PETSC_ERR_FP 72
#define PETSC_ERR_ARG_WRONGSTATE 73
#define PETSC_ERR_ARG_INCOMP 75
#define PETSC_ERR_LIB 76
#define PETSC_ERR_USER 83
#define PETSC_ERR_PLIB 77
#define PETSC_ERR_ARG_NULL 85
#define PETSC_ERR_ARG_UNKNOWN_TYPE 86
#define PETSC_ERR_ARG_TYPENOTSET 89
#define PETSC_ERR_NOT_CONVERGED 91

typedef enum { NOT_SET_VALUES, INSERT_VALUES, ADD_VALUES, MAX_VALUES } InsertMode;          /* petscsys.h */
typedef enum { SCATTER_FORWARD = 0, SCATTER_REVERSE = 1 } ScatterMode;                         /* petscvec.h:42 */
typedef enum { NORM_1 = 0, NORM_2 = 1, NORM_FROBENIUS = 2, NORM_INFINITY = 3, NORM_1_AND_2 = 4 } NormType; /* petscvec.h:155 */
typedef enum { MAT_FLUSH_ASSEMBLY = 1, MAT_FINAL_ASSEMBLY = 0 } MatAssemblyType;               /* petscmat.h:347 */
typedef enum { DIFFERENT_NONZERO_PATTERN, SUBSET_NONZERO_PATTERN, SAME_NONZERO_PATTERN, SAME_PRECONDITIONER } MatStructure;
typedef enum { PC_SIDE_DEFAULT = -1, PC_LEFT, PC_RIGHT, PC_SYMMETRIC } PCSide;
typedef enum { KSP_NORM_DEFAULT = -1, KSP_NORM_NONE = 0, KSP_NORM_PRECONDITIONED = 1, KSP_NORM_UNPRECONDITIONED = 2, KSP_NORM_NATURAL = 3 } KSPNormType;
typedef enum { KSP_GMRES_CGS_REFINE_NEVER, KSP_GMRES_CGS_REFINE_IFNEEDED, KSP_GMRES_CGS_REFINE_ALWAYS } KSPGMRESCGSRefinementType;
typedef enum { /* petscksp.h:403-430 */
  KSP_CONVERGED_RTOL_NORMAL = 1, KSP_CONVERGED_ATOL_NORMAL = 9, KSP_CONVERGED_RTOL = 2, KSP_CONVERGED_ATOL = 3,
  KSP_CONVERGED_ITS = 4, KSP_CONVERGED_HAPPY_BREAKDOWN = 8,
  KSP_DIVERGED_NULL = -2, KSP_DIVERGED_ITS = -3, KSP_DIVERGED_DTOL = -4, KSP_DIVERGED_BREAKDOWN = -5,
  KSP_DIVERGED_BREAKDOWN_BICG = -6, KSP_DIVERGED_NONSYMMETRIC = -7, KSP_DIVERGED_INDEFINITE_PC = -8,
  KSP_DIVERGED_NAN = -9, KSP_DIVERGED_INDEFINITE_MAT = -10, KSP_CONVERGED_ITERATING = 0
} KSPConvergedReason;

typedef struct _p_PetscComm  *MPI_Comm;     /* stands in for MPI_Comm */
typedef struct _p_Vec        *Vec;
typedef struct _p_Mat        *Mat;
typedef struct _p_VecScatter *VecScatter;
typedef struct _p_KSP        *KSP;
typedef struct _p_PC         *PC;
typedef struct _p_PetscViewer *PetscViewer;
typedef enum { FILE_MODE_READ, FILE_MODE_WRITE } PetscFileMode;
typedef const char *VecType;
typedef const char *MatType;
typedef const char *KSPType;
typedef const char *PCType;

#define VECSEQHIPMI355X     "seqhipmi355x"
#define VECMPIHIPMI355X     "mpihipmi355x"
#define VECHIPMI355X        "hipmi355x"
#define MATSEQAIJHIPMI355X  "seqaijhipmi355x"
#define MATMPIAIJHIPMI355X  "mpiaijhipmi355x"
#define MATAIJHIPMI355X     "aijhipmi355x"
#define MATSEQBAIJHIPMI355X "seqbaijhipmi355x"
#define KSPCG      "cg"
#define KSPGMRES   "gmres"
#define KSPBCGS    "bcgs"
#define KSPPREONLY "preonly"
#define KSPGROPPCG "groppcg"   /* Gropp's overlapped CG (src/ksp/ksp/impls/cg/groppcg/groppcg.c), SURVEY 8f.4 */
#define PCNONE     "none"
#define PCJACOBI   "jacobi"
#define PCBJACOBI  "bjacobi"
#define PCILU      "ilu"       /* ILU(0), natural ordering, sequential AIJ */

/* ---- Sys ----------------------------------------------------------------------------------- */
extern MPI_Comm PETSC_COMM_SELF, PETSC_COMM_WORLD;
/* registers the HIPMI355X types (the analogue of a PetscDLLibraryRegister entry, src/sys/dll) and
 * binds this process to GPU `device` (-1: LOCAL_RANK env or 0).  Safe to call without a GPU: type
 * registration and all host-side set-up work; the first device op reports PETSC_ERR_LIB. */
PetscErrorCode PetscHIPMI355XInitialize(int device);
PetscErrorCode PetscHIPMI355XFinalize(void);
PetscErrorCode PetscHIPMI355XRegisterAll(void);
const char    *PetscHIPMI355XVersion(void);
/* last error message (PetscError traceback text, src/sys/error/err.c) */
const char    *PetscGetLastErrorMessage(void);
/* Host collectives supplied by the launcher (torch.distributed / MPI): the reference uses MPI for these */
typedef int (*PetscCommAllgatherFn)(void *ctx, const void *sendbuf, int nbytes, void *recvbuf);
typedef int (*PetscCommAllreduceFn)(void *ctx, void *buf, int count, int is_double, int op /*0 sum,1 max,2 min*/);
typedef int (*PetscCommBarrierFn)(void *ctx);
PetscErrorCode PetscCommCreate(int rank, int size, void *ctx, PetscCommAllgatherFn, PetscCommAllreduceFn, PetscCommBarrierFn, MPI_Comm *comm);
/* optional host-staged neighbour exchange (used only when no RCCL communicator is attached, e.g. several ranks
 * sharing one GPU): post every receive and send of one halo exchange, return when all have completed */
typedef int (*PetscCommExchangeFn)(void *ctx, int nsend, const int *speers, void *const *sbufs, const int *sbytes,
                                   int nrecv, const int *rpeers, void *const *rbufs, const int *rbytes);
PetscErrorCode PetscCommSetExchange(MPI_Comm comm, PetscCommExchangeFn fn);
PetscErrorCode PetscCommSetWorld(MPI_Comm comm);
/* attach the RCCL communicator (mi355x_comm.h) used for device-side halo exchange and reductions */
PetscErrorCode PetscCommSetDeviceComm(MPI_Comm comm, void *mi355x_comm);
/* one communicator per HIP stream: `reduce` serves the all-reduces queued on the compute stream, `halo` the grouped
 * send/recv (and split-phase all-reduces) queued on the halo stream, so that RCCL does not serialise the two */
PetscErrorCode PetscCommSetDeviceComms(MPI_Comm comm, void *reduce, void *halo);
PetscErrorCode PetscCommGetDeviceTransport(MPI_Comm comm, int *kind /*0 single,1 RCCL,2 host-staged*/, int *nranks, int *distinct_halo_comm);
PetscErrorCode PetscCommDestroy(MPI_Comm *comm);
PetscErrorCode MPI_Comm_rank(MPI_Comm comm, PetscMPIInt *rank);
PetscErrorCode MPI_Comm_size(MPI_Comm comm, PetscMPIInt *size);
/* flop counter fed by PetscLogFlops with the reference's formulas (include/petsclog.h:294) */
PetscErrorCode PetscGetFlops(PetscLogDouble *flops);
PetscErrorCode PetscSplitOwnership(MPI_Comm comm, PetscInt *n, PetscInt *N);   /* src/sys/utils/psplit.c */
/* options database subset (src/sys/objects/options.c): "-ksp_type cg -pc_type jacobi ..." */
PetscErrorCode PetscOptionsInsertString(const char *str);
PetscErrorCode PetscOptionsSetValue(const char *name, const char *value);
PetscErrorCode PetscOptionsClear(void);

/* ---- Vec (include/petscvec.h; wrappers src/vec/vec/interface/rvector.c) ---------------------- */
PetscErrorCode VecCreate(MPI_Comm comm, Vec *vec);
PetscErrorCode VecSetSizes(Vec v, PetscInt n, PetscInt N);
PetscErrorCode VecSetType(Vec v, VecType type);
PetscErrorCode VecSetFromOptions(Vec v);                       /* -vec_type */
PetscErrorCode VecGetType(Vec v, VecType *type);
PetscErrorCode VecDuplicate(Vec v, Vec *newv);
PetscErrorCode VecDuplicateVecs(Vec v, PetscInt m, Vec **V);
PetscErrorCode VecDestroyVecs(PetscInt m, Vec **V);
PetscErrorCode VecDestroy(Vec *v);
PetscErrorCode VecGetSize(Vec v, PetscInt *N);
PetscErrorCode VecGetLocalSize(Vec v, PetscInt *n);
PetscErrorCode VecGetOwnershipRange(Vec v, PetscInt *low, PetscInt *high);
PetscErrorCode VecSetValues(Vec v, PetscInt ni, const PetscInt ix[], const PetscScalar y[], InsertMode mode);
PetscErrorCode VecAssemblyBegin(Vec v);
PetscErrorCode VecAssemblyEnd(Vec v);
PetscErrorCode VecGetArray(Vec v, PetscScalar **a);            /* host pointer; syncs device -> host */
PetscErrorCode VecRestoreArray(Vec v, PetscScalar **a);        /* marks host copy newer */
PetscErrorCode VecGetArrayRead(Vec v, const PetscScalar **a);
PetscErrorCode VecRestoreArrayRead(Vec v, const PetscScalar **a);
PetscErrorCode VecPlaceArray(Vec v, const PetscScalar *a);
PetscErrorCode VecResetArray(Vec v);
/* device pointer access (the analogue of VecCUSPGetArrayReadWrite, cuspvecimpl.h:95) */
PetscErrorCode VecHIPMI355XGetArray(Vec v, PetscScalar **d);
PetscErrorCode VecHIPMI355XRestoreArray(Vec v, PetscScalar **d);
PetscErrorCode VecHIPMI355XGetArrayRead(Vec v, const PetscScalar **d);
PetscErrorCode VecSet(Vec x, PetscScalar alpha);
PetscErrorCode VecCopy(Vec x, Vec y);
PetscErrorCode VecSwap(Vec x, Vec y);
PetscErrorCode VecScale(Vec x, PetscScalar alpha);
PetscErrorCode VecAXPY(Vec y, PetscScalar alpha, Vec x);
PetscErrorCode VecAYPX(Vec y, PetscScalar alpha, Vec x);
PetscErrorCode VecAXPBY(Vec y, PetscScalar alpha, PetscScalar beta, Vec x);
PetscErrorCode VecWAXPY(Vec w, PetscScalar alpha, Vec x, Vec y);
PetscErrorCode VecAXPBYPCZ(Vec z, PetscScalar alpha, PetscScalar beta, PetscScalar gamma, Vec x, Vec y);
PetscErrorCode VecMAXPY(Vec y, PetscInt nv, const PetscScalar alpha[], Vec x[]);
PetscErrorCode VecPointwiseMult(Vec w, Vec x, Vec y);
PetscErrorCode VecPointwiseDivide(Vec w, Vec x, Vec y);
PetscErrorCode VecReciprocal(Vec x);
PetscErrorCode VecDot(Vec x, Vec y, PetscScalar *val);
PetscErrorCode VecTDot(Vec x, Vec y, PetscScalar *val);
PetscErrorCode VecMDot(Vec x, PetscInt nv, const Vec y[], PetscScalar val[]);
PetscErrorCode VecMTDot(Vec x, PetscInt nv, const Vec y[], PetscScalar val[]);
PetscErrorCode VecNorm(Vec x, NormType type, PetscReal *val);
PetscErrorCode VecNormalize(Vec x, PetscReal *val);
PetscErrorCode VecDotNorm2(Vec s, Vec t, PetscScalar *dp, PetscReal *nm);
/* split-phase reductions (src/vec/vec/utils/comb.c:402-721); the all-reduce runs on the halo stream between Begin and End */
PetscErrorCode VecDotBegin(Vec x, Vec y, PetscScalar *result);
PetscErrorCode VecDotEnd(Vec x, Vec y, PetscScalar *result);
PetscErrorCode VecNormBegin(Vec x, NormType type, PetscReal *result);   /* NORM_2 */
PetscErrorCode VecNormEnd(Vec x, NormType type, PetscReal *result);
PetscErrorCode PetscCommSplitReductionBegin(MPI_Comm comm);

/* ---- VecScatter (parallel -> sequential general, the MPIAIJ halo) ----------------------------- */
PetscErrorCode VecScatterBegin(VecScatter ctx, Vec x, Vec y, InsertMode addv, ScatterMode mode);
PetscErrorCode VecScatterEnd(VecScatter ctx, Vec x, Vec y, InsertMode addv, ScatterMode mode);
PetscErrorCode VecScatterDestroy(VecScatter *ctx);
/* introspection of the index lists (to/from of VecScatter_MPI_General, vecimpl.h:509-537) for parity tests */
PetscErrorCode VecScatterGetLists(VecScatter ctx, PetscInt *nrecv, const PetscInt **rprocs, const PetscInt **rstarts, const PetscInt **rindices,
                                  PetscInt *nsend, const PetscInt **sprocs, const PetscInt **sstarts, const PetscInt **sindices,
                                  PetscInt *nlocal, const PetscInt **lto, const PetscInt **lfrom);

/* ---- Mat (include/petscmat.h; wrappers src/mat/interface/matrix.c) --------------------------- */
PetscErrorCode MatCreate(MPI_Comm comm, Mat *A);
PetscErrorCode MatSetSizes(Mat A, PetscInt m, PetscInt n, PetscInt M, PetscInt N);
PetscErrorCode MatSetType(Mat A, MatType type);
PetscErrorCode MatSetFromOptions(Mat A);                      /* -mat_type */
PetscErrorCode MatGetType(Mat A, MatType *type);
PetscErrorCode MatSetUp(Mat A);
PetscErrorCode MatSeqAIJSetPreallocation(Mat A, PetscInt nz, const PetscInt nnz[]);
PetscErrorCode MatMPIAIJSetPreallocation(Mat A, PetscInt d_nz, const PetscInt d_nnz[], PetscInt o_nz, const PetscInt o_nnz[]);
PetscErrorCode MatSetValues(Mat A, PetscInt m, const PetscInt idxm[], PetscInt n, const PetscInt idxn[], const PetscScalar v[], InsertMode addv);
PetscErrorCode MatSetValuesBatch(Mat A, PetscInt nb, PetscInt bs, PetscInt rows[], const PetscScalar v[]);   /* matrix.c:1698; device-side value assembly when the pattern is unchanged */
PetscErrorCode MatAssemblyBegin(Mat A, MatAssemblyType type);
PetscErrorCode MatAssemblyEnd(Mat A, MatAssemblyType type);
/* bulk creation from CSR (MatCreateSeqAIJWithArrays src/mat/impls/aij/seq/aij.c, MatCreateMPIAIJWithArrays
 * src/mat/impls/aij/mpi/mpiaij.c; i/j/a are copied; j holds global column indices, ascending per row) */
PetscErrorCode MatCreateSeqAIJWithArrays(MPI_Comm comm, PetscInt m, PetscInt n, PetscInt i[], PetscInt j[], PetscScalar a[], Mat *mat);
PetscErrorCode MatCreateMPIAIJWithArrays(MPI_Comm comm, PetscInt m, PetscInt n, PetscInt M, PetscInt N, const PetscInt i[], const PetscInt j[], const PetscScalar a[], Mat *mat);
PetscErrorCode MatCreateSeqBAIJWithArrays(MPI_Comm comm, PetscInt bs, PetscInt m, PetscInt n, PetscInt i[], PetscInt j[], PetscScalar a[], Mat *mat);
PetscErrorCode MatDestroy(Mat *A);
PetscErrorCode MatGetSize(Mat A, PetscInt *M, PetscInt *N);
PetscErrorCode MatGetLocalSize(Mat A, PetscInt *m, PetscInt *n);
PetscErrorCode MatGetOwnershipRange(Mat A, PetscInt *rstart, PetscInt *rend);
PetscErrorCode MatGetVecs(Mat A, Vec *right, Vec *left);
PetscErrorCode MatMult(Mat A, Vec x, Vec y);
PetscErrorCode MatMultAdd(Mat A, Vec x, Vec y, Vec z);
PetscErrorCode MatMultTranspose(Mat A, Vec x, Vec y);
PetscErrorCode MatMultTransposeAdd(Mat A, Vec x, Vec y, Vec z);
PetscErrorCode MatGetDiagonal(Mat A, Vec d);
PetscErrorCode MatScale(Mat A, PetscScalar a);
PetscErrorCode MatZeroEntries(Mat A);
PetscErrorCode MatDiagonalScale(Mat A, Vec l, Vec r);   /* A <- diag(l) A diag(r); l or r may be NULL (aij.c:2055, mpiaij.c:2183) */
/* introspection for parity tests: host CSR of a SeqAIJ block; MPIAIJ pieces (Mat_MPIAIJ, mpiaij.h:35-77) */
PetscErrorCode MatSeqAIJGetArrays(Mat A, PetscInt *m, const PetscInt **i, const PetscInt **j, const PetscScalar **a);
PetscErrorCode MatMPIAIJGetSeqAIJ(Mat A, Mat *Ad, Mat *Ao, const PetscInt **garray);
PetscErrorCode MatMPIAIJGetScatter(Mat A, VecScatter *ctx, Vec *lvec, PetscInt *ec);
/* per-MatMult device timing (HIP events on the compute stream around the SpMV launches) for bench.py */
PetscErrorCode MatHIPMI355XSetTiming(Mat A, PetscBool on);
PetscErrorCode MatHIPMI355XGetTiming(Mat A, PetscInt *nlaunches, PetscLogDouble *total_ms);
PetscErrorCode MatHIPMI355XGetUploadCount(Mat A, PetscInt *n);   /* value uploads host -> device of a sequential matrix so far */
PetscErrorCode MatHIPMI355XGetInodeInfo(Mat A, PetscInt *nodes, PetscInt *groups, PetscInt *shared_indices);   /* Mat_CheckInode's node count; groups / column indices the device plan stores once per group */
PetscErrorCode MatHIPMI355XGetIndexCompression(Mat A, PetscInt *noffsets);   /* 0: plain CSR indices; else #offsets of the 1-byte dictionary */

/* ---- binary IO (PETSc binary format, big-endian; src/mat/impls/aij/seq/aij.c:4093-4157, src/vec/vec/utils/vecio.c) ---- */
PetscErrorCode PetscViewerBinaryOpen(MPI_Comm comm, const char name[], PetscFileMode mode, PetscViewer *viewer);
PetscErrorCode PetscViewerDestroy(PetscViewer *viewer);
PetscErrorCode MatLoad(Mat A, PetscViewer viewer);     /* AIJ types, sequential and parallel (each rank reads its rows) */
PetscErrorCode MatView(Mat A, PetscViewer viewer);     /* sequential AIJ */
PetscErrorCode VecLoad(Vec v, PetscViewer viewer);
PetscErrorCode VecView(Vec v, PetscViewer viewer);     /* sequential */

/* example-driver support: bulk assembly of the 3-D 7-point Poisson operator (rows [rstart,rend), global
 * ascending columns); the 3-D analogue of src/ksp/ksp/examples/tutorials/ex2.c:96-103 */
PetscErrorCode PetscHIPMI355XGenPoisson7(PetscInt nx, PetscInt ny, PetscInt nz, long rstart, long rend, PetscInt *ai, PetscInt *aj, PetscScalar *aa, long *nnz_out);

/* ---- PC (include/petscpc.h) ------------------------------------------------------------------ */
PetscErrorCode PCCreate(MPI_Comm comm, PC *pc);
PetscErrorCode PCSetType(PC pc, PCType type);
PetscErrorCode PCGetType(PC pc, PCType *type);
PetscErrorCode PCSetOperators(PC pc, Mat Amat, Mat Pmat, MatStructure flag);
PetscErrorCode PCSetUp(PC pc);
PetscErrorCode PCApply(PC pc, Vec x, Vec y);
PetscErrorCode PCSetFromOptions(PC pc);
PetscErrorCode PCDestroy(PC *pc);
PetscErrorCode PCILUGetLevels_HIPMI355X(PC pc, PetscInt *nlevL, PetscInt *nlevU);   /* dependency levels of the two triangular solves */
PetscErrorCode PCILUGetSolver_HIPMI355X(PC pc, PetscInt *syncfree, PetscInt *aborted);   /* 1: two-launch sync-free solves (-pc_factor_hipmi355x_trisolve syncfree, default above 16 levels); 0: one launch per level */
PetscErrorCode PCBJacobiGetSubKSP(PC pc, PetscInt *n_local, PetscInt *first_local, KSP **ksp);

/* ---- KSP (include/petscksp.h) ---------------------------------------------------------------- */
PetscErrorCode KSPCreate(MPI_Comm comm, KSP *ksp);
PetscErrorCode KSPSetType(KSP ksp, KSPType type);
PetscErrorCode KSPGetType(KSP ksp, KSPType *type);
PetscErrorCode KSPSetOperators(KSP ksp, Mat Amat, Mat Pmat, MatStructure flag);
PetscErrorCode KSPGetPC(KSP ksp, PC *pc);
PetscErrorCode KSPSetTolerances(KSP ksp, PetscReal rtol, PetscReal abstol, PetscReal dtol, PetscInt maxits);
PetscErrorCode KSPSetInitialGuessNonzero(KSP ksp, PetscBool flg);
PetscErrorCode KSPSetNormType(KSP ksp, KSPNormType normtype);
PetscErrorCode KSPSetPCSide(KSP ksp, PCSide side);   /* -ksp_pc_side <left|right>; right: KSPGMRES only */
PetscErrorCode KSPSetOptionsPrefix(KSP ksp, const char prefix[]);
PetscErrorCode KSPSetFromOptions(KSP ksp);   /* -ksp_type -ksp_rtol -ksp_atol -ksp_max_it -ksp_gmres_restart -ksp_gmres_cgs_refinement_type -pc_type -sub_* */
PetscErrorCode KSPGMRESSetRestart(KSP ksp, PetscInt restart);
PetscErrorCode KSPGMRESSetCGSRefinementType(KSP ksp, KSPGMRESCGSRefinementType type);
PetscErrorCode KSPSetUp(KSP ksp);
PetscErrorCode KSPSolve(KSP ksp, Vec b, Vec x);
PetscErrorCode KSPGetIterationNumber(KSP ksp, PetscInt *its);
PetscErrorCode KSPGetResidualNorm(KSP ksp, PetscReal *rnorm);
PetscErrorCode KSPGetConvergedReason(KSP ksp, KSPConvergedReason *reason);
PetscErrorCode KSPSetResidualHistory(KSP ksp, PetscReal a[], PetscInt na, PetscBool reset);
PetscErrorCode KSPGetResidualHistory(KSP ksp, PetscReal *a[], PetscInt *na);
/* the residual norms KSPMonitor would be called with (what -ksp_monitor_short prints) */
/* -ksp_monitor / -ksp_monitor_short (iterativ.c:178,484): the reference's text, on rank 0 */
PetscErrorCode KSPMonitorDefault(KSP ksp, PetscInt n, PetscReal rnorm, void *dummy);
PetscErrorCode KSPMonitorDefaultShort(KSP ksp, PetscInt n, PetscReal rnorm, void *dummy);
PetscErrorCode KSPMonitorSet(KSP ksp, PetscErrorCode (*monitor)(KSP, PetscInt, PetscReal, void *), void *mctx, PetscErrorCode (*destroy)(void **));
PetscErrorCode KSPDestroy(KSP *ksp);

#ifdef __cplusplus
}
#endif
#endif
