/*
 * petscmini.h -- public interface of the HARNESS library (libpetscharness.so): a stand-in for the slice of PETSc's
 * object model the Krylov hot path runs through, for boxes where no PETSc exists (the GPU test box; the reference
 * cannot be built here).  It mirrors, name for name and argument for argument, erdc/petsc-dev 3.3.0-dev's
 * include/petscsys.h, petscvec.h, petscmat.h, petscpc.h, petscksp.h for the calls of
 * src/ksp/ksp/examples/tutorials/ex2.c-style programs: type registries keyed by strings, per-object function tables,
 * composed functions, options database, KSPSolve_CG/GMRES/BCGS, PCNONE/JACOBI/BJACOBI drivers.  It contains NO
 * device code and knows no HIPMI355X name: the Vec/Mat implementations come from the plugin
 * (include/petschipmi355x.h, libpetschipmi355x.so), which registers them here exactly as it registers them with a
 * real PETSc.  Inside a PETSc tree this header and this library are not used at all.
 *
 * Names: the communicator type is PetscComm (own name), so that this header can be included next to <mpi.h>;
 * petscmini_mpinames.h maps MPI_Comm / MPI_Comm_rank / MPI_Comm_size onto it for example programs written against
 * PETSc's spelling.
 */
#ifndef PETSCMINI_H
#define PETSCMINI_H
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
typedef enum { MAT_DO_NOT_COPY_VALUES, MAT_COPY_VALUES, MAT_SHARE_NONZERO_PATTERN } MatDuplicateOption;   /* petscmat.h:440 */
typedef enum { DIFFERENT_NONZERO_PATTERN, SUBSET_NONZERO_PATTERN, SAME_NONZERO_PATTERN, SAME_PRECONDITIONER } MatStructure;
/* matrix factorisation interface (include/petscmat.h:100-131,1042-1088): the factored matrix is a Mat of its own, obtained from the
 * operator with MatGetFactor, filled by a symbolic and a numeric call, applied with MatSolve */
typedef enum { MAT_FACTOR_NONE, MAT_FACTOR_LU, MAT_FACTOR_CHOLESKY, MAT_FACTOR_ILU, MAT_FACTOR_ICC, MAT_FACTOR_ILUDT } MatFactorType;
typedef enum { MAT_SHIFT_NONE, MAT_SHIFT_NONZERO, MAT_SHIFT_POSITIVE_DEFINITE, MAT_SHIFT_INBLOCKS } MatFactorShiftType;
typedef struct {
  PetscReal diagonal_fill, usedt, dt, dtcol, dtcount;
  PetscReal fill;            /* expected fill */
  PetscReal levels;          /* ICC/ILU(levels) */
  PetscReal pivotinblocks;
  PetscReal zeropivot;       /* pivot is called zero if less than this */
  PetscReal shifttype;       /* MatFactorShiftType, stored as a real (petscmat.h:1072) */
  PetscReal shiftamount;
} MatFactorInfo;
#define MatSolverPackage char *
#define MATSOLVERPETSC "petsc"
typedef enum { PC_SIDE_DEFAULT = -1, PC_LEFT, PC_RIGHT, PC_SYMMETRIC } PCSide;
#define PC_SIDE_MAX 3    /* PC_SYMMETRIC + 1, petscpc.h:96 */
typedef enum { KSP_NORM_DEFAULT = -1, KSP_NORM_NONE = 0, KSP_NORM_PRECONDITIONED = 1, KSP_NORM_UNPRECONDITIONED = 2, KSP_NORM_NATURAL = 3 } KSPNormType;
#define KSP_NORM_MAX 4   /* KSP_NORM_NATURAL + 1, petscksp.h:349 */
typedef enum { KSP_GMRES_CGS_REFINE_NEVER, KSP_GMRES_CGS_REFINE_IFNEEDED, KSP_GMRES_CGS_REFINE_ALWAYS } KSPGMRESCGSRefinementType;
typedef enum { /* petscksp.h:403-430 */
  KSP_CONVERGED_RTOL_NORMAL = 1, KSP_CONVERGED_ATOL_NORMAL = 9, KSP_CONVERGED_RTOL = 2, KSP_CONVERGED_ATOL = 3,
  KSP_CONVERGED_ITS = 4, KSP_CONVERGED_HAPPY_BREAKDOWN = 8,
  KSP_DIVERGED_NULL = -2, KSP_DIVERGED_ITS = -3, KSP_DIVERGED_DTOL = -4, KSP_DIVERGED_BREAKDOWN = -5,
  KSP_DIVERGED_BREAKDOWN_BICG = -6, KSP_DIVERGED_NONSYMMETRIC = -7, KSP_DIVERGED_INDEFINITE_PC = -8,
  KSP_DIVERGED_NAN = -9, KSP_DIVERGED_INDEFINITE_MAT = -10, KSP_CONVERGED_ITERATING = 0
} KSPConvergedReason;

typedef struct _p_PetscComm  *PetscComm;    /* stands in for PetscComm (own name: coexists with <mpi.h>) */
typedef struct _p_Vec        *Vec;
typedef struct _p_Mat        *Mat;
typedef struct _p_VecScatter *VecScatter;
typedef struct _p_KSP        *KSP;
typedef struct _p_PC         *PC;
typedef struct _p_PetscViewer *PetscViewer;
typedef struct _p_IS         *IS;           /* index set: only ever NULL here (the natural ordering) */
typedef enum { FILE_MODE_READ, FILE_MODE_WRITE } PetscFileMode;
typedef const char *VecType;
typedef const char *MatType;
typedef const char *KSPType;
typedef const char *PCType;

/* the reference's type names (include/petscvec.h:75-80, petscmat.h:33-62); WHICH implementation answers to them is up to
 * whoever registered it (VecRegister / MatRegister): this library has none of its own */
#define VECSEQ      "seq"
#define VECMPI      "mpi"
#define VECSTANDARD "standard"
#define MATSEQAIJ   "seqaij"
#define MATMPIAIJ   "mpiaij"
#define MATAIJ      "aij"
#define MATSEQBAIJ  "seqbaij"
#define KSPCG      "cg"
#define KSPGMRES   "gmres"
#define KSPBCGS    "bcgs"
#define KSPPREONLY "preonly"
#define KSPPIPECG  "pipecg"    /* pipelined CG (src/ksp/ksp/impls/cg/pipecg/pipecg.c), SURVEY 8f.4 */
#define KSPGROPPCG "groppcg"   /* Gropp's overlapped CG (src/ksp/ksp/impls/cg/groppcg/groppcg.c), SURVEY 8f.4 */
#define PCNONE     "none"
#define PCJACOBI   "jacobi"
#define PCBJACOBI  "bjacobi"
#define PCPBJACOBI "pbjacobi"  /* point-block Jacobi (src/ksp/pc/impls/pbjacobi/pbjacobi.c), SURVEY 8f.4 */
#define PCILU      "ilu"       /* ILU(0), natural ordering, sequential AIJ */
#define PCICC      "icc"       /* ICC(0), natural ordering, sequential AIJ */

/* ---- Sys ----------------------------------------------------------------------------------- */
extern PetscComm PETSC_COMM_SELF, PETSC_COMM_WORLD;
/* registers the harness's own KSP and PC types (idempotent; every Create calls it) */
PetscErrorCode PetscMiniInitialize(void);
/* last error message (PetscError traceback text, src/sys/error/err.c) */
const char    *PetscGetLastErrorMessage(void);
/* Host collectives supplied by the launcher (torch.distributed / MPI): the reference uses MPI for these */
typedef int (*PetscCommAllgatherFn)(void *ctx, const void *sendbuf, int nbytes, void *recvbuf);
typedef int (*PetscCommAllreduceFn)(void *ctx, void *buf, int count, int is_double, int op /*0 sum,1 max,2 min*/);
typedef int (*PetscCommBarrierFn)(void *ctx);
PetscErrorCode PetscCommCreate(int rank, int size, void *ctx, PetscCommAllgatherFn, PetscCommAllreduceFn, PetscCommBarrierFn, PetscComm *comm);
/* optional host-staged neighbour exchange (used only when no RCCL communicator is attached, e.g. several ranks
 * sharing one GPU): post every receive and send of one halo exchange, return when all have completed */
typedef int (*PetscCommExchangeFn)(void *ctx, int nsend, const int *speers, void *const *sbufs, const int *sbytes,
                                   int nrecv, const int *rpeers, void *const *rbufs, const int *rbytes);
PetscErrorCode PetscCommSetExchange(PetscComm comm, PetscCommExchangeFn fn);
PetscErrorCode PetscCommSetWorld(PetscComm comm);
/* two opaque slots a Vec/Mat plugin may hang communicator-wide data on (the HIPMI355X plugin: its RCCL communicators) */
PetscErrorCode PetscCommSetPluginData(PetscComm comm, int slot, void *data);
PetscErrorCode PetscCommGetPluginData(PetscComm comm, int slot, void **data);
PetscErrorCode PetscCommDestroy(PetscComm *comm);
PetscErrorCode PetscCommRank(PetscComm comm, PetscMPIInt *rank);
PetscErrorCode PetscCommSize(PetscComm comm, PetscMPIInt *size);
/* flop counter fed by PetscLogFlops with the reference's formulas (include/petsclog.h:294) */
PetscErrorCode PetscGetFlops(PetscLogDouble *flops);
PetscErrorCode PetscSplitOwnership(PetscComm comm, PetscInt *n, PetscInt *N);   /* src/sys/utils/psplit.c */
/* options database subset (src/sys/objects/options.c): "-ksp_type cg -pc_type jacobi ..." */
PetscErrorCode PetscOptionsInsertString(const char *str);
PetscErrorCode PetscOptionsSetValue(const char *name, const char *value);
PetscErrorCode PetscOptionsClear(void);

/* ---- registries and composed functions: what a Vec/Mat/PC plugin binds to ------------------------------------------
 * VecRegister (src/vec/vec/interface/vecreg.c:100), MatRegister (src/mat/interface/matreg.c:137), PCRegister
 * (src/ksp/pc/interface/pcregis.c), PetscObjectComposeFunction / PetscObjectQueryFunction (src/sys/objects/inherit.c):
 * type-specific methods such as "MatSeqAIJSetPreallocation_C" are looked up by name on the object. */
typedef struct _p_PetscObject *PetscObject;
typedef void (*PetscVoidFunction)(void);
/* signatures of the reference (petscvec.h:314, petscmat.h:176, petscpc.h, petscksp.h, petscsys.h:1360): `path` and the
 * function's own name serve dynamic loading there and are ignored here */
PetscErrorCode VecRegister(const char sname[], const char path[], const char name[], PetscErrorCode (*create)(Vec));
PetscErrorCode MatRegister(const char sname[], const char path[], const char name[], PetscErrorCode (*create)(Mat));
PetscErrorCode PCRegister(const char sname[], const char path[], const char name[], PetscErrorCode (*create)(PC));
PetscErrorCode KSPRegister(const char sname[], const char path[], const char name[], PetscErrorCode (*create)(KSP));
PetscErrorCode PetscObjectComposeFunction(PetscObject obj, const char name[], const char fname[], PetscVoidFunction fn);
PetscErrorCode PetscObjectQueryFunction(PetscObject obj, const char name[], PetscVoidFunction *fn);
PetscErrorCode PetscObjectChangeTypeName(PetscObject obj, const char type_name[]);

/* ---- Vec (include/petscvec.h; wrappers src/vec/vec/interface/rvector.c) ---------------------- */
PetscErrorCode VecCreate(PetscComm comm, Vec *vec);
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
PetscErrorCode VecReplaceArray(Vec v, const PetscScalar *a);   /* rvector.c:1610: the vector takes ownership of a (allocated with PetscMalloc) */
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
/* split-phase reductions (src/vec/vec/utils/comb.c:402-721) */
PetscErrorCode VecDotBegin(Vec x, Vec y, PetscScalar *result);
PetscErrorCode VecDotEnd(Vec x, Vec y, PetscScalar *result);
PetscErrorCode VecNormBegin(Vec x, NormType type, PetscReal *result);
PetscErrorCode VecNormEnd(Vec x, NormType type, PetscReal *result);
PetscErrorCode PetscCommSplitReductionBegin(PetscComm comm);
/* ---- Mat (include/petscmat.h; wrappers src/mat/interface/matrix.c) --------------------------- */
PetscErrorCode MatCreate(PetscComm comm, Mat *A);
PetscErrorCode MatSetSizes(Mat A, PetscInt m, PetscInt n, PetscInt M, PetscInt N);
PetscErrorCode MatSetType(Mat A, MatType type);
PetscErrorCode MatSetFromOptions(Mat A);                      /* -mat_type, then the type's own options (ops->setfromoptions, gcreate.c:167-205) */
PetscErrorCode MatDuplicate(Mat A, MatDuplicateOption op, Mat *B);   /* matrix.c:4023 */
PetscErrorCode MatGetType(Mat A, MatType *type);
PetscErrorCode MatSetUp(Mat A);
PetscErrorCode MatSeqAIJSetPreallocation(Mat A, PetscInt nz, const PetscInt nnz[]);
PetscErrorCode MatMPIAIJSetPreallocation(Mat A, PetscInt d_nz, const PetscInt d_nnz[], PetscInt o_nz, const PetscInt o_nnz[]);
PetscErrorCode MatSeqAIJSetPreallocationCSR(Mat A, const PetscInt i[], const PetscInt j[], const PetscScalar v[]);   /* aij.c:3795 */
PetscErrorCode MatMPIAIJSetPreallocationCSR(Mat A, const PetscInt i[], const PetscInt j[], const PetscScalar v[]);   /* mpiaij.c:3960 */
PetscErrorCode MatGetDiagonalBlock(Mat A, Mat *a);
PetscErrorCode MatSetValues(Mat A, PetscInt m, const PetscInt idxm[], PetscInt n, const PetscInt idxn[], const PetscScalar v[], InsertMode addv);
PetscErrorCode MatSetValuesBatch(Mat A, PetscInt nb, PetscInt bs, PetscInt rows[], const PetscScalar v[]);   /* matrix.c:1698; device-side value assembly when the pattern is unchanged */
PetscErrorCode MatAssemblyBegin(Mat A, MatAssemblyType type);
PetscErrorCode MatAssemblyEnd(Mat A, MatAssemblyType type);
/* bulk creation from CSR (MatCreateSeqAIJWithArrays src/mat/impls/aij/seq/aij.c, MatCreateMPIAIJWithArrays
 * src/mat/impls/aij/mpi/mpiaij.c; i/j/a are copied; j holds global column indices, ascending per row) */
PetscErrorCode MatCreateSeqAIJWithArrays(PetscComm comm, PetscInt m, PetscInt n, PetscInt i[], PetscInt j[], PetscScalar a[], Mat *mat);
PetscErrorCode MatCreateMPIAIJWithArrays(PetscComm comm, PetscInt m, PetscInt n, PetscInt M, PetscInt N, const PetscInt i[], const PetscInt j[], const PetscScalar a[], Mat *mat);
PetscErrorCode MatCreateSeqBAIJWithArrays(PetscComm comm, PetscInt bs, PetscInt m, PetscInt n, PetscInt i[], PetscInt j[], PetscScalar a[], Mat *mat);
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
PetscErrorCode MatSetOptionsPrefix(Mat A, const char prefix[]);
/* matrix.c:3937-4010 (MatGetFactor looks "MatGetFactor_<package>_C" up on the operator), 2766-3144 (symbolic / numeric), 3196 (MatSolve) */
PetscErrorCode MatFactorInfoInitialize(MatFactorInfo *info);
PetscErrorCode MatGetFactor(Mat mat, const MatSolverPackage type, MatFactorType ftype, Mat *f);
PetscErrorCode MatGetFactorAvailable(Mat mat, const MatSolverPackage type, MatFactorType ftype, PetscBool *flg);
PetscErrorCode MatILUFactorSymbolic(Mat fact, Mat mat, IS row, IS col, const MatFactorInfo *info);
PetscErrorCode MatLUFactorNumeric(Mat fact, Mat mat, const MatFactorInfo *info);
PetscErrorCode MatICCFactorSymbolic(Mat fact, Mat mat, IS perm, const MatFactorInfo *info);
PetscErrorCode MatCholeskyFactorNumeric(Mat fact, Mat mat, const MatFactorInfo *info);
PetscErrorCode MatSolve(Mat mat, Vec b, Vec x);
PetscErrorCode MatDiagonalScale(Mat A, Vec l, Vec r);   /* A <- diag(l) A diag(r); l or r may be NULL (aij.c:2055, mpiaij.c:2183) */
/* ---- binary IO (PETSc binary format, big-endian; src/mat/impls/aij/seq/aij.c:4093-4157, src/vec/vec/utils/vecio.c) ---- */
PetscErrorCode PetscViewerBinaryOpen(PetscComm comm, const char name[], PetscFileMode mode, PetscViewer *viewer);
PetscErrorCode PetscViewerDestroy(PetscViewer *viewer);
PetscErrorCode MatLoad(Mat A, PetscViewer viewer);     /* AIJ types, sequential and parallel (each rank reads its rows) */
PetscErrorCode MatView(Mat A, PetscViewer viewer);     /* sequential AIJ */
PetscErrorCode VecLoad(Vec v, PetscViewer viewer);
PetscErrorCode VecView(Vec v, PetscViewer viewer);     /* sequential */

/* example-driver support: bulk assembly of the 3-D 7-point Poisson operator (rows [rstart,rend), global
 * ascending columns); the 3-D analogue of src/ksp/ksp/examples/tutorials/ex2.c:96-103 */
PetscErrorCode PetscMiniGenPoisson7(PetscInt nx, PetscInt ny, PetscInt nz, long rstart, long rend, PetscInt *ai, PetscInt *aj, PetscScalar *aa, long *nnz_out);

/* ---- PC (include/petscpc.h) ------------------------------------------------------------------ */
PetscErrorCode PCCreate(PetscComm comm, PC *pc);
PetscErrorCode PCSetType(PC pc, PCType type);
PetscErrorCode PCGetType(PC pc, PCType *type);
PetscErrorCode PCSetOperators(PC pc, Mat Amat, Mat Pmat, MatStructure flag);
PetscErrorCode PCGetOperators(PC pc, Mat *Amat, Mat *Pmat, MatStructure *flag);
PetscErrorCode PCSetUp(PC pc);
PetscErrorCode PCSetUpOnBlocks(PC pc);
PetscErrorCode PCApply(PC pc, Vec x, Vec y);
PetscErrorCode PCSetFromOptions(PC pc);
PetscErrorCode PCDestroy(PC *pc);
PetscErrorCode PCFactorGetMatrix(PC pc, Mat *mat);   /* the factored matrix of a PCILU / PCICC (precon.c PCFactorGetMatrix) */
PetscErrorCode PCBJacobiGetSubKSP(PC pc, PetscInt *n_local, PetscInt *first_local, KSP **ksp);

/* ---- KSP (include/petscksp.h) ---------------------------------------------------------------- */
PetscErrorCode KSPCreate(PetscComm comm, KSP *ksp);
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
PetscErrorCode KSPSetUpOnBlocks(KSP ksp);
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
