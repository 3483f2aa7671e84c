/* Private object layouts of the host library.  The function tables follow the reference's
 * struct _VecOps (include/petsc-private/vecimpl.h:221-294), struct _MatOps
 * (include/petsc-private/matimpl.h:17-188), struct _PCOps / _KSPOps, reduced to the slots the
 * Krylov path dispatches through. */
#ifndef PETSCIMPL_H
#define PETSCIMPL_H
#include "petschipmi355x.h"
#include "mi355x_kernels.h"
#include "mi355x_comm.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ---- error handling (include/petscerror.h:120,251) ---- */
PetscErrorCode PetscError(int line, const char *func, const char *file, PetscErrorCode n, const char *fmt, ...);
#define SETERRQ(comm, n, ...) return PetscError(__LINE__, __func__, __FILE__, (n), __VA_ARGS__)
#define CHKERRQ(n) do { if ((n)) return PetscError(__LINE__, __func__, __FILE__, (n), " "); } while (0)
/* map a kernel-library (hipError_t) failure to PETSC_ERR_LIB like CHKERRCUSP, cuspvecimpl.h:79 */
#define CHKHIP(e) do { int e__ = (e); if (e__) return PetscError(__LINE__, __func__, __FILE__, PETSC_ERR_LIB, "HIP/RCCL error %d: %s", e__, mi355x_comm_error_string(e__)); } while (0)
#define PetscFunctionBegin
#define PetscFunctionReturn(a) return (a)
PetscErrorCode PetscMallocFn(size_t bytes, void **p);
#define PetscMalloc(bytes, p) PetscMallocFn((size_t)(bytes), (void **)(p))
#define PetscFree(p) (free(p), (p) = NULL, 0)
PetscErrorCode PetscLogFlops(PetscLogDouble f);
#define PetscMax(a, b) (((a) < (b)) ? (b) : (a))
#define PetscMin(a, b) (((a) < (b)) ? (a) : (b))
#define PetscAbsScalar(a) fabs(a)
#define PetscSqrtReal(a) sqrt(a)
#define PetscIsInfOrNanScalar(a) (isnan(a) || isinf(a))

/* ---- communicator ---- */
struct _p_PetscComm {
  int rank, size;
  void *ctx;
  PetscCommAllgatherFn allgather;
  PetscCommAllreduceFn allreduce;
  PetscCommBarrierFn barrier;
  PetscCommExchangeFn exchange;
  /* RCCL communicators for device buffers, one per HIP stream (operations of one communicator are serialised by RCCL
   * in issue order whatever stream they are given, so the halo and the reductions would not overlap on one):
   * dcomm carries what is queued on the compute stream (all-reduces of dots / norms), dcomm_halo what is queued on
   * the halo stream (grouped ncclSend/ncclRecv of VecScatter, the split-phase all-reduce of comb.c's Begin/End) */
  mi355x_comm_t dcomm, dcomm_halo;
};

/* ---- device context of this process (one GPU, two streams) ---- */
typedef struct {
  int initialized, device;
  mi355x_handle_t h;        /* compute stream */
  mi355x_handle_t hcomm;    /* halo stream ("second HIP stream" of the north star) */
} PetscDeviceCtx;
PetscErrorCode PetscDeviceGet(PetscDeviceCtx **ctx);   /* lazily creates handles; PETSC_ERR_LIB without a GPU */

/* ---- layout (PetscLayout, include/petsc-private/vecimpl.h:21-32) ---- */
typedef struct {
  PetscInt n, N, rstart, rend;
  PetscInt *range;        /* size+1 */
  int refcnt;
} PetscLayout;
PetscErrorCode PetscLayoutCreateSetUp(MPI_Comm comm, PetscInt n, PetscInt N, PetscLayout **map);
PetscErrorCode PetscLayoutReference(PetscLayout *in, PetscLayout **out);
PetscErrorCode PetscLayoutDestroy(PetscLayout **map);

/* ---- options ---- */
PetscErrorCode PetscOptionsGetString(const char *pre, const char *name, char *value, size_t len, PetscBool *set);
PetscErrorCode PetscOptionsGetInt(const char *pre, const char *name, PetscInt *value, PetscBool *set);
PetscErrorCode PetscOptionsGetReal(const char *pre, const char *name, PetscReal *value, PetscBool *set);

/* ---- Vec ---- */
typedef struct _VecOps {
  PetscErrorCode (*duplicate)(Vec, Vec *);
  PetscErrorCode (*dot)(Vec, Vec, PetscScalar *);
  PetscErrorCode (*mdot)(Vec, PetscInt, const Vec[], PetscScalar *);
  PetscErrorCode (*norm)(Vec, NormType, PetscReal *);
  PetscErrorCode (*tdot)(Vec, Vec, PetscScalar *);
  PetscErrorCode (*mtdot)(Vec, PetscInt, const Vec[], PetscScalar *);
  PetscErrorCode (*scale)(Vec, PetscScalar);
  PetscErrorCode (*copy)(Vec, Vec);
  PetscErrorCode (*set)(Vec, PetscScalar);
  PetscErrorCode (*swap)(Vec, Vec);
  PetscErrorCode (*axpy)(Vec, PetscScalar, Vec);
  PetscErrorCode (*axpby)(Vec, PetscScalar, PetscScalar, Vec);
  PetscErrorCode (*maxpy)(Vec, PetscInt, const PetscScalar *, Vec *);
  PetscErrorCode (*aypx)(Vec, PetscScalar, Vec);
  PetscErrorCode (*waxpy)(Vec, PetscScalar, Vec, Vec);
  PetscErrorCode (*axpbypcz)(Vec, PetscScalar, PetscScalar, PetscScalar, Vec, Vec);
  PetscErrorCode (*pointwisemult)(Vec, Vec, Vec);
  PetscErrorCode (*pointwisedivide)(Vec, Vec, Vec);
  PetscErrorCode (*setvalues)(Vec, PetscInt, const PetscInt[], const PetscScalar[], InsertMode);
  PetscErrorCode (*assemblybegin)(Vec);
  PetscErrorCode (*assemblyend)(Vec);
  PetscErrorCode (*getarray)(Vec, PetscScalar **);
  PetscErrorCode (*restorearray)(Vec, PetscScalar **);
  PetscErrorCode (*getarrayread)(Vec, const PetscScalar **);
  PetscErrorCode (*placearray)(Vec, const PetscScalar *);
  PetscErrorCode (*resetarray)(Vec);
  PetscErrorCode (*destroy)(Vec);
  PetscErrorCode (*reciprocal)(Vec);
  PetscErrorCode (*dotnorm2)(Vec, Vec, PetscScalar *, PetscScalar *);
  PetscErrorCode (*create)(Vec);   /* type constructor, kept for duplicate */
} VecOps;

struct _p_Vec {
  MPI_Comm comm;
  char type_name[32];
  VecOps ops[1];
  PetscLayout *map;
  void *data;
  int state;                      /* PetscObjectStateIncrease, include/petsc-private/petscimpl.h:440 */
  /* norm cache keyed on state (rvector.c:205-224) */
  int norm_state[4];
  PetscReal norm_val[4];
};
#define PetscObjectStateIncrease(v) ((v)->state++)

/* coherence flags, as PETSC_CUSP_UNALLOCATED/CPU/GPU/BOTH (include/petsc-private/vecimpl.h) */
enum { VALID_NONE = 0, VALID_HOST = 1, VALID_DEVICE = 2, VALID_BOTH = 3 };
typedef struct {
  PetscScalar *host;        /* host mirror, allocated on first host access */
  PetscScalar *dev;         /* HBM */
  int valid;
  PetscScalar *placed_save; /* VecPlaceArray */
  int host_owned;
} Vec_HIPMI355X;

typedef PetscErrorCode (*VecCreateFn)(Vec);
PetscErrorCode VecRegister(const char *name, VecCreateFn fn);
PetscErrorCode VecCreate_SeqHIPMI355X(Vec v);
PetscErrorCode VecCreate_MPIHIPMI355X(Vec v);
PetscErrorCode VecCreate_HIPMI355X(Vec v);
PetscErrorCode VecCreateSeqHIPMI355X(MPI_Comm comm, PetscInt n, Vec *v);
/* device pointers with coherence (analogue of VecCUSPGetArrayRead/Write, cuspvecimpl.h:95-150) */
PetscErrorCode VecHIPGetRead(Vec v, const PetscScalar **d);
PetscErrorCode VecHIPGetWrite(Vec v, PetscScalar **d);       /* contents will be overwritten */
PetscErrorCode VecHIPGetReadWrite(Vec v, PetscScalar **d);
PetscErrorCode VecHIPRestoreWrite(Vec v);                    /* device newer; state++ */
PetscErrorCode VecCGUpdate_HIPMI355X(Vec x, Vec r, Vec z, Vec p, Vec w, Vec d, PetscScalar a, PetscScalar *zz, PetscScalar *zr, PetscScalar *rr, PetscBool *done);
PetscErrorCode VecCGUpdateCheck_HIPMI355X(Vec x, Vec r, Vec z, Vec p, Vec w, Vec d, PetscBool *ok);
PetscErrorCode VecTDotBegin_HIPMI355X(Vec x, Vec y, PetscBool *ok);   /* result stays on the device; pairs with VecCGUpdateDev */
PetscErrorCode VecCGUpdateDevBegin_HIPMI355X(Vec x, Vec r, Vec z, Vec p, Vec w, Vec d, PetscScalar beta, PetscScalar dpiold, PetscBool check_sign);
PetscErrorCode VecCGUpdateDevEnd_HIPMI355X(Vec x, PetscScalar *zz, PetscScalar *zr, PetscScalar *rr, PetscScalar *dpi);
#define PETSC_HIP_DPI_SLOT 8   /* device scratch slot holding p'w between the dot (or the SpMV by-product) and the CG update */
PetscErrorCode MatMultTDotBegin_HIPMI355X(Mat A, Vec x, Vec y, PetscBool *ok);   /* y = A x, x'y left on the device */
PetscErrorCode VecAYPXDev_HIPMI355X(Vec p, PetscScalar den, Vec z);   /* p = z + (z'r on the device / den) p */
PetscErrorCode VecPMultDot_HIPMI355X(Vec w, Vec x, Vec d, Vec y, PetscScalar *val, PetscBool *done);
PetscErrorCode VecPMultDotNorm2_HIPMI355X(Vec w, Vec x, Vec d, Vec s, PetscScalar *dp, PetscReal *nm, PetscBool *done);
PetscErrorCode VecBCGSUpdate_HIPMI355X(Vec x, Vec r, Vec p, Vec s, Vec t, Vec rp, PetscScalar alpha, PetscScalar omega, PetscScalar *rr, PetscScalar *rho, PetscBool *done);
PetscBool PCIsNone_Private(PC pc);
PetscErrorCode PCJacobiGetInverseDiagonal_Private(PC pc, Vec *d);   /* NULL unless pc is a set-up PCJACOBI */

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
struct _p_VecScatter {
  MPI_Comm comm;
  VecScatterSide to, from;
  PetscBool inuse;            /* guard, vscat.c:1637 */
  mi355x_event_t ev_packed, ev_done;
  int device_ready;
  int ready_marked;           /* ev_packed already recorded by VecScatterMarkReady */
  /* every rank's request list (kept from set-up) for the host-staged transport */
  PetscScalar *h_send, *h_recv;   /* host staging buffers of the host-staged transport */
  PetscScalar *d_local_tmp;       /* device staging of the local (self) part, local_n doubles */
};
PetscErrorCode VecScatterMarkReady(VecScatter ctx, Vec x);
PetscErrorCode VecScatterCreate_PtoS_MPIAIJ(MPI_Comm comm, PetscLayout *xmap, PetscInt ec, const PetscInt *garray, VecScatter *ctx);

/* ---- Mat ---- */
typedef struct _MatOps {
  PetscErrorCode (*setvalues)(Mat, PetscInt, const PetscInt[], PetscInt, const PetscInt[], const PetscScalar[], InsertMode);
  PetscErrorCode (*mult)(Mat, Vec, Vec);                 /* slot 3 */
  PetscErrorCode (*multadd)(Mat, Vec, Vec, Vec);         /* slot 4 */
  PetscErrorCode (*multtranspose)(Mat, Vec, Vec);        /* slot 5 */
  PetscErrorCode (*multtransposeadd)(Mat, Vec, Vec, Vec);/* slot 6 */
  PetscErrorCode (*getdiagonal)(Mat, Vec);               /* slot 17 */
  PetscErrorCode (*assemblybegin)(Mat, MatAssemblyType);
  PetscErrorCode (*assemblyend)(Mat, MatAssemblyType);   /* slot 21 */
  PetscErrorCode (*zeroentries)(Mat);                    /* slot 23 */
  PetscErrorCode (*setup)(Mat);
  PetscErrorCode (*scale)(Mat, PetscScalar);
  PetscErrorCode (*diagonalscale)(Mat, Vec, Vec);        /* slot 18 */
  PetscErrorCode (*setvaluesbatch)(Mat, PetscInt, PetscInt, PetscInt[], const PetscScalar[]);
  PetscErrorCode (*destroy)(Mat);                        /* slot 60 */
  PetscErrorCode (*getvecs)(Mat, Vec *, Vec *);          /* slot 88 */
} MatOps;

struct _p_Mat {
  MPI_Comm comm;
  char type_name[32];
  char pending_type[32];      /* MatSetType() before the sizes are known (allowed, matreg.c): applied by MatSetSizes()/MatLoad() */
  MatOps ops[1];
  PetscLayout *rmap, *cmap;
  PetscInt m_req, n_req, M_req, N_req;   /* MatSetSizes arguments */
  PetscBool assembled, was_assembled, preallocated;
  int state;
  void *data;
  void *spptr;        /* device mirror, as Mat->spptr (matimpl.h:323) */
  PetscBool timing;
  PetscInt time_n, time_cap;
  mi355x_event_t *time_ev;   /* pairs */
  PetscLogDouble time_ms;
};

/* host CSR container (Mat_SeqAIJ, src/mat/impls/aij/seq/aij.h:10-39,99-115) */
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
} Mat_SeqAIJ;

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
  PetscInt n_uploads;        /* value uploads so far */
  /* MatSetValuesBatch map for one connectivity (rows array): contributions grouped by nonzero, in call order */
  PetscInt bm_nb, bm_bs, bm_nseg; unsigned long long bm_hash; size_t bm_T, bm_vcap;
  PetscInt *bm_order, *bm_segptr, *bm_segslot;   /* device */
  PetscScalar *bm_v;                             /* device staging of the element values */
} Mat_SeqAIJHIP;

/* Mat_MPIAIJ, src/mat/impls/aij/mpi/mpiaij.h:35-77 */
typedef struct {
  Mat A, B;                 /* diagonal / off-diagonal blocks (SeqAIJHIPMI355X) */
  PetscInt *garray, ec;
  Vec lvec;
  VecScatter Mvctx;
  PetscInt rstart, rend, cstart, cend;
} Mat_MPIAIJ;

typedef PetscErrorCode (*MatCreateFn)(Mat);
PetscErrorCode MatRegister(const char *name, MatCreateFn fn);
PetscErrorCode MatCreate_SeqAIJHIPMI355X(Mat);
PetscErrorCode MatCreate_MPIAIJHIPMI355X(Mat);
PetscErrorCode MatCreate_AIJHIPMI355X(Mat);
PetscErrorCode MatCreate_SeqBAIJHIPMI355X(Mat);
PetscErrorCode MatSeqAIJHIPUpload(Mat A);
PetscErrorCode MatSeqAIJHIPSetCompressedRow(Mat A, PetscBool flg);
PetscErrorCode MatSetUpMultiply_MPIAIJ(Mat mat);
PetscErrorCode MatTimingBegin(Mat A, mi355x_handle_t h);
PetscErrorCode MatTimingEnd(Mat A, mi355x_handle_t h);

/* ---- PC / KSP ---- */
typedef struct {
  PetscErrorCode (*setup)(PC);
  PetscErrorCode (*apply)(PC, Vec, Vec);
  PetscErrorCode (*setfromoptions)(PC);
  PetscErrorCode (*destroy)(PC);
} PCOps;
struct _p_PC {
  MPI_Comm comm;
  char type_name[32];
  char prefix[64];
  PCOps ops[1];
  Mat mat, pmat;
  int setupcalled;
  void *data;
};

typedef struct {
  PetscErrorCode (*setup)(KSP);
  PetscErrorCode (*solve)(KSP);
  PetscErrorCode (*setfromoptions)(KSP);
  PetscErrorCode (*destroy)(KSP);
} KSPOps;
struct _p_KSP {
  MPI_Comm comm;
  char type_name[32];
  char prefix[64];
  KSPOps ops[1];
  PC pc;
  Vec vec_sol, vec_rhs;
  Vec *work; PetscInt nwork;
  PetscReal rtol, abstol, divtol, ttol, rnorm0, rnorm;
  PetscInt max_it, its, chknorm;
  PetscBool guess_zero;
  KSPNormType normtype;
  PCSide pc_side;
  KSPConvergedReason reason;
  int setupcalled;
  PetscReal *res_hist; PetscInt res_hist_len, res_hist_max; PetscBool res_hist_reset; PetscReal *res_hist_alloc;
  PetscErrorCode (*monitor)(KSP, PetscInt, PetscReal, void *); void *mctx;
  PetscBool printreason;   /* -ksp_converged_reason */
  void *data;
};
PetscErrorCode KSPDefaultConverged(KSP ksp, PetscInt n, PetscReal rnorm, KSPConvergedReason *reason);
PetscErrorCode KSPMonitor(KSP ksp, PetscInt it, PetscReal rnorm);
PetscErrorCode KSPLogResidualHistory(KSP ksp, PetscReal norm);
PetscErrorCode KSPDefaultGetWork(KSP ksp, PetscInt nw);
PetscErrorCode KSPInitialResidual(KSP ksp, Vec vsoln, Vec vt1, Vec vt2, Vec vres, Vec vb);
PetscErrorCode KSP_MatMult(KSP ksp, Mat A, Vec x, Vec y);
PetscErrorCode KSP_PCApply(KSP ksp, Vec x, Vec y);
PetscErrorCode KSP_PCApplyBAorAB(KSP ksp, Vec x, Vec y, Vec w);
PetscErrorCode KSPCreate_CG(KSP), KSPCreate_GROPPCG(KSP), KSPCreate_GMRES(KSP), KSPCreate_BCGS(KSP), KSPCreate_PREONLY(KSP);
PetscErrorCode PCCreate_None(PC), PCCreate_Jacobi(PC), PCCreate_BJacobi(PC), PCCreate_ILU(PC);

#endif
