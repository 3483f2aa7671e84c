/* Private object layouts of the HARNESS library: the stand-in for include/petsc-private/{petscimpl,vecimpl,matimpl,
 * pcimpl,kspimpl}.h.  Objects start with a common header (PETSCHEADER, petscimpl.h:110-112) and dispatch through
 * per-object function tables whose slot NAMES and SIGNATURES are those of the reference (struct _VecOps
 * include/petsc-private/vecimpl.h:221-294, struct _MatOps include/petsc-private/matimpl.h:17-188), reduced to the
 * slots the Krylov path dispatches through -- tests/test_integration_shim.py checks every one of them against the
 * reference's headers.  Nothing here knows about devices or about the HIPMI355X types. */
#ifndef PETSCIMPL_H
#define PETSCIMPL_H
#include "petscmini.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ---- error handling (include/petscerror.h:120,251) ---- */
PetscErrorCode PetscError(int line, const char *func, const char *file, PetscErrorCode n, const char *fmt, ...);
#define SETERRQ(comm, n, ...) return PetscError(__LINE__, __func__, __FILE__, (n), __VA_ARGS__)
#define CHKERRQ(n) do { if ((n)) return PetscError(__LINE__, __func__, __FILE__, (n), " "); } while (0)
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
  void *plugin[2];         /* PetscCommSetPluginData: opaque to this library */
};

/* ---- common object header (PETSCHEADER, include/petsc-private/petscimpl.h:60-112) ---- */
struct _n_PetscFList { char name[64]; PetscVoidFunction fn; struct _n_PetscFList *next; };
#define PETSC_OBJECT_FIELDS \
  PetscComm comm;                 \
  char type_name[32];             \
  char prefix[64];                \
  int state;                      /* PetscObjectStateIncrease, petscimpl.h:440 */ \
  struct _n_PetscFList *qlist;    /* composed functions (PetscObjectComposeFunction) */
struct _p_PetscObject { PETSC_OBJECT_FIELDS };
/* `hdr` as in the reference; the fields are also reachable directly (obj->comm), which the wrappers use */
#define PETSCHEADER(ObjectOps) union { struct _p_PetscObject hdr; struct { PETSC_OBJECT_FIELDS }; }; ObjectOps ops[1]
#define PetscObjectStateIncrease(obj) (((PetscObject)(obj))->state++)
PetscErrorCode PetscObjectListDestroy_Private(PetscObject obj);

/* ---- layout (PetscLayout, include/petsc-private/vecimpl.h:21-32) ---- */
typedef struct _n_PetscLayout *PetscLayout;
struct _n_PetscLayout {
  PetscInt n, N, rstart, rend;
  PetscInt *range;        /* size+1 */
  int refcnt;
};
PetscErrorCode PetscLayoutCreateSetUp(PetscComm comm, PetscInt n, PetscInt N, PetscLayout *map);
PetscErrorCode PetscLayoutReference(PetscLayout in, PetscLayout *out);
PetscErrorCode PetscLayoutDestroy(PetscLayout *map);

/* ---- options ---- */
PetscErrorCode PetscOptionsGetString(const char *pre, const char *name, char *value, size_t len, PetscBool *set);
PetscErrorCode PetscOptionsGetInt(const char *pre, const char *name, PetscInt *value, PetscBool *set);
PetscErrorCode PetscOptionsGetReal(const char *pre, const char *name, PetscReal *value, PetscBool *set);

/* ---- Vec ---- */
struct _VecOps {
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
  PetscErrorCode (*placearray)(Vec, const PetscScalar *);
  PetscErrorCode (*replacearray)(Vec, const PetscScalar *);   /* vecimpl.h:259 */
  PetscErrorCode (*resetarray)(Vec);
  PetscErrorCode (*destroy)(Vec);
  PetscErrorCode (*reciprocal)(Vec);
  PetscErrorCode (*dotnorm2)(Vec, Vec, PetscScalar *, PetscScalar *);
};
typedef struct _VecOps VecOps;

struct _p_Vec {
  PETSCHEADER(struct _VecOps);
  PetscLayout map;
  void *data;
  PetscBool petscnative;          /* PETSC_FALSE: host access goes through ops->getarray / restorearray (vecimpl.h:375-434) */
  /* norm cache keyed on state (rvector.c:205-224; the reference keeps it in composed data) */
  int norm_state[4];
  PetscReal norm_val[4];
};

/* ---- Mat ---- */
struct _MatOps {
  PetscErrorCode (*setvalues)(Mat, PetscInt, const PetscInt[], PetscInt, const PetscInt[], const PetscScalar[], InsertMode);
  PetscErrorCode (*mult)(Mat, Vec, Vec);                 /* slot 3 */
  PetscErrorCode (*multadd)(Mat, Vec, Vec, Vec);         /* slot 4 */
  PetscErrorCode (*multtranspose)(Mat, Vec, Vec);        /* slot 5 */
  PetscErrorCode (*multtransposeadd)(Mat, Vec, Vec, Vec);/* slot 6 */
  PetscErrorCode (*getdiagonal)(Mat, Vec);               /* slot 17 */
  PetscErrorCode (*diagonalscale)(Mat, Vec, Vec);        /* slot 18 */
  PetscErrorCode (*assemblybegin)(Mat, MatAssemblyType);
  PetscErrorCode (*assemblyend)(Mat, MatAssemblyType);   /* slot 21 */
  PetscErrorCode (*zeroentries)(Mat);                    /* slot 23 */
  PetscErrorCode (*setup)(Mat);                          /* slot 29 */
  PetscErrorCode (*scale)(Mat, PetscScalar);
  PetscErrorCode (*duplicate)(Mat, MatDuplicateOption, Mat *);   /* slot 34 */
  PetscErrorCode (*setfromoptions)(Mat);                 /* slot 76 */
  PetscErrorCode (*destroy)(Mat);                        /* slot 60 */
  PetscErrorCode (*getvecs)(Mat, Vec *, Vec *);          /* slot 88 */
  PetscErrorCode (*setvaluesbatch)(Mat, PetscInt, PetscInt, PetscInt[], const PetscScalar[]);
  /* the factorisation slots PCILU / PCICC drive (matimpl.h: solve 8, lufactornumeric 30/..., ilufactorsymbolic, iccfactorsymbolic) */
  PetscErrorCode (*solve)(Mat, Vec, Vec);
  PetscErrorCode (*lufactornumeric)(Mat, Mat, const MatFactorInfo *);
  PetscErrorCode (*choleskyfactornumeric)(Mat, Mat, const MatFactorInfo *);
  PetscErrorCode (*ilufactorsymbolic)(Mat, Mat, IS, IS, const MatFactorInfo *);
  PetscErrorCode (*iccfactorsymbolic)(Mat, Mat, IS, const MatFactorInfo *);
};
typedef struct _MatOps MatOps;

struct _p_Mat {
  PETSCHEADER(struct _MatOps);
  char pending_type[32];      /* MatSetType() before the sizes are known (allowed, matreg.c): applied by MatSetSizes()/MatLoad() */
  PetscLayout rmap, cmap;
  PetscInt m_req, n_req, M_req, N_req;   /* MatSetSizes arguments */
  PetscBool assembled, was_assembled, preallocated;
  MatFactorType factortype;   /* MAT_FACTOR_NONE for an operator (matimpl.h:306) */
  void *data;
  void *spptr;        /* for the implementation's accelerator mirror, as Mat->spptr (matimpl.h:323) */
};

/* ---- PC / KSP ---- */
struct _PCOps {
  PetscErrorCode (*setup)(PC);
  PetscErrorCode (*apply)(PC, Vec, Vec);
  PetscErrorCode (*setfromoptions)(PC);
  PetscErrorCode (*destroy)(PC);
  PetscErrorCode (*getfactoredmatrix)(PC, Mat *);
  PetscErrorCode (*setuponblocks)(PC);     /* PCSetUpOnBlocks, precon.c:857: what KSPSolve asks for after KSPSetUp (itfunc.c:377) */
};
typedef struct _PCOps PCOps;
struct _p_PC {
  PETSCHEADER(struct _PCOps);
  Mat mat, pmat;
  int setupcalled;
  void *data;
};

struct _KSPOps {
  PetscErrorCode (*setup)(KSP);
  PetscErrorCode (*solve)(KSP);
  PetscErrorCode (*setfromoptions)(KSP);
  PetscErrorCode (*destroy)(KSP);
};
typedef struct _KSPOps KSPOps;
struct _p_KSP {
  PETSCHEADER(struct _KSPOps);
  PC pc;
  Vec vec_sol, vec_rhs;
  Vec *work; PetscInt nwork;
  PetscReal rtol, abstol, divtol, ttol, rnorm0, rnorm;
  PetscInt max_it, its, chknorm;
  PetscBool guess_zero;
  KSPNormType normtype;
  PCSide pc_side;
  PetscInt normsupporttable[KSP_NORM_MAX][PC_SIDE_MAX];   /* KSPSetSupportedNorm (kspimpl.h:52, itcreate.c:309-372) */
  PetscErrorCode (*converged)(KSP, PetscInt, PetscReal, KSPConvergedReason *, void *);   /* kspimpl.h:87 */
  void *cnvP;
  KSPConvergedReason reason;
  int setupcalled;
  PetscReal *res_hist; PetscInt res_hist_len, res_hist_max; PetscBool res_hist_reset; PetscReal *res_hist_alloc;
  PetscErrorCode (*monitor)(KSP, PetscInt, PetscReal, void *); void *mctx;
  PetscBool printreason;   /* -ksp_converged_reason */
  void *data;
};
PetscErrorCode KSPDefaultConverged(KSP ksp, PetscInt n, PetscReal rnorm, KSPConvergedReason *reason, void *ctx);   /* iterativ.c:702 */
PetscErrorCode KSPSkipConverged(KSP ksp, PetscInt n, PetscReal rnorm, KSPConvergedReason *reason, void *ctx);      /* iterativ.c:536 */
PetscErrorCode KSPSetSupportedNorm(KSP ksp, KSPNormType normtype, PCSide pcside, PetscInt priority);               /* itcreate.c:309 */
PetscErrorCode KSPMonitor(KSP ksp, PetscInt it, PetscReal rnorm);
PetscErrorCode KSPLogResidualHistory(KSP ksp, PetscReal norm);
PetscErrorCode KSPDefaultGetWork(KSP ksp, PetscInt nw);
PetscErrorCode KSPInitialResidual(KSP ksp, Vec vsoln, Vec vt1, Vec vt2, Vec vres, Vec vb);
PetscErrorCode KSP_MatMult(KSP ksp, Mat A, Vec x, Vec y);
PetscErrorCode KSP_PCApply(KSP ksp, Vec x, Vec y);
PetscErrorCode KSP_PCApplyBAorAB(KSP ksp, Vec x, Vec y, Vec w);
PetscErrorCode KSPCreate_CG(KSP), KSPCreate_GROPPCG(KSP), KSPCreate_PIPECG(KSP), KSPCreate_GMRES(KSP), KSPCreate_BCGS(KSP), KSPCreate_PREONLY(KSP);
PetscErrorCode PCCreate_None(PC), PCCreate_Jacobi(PC), PCCreate_BJacobi(PC), PCCreate_ILU(PC), PCCreate_ICC(PC);

#include "petsckrylovfused.h"   /* the optional fused-kernel tables a Vec / Mat type may compose */

#endif
