/* Object-model flavour header for building the plugin (petsc-dev_amd/host/*.c) INSIDE a PETSc 3.3 tree
 * (-DPETSCHIPMI355X_WITH_PETSC): PETSc's own private headers instead of the harness's stand-ins, and the handful of
 * names through which the plugin's sources reach what differs between the two object models.
 * Selected by petsc-dev_amd/host/hipmi355ximpl.h; see INTEGRATION.md.
 *
 * What the sources use unchanged in both flavours (checked against this tree's headers by
 * tests/test_integration_shim.py): ((PetscObject)obj)->comm / type_name / state / prefix, v->map->{n,N,rstart,rend,range},
 * v->data, v->petscnative, A->rmap / cmap, A->data, A->spptr, A->assembled / was_assembled / preallocated, every
 * v->ops-> / A->ops-> slot assigned by the constructors, PetscObjectComposeFunction / QueryFunction,
 * PetscObjectChangeTypeName, PetscLayoutReference, PetscMalloc / CHKERRQ / PetscLogFlops, PetscOptionsGetString / GetInt. */
#ifndef HIPMI355X_PETSC33_H
#define HIPMI355X_PETSC33_H
#include <petsc-private/vecimpl.h>     /* struct _p_Vec, struct _VecOps, PetscLayout, VecGetArray hooks (vecimpl.h:221-294,339-434) */
#include <petsc-private/matimpl.h>     /* struct _p_Mat, struct _MatOps (matimpl.h:17-188,300-330) */
#include <petsc-private/pcimpl.h>
#include <petsc-private/kspimpl.h>    /* struct _p_KSP, KSPDefaultGetWork, KSPLogResidualHistory, KSP_MatMult / KSP_PCApply (kspimpl.h:8-110,131-191): host/kspfused.c */

/* variadic SETERRQ: PETSc 3.3 spells the argument count (SETERRQ1..8, petscerror.h:120-212); PetscError itself is variadic */
#undef SETERRQ
#define SETERRQ(comm, n, ...) return PetscError(comm, __LINE__, PETSC_FUNCTION_NAME, __FILE__, __SDIR__, n, PETSC_ERROR_INITIAL, __VA_ARGS__)
#define HipFree(p) ((void)PetscFree(p))
/* PetscOptionsGetString / GetInt take the prefix first in both; the harness's NULL prefix is PETSC_NULL here */

/* object header (petsc-private/petscimpl.h:60-112) */
#define HipObjComm(obj)     (((PetscObject)(obj))->comm)
#define HipObjTypeName(obj) (((PetscObject)(obj))->type_name)
#define HipObjPrefix(obj)   (((PetscObject)(obj))->prefix ? ((PetscObject)(obj))->prefix : "")
#define HipObjState(obj)    (((PetscObject)(obj))->state)
#define HipStateIncrease(obj) ((void)PetscObjectStateIncrease((PetscObject)(obj)))

/* communicator: MPI for the host-side set-up collectives, RCCL communicators cached on the MPI communicator as an
 * attribute (hipmi355xcomm.c) */
typedef struct _n_HipCommData *HipCommData;
struct _n_HipCommData { int size, rank; void *dcomm, *dcomm_halo; };
PetscErrorCode HipCommGetData(MPI_Comm comm, HipCommData *d);          /* creates + caches on first use */
int HipCommSize(MPI_Comm comm);
int HipCommRank(MPI_Comm comm);
void *HipCommDevice_(MPI_Comm comm, int halo);
#define HipCommDevice(comm)     ((mi355x_comm_t)HipCommDevice_(comm, 0))
#define HipCommDeviceHalo(comm) ((mi355x_comm_t)HipCommDevice_(comm, 1))
int HipCommAllgather(MPI_Comm comm, const void *sbuf, int nbytes, void *rbuf);                   /* MPI_Allgather of bytes */
int HipCommAllreduce(MPI_Comm comm, void *buf, int count, int is_double, int op /*0 sum,1 max,2 min*/);   /* MPI_Allreduce in place */
#define HipCommHasExchange(comm) 1
int HipCommExchange(MPI_Comm comm, int ns, const int *speers, void *const *sbufs, const int *sbytes,
                    int nr, const int *rpeers, void *const *rbufs, const int *rbytes);           /* MPI_Irecv / MPI_Isend / MPI_Waitall */

/* the reference's generic names the harness flavour registers as aliases are PETSc's own CPU types here */
#define VECSEQ "seq"
#define VECMPI "mpi"
#define VECSTANDARD "standard"
#endif
