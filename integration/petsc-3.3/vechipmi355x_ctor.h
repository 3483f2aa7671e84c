/* Source fragment, included by petsc-dev_amd/host/vechip.c when built inside a PETSc 3.3 tree (it needs that file's static
 * ops in scope).  The part of the Vec constructors that is specific to PETSc's REAL struct _p_Vec / struct _VecOps
 * (include/petsc-private/vecimpl.h:221-294,339-351): the slots every Vec type has to fill and this type takes from PETSc
 * unchanged.  VecCreate_SeqCUSP reaches them by calling the parent constructor VecCreate_Seq_Private (veccusp.cu:1914);
 * this type does not carry a Vec_Seq (its data lives in HBM, the host mirror is optional), so it fills them itself.
 * The numerical slots, petscnative = PETSC_FALSE and the composed methods are set by VecCreate_HIP_common for both
 * object models. */
#include <../src/vec/vec/impls/dvecimpl.h>       /* VecGetSize_Seq, VecView_Seq */
#include <../src/vec/vec/impls/mpi/pvecimpl.h>   /* VecGetSize_MPI, VecView_MPI */

static PetscErrorCode VecAssemblyNoop_HIP(Vec v) { (void)v; return 0; }   /* VecAssemblyBegin_HIP (vechip.c) has delivered the stash already */

static PetscErrorCode VecCreate_HIP_petsc33(Vec v, PetscBool mpi) {
  PetscFunctionBegin;
  v->ops->duplicatevecs = VecDuplicateVecs_Default;
  v->ops->destroyvecs   = VecDestroyVecs_Default;
  v->ops->getsize       = mpi ? VecGetSize_MPI : VecGetSize_Seq;
  v->ops->getlocalsize  = VecGetSize_Seq;
  v->ops->view          = mpi ? VecView_MPI : VecView_Seq;     /* read through VecGetArrayRead -> ops->getarray (petscnative == PETSC_FALSE) */
  v->ops->load          = VecLoad_Default;
  v->ops->assemblybegin = VecAssemblyBegin_HIP;               /* off-process VecSetValues: the type's own stash (it carries no Vec_MPI) */
  v->ops->assemblyend   = VecAssemblyNoop_HIP;
  v->ops->dot_local     = VecDot_HIP_local;                    /* the split-phase reductions of comb.c use the _local slots */
  v->ops->tdot_local    = VecDot_HIP_local;
  v->ops->norm_local    = VecNorm_HIP_local;
  v->ops->mdot_local    = VecMDot_HIP_local;
  v->ops->mtdot_local   = VecMDot_HIP_local;
  v->array_gotten       = PETSC_FALSE;
  PetscFunctionReturn(0);
}
