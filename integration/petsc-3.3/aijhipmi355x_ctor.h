/* Source fragment, included by petsc-dev_amd/host/aijhip.c when built inside a PETSc 3.3 tree (it needs that file's static
 * ops in scope).  MATSEQAIJHIPMI355X as a subclass of MATSEQAIJ, exactly the way MatCreate_SeqAIJCUSP does it
 * (src/mat/impls/aij/seq/seqcusp/aijcusp.cu:657-681): parent constructor first, then the slots are overridden, the device
 * mirror hangs off B->spptr, inode.use = PETSC_FALSE so that MatAssemblyEnd_SeqAIJ does not re-install the CPU inode
 * routines over ops->mult (aijcusp.cu:462-466), type name changed last.  The parent owns the CSR container (aij.h:10-39)
 * and its assembly; the kernels' callers read it through a view (HipAIJ) refreshed after every assembly.
 * Coherence: ((PetscObject)A)->state is compared with the state at the last upload (matrix.c bumps it at all 47 mutators),
 * instead of the CUSP-only valid_GPU_matrix flag (matimpl.h:320-322). */
#include <../src/mat/impls/aij/seq/aij.h>

typedef struct {                      /* the parent's routines this type calls through (saved from the table MatCreate_SeqAIJ filled) */
  PetscErrorCode (*assemblyend)(Mat, MatAssemblyType);
  PetscErrorCode (*destroy)(Mat);
} HipAIJParentOps;
static HipAIJParentOps seqaij_parent;

/* view of the parent's container: same member names, no copies */
static PetscErrorCode hipaij_refresh_view(Mat A) {
  Mat_SeqAIJ *aij = (Mat_SeqAIJ *)A->data;
  HipAIJ *v = HipAIJGet(A);
  PetscInt m = A->rmap->n;
  PetscFunctionBegin;
  v->m = m; v->n = A->cmap->n;
  v->i = aij->i; v->j = aij->j; v->a = aij->a; v->ilen = aij->ilen; v->imax = aij->imax;
  v->nz = aij->nz; v->maxnz = aij->maxnz; v->bs = 1;
  v->compact = A->assembled;                       /* MatAssemblyEnd_SeqAIJ has squeezed the rows (aij.c:860-930) */
  v->nonzerorows = 0;
  for (PetscInt r = 0; r < m; r++) v->nonzerorows += (aij->i[r + 1] > aij->i[r]);
  PetscFunctionReturn(0);
}

/* lazily, from MatSeqAIJHIPUpload: the parent's container may have been filled or replaced without this type's MatAssemblyEnd
 * (MatDuplicate_SeqAIJ, MatCopy_SeqAIJ, MatConvert, MatDuplicateNoCreate_SeqAIJ set assembled themselves).  New arrays or another
 * nonzero count: the pattern is not the one the device copy was built for -- rebuild; same arrays: the object state decides. */
static PetscErrorCode hipbaij_refresh_view_if_stale(Mat A);   /* baijhipmi355x_ctor.h: the same for a matrix whose parent is MATSEQBAIJ */
static PetscErrorCode hipaij_refresh_view_if_stale(Mat A) {
  Mat_SeqAIJ *aij = (Mat_SeqAIJ *)A->data;
  HipAIJ *v = HipAIJGet(A);
  PetscErrorCode ierr;
  PetscFunctionBegin;
  if (SD(A)->baij_parent) { ierr = hipbaij_refresh_view_if_stale(A);CHKERRQ(ierr); PetscFunctionReturn(0); }
  if (!A->assembled) PetscFunctionReturn(0);
  if (v->i != aij->i || v->j != aij->j || v->a != aij->a || v->nz != aij->nz || !v->compact) {
    const PetscBool new_pattern = (PetscBool)(v->i != aij->i || v->j != aij->j || v->nz != aij->nz);
    ierr = hipaij_refresh_view(A);CHKERRQ(ierr);
    SD(A)->uploaded_state = -1;
    if (new_pattern) SD(A)->pattern_nz = -1;
  }
  PetscFunctionReturn(0);
}

static PetscErrorCode MatAssemblyEnd_SeqAIJHIPMI355X(Mat A, MatAssemblyType mode) {   /* MatAssemblyEnd_SeqAIJCUSP, aijcusp.cu:452-470 */
  PetscErrorCode ierr;
  PetscFunctionBegin;
  ierr = (*seqaij_parent.assemblyend)(A, mode);CHKERRQ(ierr);
  if (mode == MAT_FLUSH_ASSEMBLY) PetscFunctionReturn(0);
  A->assembled = PETSC_TRUE;                       /* matrix.c sets it after this slot returns; the view wants it now */
  ierr = hipaij_refresh_view(A);CHKERRQ(ierr);
  SD(A)->uploaded_state = -1;                      /* host values are newer (a same-pattern assembly re-sends `a` only) */
  PetscFunctionReturn(0);
}

static PetscErrorCode MatDestroy_SeqAIJHIPMI355X(Mat A) {   /* MatDestroy_SeqAIJCUSP, aijcusp.cu:575-590: mirror first, spptr zeroed, then the parent */
  PetscErrorCode ierr;
  PetscFunctionBegin;
  if (SD(A)) {
    Mat_SeqAIJHIP *d = SD(A);
    device_free(A);
    if (d->time_ev) { for (PetscInt k = 0; k < 2 * d->time_cap; k++) mi355x_event_destroy(d->time_ev[k]); HipFree(d->time_ev); }
    HipFree(d->view.inode_size);
    ierr = PetscFree(A->spptr);CHKERRQ(ierr);
    A->spptr = 0;
  }
  ierr = (*seqaij_parent.destroy)(A);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}

EXTERN_C_BEGIN
PetscErrorCode MatCreate_SeqAIJHIPMI355X(Mat B) {
  PetscErrorCode ierr;
  Mat_SeqAIJ *aij;
  Mat_SeqAIJHIP *d;
  PetscFunctionBegin;
  ierr = MatCreate_SeqAIJ(B);CHKERRQ(ierr);
  aij = (Mat_SeqAIJ *)B->data;
  aij->inode.use = PETSC_FALSE;                    /* this type runs its own Mat_CheckInode (seqaij_check_inode) and keeps ops->mult */
  seqaij_parent.assemblyend = B->ops->assemblyend;
  seqaij_parent.destroy = B->ops->destroy;
  ierr = PetscNewLog(B, Mat_SeqAIJHIP, &d);CHKERRQ(ierr);
  d->uploaded_state = -1; d->t_state = -1; d->pattern_nz = -1;
  B->spptr = d;
  B->ops->mult             = MatMult_SeqAIJHIP;
  B->ops->multadd          = MatMultAdd_SeqAIJHIP;
  B->ops->multtranspose    = MatMultTranspose_SeqAIJHIP;
  B->ops->multtransposeadd = MatMultTransposeAdd_SeqAIJHIP;
  B->ops->getdiagonal      = MatGetDiagonal_SeqAIJHIP;
  B->ops->scale            = MatScale_SeqAIJHIP;          /* host copy and device copy updated side by side */
  B->ops->zeroentries      = MatZeroEntries_SeqAIJHIP;
  B->ops->diagonalscale    = MatDiagonalScale_SeqAIJHIP;
  B->ops->setvaluesbatch   = MatSetValuesBatch_SeqAIJHIP;
  B->ops->setfromoptions   = MatSetFromOptions_SeqAIJHIP;  /* the type's -mat_hipmi355x_* options under the matrix's prefix (slot 76); MatSetFromOptions_SeqAIJ has none of its own in 3.3 */
  /* ops->duplicate stays MatDuplicate_SeqAIJ (aij.c:3964): it creates the new matrix with MatSetType(type_name), i.e. through THIS
   * constructor, and fills the parent's container; the view is refreshed when the copy is first used (hipaij_refresh_view_if_stale) */
  B->ops->assemblyend      = MatAssemblyEnd_SeqAIJHIPMI355X;
  B->ops->destroy          = MatDestroy_SeqAIJHIPMI355X;
  B->ops->getvecs          = MatGetVecs_HIP;
  ierr = PetscObjectComposeFunction((PetscObject)B, "MatMultTDotBegin_C", "MatMultTDotBegin_HIPMI355X", (PetscVoidFunction)MatMultTDotBegin_HIPMI355X);CHKERRQ(ierr);
  ierr = PetscObjectComposeFunction((PetscObject)B, "MatMultDiagonalScale_C", "MatMultDiagonalScale_HIPMI355X", (PetscVoidFunction)MatMultDiagonalScale_HIPMI355X);CHKERRQ(ierr);
  /* PETSc's own PCILU / PCICC / PCBJACOBI ask MatGetFactor(A, "petsc", ...): answered with a factored matrix whose ops->solve is the
   * device triangular solve (host/ilu.c), as MatCreate_SeqAIJCUSPARSE overloads the same name (aijcusparse.cu:837-840) */
  ierr = PetscObjectComposeFunction((PetscObject)B, "MatGetFactor_petsc_C", "MatGetFactor_seqaijhipmi355x_petsc", (PetscVoidFunction)MatGetFactor_seqaijhipmi355x_petsc);CHKERRQ(ierr);
  ierr = PetscObjectChangeTypeName((PetscObject)B, MATSEQAIJHIPMI355X);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}
EXTERN_C_END
