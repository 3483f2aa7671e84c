/* Source fragment, included by petsc-dev_amd/host/mpiaijhip.c when built inside a PETSc 3.3 tree.  MATMPIAIJHIPMI355X as a
 * subclass of MATMPIAIJ the way MatCreate_MPIAIJCUSP does it (src/mat/impls/aij/mpi/mpicusp/mpiaijcusp.cu:204-235):
 * parent constructor, "MatMPIAIJSetPreallocation_C" recomposed so that the two blocks are created as the sequential GPU
 * type (mpiaijcusp.cu:12-58), slots overridden, type name changed.  The parent owns Mat_MPIAIJ (mpiaij.h:35-77), the
 * assembly, the stash and MatSetUpMultiply_MPIAIJ; after each assembly this type aliases A, B, garray into its own
 * HipMPIAIJ (A->spptr) and builds, from garray, the device-side halo scatter that MatMult uses instead of Mvctx
 * (the reference's CUSP path patched vpscat.h for that: VecScatterInitializeForGPU, vscatcusp.c:29-112). */
#include <../src/mat/impls/aij/mpi/mpiaij.h>

static PetscErrorCode (*mpiaij_parent_assemblyend)(Mat, MatAssemblyType);
static PetscErrorCode (*mpiaij_parent_destroy)(Mat);

static PetscErrorCode MatMPIAIJSetPreallocation_MPIAIJHIPMI355X(Mat B, PetscInt d_nz, const PetscInt d_nnz[], PetscInt o_nz, const PetscInt o_nnz[]) {
  Mat_MPIAIJ *b = (Mat_MPIAIJ *)B->data;
  PetscErrorCode ierr;
  PetscFunctionBegin;
  if (d_nz == PETSC_DEFAULT || d_nz == PETSC_DECIDE) d_nz = 5;
  if (o_nz == PETSC_DEFAULT || o_nz == PETSC_DECIDE) o_nz = 2;
  if (d_nz < 0) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_ARG_OUTOFRANGE, "d_nz cannot be less than 0: value %D", d_nz);
  if (o_nz < 0) SETERRQ(PETSC_COMM_SELF, PETSC_ERR_ARG_OUTOFRANGE, "o_nz cannot be less than 0: value %D", o_nz);
  ierr = PetscLayoutSetUp(B->rmap);CHKERRQ(ierr);
  ierr = PetscLayoutSetUp(B->cmap);CHKERRQ(ierr);
  if (!B->preallocated) {             /* explicitly create the two blocks as MATSEQAIJHIPMI355X (mpiaijcusp.cu:36-46) */
    ierr = MatCreate(PETSC_COMM_SELF, &b->A);CHKERRQ(ierr);
    ierr = MatSetSizes(b->A, B->rmap->n, B->cmap->n, B->rmap->n, B->cmap->n);CHKERRQ(ierr);
    ierr = MatSetType(b->A, MATSEQAIJHIPMI355X);CHKERRQ(ierr);
    ierr = PetscLogObjectParent(B, b->A);CHKERRQ(ierr);
    ierr = MatCreate(PETSC_COMM_SELF, &b->B);CHKERRQ(ierr);
    ierr = MatSetSizes(b->B, B->rmap->n, B->cmap->N, B->rmap->n, B->cmap->N);CHKERRQ(ierr);
    ierr = MatSetType(b->B, MATSEQAIJHIPMI355X);CHKERRQ(ierr);
    ierr = PetscLogObjectParent(B, b->B);CHKERRQ(ierr);
  }
  ierr = MatSeqAIJSetPreallocation(b->A, d_nz, d_nnz);CHKERRQ(ierr);
  ierr = MatSeqAIJSetPreallocation(b->B, o_nz, o_nnz);CHKERRQ(ierr);
  B->preallocated = PETSC_TRUE;
  PetscFunctionReturn(0);
}

static PetscErrorCode MatAssemblyEnd_MPIAIJHIPMI355X(Mat A, MatAssemblyType mode) {
  Mat_MPIAIJ *b = (Mat_MPIAIJ *)A->data;
  HipMPIAIJ *h = MA(A);
  PetscErrorCode ierr;
  PetscFunctionBegin;
  ierr = (*mpiaij_parent_assemblyend)(A, mode);CHKERRQ(ierr);       /* MatAssemblyEnd_MPIAIJ, mpiaij.c:654-720: stash, blocks, MatSetUpMultiply */
  if (mode == MAT_FLUSH_ASSEMBLY) PetscFunctionReturn(0);
  h->A = b->A; h->B = b->B; h->garray = b->garray; h->ec = b->B->cmap->n;
  h->rstart = A->rmap->rstart; h->rend = A->rmap->rend; h->cstart = A->cmap->rstart; h->cend = A->cmap->rend;
  ierr = MatSeqAIJHIPSetCompressedRow(h->B, PETSC_TRUE);CHKERRQ(ierr);
  if (!h->hscat) {                                                    /* first assembly (a later pattern change goes through MatDisAssemble_MPIAIJ and a new Mat) */
    ierr = VecCreate(PETSC_COMM_SELF, &h->lvec);CHKERRQ(ierr);
    ierr = VecSetSizes(h->lvec, h->ec, h->ec);CHKERRQ(ierr);
    ierr = VecSetType(h->lvec, VECSEQHIPMI355X);CHKERRQ(ierr);
    ierr = HipScatterCreate_PtoS_MPIAIJ(HipObjComm(A), A->cmap, h->ec, h->garray, &h->hscat);CHKERRQ(ierr);
  }
  PetscFunctionReturn(0);
}

static PetscErrorCode MatDestroy_MPIAIJHIPMI355X(Mat A) {
  HipMPIAIJ *h = MA(A);
  PetscErrorCode ierr;
  PetscFunctionBegin;
  if (h) {
    ierr = HipScatterDestroy(&h->hscat);CHKERRQ(ierr);
    ierr = VecDestroy(&h->lvec);CHKERRQ(ierr);
    ierr = PetscFree(A->spptr);CHKERRQ(ierr);
    A->spptr = 0;
  }
  ierr = (*mpiaij_parent_destroy)(A);CHKERRQ(ierr);                  /* MatDestroy_MPIAIJ frees A, B, garray, lvec, Mvctx */
  PetscFunctionReturn(0);
}

EXTERN_C_BEGIN
PetscErrorCode MatCreate_MPIAIJHIPMI355X(Mat B) {
  PetscErrorCode ierr;
  HipMPIAIJ *h;
  PetscFunctionBegin;
  ierr = MatCreate_MPIAIJ(B);CHKERRQ(ierr);
  ierr = PetscObjectComposeFunction((PetscObject)B, "MatMPIAIJSetPreallocation_C", "MatMPIAIJSetPreallocation_MPIAIJHIPMI355X",
                                    (PetscVoidFunction)MatMPIAIJSetPreallocation_MPIAIJHIPMI355X);CHKERRQ(ierr);
  mpiaij_parent_assemblyend = B->ops->assemblyend;
  mpiaij_parent_destroy = B->ops->destroy;
  ierr = PetscNewLog(B, HipMPIAIJ, &h);CHKERRQ(ierr);
  B->spptr = h;
  B->ops->mult             = MatMult_MPIAIJHIP;
  B->ops->multadd          = MatMultAdd_MPIAIJHIP;
  B->ops->multtranspose    = MatMultTranspose_MPIAIJHIP;
  B->ops->multtransposeadd = MatMultTransposeAdd_MPIAIJHIP;
  B->ops->diagonalscale    = MatDiagonalScale_MPIAIJHIP;
  B->ops->assemblyend      = MatAssemblyEnd_MPIAIJHIPMI355X;
  B->ops->destroy          = MatDestroy_MPIAIJHIPMI355X;
  B->ops->getvecs          = MatGetVecs_HIPMI355X;
  ierr = PetscObjectChangeTypeName((PetscObject)B, MATMPIAIJHIPMI355X);CHKERRQ(ierr);
  PetscFunctionReturn(0);
}
EXTERN_C_END
