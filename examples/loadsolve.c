/* Load a matrix (and, if the file holds one, a right-hand side) in PETSc binary format and solve with the solver and
 * preconditioner named on the command line -- the flow of the reference's KSP tutorial ex10 for a staged SuiteSparse
 * matrix:   mm2petsc -fin Flan_1565.mtx -fout flan.petsc ; loadsolve -f flan.petsc -ksp_type gmres -pc_type bjacobi
 * Without a vector in the file the right-hand side is A*1. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "petschipmi355x.h"
#include "petscmini_mpinames.h"   /* MPI_Comm spelling on the harness */

#define CHK(call) do { PetscErrorCode e_ = (call); if (e_) { fprintf(stderr, "error %d: %s\n", (int)e_, PetscGetLastErrorMessage()); return 1; } } while (0)

int main(int argc, char **argv) {
  const char *file = NULL;
  Mat A; Vec b, x, u; KSP ksp; PetscViewer fd;
  PetscInt its; PetscReal rnorm, bnorm; KSPConvergedReason reason;

  CHK(PetscHIPMI355XInitialize(-1));
  for (int k = 1; k < argc; k++) {
    if (argv[k][0] != '-') continue;
    const char *val = (k + 1 < argc && (argv[k + 1][0] != '-' || (argv[k + 1][1] >= '0' && argv[k + 1][1] <= '9') || argv[k + 1][1] == '.')) ? argv[k + 1] : NULL;
    if (!strcmp(argv[k], "-f") && val) file = val;
    else CHK(PetscOptionsSetValue(argv[k], val ? val : ""));
    if (val) k++;
  }
  if (!file) { fprintf(stderr, "usage: %s -f <matrix in PETSc binary format> [-ksp_type ..] [-pc_type ..] ...\n", argv[0]); return 2; }
  CHK(PetscViewerBinaryOpen(PETSC_COMM_WORLD, file, FILE_MODE_READ, &fd));
  CHK(MatCreate(PETSC_COMM_WORLD, &A));
  CHK(MatSetType(A, MATAIJHIPMI355X));
  CHK(MatLoad(A, fd));
  CHK(MatGetVecs(A, &x, &b));
  if (VecLoad(b, fd)) {                            /* no vector behind the matrix: b = A * 1 */
    CHK(VecDuplicate(x, &u)); CHK(VecSet(u, 1.0)); CHK(MatMult(A, u, b)); CHK(VecDestroy(&u));
  }
  CHK(PetscViewerDestroy(&fd));
  CHK(KSPCreate(PETSC_COMM_WORLD, &ksp));
  CHK(KSPSetOperators(ksp, A, A, DIFFERENT_NONZERO_PATTERN));
  CHK(KSPSetFromOptions(ksp));
  CHK(KSPSolve(ksp, b, x));
  CHK(KSPGetIterationNumber(ksp, &its));
  CHK(KSPGetResidualNorm(ksp, &rnorm));
  CHK(KSPGetConvergedReason(ksp, &reason));
  /* true residual b - A x */
  CHK(VecDuplicate(b, &u)); CHK(MatMult(A, x, u)); CHK(VecAYPX(u, -1.0, b)); CHK(VecNorm(u, NORM_2, &rnorm)); CHK(VecNorm(b, NORM_2, &bnorm));
  printf("Number of iterations = %3d\n", (int)its);
  printf("Residual norm %s 1.e-6 |b| (reason %d)\n", rnorm <= 1.e-6 * bnorm ? "<" : ">=", (int)reason);
  CHK(VecDestroy(&u)); CHK(KSPDestroy(&ksp)); CHK(VecDestroy(&x)); CHK(VecDestroy(&b)); CHK(MatDestroy(&A));
  CHK(PetscHIPMI355XFinalize());
  return 0;
}
