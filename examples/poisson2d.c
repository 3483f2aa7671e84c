/* A standalone C program on the host library's PETSc-named API: the kind of user code that runs on the reference and,
 * after switching the header and the two type names, on this library.  It solves the 5-point Laplacian on an m x n
 * grid with right-hand side A*1 (the problem of the reference's KSP tutorial ex2, so its output files apply) and prints
 * what that tutorial prints.  Every "-option value" on the command line goes to the options database
 * (-ksp_type, -pc_type, -ksp_monitor_short, -ksp_gmres_cgs_refinement_type ...); -m and -n set the grid.
 *
 *   gcc -O2 -I../include poisson2d.c -L../petsc-dev_amd/host -lpetschipmi355x -Wl,-rpath,... -lm -o poisson2d
 *   ./poisson2d -m 5 -n 5 -ksp_monitor_short -ksp_gmres_cgs_refinement_type refine_always
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "petschipmi355x.h"
#include "petscmini_mpinames.h"   /* MPI_Comm spelling on the harness */

#define CHK(call) do { PetscErrorCode e_ = (call); if (e_) { fprintf(stderr, "error %d: %s\n", (int)e_, PetscGetLastErrorMessage()); return 1; } } while (0)

int main(int argc, char **argv) {
  PetscInt m = 8, n = 7, its;
  PetscReal norm;
  Mat A;
  Vec x, b, u;
  KSP ksp;

  CHK(PetscHIPMI355XInitialize(-1));
  for (int k = 1; k < argc; k++) {
    if (argv[k][0] != '-') continue;
    const char *val = (k + 1 < argc && (argv[k + 1][0] != '-' || (argv[k + 1][1] >= '0' && argv[k + 1][1] <= '9') || argv[k + 1][1] == '.')) ? argv[k + 1] : NULL;
    if (!strcmp(argv[k], "-m") && val) m = atoi(val);
    else if (!strcmp(argv[k], "-n") && val) n = atoi(val);
    else CHK(PetscOptionsSetValue(argv[k], val ? val : ""));
    if (val) k++;
  }

  /* operator: one MatSetValues call per grid point, at most five entries */
  CHK(MatCreate(PETSC_COMM_WORLD, &A));
  CHK(MatSetSizes(A, PETSC_DECIDE, PETSC_DECIDE, m * n, m * n));
  CHK(MatSetType(A, MATAIJHIPMI355X));
  CHK(MatSetFromOptions(A));
  CHK(MatSetUp(A));
  for (PetscInt row = 0; row < m * n; row++) {
    const PetscInt gi = row / n, gj = row % n;
    PetscInt cols[5], nc = 0;
    PetscScalar vals[5];
    if (gi > 0)     { cols[nc] = row - n; vals[nc++] = -1.0; }
    if (gi < m - 1) { cols[nc] = row + n; vals[nc++] = -1.0; }
    if (gj > 0)     { cols[nc] = row - 1; vals[nc++] = -1.0; }
    if (gj < n - 1) { cols[nc] = row + 1; vals[nc++] = -1.0; }
    cols[nc] = row; vals[nc++] = 4.0;
    CHK(MatSetValues(A, 1, &row, nc, cols, vals, ADD_VALUES));
  }
  CHK(MatAssemblyBegin(A, MAT_FINAL_ASSEMBLY));
  CHK(MatAssemblyEnd(A, MAT_FINAL_ASSEMBLY));

  CHK(MatGetVecs(A, &u, &b));
  CHK(VecDuplicate(b, &x));
  CHK(VecSet(u, 1.0));
  CHK(MatMult(A, u, b));                       /* exact solution: all ones */

  CHK(KSPCreate(PETSC_COMM_WORLD, &ksp));
  CHK(KSPSetOperators(ksp, A, A, DIFFERENT_NONZERO_PATTERN));
  CHK(KSPSetTolerances(ksp, 1.e-2 / ((m + 1) * (n + 1)), 1.e-50, PETSC_DEFAULT, PETSC_DEFAULT));
  CHK(KSPSetFromOptions(ksp));
  CHK(KSPSolve(ksp, b, x));

  CHK(VecAXPY(x, -1.0, u));
  CHK(VecNorm(x, NORM_2, &norm));
  CHK(KSPGetIterationNumber(ksp, &its));
  printf("Norm of error %g iterations %d\n", (double)norm, (int)its);

  CHK(KSPDestroy(&ksp));
  CHK(VecDestroy(&u)); CHK(VecDestroy(&x)); CHK(VecDestroy(&b));
  CHK(MatDestroy(&A));
  CHK(PetscHIPMI355XFinalize());
  return 0;
}
