/* Operator generators for the benchmark/test drivers (the assembly loops of the reference's example
 * programs, done in bulk): they only build inputs for MatCreate{Seq,MPI}AIJWithArrays. */
#include "petscimpl.h"

/* 3-D 7-point Laplacian "P7" (SURVEY.md 8): nx*ny*nz rows, natural ordering (i fastest), diagonal 6,
 * -1 at +-1, +-nx, +-nx*ny truncated at the faces -- the 3-D analogue of the 5-point stencil loop in
 * src/ksp/ksp/examples/tutorials/ex2.c:96-103.  Rows [rstart,rend), global column indices ascending.
 * Pass aj = aa = NULL to count.  Returns nnz through *nnz_out. */
PetscErrorCode PetscMiniGenPoisson7(PetscInt nx, PetscInt ny, PetscInt nz, long rstart, long rend, PetscInt *ai, PetscInt *aj, PetscScalar *aa, long *nnz_out) {
  long nnz = 0;
  const long plane = (long)nx * ny;
  if ((long)nx * ny * nz > 2147483647L) SETERRQ(0, PETSC_ERR_ARG_OUTOFRANGE, "grid %d x %d x %d exceeds 32-bit PetscInt", nx, ny, nz);
  if (ai) ai[0] = 0;
  for (long r = rstart; r < rend; r++) {
    const long k = r / plane, rem = r - k * plane, j = rem / nx, i = rem - j * nx;
#define PUT(c, v) do { if (aj) { aj[nnz] = (PetscInt)(c); aa[nnz] = (v); } nnz++; } while (0)
    if (k > 0) PUT(r - plane, -1.0);
    if (j > 0) PUT(r - nx, -1.0);
    if (i > 0) PUT(r - 1, -1.0);
    PUT(r, 6.0);
    if (i < nx - 1) PUT(r + 1, -1.0);
    if (j < ny - 1) PUT(r + nx, -1.0);
    if (k < nz - 1) PUT(r + plane, -1.0);
#undef PUT
    if (ai) ai[r - rstart + 1] = (PetscInt)nnz;
  }
  if (nnz_out) *nnz_out = nnz;
  return 0;
}
