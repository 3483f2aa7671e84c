/*
 * gen_oracle.c -- TEST INFRASTRUCTURE.  Generators for the benchmark operators named in SURVEY.md
 * section 8 (they build inputs; they are not part of the reference path).
 */
#include "oracle.h"

/* P7(nx,ny,nz): 3-D 7-point Laplacian, natural ordering (i fastest), diagonal 6, off-diagonals -1 at
 * +-1, +-nx, +-nx*ny truncated at the faces -- the 3-D analogue of the stencil in
 * src/ksp/ksp/examples/tutorials/ex2.c:96-103.  Rows [rstart,rend) only, global column indices,
 * columns ascending.  ai has (rend-rstart)+1 entries.  Returns nnz. Pass aj=aa=NULL to count. */
long orc_gen_p7(int nx, int ny, int nz, long rstart, long rend, int *ai, int *aj, double *aa);
long orc_gen_p7(int nx, int ny, int nz, long rstart, long rend, int *ai, int *aj, double *aa) {
  long nnz = 0;
  const long plane = (long)nx * ny;
  (void)nz;
  if (ai) ai[0] = 0;
  for (long r = rstart; r < rend; r++) {
    const long k = r / plane, rem = r - k * plane, j = rem / nx, i = rem - j * nx;
#define PUT(c, v) do { if (aj) { aj[nnz] = (int)(c); aa[nnz] = (v); } nnz++; } while (0)
    if (k > 0) PUT(r - plane, -1.0);
    if (j > 0) PUT(r - nx, -1.0);
    if (i > 0) PUT(r - 1, -1.0);
    PUT(r, 6.0);
    if (i < nx - 1) PUT(r + 1, -1.0);
    if (j < ny - 1) PUT(r + nx, -1.0);
    if (k < nz - 1) PUT(r + plane, -1.0);
#undef PUT
    if (ai) ai[r - rstart + 1] = (int)nnz;
  }
  return nnz;
}
