// Level-scheduled sparse triangular solves for the ILU(0) factors (MatSolve_SeqAIJ_NaturalOrdering,
// reference src/mat/impls/aij/seq/aijfact.c:3126-3172, factor layout of :1628-1700: L rows forward, U rows
// stored from the last row backwards, each followed by its inverted diagonal at bdiag[i]).
// Rows of one level are independent; one lane per row subtracts its products in column order, exactly the
// order of PetscSparseDenseMinusDot (aij.h:337-339), so every x[i] carries the reference's bits.
#include "common.hpp"

__global__ __launch_bounds__(MI355X_BLOCK) void ilu0_lower_level_kernel(int nrows, const int *__restrict__ rows,
                                                                       const int *__restrict__ bi,
                                                                       const int *__restrict__ bj,
                                                                       const double *__restrict__ ba,
                                                                       const double *b, double *x) {
  const int t = blockIdx.x * MI355X_BLOCK + threadIdx.x;
  if (t >= nrows) return;
  const int i = rows[t];
  double sum = b[i];
  for (int q = bi[i]; q < bi[i + 1]; ++q) sum -= ba[q] * x[bj[q]];
  x[i] = sum;
}

__global__ __launch_bounds__(MI355X_BLOCK) void ilu0_upper_level_kernel(int nrows, const int *__restrict__ rows,
                                                                       const int *__restrict__ bj,
                                                                       const double *__restrict__ ba,
                                                                       const int *__restrict__ bdiag, double *x) {
  const int t = blockIdx.x * MI355X_BLOCK + threadIdx.x;
  if (t >= nrows) return;
  const int i = rows[t];
  const int s0 = bdiag[i + 1] + 1, nz = bdiag[i] - bdiag[i + 1] - 1;
  double sum = x[i];
  for (int q = 0; q < nz; ++q) sum -= ba[s0 + q] * x[bj[s0 + q]];
  x[i] = sum * ba[s0 + nz];
}

extern "C" {

int mi355x_ilu0_lower_level(mi355x_handle_t h, int nrows, const int *rows, const int *bi, const int *bj,
                            const double *ba, const double *b, double *x) {
  if (nrows <= 0) return 0;
  hipLaunchKernelGGL(ilu0_lower_level_kernel, dim3((nrows + MI355X_BLOCK - 1) / MI355X_BLOCK), dim3(MI355X_BLOCK), 0,
                     h->stream, nrows, rows, bi, bj, ba, b, x);
  MI355X_LAUNCH_CHECK();
  return 0;
}

int mi355x_ilu0_upper_level(mi355x_handle_t h, int nrows, const int *rows, const int *bj, const double *ba,
                            const int *bdiag, double *x) {
  if (nrows <= 0) return 0;
  hipLaunchKernelGGL(ilu0_upper_level_kernel, dim3((nrows + MI355X_BLOCK - 1) / MI355X_BLOCK), dim3(MI355X_BLOCK), 0,
                     h->stream, nrows, rows, bj, ba, bdiag, x);
  MI355X_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
