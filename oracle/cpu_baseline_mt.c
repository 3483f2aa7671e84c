/* TEST INFRASTRUCTURE (see oracle.h): the CPU baseline bench.py times next to the GPU number.
 *
 * The reference runs this path under MPI with one rank per core: every rank owns a contiguous block of rows
 * (PetscLayout, vecimpl.h:21-32), MatMult_MPIAIJ multiplies its rows (mpiaij.c:1102-1116), VecDot/VecNorm reduce a
 * per-rank partial with MPI_Allreduce (pbvec.c:9-35, pvec2.c:46-83), and KSPSolve_CG (cg.c:92-286) + PCApply_Jacobi
 * (jacobi.c:266-277) drive them.  This file restates exactly that arrangement with one thread per "rank" inside one
 * process (shared x instead of a halo exchange, per-thread partials summed in rank order instead of MPI_Allreduce),
 * so the baseline can use all the host cores the GPU box gives the job.  Preconditioned-norm CG, zero initial guess,
 * a fixed number of iterations (no convergence test inside the timed loop).  Returns the seconds the iterations took. */
#define _POSIX_C_SOURCE 199309L
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
#include <omp.h>
#include "oracle.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

double orc_cg_jacobi_mt(int n, const int *ai, const int *aj, const double *aa, const double *b, int its, int nthreads,
                        double *x, double *rnorm_out) {
  double *r = malloc(sizeof(double) * n), *z = malloc(sizeof(double) * n), *p = malloc(sizeof(double) * n);
  double *d = malloc(sizeof(double) * n), *part = calloc((size_t)nthreads * 8, sizeof(double));
  double t_elapsed = 0.0, dp = 0.0;
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
  {
    const int t = omp_get_thread_num(), T = omp_get_num_threads();
    const int lo = (int)((long)n * t / T), hi = (int)((long)n * (t + 1) / T);   /* this "rank"'s rows */
    double beta = 0.0, betaold = 1.0, dpi, a, bb, s;
    /* set-up: Jacobi diagonal (zeros -> 1, jacobi.c:182-190), x = 0, r = b, z = D^-1 r */
    for (int i = lo; i < hi; i++) {
      double di = 0.0;
      for (int k = ai[i]; k < ai[i + 1]; k++) if (aj[k] == i) { di = aa[k]; break; }
      d[i] = (di == 0.0) ? 1.0 : 1.0 / di;
      x[i] = 0.0; r[i] = b[i]; z[i] = r[i] * d[i];
    }
    s = 0.0; for (int i = lo; i < hi; i++) s += z[i] * r[i];
    part[8 * t] = s;
#pragma omp barrier
    for (int q = 0; q < T; q++) beta += part[8 * q];
#pragma omp barrier
#pragma omp master
    t_elapsed = now();
    for (int it = 0; it < its; it++) {
      if (!it) { for (int i = lo; i < hi; i++) p[i] = z[i]; }
      else { bb = beta / betaold; for (int i = lo; i < hi; i++) p[i] = z[i] + bb * p[i]; }
#pragma omp barrier                                            /* p complete before anyone gathers from it */
      s = 0.0;
      for (int i = lo; i < hi; i++) {                           /* w = A p (kept in z, cg.c:122) and p'w */
        double sum = 0.0;
        for (int k = ai[i]; k < ai[i + 1]; k++) sum += aa[k] * p[aj[k]];
        z[i] = sum;
        s += p[i] * sum;
      }
      part[8 * t] = s;
#pragma omp barrier
      dpi = 0.0; for (int q = 0; q < T; q++) dpi += part[8 * q];
#pragma omp barrier
      betaold = beta;
      a = beta / dpi;
      double s0 = 0.0, s1 = 0.0;
      for (int i = lo; i < hi; i++) {
        x[i] += a * p[i];
        r[i] -= a * z[i];
        z[i] = r[i] * d[i];
        s0 += z[i] * z[i];
        s1 += z[i] * r[i];
      }
      part[8 * t] = s0; part[8 * t + 1] = s1;
#pragma omp barrier
      double zz = 0.0; beta = 0.0;
      for (int q = 0; q < T; q++) { zz += part[8 * q]; beta += part[8 * q + 1]; }
#pragma omp barrier
      if (t == 0) dp = sqrt(zz);
    }
#pragma omp barrier
#pragma omp master
    t_elapsed = now() - t_elapsed;
  }
  if (rnorm_out) *rnorm_out = dp;
  free(r); free(z); free(p); free(d); free(part);
  return t_elapsed;
}
