/* The optional fused-kernel interface between a Krylov driver and the Vec / Mat types it runs on.  A type that has such
 * kernels composes them on its objects under the names below (PetscObjectComposeFunction, the reference's mechanism for
 * type-specific methods: e.g. "MatMPIAIJSetPreallocation_C", mpiaij.c:5421); a driver that knows them queries by name
 * (PetscObjectQueryFunction) and falls back to the reference's op-by-op sequence when the name is absent.  The reference's own
 * KSPSolve_CG / _GMRES / _BCGS never ask, and run op by op over the same types.  Every fused form is bit-identical to the calls it
 * replaces.  Included by the harness drivers and by the plug-in; needs Vec / Mat / PetscScalar declared before it. */
#if !defined(PETSCKRYLOVFUSED_H)
#define PETSCKRYLOVFUSED_H

/* ---- optional, type-specific methods the drivers look up by name (PetscObjectQueryFunction), never link against ----
 * "VecKrylovFusedOps_C" on a Vec returns a table of fused Krylov kernels (several BLAS-1 calls of KSPSolve_CG / _BCGS in
 * one sweep, results bit-identical to the separate calls); absent -> the drivers run the reference's op-by-op sequence.
 * "MatMultTDotBegin_C" on a Mat: y = A x with x'y left on the device for the fused CG update.
 * "MatMultDiagonalScale_C" on a Mat: y = d .* (A x), MatMult followed by PCApply_Jacobi, in one kernel. */
typedef struct {
  PetscErrorCode (*cg_update)(Vec x, Vec r, Vec z, Vec p, Vec w, Vec d, PetscScalar a, PetscScalar *zz, PetscScalar *zr, PetscScalar *rr, PetscBool *done);
  PetscErrorCode (*cg_update_check)(Vec x, Vec r, Vec z, Vec p, Vec w, Vec d, PetscBool *ok);
  PetscErrorCode (*tdot_begin)(Vec x, Vec y, PetscBool *ok);
  PetscErrorCode (*cg_update_dev_begin)(Vec x, Vec r, Vec z, Vec p, Vec w, Vec d, PetscScalar beta, PetscScalar dpiold, PetscBool check_sign);
  PetscErrorCode (*cg_update_dev_end)(Vec x, PetscScalar *zz, PetscScalar *zr, PetscScalar *rr, PetscScalar *dpi);
  PetscErrorCode (*aypx_dev)(Vec p, PetscScalar den, Vec z);
  PetscErrorCode (*pmult_dot)(Vec w, Vec x, Vec d, Vec y, PetscScalar *val, PetscBool *done);
  PetscErrorCode (*pmult_dotnorm2)(Vec w, Vec x, Vec d, Vec s, PetscScalar *dp, PetscReal *nm, PetscBool *done);
  PetscErrorCode (*bcgs_update)(Vec x, Vec r, Vec p, Vec s, Vec t, Vec rp, PetscScalar alpha, PetscScalar omega, PetscScalar *rr, PetscScalar *rho, PetscBool *done);
  /* KSPGMRESClassicalGramSchmidtOrthogonalization without refinement (borthog2.c:60-66) and the VecNormalize that follows it
   * (gmres.c:146): dots[j] = <w, V[j]>, w -= sum_j dots[j] V[j], *nrm = |w|, w /= |w|; one host wait instead of three */
  PetscErrorCode (*gmres_orthog_normalize)(Vec w, PetscInt nv, const Vec V[], PetscScalar *dots, PetscReal *nrm, PetscBool *done);
} VecKrylovFusedOps;
typedef const VecKrylovFusedOps *(*VecKrylovFusedOpsGetFn)(void);
typedef PetscErrorCode (*MatMultTDotBeginFn)(Mat A, Vec x, Vec y, PetscBool *ok);
typedef PetscErrorCode (*MatMultDiagonalScaleFn)(Mat A, Vec d, Vec x, Vec y, PetscBool *ok);

#endif
