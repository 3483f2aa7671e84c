/*
 * petschipmi355x.h -- public interface of the PLUGIN (libpetschipmi355x.so): the MI355X-native Vec/Mat types behind
 * PETSc's VecSetType / MatSetType / KSPSolve plugin surface (BASELINE.json north star),
 *     VECSEQHIPMI355X / VECMPIHIPMI355X / VECHIPMI355X
 *     MATSEQAIJHIPMI355X / MATMPIAIJHIPMI355X / MATAIJHIPMI355X , MATSEQBAIJHIPMI355X , PCILU on the device,
 * which subclass by filling the per-object function tables (struct _VecOps include/petsc-private/vecimpl.h:221-294,
 * struct _MatOps include/petsc-private/matimpl.h:17-188) and registering under a string, as the reference's own GPU
 * types do (VecCreate_SeqCUSP veccusp.cu:1905-1947, MatCreate_SeqAIJCUSP aijcusp.cu:657-681, mpiaijcusp.cu:204-235).
 * All compute goes to the HIP kernels of mi355x_kernels.h; there is no CPU compute path.
 *
 * The object model is NOT part of the plugin.  Built inside a PETSc tree (PETSCHIPMI355X_WITH_PETSC, see
 * integration/petsc-3.3/) it is PETSc's; on a box without PETSc it is the harness library (include/petscmini.h).
 */
#ifndef PETSCHIPMI355X_H
#define PETSCHIPMI355X_H
#if defined(PETSCHIPMI355X_WITH_PETSC)
#include <petscksp.h>
#else
#include "petscmini.h"
#endif
#include "petsckrylovfused.h"   /* fused-kernel tables the types compose ("VecKrylovFusedOps_C", "MatMultTDotBegin_C", "MatMultDiagonalScale_C") */
#ifdef __cplusplus
extern "C" {
#endif

#define VECSEQHIPMI355X     "seqhipmi355x"
#define VECMPIHIPMI355X     "mpihipmi355x"
#define VECHIPMI355X        "hipmi355x"
#define MATSEQAIJHIPMI355X  "seqaijhipmi355x"
#define MATMPIAIJHIPMI355X  "mpiaijhipmi355x"
#define MATAIJHIPMI355X     "aijhipmi355x"
#define MATSEQBAIJHIPMI355X "seqbaijhipmi355x"
/* the plug-in's own Krylov solvers (petsc-dev_amd/host/kspfused.c), registered with KSPRegister: -ksp_type cghipmi355x ... */
#define KSPCGHIPMI355X      "cghipmi355x"
#define KSPGMRESHIPMI355X   "gmreshipmi355x"
#define KSPBCGSHIPMI355X    "bcgshipmi355x"

/* One call after PetscInitialize (or a PetscDLLibraryRegister entry, src/sys/dll): registers the types above with
 * VecRegister / MatRegister / PCRegister.  On the harness the same constructors are also registered under the
 * reference's generic names (seq, mpi, standard, seqaij, mpiaij, aij, seqbaij; PCILU), which have no CPU implementation
 * there.  Initialize additionally binds this process to GPU `device` (-1: LOCAL_RANK env or 0); safe without a GPU --
 * host-side set-up works, the first device op reports PETSC_ERR_LIB. */
PetscErrorCode PetscHIPMI355XRegisterAll(void);
PetscErrorCode PetscHIPMI355XInitialize(int device);
PetscErrorCode PetscHIPMI355XFinalize(void);
const char    *PetscHIPMI355XVersion(void);

#if !defined(PETSCHIPMI355X_WITH_PETSC)
/* RCCL communicators (mi355x_comm.h) for the device-side halo exchange and reductions of `comm`, one per HIP stream:
 * `reduce` serves the all-reduces queued on the compute stream, `halo` the grouped send/recv (and split-phase
 * all-reduces) queued on the halo stream, so that RCCL does not serialise the two.  (With a real PETSc the plugin
 * creates both itself from the MPI communicator, integration/petsc-3.3/hipmi355xcomm.c.) */
PetscErrorCode PetscCommSetDeviceComm(PetscComm comm, void *mi355x_comm);
PetscErrorCode PetscCommSetDeviceComms(PetscComm comm, void *reduce, void *halo);
PetscErrorCode PetscCommGetDeviceTransport(PetscComm comm, int *kind /*0 single,1 RCCL,2 host-staged*/, int *nranks, int *distinct_halo_comm);
/* bench.py: microseconds per one-double ncclAllReduce on the compute stream's communicator (what VecDot_MPI / VecNorm_MPI's
 * MPI_Allreduce, pbvec.c:9-35, pvec2.c:46-83, became), synchronised after each / queued back to back.  Collective; zeros without RCCL. */
PetscErrorCode PetscCommDeviceAllreduceLatency(PetscComm comm, PetscInt reps, PetscLogDouble *sync_each_us, PetscLogDouble *back_to_back_us);

/* PETSc names the harness does not implement because only the plugin's MPIAIJ type needs them */
/* VecScatter, parallel -> sequential general: the MPIAIJ halo (src/vec/vec/utils/vpscat.c, vpscat.h) */
/* Runs the element-wise operations the Vec type has noted but not launched yet (host/vechip.c, "deferred element-wise operations":
 * an unchanged KSPSolve_CG's update sequence is recognised and run as one fused sweep; -vec_hipmi355x_defer 0 switches the noting
 * off).  Every access to a vector's storage does this by itself; the call exists for timing code. */
PetscErrorCode VecHIPMI355XFlushDeferred(void);
PetscErrorCode VecHIPMI355XSetDeferral(PetscInt on);   /* 1 / 0; negative: back to what -vec_hipmi355x_defer says */
PetscErrorCode VecScatterBegin(VecScatter ctx, Vec x, Vec y, InsertMode addv, ScatterMode mode);
PetscErrorCode VecScatterEnd(VecScatter ctx, Vec x, Vec y, InsertMode addv, ScatterMode mode);
PetscErrorCode VecScatterDestroy(VecScatter *ctx);
#endif

/* device pointer access (the analogue of VecCUSPGetArrayReadWrite, cuspvecimpl.h:95) */
PetscErrorCode VecHIPMI355XGetArray(Vec v, PetscScalar **d);
PetscErrorCode VecHIPMI355XRestoreArray(Vec v, PetscScalar **d);
PetscErrorCode VecHIPMI355XGetArrayRead(Vec v, const PetscScalar **d);

/* introspection for parity tests: the index lists of a scatter (to/from of VecScatter_MPI_General, vecimpl.h:509-537),
 * the host CSR of a SeqAIJ block, the pieces of an MPIAIJ matrix (Mat_MPIAIJ, mpiaij.h:35-77) */
PetscErrorCode VecScatterGetLists(VecScatter ctx, PetscInt *nrecv, const PetscInt **rprocs, const PetscInt **rstarts, const PetscInt **rindices,
                                  PetscInt *nsend, const PetscInt **sprocs, const PetscInt **sstarts, const PetscInt **sindices,
                                  PetscInt *nlocal, const PetscInt **lto, const PetscInt **lfrom);
PetscErrorCode MatSeqAIJGetArrays(Mat A, PetscInt *m, const PetscInt **i, const PetscInt **j, const PetscScalar **a);
PetscErrorCode MatMPIAIJGetSeqAIJ(Mat A, Mat *Ad, Mat *Ao, const PetscInt **garray);
PetscErrorCode MatMPIAIJGetScatter(Mat A, VecScatter *ctx, Vec *lvec, PetscInt *ec);
/* per-MatMult device timing (HIP events on the compute stream around the SpMV launches) for bench.py */
PetscErrorCode MatHIPMI355XSetTiming(Mat A, PetscBool on);
PetscErrorCode MatHIPMI355XGetTiming(Mat A, PetscInt *nlaunches, PetscLogDouble *total_ms);
/* the same for a MATMPIAIJHIPMI355X matrix's MatMult (mpiaij.c:1102-1116): event pairs around the diagonal-block product (compute
 * stream) AND around the halo exchange (halo stream; VecScatterBegin_1/End_1, vpscat.h:14-233): busy time of the halo stream, how much of
 * it lies inside the diagonal product, how long the compute stream waits for it afterwards; bytes this rank sends per product, to how many ranks */
PetscErrorCode MatMPIAIJHIPMI355XSetHaloTiming(Mat A, PetscBool on);
PetscErrorCode MatMPIAIJHIPMI355XGetHaloTiming(Mat A, PetscInt *nproducts, PetscLogDouble *halo_ms, PetscLogDouble *overlap_ms, PetscLogDouble *exposed_ms,
                                               PetscLogDouble *send_bytes, PetscInt *neighbours);
PetscErrorCode MatHIPMI355XGetUploadCount(Mat A, PetscInt *n);   /* value uploads host -> device of a sequential matrix so far */
PetscErrorCode MatHIPMI355XGetTransposeCounts(Mat A, PetscInt *host_builds, PetscInt *device_refreshes);   /* of the cached explicit A^T behind MatMultTranspose */
PetscErrorCode MatHIPMI355XGetInodeInfo(Mat A, PetscInt *nodes, PetscInt *groups, PetscInt *shared_indices);   /* Mat_CheckInode's node count; groups / column indices the device plan stores once per group */
PetscErrorCode MatHIPMI355XGetBlockedInfo(Mat A, PetscInt *bs, PetscInt *nblocks);   /* blocked companion of an AIJ matrix whose nodes are complete bs x bs blocks (products by the BAIJ kernel); 0, 0: not in use */
PetscErrorCode MatHIPMI355XGetTiledInfo(Mat A, PetscInt *staged, PetscInt *remainder);   /* column-tiled product (x tiles in LDS): nonzeros gathering from LDS / left to the row-block kernel; 0, 0: not in use */
PetscErrorCode MatHIPMI355XGetRowPatterns(Mat A, PetscInt *npat);   /* 0: none; else the size of the row-pattern dictionary the SpMV runs with */
PetscErrorCode MatHIPMI355XSetValuePatterns(Mat A, PetscBool on);   /* A/B switch for one matrix until its next upload (the option -mat_hipmi355x_value_patterns decides at every upload) */
PetscErrorCode VecHIPMI355XSetCGUpdateTiming(PetscBool on);   /* hipEvent pairs around the fused CG update (bench.py's roofline for that kernel) */
PetscErrorCode VecHIPMI355XGetCGUpdateTiming(PetscInt *nlaunches, PetscLogDouble *total_ms);
PetscErrorCode MatHIPMI355XGetValuePatterns(Mat A, PetscInt *nvpat);   /* 0: the SpMV streams the values; else the number of distinct rows (offsets + values) of the dictionary it runs from */
PetscErrorCode MatHIPMI355XGetIndexCompression(Mat A, PetscInt *noffsets);   /* 0: plain CSR indices; else #offsets of the 1-byte dictionary */
PetscErrorCode PCICCGetInfo_HIPMI355X(PC pc, PetscInt *nlevL, PetscInt *nlevU, PetscInt *nshift);   /* levels of the two sweeps of ICC(0); positive-definite shifts the factorisation took */
PetscErrorCode PCILUGetShiftCount_HIPMI355X(PC pc, PetscInt *nshift);   /* restarts of ILU(0) with a larger diagonal shift (MAT_SHIFT_NONZERO, PCILU's default) */
PetscErrorCode PCILUGetLevels_HIPMI355X(PC pc, PetscInt *nlevL, PetscInt *nlevU);   /* dependency levels of the two triangular solves */
PetscErrorCode PCILUGetNodeInfo_HIPMI355X(PC pc, PetscInt *nodes, PetscInt *nlevL, PetscInt *nlevU);   /* node-blocked triangular solves (factor of a matrix with inodes): nodes in the plans (0: row-granular), dependency levels over nodes */
PetscErrorCode PCFactorDebugSetAborted_HIPMI355X(PC pc);   /* tests: raise the "a dependency wait gave up" flag of this PC's sync-free plans */
PetscErrorCode PCILUGetSolver_HIPMI355X(PC pc, PetscInt *syncfree, PetscInt *aborted);   /* 1: two-launch sync-free solves (-pc_factor_hipmi355x_trisolve syncfree, default above 16 levels); 0: one launch per level */

#ifdef __cplusplus
}
#endif
#endif
