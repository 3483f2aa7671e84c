/* Plugin "Sys": the device context of this process, type registration, the RCCL communicators hung on a communicator.
 * (The role of src/sys/objects/init.c:614 PetscOptionsCheckInitial's CUSP part and src/vec/vec/interface/dlregisvec.c
 * in the reference.) */
#include "hipmi355ximpl.h"
#include <time.h>

const char *PetscHIPMI355XVersion(void) { return "petsc-hipmi355x 0.2 (gfx950)"; }

/* ---------------------------------------------------------------- device */
static PetscDeviceCtx devctx = {0, -1, NULL, NULL};
static int requested_device = -1;

PetscErrorCode PetscDeviceGet(PetscDeviceCtx **ctx) {
  if (!devctx.initialized) {
    int n = 0, dev = requested_device;
    int rc = mi355x_device_count(&n);
    if (rc || n < 1) SETERRQ(0, PETSC_ERR_LIB, "no gfx950 device is available to the HIPMI355X types (hip rc=%d, devices=%d); there is no CPU path", rc, n);
    if (dev < 0) {
      const char *lr = getenv("LOCAL_RANK");
      dev = lr ? atoi(lr) % n : 0;
    }
    CHKHIP(mi355x_set_device(dev));
    CHKHIP(mi355x_handle_create(&devctx.h));
    CHKHIP(mi355x_handle_create(&devctx.hcomm));
    devctx.device = dev;
    devctx.initialized = 1;
  }
  *ctx = &devctx;
  return 0;
}

/* ---------------------------------------------------------------- host threads for the set-up passes */
#include <pthread.h>
#include <unistd.h>
/* the set-up's bulk loops over rows (16.7 M of them for P7(256)) on host threads: contiguous ranges, nothing shared */
typedef struct { HipRangeFn fn; void *ctx; PetscInt lo, hi; } HipRangeArg;
static void *hip_range_thread(void *a_) { HipRangeArg *a = (HipRangeArg *)a_; a->fn(a->ctx, a->lo, a->hi); return NULL; }
/* the affinity mask's CPUs, cut to the cgroup quota and shared among the ranks of this node (mi355x_host_threads): eight ranks of a
 * node do not start 8 x 16 spinning threads */
int HipHostThreads(int cap) { return mi355x_host_threads(cap); }
void HipParallelRanges(PetscInt n, HipRangeFn fn, void *ctx) {
  HipRangeArg args[16]; pthread_t th[16]; int started[16];
  int nth = HipHostThreads(16);
  if (n < 200000) nth = 1;
  for (int t = 0; t < nth; t++) { args[t].fn = fn; args[t].ctx = ctx; args[t].lo = (PetscInt)((long)n * t / nth); args[t].hi = (PetscInt)((long)n * (t + 1) / nth); }
  for (int t = 1; t < nth; t++) started[t] = !pthread_create(&th[t], NULL, hip_range_thread, &args[t]);
  fn(ctx, args[0].lo, args[0].hi);
  for (int t = 1; t < nth; t++) { if (started[t]) pthread_join(th[t], NULL); else fn(ctx, args[t].lo, args[t].hi); }
}


/* ---------------------------------------------------------------- registration */
PetscErrorCode PetscHIPMI355XRegisterAll(void) {
  PetscErrorCode ierr;
  static int done = 0;
  if (done) return 0;
  done = 1;
  ierr = VecRegister(VECSEQHIPMI355X, 0, "VecCreate_SeqHIPMI355X", VecCreate_SeqHIPMI355X);CHKERRQ(ierr);
  ierr = VecRegister(VECMPIHIPMI355X, 0, "VecCreate_MPIHIPMI355X", VecCreate_MPIHIPMI355X);CHKERRQ(ierr);
  ierr = VecRegister(VECHIPMI355X, 0, "VecCreate_HIPMI355X", VecCreate_HIPMI355X);CHKERRQ(ierr);
  ierr = MatRegister(MATSEQAIJHIPMI355X, 0, "MatCreate_SeqAIJHIPMI355X", MatCreate_SeqAIJHIPMI355X);CHKERRQ(ierr);
  ierr = MatRegister(MATMPIAIJHIPMI355X, 0, "MatCreate_MPIAIJHIPMI355X", MatCreate_MPIAIJHIPMI355X);CHKERRQ(ierr);
  ierr = MatRegister(MATAIJHIPMI355X, 0, "MatCreate_AIJHIPMI355X", MatCreate_AIJHIPMI355X);CHKERRQ(ierr);
  ierr = MatRegister(MATSEQBAIJHIPMI355X, 0, "MatCreate_SeqBAIJHIPMI355X", MatCreate_SeqBAIJHIPMI355X);CHKERRQ(ierr);
  /* the plug-in's solvers are KSP types of their own; KSPCG / KSPGMRES / KSPBCGS stay whatever the object model has */
  ierr = KSPRegister(KSPCGHIPMI355X, 0, "KSPCreate_CGHIPMI355X", KSPCreate_CGHIPMI355X);CHKERRQ(ierr);
  ierr = KSPRegister(KSPGMRESHIPMI355X, 0, "KSPCreate_GMRESHIPMI355X", KSPCreate_GMRESHIPMI355X);CHKERRQ(ierr);
  ierr = KSPRegister(KSPBCGSHIPMI355X, 0, "KSPCreate_BCGSHIPMI355X", KSPCreate_BCGSHIPMI355X);CHKERRQ(ierr);
#if !defined(PETSCHIPMI355X_WITH_PETSC)
  /* the harness has no CPU types: the reference's generic names select the HIPMI355X implementation of the same shape
   * (with a real PETSc they keep meaning the CPU types, and -vec_type hipmi355x -mat_type aijhipmi355x select these) */
  ierr = VecRegister(VECSEQ, 0, "VecCreate_SeqHIPMI355X", VecCreate_SeqHIPMI355X);CHKERRQ(ierr);
  ierr = VecRegister(VECMPI, 0, "VecCreate_MPIHIPMI355X", VecCreate_MPIHIPMI355X);CHKERRQ(ierr);
  ierr = VecRegister(VECSTANDARD, 0, "VecCreate_HIPMI355X", VecCreate_HIPMI355X);CHKERRQ(ierr);
  ierr = MatRegister(MATSEQAIJ, 0, "MatCreate_SeqAIJHIPMI355X", MatCreate_SeqAIJHIPMI355X);CHKERRQ(ierr);
  ierr = MatRegister(MATMPIAIJ, 0, "MatCreate_MPIAIJHIPMI355X", MatCreate_MPIAIJHIPMI355X);CHKERRQ(ierr);
  ierr = MatRegister(MATAIJ, 0, "MatCreate_AIJHIPMI355X", MatCreate_AIJHIPMI355X);CHKERRQ(ierr);
  ierr = MatRegister(MATSEQBAIJ, 0, "MatCreate_SeqBAIJHIPMI355X", MatCreate_SeqBAIJHIPMI355X);CHKERRQ(ierr);
  ierr = PCRegister(PCPBJACOBI, 0, "PCCreate_PBJacobi_HIPMI355X", PCCreate_PBJacobi_HIPMI355X);CHKERRQ(ierr);
#else
  ierr = PCRegister("pbjacobihipmi355x", 0, "PCCreate_PBJacobi_HIPMI355X", PCCreate_PBJacobi_HIPMI355X);CHKERRQ(ierr);
#endif
  /* PCILU / PCICC are NOT registered by this library: the object model's own (PETSc's; the harness's pcfactor.c) reach the
   * device solves through MatGetFactor -> "MatGetFactor_petsc_C", composed on every MATSEQAIJHIPMI355X (host/ilu.c) */
  return 0;
}

PetscErrorCode PetscHIPMI355XInitialize(int device) {
  requested_device = device;
  return PetscHIPMI355XRegisterAll();
}
PetscErrorCode PetscHIPMI355XFinalize(void) {
  HipFactorJoinHelpers();
  (void)VecHIPMI355XFlushDeferred();
  if (devctx.initialized) {
    mi355x_handle_destroy(devctx.h);
    mi355x_handle_destroy(devctx.hcomm);
    devctx.initialized = 0;
  }
  return 0;
}

#if !defined(PETSCHIPMI355X_WITH_PETSC)
/* ---------------------------------------------------------------- RCCL communicators of a communicator */
PetscErrorCode PetscCommSetDeviceComm(MPI_Comm comm, void *dcomm) {
  PetscErrorCode ierr;
  ierr = PetscCommSetPluginData(comm, 0, dcomm);CHKERRQ(ierr);
  ierr = PetscCommSetPluginData(comm, 1, dcomm);CHKERRQ(ierr);
  return 0;
}
/* two communicators over the same ranks: `reduce` for the compute stream, `halo` for the halo stream */
PetscErrorCode PetscCommSetDeviceComms(MPI_Comm comm, void *reduce, void *halo) {
  PetscErrorCode ierr;
  if ((reduce == NULL) != (halo == NULL)) SETERRQ(comm, PETSC_ERR_ARG_WRONG, "both RCCL communicators or none");
  ierr = PetscCommSetPluginData(comm, 0, reduce);CHKERRQ(ierr);
  ierr = PetscCommSetPluginData(comm, 1, halo);CHKERRQ(ierr);
  return 0;
}
/* what the device-side collectives of this communicator travel over: 0 = one rank, nothing to exchange; 1 = RCCL
 * (nranks = the size RCCL reports for the reduction communicator, distinct = 1 when the halo has a communicator of its
 * own); 2 = host-staged (several ranks, no RCCL communicator attached) */
PetscErrorCode PetscCommGetDeviceTransport(MPI_Comm comm, int *kind, int *nranks, int *distinct) {
  int r = 0, n = 0;
  *kind = HipCommDevice(comm) ? 1 : (HipCommSize(comm) > 1 ? 2 : 0);
  if (HipCommDevice(comm)) CHKHIP(mi355x_comm_rank(HipCommDevice(comm), &r, &n));
  if (nranks) *nranks = n;
  if (distinct) *distinct = (HipCommDevice(comm) && HipCommDeviceHalo(comm) != HipCommDevice(comm)) ? 1 : 0;
  return 0;
}
/* bench.py: what ONE scalar all-reduce of the solvers costs on this communicator (VecDot_MPI / VecNorm_MPI's MPI_Allreduce,
 * pbvec.c:9-35, pvec2.c:46-83, here ncclAllReduce of one double on the compute stream): reps reductions with a stream
 * synchronise after each (sync_each_us: what a solver that reads every result on the host pays) and reps queued back to back
 * (back_to_back_us: the device-side cost per reduction).  Collective.  Zeros without RCCL communicators. */
PetscErrorCode PetscCommDeviceAllreduceLatency(MPI_Comm comm, PetscInt reps, PetscLogDouble *sync_each_us, PetscLogDouble *back_to_back_us) {
  PetscErrorCode ierr;
  PetscDeviceCtx *dc;
  struct timespec t0, t1;
  *sync_each_us = 0.0; *back_to_back_us = 0.0;
  if (!HipCommDevice(comm) || reps < 1) return 0;
  ierr = PetscDeviceGet(&dc);CHKERRQ(ierr);
  double *ds = mi355x_handle_device_scratch(dc->h) + 16;
  const double one = 1.0;
  CHKHIP(mi355x_memcpy_h2d(dc->h, ds, &one, sizeof(one)));
  for (int w = 0; w < 5; w++) CHKHIP(mi355x_comm_allreduce_max(HipCommDevice(comm), dc->h, ds, 1));
  CHKHIP(mi355x_handle_synchronize(dc->h));
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (PetscInt r = 0; r < reps; r++) { CHKHIP(mi355x_comm_allreduce_max(HipCommDevice(comm), dc->h, ds, 1)); CHKHIP(mi355x_handle_synchronize(dc->h)); }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  *sync_each_us = ((double)(t1.tv_sec - t0.tv_sec) * 1e9 + (double)(t1.tv_nsec - t0.tv_nsec)) / 1e3 / (double)reps;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (PetscInt r = 0; r < reps; r++) CHKHIP(mi355x_comm_allreduce_max(HipCommDevice(comm), dc->h, ds, 1));
  CHKHIP(mi355x_handle_synchronize(dc->h));
  clock_gettime(CLOCK_MONOTONIC, &t1);
  *back_to_back_us = ((double)(t1.tv_sec - t0.tv_sec) * 1e9 + (double)(t1.tv_nsec - t0.tv_nsec)) / 1e3 / (double)reps;
  return 0;
}
#endif

/* ---------------------------------------------------------------- stash of off-process entries */
PetscErrorCode HipStashAdd(HipStash *s, PetscInt i, PetscInt j, PetscScalar v, int mode) {
  PetscErrorCode ierr;
  if (s->mode && s->mode != mode) SETERRQ(0, PETSC_ERR_ARG_WRONGSTATE, "Cannot mix add values and insert values");   /* matrix.c:1010 */
  s->mode = mode;
  if (s->n == s->cap) {
    const PetscInt ncap = s->cap ? 2 * s->cap : 1024;
    PetscInt *ni, *nj; PetscScalar *nv;
    ierr = PetscMalloc(sizeof(PetscInt) * (size_t)ncap, &ni);CHKERRQ(ierr);
    ierr = PetscMalloc(sizeof(PetscInt) * (size_t)ncap, &nj);CHKERRQ(ierr);
    ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)ncap, &nv);CHKERRQ(ierr);
    if (s->n) { memcpy(ni, s->i, sizeof(PetscInt) * (size_t)s->n); memcpy(nj, s->j, sizeof(PetscInt) * (size_t)s->n); memcpy(nv, s->v, sizeof(PetscScalar) * (size_t)s->n); }
    HipFree(s->i); HipFree(s->j); HipFree(s->v);
    s->i = ni; s->j = nj; s->v = nv; s->cap = ncap;
  }
  s->i[s->n] = i; s->j[s->n] = j; s->v[s->n] = v; s->n++;
  return 0;
}
void HipStashFree(HipStash *s) { HipFree(s->i); HipFree(s->j); HipFree(s->v); memset(s, 0, sizeof(*s)); }

/* Collective.  Every rank receives the stashed entries of ALL ranks, rank after rank, each rank's in the order they were set
 * (the owner picks out its rows); the stash is emptied.  *mode: the InsertMode the ranks used (0: nobody stashed); ranks
 * that inserted while others added are an error, as MatAssemblyBegin_MPIAIJ's MPI_BOR test (mpiaij.c:598-603). */
PetscErrorCode HipStashExchange(MPI_Comm comm, HipStash *s, PetscInt *nrecv, PetscInt **ri, PetscInt **rj, PetscScalar **rv, int *mode) {
  PetscErrorCode ierr;
  const int size = HipCommSize(comm);
  PetscInt hdr[2], *hdrs, maxn = 0, total = 0;
  *nrecv = 0; *ri = NULL; *rj = NULL; *rv = NULL; *mode = s->mode;
  if (size == 1) { s->n = 0; s->mode = 0; return 0; }
  hdr[0] = s->n; hdr[1] = s->mode;
  ierr = PetscMalloc(sizeof(PetscInt) * 2 * (size_t)size, &hdrs);CHKERRQ(ierr);
  if (HipCommAllgather(comm, hdr, (int)sizeof(hdr), hdrs)) SETERRQ(comm, PETSC_ERR_LIB, "allgather failed");
  for (int r = 0; r < size; r++) {
    if (hdrs[2 * r] > maxn) maxn = hdrs[2 * r];
    total += hdrs[2 * r];
    if (hdrs[2 * r + 1]) {
      if (*mode && *mode != (int)hdrs[2 * r + 1]) { HipFree(hdrs); SETERRQ(comm, PETSC_ERR_ARG_WRONGSTATE, "Some processors inserted others added"); }
      *mode = (int)hdrs[2 * r + 1];
    }
  }
  if (maxn) {
    const size_t rec = 2 * sizeof(PetscInt) + sizeof(PetscScalar);
    char *mine, *all;
    ierr = PetscMalloc(rec * (size_t)maxn, &mine);CHKERRQ(ierr);
    ierr = PetscMalloc(rec * (size_t)maxn * (size_t)size, &all);CHKERRQ(ierr);
    memset(mine, 0, rec * (size_t)maxn);
    memcpy(mine, s->i, sizeof(PetscInt) * (size_t)s->n);
    memcpy(mine + sizeof(PetscInt) * (size_t)maxn, s->j, sizeof(PetscInt) * (size_t)s->n);
    memcpy(mine + 2 * sizeof(PetscInt) * (size_t)maxn, s->v, sizeof(PetscScalar) * (size_t)s->n);
    if (HipCommAllgather(comm, mine, (int)(rec * (size_t)maxn), all)) SETERRQ(comm, PETSC_ERR_LIB, "allgather failed");
    ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(total, 1), ri);CHKERRQ(ierr);
    ierr = PetscMalloc(sizeof(PetscInt) * (size_t)PetscMax(total, 1), rj);CHKERRQ(ierr);
    ierr = PetscMalloc(sizeof(PetscScalar) * (size_t)PetscMax(total, 1), rv);CHKERRQ(ierr);
    PetscInt k = 0;
    for (int r = 0; r < size; r++) {
      const char *blk = all + rec * (size_t)maxn * (size_t)r;
      const PetscInt cnt = hdrs[2 * r];
      memcpy(*ri + k, blk, sizeof(PetscInt) * (size_t)cnt);
      memcpy(*rj + k, blk + sizeof(PetscInt) * (size_t)maxn, sizeof(PetscInt) * (size_t)cnt);
      memcpy(*rv + k, blk + 2 * sizeof(PetscInt) * (size_t)maxn, sizeof(PetscScalar) * (size_t)cnt);
      k += cnt;
    }
    *nrecv = total;
    HipFree(mine); HipFree(all);
  }
  HipFree(hdrs);
  s->n = 0; s->mode = 0;
  return 0;
}
