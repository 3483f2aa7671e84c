/* Plugin "Sys": the device context of this process, type registration, the RCCL communicators hung on a communicator.
 * (The role of src/sys/objects/init.c:614 PetscOptionsCheckInitial's CUSP part and src/vec/vec/interface/dlregisvec.c
 * in the reference.) */
#include "hipmi355ximpl.h"

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
  ierr = PCRegister(PCILU, 0, "PCCreate_ILU_HIPMI355X", PCCreate_ILU_HIPMI355X);CHKERRQ(ierr);
  ierr = PCRegister(PCPBJACOBI, 0, "PCCreate_PBJacobi_HIPMI355X", PCCreate_PBJacobi_HIPMI355X);CHKERRQ(ierr);
#else
  ierr = PCRegister("pbjacobihipmi355x", 0, "PCCreate_PBJacobi_HIPMI355X", PCCreate_PBJacobi_HIPMI355X);CHKERRQ(ierr);
  /* PETSc's own PCILU keeps its name (host MatSolve); the device-side ILU(0) apply is offered next to it */
  ierr = PCRegister("iluhipmi355x", 0, "PCCreate_ILU_HIPMI355X", PCCreate_ILU_HIPMI355X);CHKERRQ(ierr);
#endif
  return 0;
}

PetscErrorCode PetscHIPMI355XInitialize(int device) {
  requested_device = device;
  return PetscHIPMI355XRegisterAll();
}
PetscErrorCode PetscHIPMI355XFinalize(void) {
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
#endif
