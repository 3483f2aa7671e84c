/* placeholder so the host library links before the object model lands */
const char *PetscHIPMI355XVersion(void) { return "petsc-hipmi355x 0.1 (round 1)"; }
