/* PETSc's MPI spellings for example programs written against the reference (src/ksp/ksp/examples/tutorials/ex2.c style)
 * and built on the harness (include/petscmini.h), where no MPI exists.  Opt-in: NEVER include this next to <mpi.h>. */
#ifndef PETSCMINI_MPINAMES_H
#define PETSCMINI_MPINAMES_H
#include "petscmini.h"
typedef PetscComm MPI_Comm;
#define MPI_Comm_rank(comm, rank) PetscCommRank(comm, rank)
#define MPI_Comm_size(comm, size) PetscCommSize(comm, size)
#endif
