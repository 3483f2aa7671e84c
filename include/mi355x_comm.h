/*
 * mi355x_comm.h -- C ABI of the device-side communication layer (RCCL over xGMI) used by the
 * MPI-parallel HIPMI355X types.  It replaces, for device buffers, the MPI calls on the
 * reference's hot path:
 *   MPI_Allreduce            src/vec/vec/impls/mpi/pbvec.c:16,30 ; pvec2.c:20,62-80
 *   persistent MPI_Start/Waitany halo  src/vec/vec/utils/vpscat.h:97,121,127,210
 * One communicator per process (one process per GPU).  The 128-byte unique id is created on
 * rank 0 and distributed by the launcher (torch.distributed / any out-of-band channel).
 * All calls enqueue on the handle's HIP stream and return immediately (hipError_t/ncclResult_t
 * mapped to a non-zero int on failure).
 */
#ifndef MI355X_COMM_H
#define MI355X_COMM_H
#include "mi355x_kernels.h"
#ifdef __cplusplus
extern "C" {
#endif

#define MI355X_UNIQUE_ID_BYTES 128
typedef struct mi355x_comm_s *mi355x_comm_t;

int mi355x_comm_get_unique_id(char id[MI355X_UNIQUE_ID_BYTES]);
int mi355x_comm_init_rank(mi355x_comm_t *comm, int nranks, int rank, const char id[MI355X_UNIQUE_ID_BYTES]);
int mi355x_comm_destroy(mi355x_comm_t comm);
int mi355x_comm_rank(mi355x_comm_t comm, int *rank, int *nranks);
const char *mi355x_comm_error_string(int err);

/* in-place all-reduce of `count` doubles in device memory, on h's stream */
int mi355x_comm_allreduce_sum(mi355x_comm_t comm, mi355x_handle_t h, double *buf, size_t count);
int mi355x_comm_allreduce_max(mi355x_comm_t comm, mi355x_handle_t h, double *buf, size_t count);
/* neighbour exchange: bracket any number of send/recv with group_start/group_end */
int mi355x_comm_group_start(void);
int mi355x_comm_group_end(void);
int mi355x_comm_send(mi355x_comm_t comm, mi355x_handle_t h, const double *buf, size_t count, int peer);
int mi355x_comm_recv(mi355x_comm_t comm, mi355x_handle_t h, double *buf, size_t count, int peer);

#ifdef __cplusplus
}
#endif
#endif
