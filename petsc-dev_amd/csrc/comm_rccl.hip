// RCCL (xGMI) transport behind include/mi355x_comm.h.  Thin by design: the choreography
// (which stream, which event, what overlaps with the diagonal-block SpMV) lives in the
// host-side VecScatter/Vec implementations.
#include "common.hpp"
#include "mi355x_comm.h"
#include <rccl/rccl.h>
#include <string.h>

struct mi355x_comm_s {
  ncclComm_t comm;
  int rank, nranks;
};

#define NCCL_TRY(expr)                                   \
  do {                                                   \
    ncclResult_t r_ = (expr);                            \
    if (r_ != ncclSuccess) return 100000 + (int)r_;      \
  } while (0)

static_assert(sizeof(ncclUniqueId) <= MI355X_UNIQUE_ID_BYTES, "unique id does not fit");

extern "C" {

const char *mi355x_comm_error_string(int err) {
  if (err >= 100000) return ncclGetErrorString((ncclResult_t)(err - 100000));
  return hipGetErrorString((hipError_t)err);
}

int mi355x_comm_get_unique_id(char id[MI355X_UNIQUE_ID_BYTES]) {
  ncclUniqueId uid;
  NCCL_TRY(ncclGetUniqueId(&uid));
  memset(id, 0, MI355X_UNIQUE_ID_BYTES);
  memcpy(id, &uid, sizeof(uid));
  return 0;
}

int mi355x_comm_init_rank(mi355x_comm_t *out, int nranks, int rank, const char id[MI355X_UNIQUE_ID_BYTES]) {
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  mi355x_comm_s *c = new mi355x_comm_s();
  c->rank = rank;
  c->nranks = nranks;
  ncclResult_t r = ncclCommInitRank(&c->comm, nranks, uid, rank);
  if (r != ncclSuccess) { delete c; *out = nullptr; return 100000 + (int)r; }
  *out = c;
  return 0;
}

int mi355x_comm_destroy(mi355x_comm_t c) {
  if (!c) return 0;
  ncclCommDestroy(c->comm);
  delete c;
  return 0;
}

int mi355x_comm_rank(mi355x_comm_t c, int *rank, int *nranks) {
  *rank = c->rank;
  *nranks = c->nranks;
  return 0;
}

int mi355x_comm_allreduce_sum(mi355x_comm_t c, mi355x_handle_t h, double *buf, size_t count) {
  NCCL_TRY(ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, c->comm, h->stream));
  return 0;
}
int mi355x_comm_allreduce_max(mi355x_comm_t c, mi355x_handle_t h, double *buf, size_t count) {
  NCCL_TRY(ncclAllReduce(buf, buf, count, ncclDouble, ncclMax, c->comm, h->stream));
  return 0;
}
int mi355x_comm_group_start(void) { NCCL_TRY(ncclGroupStart()); return 0; }
int mi355x_comm_group_end(void) { NCCL_TRY(ncclGroupEnd()); return 0; }
int mi355x_comm_send(mi355x_comm_t c, mi355x_handle_t h, const double *buf, size_t count, int peer) {
  NCCL_TRY(ncclSend(buf, count, ncclDouble, peer, c->comm, h->stream));
  return 0;
}
int mi355x_comm_recv(mi355x_comm_t c, mi355x_handle_t h, double *buf, size_t count, int peer) {
  NCCL_TRY(ncclRecv(buf, count, ncclDouble, peer, c->comm, h->stream));
  return 0;
}

}  // extern "C"
