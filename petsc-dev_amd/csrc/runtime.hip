// Runtime part of the C ABI: device selection, the per-rank handle (one HIP stream +
// reduction workspace), memory, events.  Everything returns hipError_t as int.
#include "common.hpp"
#include <string.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>

extern "C" {

const char *mi355x_error_string(int err) { return hipGetErrorString((hipError_t)err); }

int mi355x_device_count(int *count) {
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) { *count = 0; return (int)e; }
  return 0;
}
int mi355x_set_device(int dev) { MI355X_TRY(hipSetDevice(dev)); return 0; }
int mi355x_get_device(int *dev) { MI355X_TRY(hipGetDevice(dev)); return 0; }
int mi355x_device_name(char *buf, size_t len) {
  int dev; hipDeviceProp_t prop;
  MI355X_TRY(hipGetDevice(&dev));
  MI355X_TRY(hipGetDeviceProperties(&prop, dev));
  snprintf(buf, len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return 0;
}
int mi355x_device_synchronize(void) { MI355X_TRY(hipDeviceSynchronize()); return 0; }
int mi355x_mem_info(size_t *free_bytes, size_t *total_bytes) { MI355X_TRY(hipMemGetInfo(free_bytes, total_bytes)); return 0; }

int mi355x_host_threads(int cap) {
  const char *e = getenv("MI355X_HOST_THREADS");
  if (e && atoi(e) > 0) return atoi(e) > 64 ? 64 : atoi(e);
  long n = 1;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) n = CPU_COUNT(&set);
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {          // cgroup v2 quota of this container: "<quota> <period>" or "max <period>"
    char q[32]; long period = 0;
    if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") && period > 0) {
      long share = (atol(q) + period - 1) / period;
      if (share >= 1 && share < n) n = share;
    }
    fclose(f);
  }
  const char *lw = getenv("LOCAL_WORLD_SIZE");                    // ranks torchrun started on this node share its cores
  if (lw && atoi(lw) > 1) n /= atoi(lw);
  if (n > cap) n = cap;
  return n < 1 ? 1 : (int)n;
}

int mi355x_handle_create(mi355x_handle_t *out) {
  mi355x_handle_s *h = new mi355x_handle_s();
  memset(h, 0, sizeof(*h));
  MI355X_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  MI355X_TRY(hipMalloc((void **)&h->partials, sizeof(double) * MI355X_REDUCE_GRID_CAP_BIG * MI355X_MAX_RED));
  MI355X_TRY(hipMalloc((void **)&h->ticket, 256));
  MI355X_TRY(hipMemset(h->ticket, 0, 256));
  // +1 slot: the completion sequence word that lets the host poll instead of synchronising the stream
  MI355X_TRY(hipHostMalloc((void **)&h->host_scratch, sizeof(double) * (MI355X_SCRATCH_DOUBLES + 8), hipHostMallocMapped | hipHostMallocCoherent));
  MI355X_TRY(hipMalloc((void **)&h->dev_scratch, sizeof(double) * MI355X_SCRATCH_DOUBLES));
  MI355X_TRY(hipMemset(h->dev_scratch, 0, sizeof(double) * MI355X_SCRATCH_DOUBLES));
  memset(h->host_scratch, 0, sizeof(double) * (MI355X_SCRATCH_DOUBLES + 8));
  h->host_seq = reinterpret_cast<volatile unsigned long long *>(h->host_scratch + MI355X_SCRATCH_DOUBLES);
  h->seq = 0;
  MI355X_TRY(hipDeviceSynchronize());
  *out = h;
  return 0;
}
int mi355x_handle_destroy(mi355x_handle_t h) {
  if (!h) return 0;
  hipStreamSynchronize(h->stream);
  hipFree(h->partials);
  hipFree(h->ticket);
  hipHostFree(h->host_scratch);
  hipFree(h->dev_scratch);
  hipStreamDestroy(h->stream);
  delete h;
  return 0;
}
int mi355x_handle_synchronize(mi355x_handle_t h) { MI355X_TRY(hipStreamSynchronize(h->stream)); return 0; }

// Wait for the most recent reduction that targeted the handle's pinned scratch: the finishing workgroup stores the
// result and then the sequence number (system-scope release), so the host can poll a cache line instead of paying
// a stream synchronisation.  Bounded: after ~2 ms of polling it falls back to hipStreamSynchronize.
int mi355x_handle_wait_result(mi355x_handle_t h) {
  const unsigned long long want = h->seq;
  for (long spin = 0; spin < 2000000; ++spin) {
    if (*h->host_seq == want) { __sync_synchronize(); return 0; }
    __builtin_ia32_pause();
  }
  MI355X_TRY(hipStreamSynchronize(h->stream));
  return 0;
}
void *mi355x_handle_stream(mi355x_handle_t h) { return (void *)h->stream; }
double *mi355x_handle_host_scratch(mi355x_handle_t h) { return h->host_scratch; }
double *mi355x_handle_device_scratch(mi355x_handle_t h) { return h->dev_scratch; }

int mi355x_malloc(void **dptr, size_t bytes) {
  if (bytes == 0) bytes = 16;
  MI355X_TRY(hipMalloc(dptr, bytes));
  return 0;
}
int mi355x_free(void *dptr) { if (dptr) MI355X_TRY(hipFree(dptr)); return 0; }
int mi355x_host_malloc(void **hptr, size_t bytes) {
  if (bytes == 0) bytes = 16;
  MI355X_TRY(hipHostMalloc(hptr, bytes, hipHostMallocMapped));
  return 0;
}
int mi355x_host_free(void *hptr) { if (hptr) MI355X_TRY(hipHostFree(hptr)); return 0; }
int mi355x_memcpy_h2d(mi355x_handle_t h, void *dst, const void *src, size_t bytes) {
  if (!bytes) return 0;
  MI355X_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
  return 0;
}
int mi355x_memcpy_d2h(mi355x_handle_t h, void *dst, const void *src, size_t bytes) {
  if (!bytes) return 0;
  MI355X_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
  return 0;
}
int mi355x_memcpy_d2d(mi355x_handle_t h, void *dst, const void *src, size_t bytes) {
  if (!bytes) return 0;
  MI355X_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, h->stream));
  return 0;
}
int mi355x_memset(mi355x_handle_t h, void *dst, int byte, size_t bytes) {
  if (!bytes) return 0;
  MI355X_TRY(hipMemsetAsync(dst, byte, bytes, h->stream));
  return 0;
}

int mi355x_event_create(mi355x_event_t *e) {
  mi355x_event_s *ev = new mi355x_event_s();
  MI355X_TRY(hipEventCreate(&ev->ev));
  *e = ev;
  return 0;
}
int mi355x_event_destroy(mi355x_event_t e) {
  if (!e) return 0;
  hipEventDestroy(e->ev);
  delete e;
  return 0;
}
int mi355x_event_record(mi355x_event_t e, mi355x_handle_t h) { MI355X_TRY(hipEventRecord(e->ev, h->stream)); return 0; }
int mi355x_event_synchronize(mi355x_event_t e) { MI355X_TRY(hipEventSynchronize(e->ev)); return 0; }
int mi355x_event_elapsed_ms(mi355x_event_t a, mi355x_event_t b, float *ms) {
  MI355X_TRY(hipEventElapsedTime(ms, a->ev, b->ev));
  return 0;
}
int mi355x_handle_wait_event(mi355x_handle_t h, mi355x_event_t e) {
  MI355X_TRY(hipStreamWaitEvent(h->stream, e->ev, 0));
  return 0;
}

// ---- hipGraph capture of a launch-bound sequence on the handle's stream (e.g. the ~1500 level kernels of one
// ILU(0) application): capture once, replay with one call.
int mi355x_graph_capture_begin(mi355x_handle_t h) {
  MI355X_TRY(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
  return 0;
}
int mi355x_graph_capture_end(mi355x_handle_t h, void **exec_out) {
  hipGraph_t g = nullptr;
  hipGraphExec_t e = nullptr;
  MI355X_TRY(hipStreamEndCapture(h->stream, &g));
  hipError_t rc = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  hipGraphDestroy(g);
  if (rc != hipSuccess) return (int)rc;
  *exec_out = (void *)e;
  return 0;
}
int mi355x_graph_launch(mi355x_handle_t h, void *exec) {
  MI355X_TRY(hipGraphLaunch((hipGraphExec_t)exec, h->stream));
  return 0;
}
int mi355x_graph_destroy(void *exec) {
  if (exec) MI355X_TRY(hipGraphExecDestroy((hipGraphExec_t)exec));
  return 0;
}

}  // extern "C"
