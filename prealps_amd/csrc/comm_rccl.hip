// comm_rccl.hip -- native RCCL binding of the two process-group hooks
// (preAlps_hip_set_comm): sums of the t x t blocks with ncclAllReduce and the
// boundary-row exchange with grouped ncclSend / ncclRecv, both enqueued on the
// library's own stream, so they are ordered with the kernels and need no host
// round trip.  RCCL is loaded lazily (dlopen) so that the library itself does
// not depend on librccl.so.  Replaces MPI_Allreduce (src/solvers/ecg.c:427,441,
// 513,563) and MPI_Isend/Irecv of whole panels (utils/cplm_v0/cplm_v0_matmult_v2.c
// :184-275) of the reference.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <cstring>

#include "pa_device.h"

namespace {
struct Api {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
} api;
ncclComm_t g_comm = nullptr;
char g_err[256] = "";

int load_api() {
  if (api.h) return 0;
  /* PREALPS_RCCL_LIB: another library with the nine nccl* entry points used here (tests/c/rccl_standin.c
   * drives this binding with several ranks on ONE device, where the real RCCL refuses to start) */
  const char* name = getenv("PREALPS_RCCL_LIB");
  if (name && *name) api.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
  else {
    api.h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!api.h) api.h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  }
  if (!api.h) { snprintf(g_err, sizeof(g_err), "cannot load %s: %s", name && *name ? name : "librccl.so", dlerror()); return 1; }
#define SYM(field, name)                                                     \
  *(void**)(&api.field) = dlsym(api.h, name);                                \
  if (!api.field) { snprintf(g_err, sizeof(g_err), "librccl.so lacks %s", name); return 1; }
  SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
  SYM(AllReduce, "ncclAllReduce") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv")
  SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
  return 0;
}
int nfail(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return 0;
  snprintf(g_err, sizeof(g_err), "%s: %s", what, api.GetErrorString ? api.GetErrorString(r) : "?");
  return 1;
}
}  // namespace

extern "C" {

const char* pa_rccl_error(void) { return g_err; }
int pa_rccl_available(void) { return load_api(); }

/* 128 bytes to be broadcast by the caller's launcher (torch.distributed, MPI, ...). */
int pa_rccl_unique_id(char* id128) {
  if (load_api()) return 1;
  ncclUniqueId id;
  if (nfail(api.GetUniqueId(&id), "ncclGetUniqueId")) return 1;
  memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

int pa_rccl_init(const char* id128, int rank, int size) {
  if (load_api()) return 1;
  if (g_comm) { api.CommDestroy(g_comm); g_comm = nullptr; }
  ncclUniqueId id;
  memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  return nfail(api.CommInitRank(&g_comm, size, id, rank), "ncclCommInitRank");
}

void pa_rccl_shutdown(void) {
  if (g_comm && api.CommDestroy) api.CommDestroy(g_comm);
  g_comm = nullptr;
}

int pa_rccl_allreduce(void* ctx, double* buf, int count) {
  (void)ctx;
  if (!g_comm) { snprintf(g_err, sizeof(g_err), "RCCL communicator not initialised"); return 1; }
  return nfail(api.AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, g_comm, (hipStream_t)pa_rt_stream()),
               "ncclAllReduce");
}

/* One communicator serves both streams: the send/recv group goes to the side stream only after
 * that stream has waited for an event recorded on the main stream behind the previous
 * collective (operator.c: ev_packed), and the next all-reduce on the main stream comes after
 * the main stream has waited for the exchange (ev_halo) -- the operations on the communicator
 * are totally ordered by these events, never concurrent. */
int pa_rccl_exchange(void* ctx, const double* send, const int* send_counts, double* recv,
                     const int* recv_counts, const int* peers, int npeers) {
  (void)ctx;
  if (!g_comm) { snprintf(g_err, sizeof(g_err), "RCCL communicator not initialised"); return 1; }
  hipStream_t st = (hipStream_t)pa_rt_stream();
  if (nfail(api.GroupStart(), "ncclGroupStart")) return 1;
  int bad = 0;
  for (int i = 0; i < npeers && !bad; ++i) {
    if (send_counts[i] > 0) bad = nfail(api.Send(send, (size_t)send_counts[i], ncclDouble, peers[i], g_comm, st), "ncclSend");
    if (!bad && recv_counts[i] > 0) bad = nfail(api.Recv(recv, (size_t)recv_counts[i], ncclDouble, peers[i], g_comm, st), "ncclRecv");
    send += send_counts[i];
    recv += recv_counts[i];
  }
  /* the group is closed whatever happened inside it (an open group would swallow every later call on this
   * thread); the first error is the one reported */
  if (bad) { char first[sizeof(g_err)]; snprintf(first, sizeof(first), "%s", g_err); api.GroupEnd(); snprintf(g_err, sizeof(g_err), "%s", first); return 1; }
  return nfail(api.GroupEnd(), "ncclGroupEnd");
}

}  // extern "C"
