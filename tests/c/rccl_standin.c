/* TEST INFRASTRUCTURE, not product code: a stand-in for librccl.so with the nine nccl* entry points that
 * prealps_amd/csrc/comm_rccl.hip binds (PREALPS_RCCL_LIB=<this library>), built by tests/test_gpu_mpi.py.
 *
 * The real RCCL refuses two ranks on one device, and the GPU box has one device, so the `rccl` branch of
 * pa_mpi_bind (id broadcast over MPI, ncclCommInitRank, the self-test, the grouped ncclSend / ncclRecv exchange
 * with several peers, the all-reduces on the library's stream) never ran there with more than one rank.
 * This library gives that branch something to talk to: every operation waits for the stream it was queued
 * on, stages through the host and goes over the MPI the test driver initialised.  Same call sequence and
 * semantics as seen from the binding (in-place all-reduce, grouped point-to-point, stream order); none of
 * RCCL's performance, and no claim about RCCL itself. */
#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

struct ncclComm { MPI_Comm mpi; int rank, size; };
#define MAGIC "prealps-rccl-standin"

typedef struct { int send; void* dev; size_t count; int peer; hipStream_t st; MPI_Comm mpi; } op_t;
static op_t g_ops[1024];
static int g_nops = 0, g_depth = 0;
static long g_calls[4];       /* all-reduce, send, recv, groups -- printed at destroy (the test reads it) */

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  strcpy(id->internal, MAGIC);
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  int r, s;
  if (strcmp(id.internal, MAGIC)) return ncclInvalidArgument;        /* the id did not come from rank 0's call */
  MPI_Comm_rank(MPI_COMM_WORLD, &r);
  MPI_Comm_size(MPI_COMM_WORLD, &s);
  if (r != rank || s != nranks) return ncclInvalidArgument;
  struct ncclComm* c = (struct ncclComm*)calloc(1, sizeof(*c));
  if (!c || MPI_Comm_dup(MPI_COMM_WORLD, &c->mpi)) return ncclSystemError;
  c->rank = rank; c->size = nranks;
  *comm = c;
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  if (comm) {
    fprintf(stderr, "[rccl stand-in] rank %d: %ld all-reduces, %ld sends, %ld receives, %ld groups\n", comm->rank,
            g_calls[0], g_calls[1], g_calls[2], g_calls[3]);
    int fin = 0;
    MPI_Finalized(&fin);
    if (!fin) MPI_Comm_free(&comm->mpi);
    free(comm);
  }
  return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "stand-in error"; }

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t st) {
  if (dt != ncclDouble || op != ncclSum || !comm) return ncclInvalidArgument;
  double* h = (double*)malloc((count ? count : 1) * sizeof(double));
  if (!h) return ncclSystemError;
  if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(h, send, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  if (MPI_Allreduce(MPI_IN_PLACE, h, (int)count, MPI_DOUBLE, MPI_SUM, comm->mpi)) return ncclSystemError;
  if (hipMemcpy(recv, h, count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  free(h);
  ++g_calls[0];
  return ncclSuccess;
}

static ncclResult_t flush(void) {
  int n = g_nops;
  g_nops = 0;
  if (!n) return ncclSuccess;
  double** h = (double**)calloc((size_t)n, sizeof(double*));
  MPI_Request* rq = (MPI_Request*)malloc((size_t)n * sizeof(MPI_Request));
  if (!h || !rq) return ncclSystemError;
  for (int i = 0; i < n; ++i) {
    h[i] = (double*)malloc((g_ops[i].count ? g_ops[i].count : 1) * sizeof(double));
    if (!h[i]) return ncclSystemError;
    if (hipStreamSynchronize(g_ops[i].st) != hipSuccess) return ncclUnhandledCudaError;
    if (g_ops[i].send && hipMemcpy(h[i], g_ops[i].dev, g_ops[i].count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
      return ncclUnhandledCudaError;
  }
  /* messages between one pair of ranks match in the order they were queued, as in RCCL */
  for (int i = 0; i < n; ++i) {
    int rc = g_ops[i].send ? MPI_Isend(h[i], (int)g_ops[i].count, MPI_DOUBLE, g_ops[i].peer, 901, g_ops[i].mpi, &rq[i])
                           : MPI_Irecv(h[i], (int)g_ops[i].count, MPI_DOUBLE, g_ops[i].peer, 901, g_ops[i].mpi, &rq[i]);
    if (rc) return ncclSystemError;
  }
  if (MPI_Waitall(n, rq, MPI_STATUSES_IGNORE)) return ncclSystemError;
  for (int i = 0; i < n; ++i) {
    if (!g_ops[i].send && hipMemcpy(g_ops[i].dev, h[i], g_ops[i].count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
      return ncclUnhandledCudaError;
    free(h[i]);
  }
  free(h); free(rq);
  return ncclSuccess;
}
static ncclResult_t queue(int send, void* dev, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st) {
  if (dt != ncclDouble || !comm || peer < 0 || peer >= comm->size || g_nops >= 1024) return ncclInvalidArgument;
  op_t o = {send, dev, count, peer, st, comm->mpi};
  g_ops[g_nops++] = o;
  ++g_calls[send ? 1 : 2];
  return g_depth ? ncclSuccess : flush();
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st) {
  return queue(1, (void*)buf, count, dt, peer, comm, st);
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st) {
  return queue(0, buf, count, dt, peer, comm, st);
}
ncclResult_t ncclGroupStart(void) { ++g_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) {
  if (g_depth <= 0) return ncclInvalidUsage;
  if (--g_depth) return ncclSuccess;
  ++g_calls[3];
  return flush();
}
