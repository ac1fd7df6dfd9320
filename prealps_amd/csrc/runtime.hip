// runtime.hip -- HIP runtime calls behind the C shim of pa_device.h.
// gfx950 only; no fallback: every failure is reported to the C host code,
// which aborts like the reference's CPLM_Abort.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "pa_device.h"

namespace {
hipStream_t g_own = nullptr;
hipStream_t g_cur = nullptr;
hipStream_t g_side = nullptr;   // halo exchange runs here, beside the interior SpMM
bool g_ready = false;
int g_skip = 0;            // 1: asynchronous device operations are no-ops (a graph replays them)
bool g_capturing = false;  // the current stream is being captured
int g_cus = 0;
char g_err[512] = "";

int fail(hipError_t e, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return 1;
}
}  // namespace

#define RT(call)                                        \
  do {                                                  \
    hipError_t e_ = (call);                             \
    if (e_ != hipSuccess) return fail(e_, #call);       \
  } while (0)

extern "C" {

const char* pa_rt_error(void) { return g_err; }
int pa_rt_ready(void) { return g_ready ? 1 : 0; }
int pa_rt_num_cus(void) { return g_cus; }
/* devices visible to this process (does not select one) */
int pa_rt_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int pa_rt_init(int device) {
  if (g_ready) return 0;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    snprintf(g_err, sizeof(g_err), "no HIP device available (%s); this library has no CPU path",
             e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return 1;
  }
  RT(hipSetDevice(device));
  hipDeviceProp_t prop;
  RT(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    snprintf(g_err, sizeof(g_err), "device %d is %s; kernels are built for gfx950 only", device,
             prop.gcnArchName);
    return 1;
  }
  g_cus = prop.multiProcessorCount;
  RT(hipStreamCreateWithFlags(&g_own, hipStreamNonBlocking));
  RT(hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking));
  g_cur = g_own;
  g_ready = true;
  return 0;
}

void pa_rt_shutdown(void) {
  if (!g_ready) return;
  (void)hipStreamSynchronize(g_cur);
  (void)hipStreamDestroy(g_own);
  (void)hipStreamDestroy(g_side);
  g_own = g_cur = g_side = nullptr;
  g_ready = false;
}

void pa_rt_set_stream(void* s) { g_cur = s ? (hipStream_t)s : g_own; }
void* pa_rt_stream(void) { return (void*)g_cur; }

int pa_rt_sync(void) {
  if (g_skip || g_capturing) { snprintf(g_err, sizeof(g_err), "stream synchronisation inside a graph segment"); return 1; }
  RT(hipStreamSynchronize(g_cur));
  return 0;
}

void* pa_rt_malloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0) bytes = 16;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) { fail(e, "hipMalloc"); return nullptr; }
  return p;
}
void pa_rt_free(void* d) { if (d) (void)hipFree(d); }

void* pa_rt_host_alloc(size_t bytes) {
  void* p = nullptr;
  hipError_t e = hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault);
  if (e != hipSuccess) { fail(e, "hipHostMalloc"); return nullptr; }
  return p;
}
/* fine-grained (host-coherent) pinned memory: words a kernel writes while the host polls them */
void* pa_rt_host_alloc_coherent(size_t bytes) {
  void* p = nullptr;
  hipError_t e = hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocCoherent | hipHostMallocMapped);
  if (e != hipSuccess) { (void)hipGetLastError(); return pa_rt_host_alloc(bytes); }
  return p;
}
void pa_rt_host_free(void* h) { if (h) (void)hipHostFree(h); }
/* 0: work pending, 1: the library stream is idle, -1: it reports an error */
int pa_rt_stream_state(void) {
  hipError_t e = hipStreamQuery(g_cur);
  if (e == hipErrorNotReady) return 0;
  if (e == hipSuccess) return 1;
  fail(e, "hipStreamQuery");
  return -1;
}

/* ---- graphs: one ECG iteration segment captured once, replayed afterwards -------------------------
 * While a segment is REPLAYED its host code still runs (pointer rotations, counters), but every
 * asynchronous device operation it would queue -- kernel launches (PA_LAUNCH in the kernel files),
 * memsets, copies, event records / waits -- is skipped: the instantiated graph queues them. */
int pa_rt_skipping(void) { return g_skip; }
void pa_rt_skip(int on) { g_skip = on ? 1 : 0; }
int pa_rt_capture_begin(void) {
  RT(hipStreamBeginCapture(g_cur, hipStreamCaptureModeRelaxed));
  g_capturing = true;
  return 0;
}
/* ends the capture on the current stream; *exec_out = an executable graph (NULL on failure) */
int pa_rt_capture_end(void** exec_out) {
  hipGraph_t graph = nullptr;
  *exec_out = nullptr;
  g_capturing = false;
  RT(hipStreamEndCapture(g_cur, &graph));
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) return fail(e, "hipGraphInstantiate");
  *exec_out = (void*)exec;
  return 0;
}
int pa_rt_graph_launch(void* exec) {
  RT(hipGraphLaunch((hipGraphExec_t)exec, g_cur));
  return 0;
}
void pa_rt_graph_free(void* exec) { if (exec) (void)hipGraphExecDestroy((hipGraphExec_t)exec); }

int pa_rt_memset(void* d, int v, size_t bytes) {
  if (g_skip) return 0;
  if (bytes) RT(hipMemsetAsync(d, v, bytes, g_cur));
  return 0;
}
int pa_rt_h2d(void* d, const void* h, size_t bytes) {
  if (!bytes) return 0;
  if (g_skip || g_capturing) { snprintf(g_err, sizeof(g_err), "synchronous copy inside a graph segment"); return 1; }
  RT(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, g_cur));
  RT(hipStreamSynchronize(g_cur));
  return 0;
}
int pa_rt_d2h(void* h, const void* d, size_t bytes) {
  if (!bytes) return 0;
  if (g_skip || g_capturing) { snprintf(g_err, sizeof(g_err), "synchronous copy inside a graph segment"); return 1; }
  RT(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, g_cur));
  RT(hipStreamSynchronize(g_cur));
  return 0;
}
int pa_rt_d2d(void* dst, const void* src, size_t bytes) {
  if (g_skip) return 0;
  if (bytes) RT(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, g_cur));
  return 0;
}
/* (the pinned source must stay as it is until the stream has passed the copy) */
int pa_rt_h2d_async(void* d, const void* pinned, size_t bytes) {
  if (g_skip) return 0;
  if (bytes) RT(hipMemcpyAsync(d, pinned, bytes, hipMemcpyHostToDevice, g_cur));
  return 0;
}
int pa_rt_d2h_async(void* pinned, const void* d, size_t bytes) {
  if (g_skip) return 0;
  if (bytes) RT(hipMemcpyAsync(pinned, d, bytes, hipMemcpyDeviceToHost, g_cur));
  return 0;
}

void* pa_rt_side_stream(void) { return (void*)g_side; }
/* make stream `s` wait for everything recorded in event `e` */
int pa_rt_stream_wait_event(void* s, void* e) {
  if (g_skip) return 0;
  RT(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)e, 0));
  return 0;
}
int pa_rt_event_record_on(void* e, void* s) {
  if (g_skip) return 0;
  RT(hipEventRecord((hipEvent_t)e, (hipStream_t)s));
  return 0;
}
void* pa_rt_event_create(void) {
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return (void*)e;
}
void pa_rt_event_destroy(void* e) { if (e) (void)hipEventDestroy((hipEvent_t)e); }
int pa_rt_event_record(void* e) {
  if (g_skip) return 0;
  RT(hipEventRecord((hipEvent_t)e, g_cur));
  return 0;
}
int pa_rt_event_wait(void* e) {
  RT(hipEventSynchronize((hipEvent_t)e));
  return 0;
}
double pa_rt_event_elapsed_s(void* a, void* b) {
  float ms = 0.f;
  if (hipEventSynchronize((hipEvent_t)b) != hipSuccess) return -1.0;
  if (hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b) != hipSuccess) return -1.0;
  return 1e-3 * (double)ms;
}

}  // extern "C"
