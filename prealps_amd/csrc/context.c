/*
 * context.c -- process-wide state of libprealps_hip.so: device selection,
 * error reporting in the reference's CPLM_Abort style, the process-group
 * hooks, phase timers and the dense descriptor helper.
 *
 * Reference behaviour mirrored here:
 *   CPLM_FAbort / CPLM_efprintf   utils/cplm_core/cplm_utils.c:17-58
 *   CPLM_MatDenseSetInfo          utils/cplm_light/cplm_matdense.c:124-135
 */
#define _GNU_SOURCE
#include <sched.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "pa_host.h"

static int g_abort_mode = 1;
static char g_last_error[1024] = "";
static int g_rank = 0, g_size = 1;
static preAlps_allreduce_fn g_allreduce = NULL;
static preAlps_exchange_fn g_exchange = NULL;
static void* g_comm_ctx = NULL;

static int g_timing = 0;
static double g_times[PA_T_COUNT];
static void* g_ev0 = NULL;
static void* g_ev1 = NULL;

/* Threads worth starting for host-side setup work: what OpenMP would use, capped by the CPU time
 * the process may actually consume (cgroup quota) and by its affinity mask -- in a container with
 * 16 CPUs' worth of quota on a 256-thread host, 256 threads only take turns (measured on the MI355X
 * box: the OpenMP loops of the CPU baseline run 4x slower with 128 threads than with 16). */
static int host_cpu_share(void);
static int g_solo = 0;
void pa_host_solo(int on) { g_solo = on ? 1 : 0; }
int pa_host_threads(void) {
  int t = 1;
#ifdef _OPENMP
  t = omp_get_max_threads();          /* OMP_NUM_THREADS, if the caller set it */
#endif
  /* the ranks of a multi-GPU run share the node's CPUs (one node: preAlps_hip_set_world) */
  int cap = host_cpu_share() / (g_size > 1 && !g_solo ? g_size : 1);
  if (cap < 1) cap = 1;
  return t < cap ? t : cap;
}
static int host_cpu_share(void) {
  static int cached = 0;
  if (cached) return cached;
  int t = 1 << 20;
  long long quota = -1, period = -1;
  FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");                        /* cgroup v2 */
  if (f) {
    char q[64];
    if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max")) quota = atoll(q);
    fclose(f);
  } else {                                                                  /* cgroup v1 */
    FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
    FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
    if (fq && fp && (fscanf(fq, "%lld", &quota) != 1 || fscanf(fp, "%lld", &period) != 1)) quota = -1;
    if (fq) fclose(fq);
    if (fp) fclose(fp);
  }
  if (quota > 0 && period > 0) {
    int c = (int)((quota + period - 1) / period);
    if (c >= 1 && c < t) t = c;
  }
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) {
    int c = CPU_COUNT(&set);
    if (c >= 1 && c < t) t = c;
  }
  cached = t < 1 ? 1 : t;
  return cached;
}

double pa_wtime(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void pa_time_abort(void);
int pa_fail_at(const char* func, const char* fmt, ...) {
  char msg[900];
  va_list va;
  va_start(va, fmt);
  vsnprintf(msg, sizeof(msg), fmt, va);
  va_end(va);
  snprintf(g_last_error, sizeof(g_last_error), "%s: %s", func, msg);
  pa_time_abort();      /* an entry point that fails inside a timed region leaves no region open */
  if (g_abort_mode) {
    /* same shape as the reference's abort banner */
    fprintf(stderr, "\nABORTING from %s : [Proc: %d] %s\n\n", func, g_rank, msg);
    fflush(stderr);
    pa_mpi_abort();       /* with an MPI attached: MPI_Abort(MPI_COMM_WORLD, 1) like CPLM_Abort, every rank ends */
    abort();
  }
  return 1;
}

void preAlps_hip_set_abort_mode(int abort_on_error) { g_abort_mode = abort_on_error ? 1 : 0; }
const char* preAlps_hip_last_error(void) { return g_last_error; }

int pa_default_device(void) {
  const char* lr = getenv("LOCAL_RANK");
  return lr ? atoi(lr) : 0;
}

int preAlps_hip_init(int device) {
  if (pa_rt_init(device) != 0) return PA_FAIL("%s", pa_rt_error());
  return 0;
}

void preAlps_hip_shutdown(void) {
  pa_rt_event_destroy(g_ev0);
  pa_rt_event_destroy(g_ev1);
  g_ev0 = g_ev1 = NULL;
  pa_rccl_shutdown();
  pa_mpi_release();
  pa_rt_shutdown();
}

int preAlps_hip_set_stream(void* s) {
  PA_REQUIRE_GPU();
  pa_rt_set_stream(s);
  return 0;
}
void* preAlps_hip_get_stream(void) { return pa_rt_stream(); }
int preAlps_hip_sync(void) {
  if (!pa_rt_ready()) return 0;
  PA_CHECK(pa_rt_sync());
  return 0;
}

/* ---- panels ------------------------------------------------------------- */
int pa_panel_stride(int enlFac) {
  int ts = 2;
  while (ts < enlFac) ts <<= 1;
  return ts;
}
int preAlps_hip_panel_stride(int enlFac) { return pa_panel_stride(enlFac); }

/* Reference semantics (lda = m for COL_MAJOR, n for ROW_MAJOR). */
int CPLM_MatDenseSetInfo(CPLM_Mat_Dense_t* A, int M, int N, int m, int n,
                         CPLM_storage_type_t storage) {
  A->info.M = M; A->info.N = N; A->info.m = m; A->info.n = n;
  A->info.lda = (storage == ROW_MAJOR) ? n : m;
  A->info.nval = m * n;
  A->info.stor_type = storage;
  return 0;
}

/* Device panels: ROW_MAJOR with the fixed stride ts as leading dimension. */
void pa_set_desc(CPLM_Mat_Dense_t* A, int M, int N, int m, int n, int ts) {
  CPLM_MatDenseSetInfo(A, M, N, m, n, ROW_MAJOR);
  A->info.lda = ts;
}

/* ---- process group -------------------------------------------------------- */
int preAlps_hip_set_world(int rank, int size) {
  if (size < 1 || rank < 0 || rank >= size) return PA_FAIL("invalid rank %d of %d", rank, size);
  g_rank = rank; g_size = size;
  return 0;
}
int preAlps_hip_set_comm(preAlps_allreduce_fn allreduce, preAlps_exchange_fn exchange, void* ctx) {
  g_allreduce = allreduce; g_exchange = exchange; g_comm_ctx = ctx;
  return 0;
}
/* Native binding: RCCL on the library stream.  The 128-byte id comes from rank 0
 * (preAlps_hip_rccl_unique_id) and is broadcast by whatever launched the processes. */
/* 0 when librccl.so and every entry point this library uses can be loaded (no communicator yet). */
int preAlps_hip_rccl_available(void) {
  if (pa_rccl_available()) return PA_FAIL("%s", pa_rccl_error());
  return 0;
}
int preAlps_hip_rccl_unique_id(char* id128) {
  if (pa_rccl_unique_id(id128)) return PA_FAIL("%s", pa_rccl_error());
  return 0;
}
int preAlps_hip_rccl_init(const char* id128, int rank, int size) {
  PA_REQUIRE_GPU();
  if (preAlps_hip_set_world(rank, size)) return 1;
  if (pa_rccl_init(id128, rank, size)) return PA_FAIL("%s", pa_rccl_error());
  g_allreduce = pa_rccl_allreduce; g_exchange = pa_rccl_exchange; g_comm_ctx = NULL;
  return 0;
}
/* Rehearsal of ONE shard of a `size`-process run in a single process (bench.py --shard-of): this
 * process plans and owns the rows of `rank`, sums over the processes are its own sums, halo rows
 * arrive as zeros -- i.e. it solves with the diagonal block A(rank, rank) of the partitioned matrix,
 * an SPD problem of its own, through exactly the kernels, launches and stream choreography (pack,
 * side-stream exchange, interior / halo-reading SpMM halves, reductions) the rank would run. */
static int loop_allreduce(void* ctx, double* dev_buf, int count) { (void)ctx; (void)dev_buf; (void)count; return 0; }
static int loop_exchange(void* ctx, const double* dev_send, const int* send_counts, double* dev_recv,
                         const int* recv_counts, const int* peers, int npeers) {
  (void)ctx; (void)dev_send; (void)send_counts; (void)peers;
  size_t n = 0;
  for (int i = 0; i < npeers; ++i) n += (size_t)recv_counts[i];
  return pa_rt_memset(dev_recv, 0, n * sizeof(double));
}
int preAlps_hip_loopback(int rank, int size) {
  if (preAlps_hip_set_world(rank, size)) return 1;
  g_allreduce = loop_allreduce; g_exchange = loop_exchange; g_comm_ctx = NULL;
  return 0;
}
int pa_comm_is_loopback(void) { return g_allreduce == loop_allreduce; }

/* Round-trip check of whatever hooks are installed: sum of (rank+1) over the ranks, and a
 * ring exchange (send our rank to rank+1, receive from rank-1).  0 = both came back right. */
int preAlps_hip_comm_selftest(void) {
  PA_REQUIRE_GPU();
  if (g_size == 1) return 0;
  double* d = (double*)pa_rt_malloc(4 * sizeof(double));
  if (!d) return PA_FAIL("%s", pa_rt_error());
  double h[4] = {(double)(g_rank + 1), 0.0, (double)g_rank, -1.0};
  int rc = pa_rt_h2d(d, h, sizeof(h));
  rc = rc || pa_allreduce(d, 1);
  int nxt = (g_rank + 1) % g_size, prv = (g_rank + g_size - 1) % g_size;
  if (!rc) {
    if (g_size == 2) { /* one peer: it is both neighbours */
      int peers[1] = {nxt}, sc[1] = {1}, rcnt[1] = {1};
      rc = pa_exchange(d + 2, sc, d + 3, rcnt, peers, 1);
    } else {
      int peers[2], sc[2], rcnt[2];
      /* peers in ascending order, like the operator's plan */
      if (prv < nxt) { peers[0] = prv; sc[0] = 0; rcnt[0] = 1; peers[1] = nxt; sc[1] = 1; rcnt[1] = 0; }
      else { peers[0] = nxt; sc[0] = 1; rcnt[0] = 0; peers[1] = prv; sc[1] = 0; rcnt[1] = 1; }
      rc = pa_exchange(d + 2, sc, d + 3, rcnt, peers, 2);
    }
  }
  rc = rc || pa_rt_d2h(h, d, sizeof(h));
  pa_rt_free(d);
  if (rc) return 1;
  double want = 0.5 * g_size * (g_size + 1);
  if (h[0] != want) return PA_FAIL("all-reduce self-test: got %g, expected %g", h[0], want);
  if (h[3] != (double)prv) return PA_FAIL("exchange self-test: got %g from rank %d", h[3], prv);
  return 0;
}
int pa_world_rank(void) { return g_rank; }
int pa_world_size(void) { return g_size; }

int pa_allreduce(double* dev_buf, int count) {
  if (g_size == 1 || count <= 0) return 0;
  if (!g_allreduce) return PA_FAIL("%d processes but no all-reduce hook (preAlps_hip_set_comm)", g_size);
  pa_time_begin(PA_T_COMM);
  int rc = g_allreduce(g_comm_ctx, dev_buf, count);
  pa_time_end(PA_T_COMM);
  if (rc) return PA_FAIL("all-reduce hook returned %d (%s)", rc, g_allreduce == pa_rccl_allreduce ? pa_rccl_error() : "user hook");
  return 0;
}

int pa_exchange(const double* dev_send, const int* send_counts, double* dev_recv,
                const int* recv_counts, const int* peers, int npeers) {
  if (g_size == 1 || npeers <= 0) return 0;
  if (!g_exchange) return PA_FAIL("%d processes but no halo-exchange hook (preAlps_hip_set_comm)", g_size);
  pa_time_begin(PA_T_COMM);
  int rc = g_exchange(g_comm_ctx, dev_send, send_counts, dev_recv, recv_counts, peers, npeers);
  pa_time_end(PA_T_COMM);
  if (rc) return PA_FAIL("halo-exchange hook returned %d (%s)", rc, g_exchange == pa_rccl_exchange ? pa_rccl_error() : "user hook");
  return 0;
}

/* ---- timing --------------------------------------------------------------- */
static int g_time_depth = 0;
void pa_time_abort(void) { g_time_depth = 0; }
static const char* k_time_keys[PA_T_COUNT] = {"operator", "precond", "gram", "trsm",
                                              "update", "small", "comm"};

void preAlps_hip_timing(int enable) {
  g_timing = enable ? 1 : 0;
  if (g_timing && pa_rt_ready() && !g_ev0) {
    g_ev0 = pa_rt_event_create();
    g_ev1 = pa_rt_event_create();
  }
}
int pa_timing_enabled(void) { return g_timing; }
void preAlps_hip_timing_reset(void) { memset(g_times, 0, sizeof(g_times)); }

void pa_time_begin(int key) {
  (void)key;
  if (!g_timing || !g_ev0) return;
  if (g_time_depth++ == 0) {
    /* every region starts on an idle stream (pa_time_end waits for its stop event): a launch onto an idle stream
     * begins 5-10 us after an event recorded in front of it, which the pair would count as kernel time -- the
     * rocprofv3 trace of the same run showed k_spmm_runs_gram at 151 us where this pair read 162.  A spacer of
     * 20 us in front of the start event gives the host the time to queue the region's first launch */
    pa_k_spacer(20);
    pa_rt_event_record(g_ev0);
  }
}
/* Returns the device seconds of the region that just closed, or -1 when timing is off or the
 * region is nested inside another one (the outermost region owns the event pair). */
double pa_time_end(int key) {
  if (!g_timing || !g_ev0) return -1.0;
  if (--g_time_depth == 0) {
    pa_rt_event_record(g_ev1);
    double s = pa_rt_event_elapsed_s(g_ev0, g_ev1);
    if (s > 0) g_times[key] += s;
    return s > 0 ? s : 0.0;
  }
  return -1.0;
}
static void* g_sw0 = NULL;
static void* g_sw1 = NULL;
int preAlps_hip_timer_start(void) {
  PA_REQUIRE_GPU();
  if (!g_sw0) { g_sw0 = pa_rt_event_create(); g_sw1 = pa_rt_event_create(); }
  if (!g_sw0 || !g_sw1) return PA_FAIL("hipEventCreate failed");
  PA_CHECK(pa_rt_event_record(g_sw0));
  return 0;
}
int preAlps_hip_timer_stop(double* seconds) {
  if (!g_sw0) return PA_FAIL("timer not started");
  PA_CHECK(pa_rt_event_record(g_sw1));
  *seconds = pa_rt_event_elapsed_s(g_sw0, g_sw1);
  if (*seconds < 0) return PA_FAIL("hipEventElapsedTime failed");
  return 0;
}

/* Streaming ceilings of this device, measured with the plainest kernels on `bytes` of HBM
 * (take it well above the 256 MiB Infinity Cache): copy counts bytes read + written. */
int preAlps_hip_hbm_probe(size_t bytes, int reps, double* copy_GBs, double* read_GBs) {
  PA_REQUIRE_GPU();
  if (bytes < (1u << 20) || reps < 1) return PA_FAIL("probe needs at least 1 MiB and one repetition");
  bytes &= ~(size_t)4095;
  double* a = (double*)pa_rt_malloc(bytes);
  double* b = (double*)pa_rt_malloc(bytes);
  int rc = !a || !b || pa_rt_memset(a, 0, bytes) || pa_rt_memset(b, 0, bytes);
  double sec = 0.0;
  for (int which = 0; which < 2 && !rc; ++which) {
    rc = pa_k_probe(which, bytes, a, b) || preAlps_hip_timer_start();   /* one warm-up launch */
    for (int r = 0; r < reps && !rc; ++r) rc = pa_k_probe(which, bytes, a, b);
    rc = rc || preAlps_hip_timer_stop(&sec);
    if (!rc) {
      double gbs = (which == 0 ? 2.0 : 1.0) * (double)bytes * reps / sec / 1e9;
      if (which == 0) { if (copy_GBs) *copy_GBs = gbs; } else if (read_GBs) *read_GBs = gbs;
    }
  }
  pa_rt_free(a); pa_rt_free(b);
  if (rc) return PA_FAIL("HBM probe failed: %s", pa_rt_error());
  return 0;
}

int preAlps_hip_get_time(const char* key, double* seconds) {
  for (int i = 0; i < PA_T_COUNT; ++i)
    if (strcmp(key, k_time_keys[i]) == 0) { *seconds = g_times[i]; return 0; }
  return 1;
}

/* ---- host <-> device panels ------------------------------------------------ */
int preAlps_hip_panel_alloc(CPLM_Mat_Dense_t* A, int M, int N, int m, int n, int enlFac) {
  PA_REQUIRE_GPU();
  int ts = pa_panel_stride(enlFac);
  if (n > ts) return PA_FAIL("panel with %d columns does not fit stride %d", n, ts);
  A->val = (double*)pa_rt_malloc((size_t)(m > 0 ? m : 1) * ts * sizeof(double));
  if (!A->val) return PA_FAIL("device allocation failed: %s", pa_rt_error());
  PA_CHECK(pa_rt_memset(A->val, 0, (size_t)m * ts * sizeof(double)));
  pa_set_desc(A, M, N, m, n, ts);
  return 0;
}
void preAlps_hip_panel_free(CPLM_Mat_Dense_t* A) {
  if (A && A->val) { pa_rt_free(A->val); A->val = NULL; }
}

int preAlps_hip_panel_to_host(const CPLM_Mat_Dense_t* A, int enlFac, double* host, int ld) {
  PA_REQUIRE_GPU();
  (void)enlFac;
  int m = A->info.m, n = A->info.n, ts = pa_desc_stride(A);
  double* tmp = (double*)malloc((size_t)(m > 0 ? m : 1) * ts * sizeof(double));
  if (!tmp) return PA_FAIL("out of host memory");
  if (pa_rt_d2h(tmp, A->val, (size_t)m * ts * sizeof(double))) { free(tmp); return PA_FAIL("%s", pa_rt_error()); }
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < m; ++i) host[i + (size_t)ld * j] = tmp[(size_t)i * ts + j];
  free(tmp);
  return 0;
}

int preAlps_hip_panel_from_host(CPLM_Mat_Dense_t* A, int enlFac, const double* host, int ld) {
  PA_REQUIRE_GPU();
  (void)enlFac;
  int m = A->info.m, n = A->info.n, ts = pa_desc_stride(A);
  double* tmp = (double*)calloc((size_t)(m > 0 ? m : 1) * ts, sizeof(double));
  if (!tmp) return PA_FAIL("out of host memory");
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < m; ++i) tmp[(size_t)i * ts + j] = host[i + (size_t)ld * j];
  int rc = pa_rt_h2d(A->val, tmp, (size_t)m * ts * sizeof(double));
  free(tmp);
  if (rc) return PA_FAIL("%s", pa_rt_error());
  return 0;
}
