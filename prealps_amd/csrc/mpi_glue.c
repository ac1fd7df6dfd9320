/*
 * mpi_glue.c -- what lets the reference's MPI driver run unchanged with one rank per GPU.
 *
 * The reference is an MPI program: examples/test_ecg_prealps_op.c:69,158 hands MPI_COMM_WORLD
 * to preAlps_OperatorBuild, which takes rank and size from it, reads the matrix on rank 0
 * and scatters row panels (utils/operator.c:38-134, utils/cplm_light/cplm_matcsr.c:382-497).
 * This library does not link an MPI: when preAlps_OperatorBuild is entered in a process whose
 * main program has initialised one, the handful of MPI entry points below are resolved at run
 * time from the symbols that program already loaded (dlsym on the global scope, then
 * libmpi.so.12), with the MPICH ABI (handles are ints with the constants below: MPICH, Intel
 * MPI, MVAPICH, Cray MPI).  From the communicator it takes rank and size, picks the device from
 * the rank's position on its node, and binds the two process-group hooks of the library:
 *   - RCCL (comm_rccl.hip) with the unique id broadcast over the communicator, when every
 *     rank of a node has a device of its own;
 *   - otherwise (several ranks on one device, which RCCL refuses; PREALPS_COMM=mpi) the same
 *     hooks staged through pinned host buffers and MPI_Allreduce / MPI_Isend / MPI_Irecv.
 * It also carries the set-up traffic of the distributed operator build (operator.c): the
 * header, ordering and scaling vectors go out by broadcast, each row panel by one send, the
 * rows a rank needs from its neighbours by an all-to-all of index lists.
 * Ranks that wait for rank 0 (which reads and partitions the matrix) sleep between polls:
 * MPICH's blocking calls spin, and the ranks of a node share its CPUs with rank 0's threads.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "pa_host.h"

/* ---- MPICH ABI (mpi.h of MPICH 3.x: lines 89-117, 287-312, 529-552, 585-591, 730) ---- */
typedef int mp_comm, mp_dtype, mp_op, mp_req, mp_info;
typedef struct { int count_lo, count_hi_and_cancelled, source, tag, error; } mp_status;
#define MP_BYTE   ((mp_dtype)0x4c00010d)
#define MP_INT    ((mp_dtype)0x4c000405)
#define MP_DOUBLE ((mp_dtype)0x4c00080b)
#define MP_SUM    ((mp_op)0x58000003)
#define MP_MIN    ((mp_op)0x58000002)
#define MP_MAX    ((mp_op)0x58000001)
#define MP_IN_PLACE ((void*)-1)
#define MP_STATUS_IGNORE ((mp_status*)1)
#define MP_INFO_NULL ((mp_info)0x1c000000)
#define MP_COMM_TYPE_SHARED 1
#define MP_COMM_WORLD ((mp_comm)0x44000000)

static struct {
  int resolved;        /* 0: not tried, 1: usable, -1: no MPI in this process */
  int (*Initialized)(int*);
  int (*Finalized)(int*);
  int (*Comm_rank)(mp_comm, int*);
  int (*Comm_size)(mp_comm, int*);
  int (*Comm_split_type)(mp_comm, int, int, mp_info, mp_comm*);
  int (*Comm_free)(mp_comm*);
  int (*Ibcast)(void*, int, mp_dtype, int, mp_comm, mp_req*);
  int (*Test)(mp_req*, int*, mp_status*);
  int (*Send)(const void*, int, mp_dtype, int, int, mp_comm);
  int (*Isend)(const void*, int, mp_dtype, int, int, mp_comm, mp_req*);
  int (*Irecv)(void*, int, mp_dtype, int, int, mp_comm, mp_req*);
  int (*Waitall)(int, mp_req*, mp_status*);
  int (*Allreduce)(const void*, void*, int, mp_dtype, mp_op, mp_comm);
  int (*Alltoall)(const void*, int, mp_dtype, void*, int, mp_dtype, mp_comm);
  int (*Alltoallv)(const void*, const int*, const int*, mp_dtype, void*, const int*, const int*, mp_dtype, mp_comm);
  int (*Barrier)(mp_comm);
  int (*Abort)(mp_comm, int);
} M;

static mp_comm g_comm;
static int g_active = 0, g_mrank = 0, g_msize = 1;
static int g_bound = 0;           /* device + hooks chosen */
static const char* g_binding = "none";

static void* sym(void* lib, const char* name) {
  void* p = dlsym(RTLD_DEFAULT, name);
  if (!p && lib) p = dlsym(lib, name);
  return p;
}

static int resolve(void) {
  if (M.resolved) return M.resolved > 0;
  M.resolved = -1;
  const char* off = getenv("PREALPS_MPI");
  if (off && !strcmp(off, "0")) return 0;
  /* only an MPI the main program has already loaded counts: never pull one in */
  void* lib = NULL;
  if (!dlsym(RTLD_DEFAULT, "MPI_Initialized")) {
    lib = dlopen("libmpi.so.12", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (!lib) return 0;
  }
#define S(f) do { *(void**)(&M.f) = sym(lib, "MPI_" #f); if (!M.f) return 0; } while (0)
  S(Initialized); S(Finalized); S(Comm_rank); S(Comm_size); S(Comm_split_type); S(Comm_free); S(Ibcast); S(Test);
  S(Send); S(Isend); S(Irecv); S(Waitall); S(Allreduce); S(Alltoall); S(Alltoallv); S(Barrier); S(Abort);
#undef S
  M.resolved = 1;
  return 1;
}

/* An MPICH handle of a communicator: kind in bits 31-30 (builtin / direct / indirect), object
 * type 1 in bits 29-26.  (An Open MPI communicator is a pointer and does not survive the int.) */
static int is_mpich_comm(int c) {
  unsigned u = (unsigned)c;
  return (u >> 30) != 0 && ((u >> 26) & 0xF) == 1;
}

/* 1: the process runs under an initialised MPI, `comm` has more than one rank and the caller
 * has not described the process group itself (preAlps_hip_set_world): the library takes rank
 * and size from the communicator, as the reference does (utils/operator.c:42-43).  0: no MPI to attach
 * to (one process).  -1: started by an MPI this library cannot speak to -- the caller must fail, NOT go on
 * alone (every rank would build and solve the whole problem). */
int pa_mpi_attach(MPI_Comm comm, int* rank, int* size) {
  if (g_active && (mp_comm)comm == g_comm) { *rank = g_mrank; *size = g_msize; return 1; }
  {
    /* an Open MPI launcher: its communicators are pointers, this library speaks the MPICH ABI only */
    const char* os = getenv("OMPI_COMM_WORLD_SIZE");
    if (os && atoi(os) > 1 && pa_world_size() == 1) {
      PA_FAIL("started by Open MPI with %s ranks: this library resolves MPI at run time with the MPICH ABI (MPICH, Intel MPI, "
              "MVAPICH, Cray MPI); with Open MPI describe the process group yourself (preAlps_hip_set_world / "
              "preAlps_hip_set_comm or preAlps_hip_rccl_init, INTEGRATION.md section 2)", os);
      return -1;
    }
  }
  if (!resolve()) return 0;
  int flag = 0;
  if (M.Initialized(&flag) || !flag) return 0;
  if (M.Finalized(&flag) || flag) return 0;
  if (!is_mpich_comm((int)comm)) return 0;
  int r = 0, s = 1;
  if (M.Comm_size((mp_comm)comm, &s) || M.Comm_rank((mp_comm)comm, &r)) return 0;
  if (s < 2) return 0;
  if (!g_active && pa_world_size() > 1) return 0;   /* the caller bound its own process group */
  g_comm = (mp_comm)comm; g_active = 1; g_mrank = r; g_msize = s;
  *rank = r; *size = s;
  return 1;
}
int pa_mpi_active(void) { return g_active; }
const char* pa_mpi_binding(void) { return g_binding; }

static void nap(void) {
  struct timespec ts = {0, 200000};   /* 0.2 ms */
  nanosleep(&ts, NULL);
}
/* Seconds a rank waits inside one set-up message (rank 0 may be reading and partitioning a large file
 * meanwhile); PREALPS_MPI_TIMEOUT overrides, 0 = for ever. */
static double wait_limit(void) {
  static double lim = -1.0;
  if (lim < 0.0) { const char* e = getenv("PREALPS_MPI_TIMEOUT"); lim = e ? atof(e) : 3600.0; if (lim < 0.0) lim = 0.0; }
  return lim;
}
static int quiet_wait(mp_req* rq) {
  int done = 0;
  const double t0 = pa_wtime(), lim = wait_limit();
  for (;;) {
    if (M.Test(rq, &done, MP_STATUS_IGNORE)) return 1;
    if (done) return 0;
    if (lim > 0.0 && pa_wtime() - t0 > lim) return PA_FAIL("no answer from the other ranks within %.0f s (PREALPS_MPI_TIMEOUT)", lim);
    nap();
  }
}

/* The reference ends a failed run with MPI_Abort(MPI_COMM_WORLD, 1) (utils/cplm_core/cplm_utils.c:42-58): when
 * an MPI is attached the library's abort goes the same way, so that no rank is left inside a collective. */
void pa_mpi_abort(void) {
  if (g_active && M.resolved > 0 && M.Abort) M.Abort(MP_COMM_WORLD, 1);
}

/* Collective: nonzero on every rank as soon as `rc` is nonzero on one -- called between the stages of the
 * set-up so that a rank that failed locally (memory, device) does not leave the others inside the next
 * collective when the library returns error codes instead of aborting. */
int pa_mpi_agree(int rc) {
  int v = rc ? 1 : 0;
  if (!g_active) return v;
  if (M.Allreduce(MP_IN_PLACE, &v, 1, MP_INT, MP_MAX, g_comm)) return 1;
  return v;
}

#define MP_CHUNK ((size_t)1 << 30)
int pa_mpi_bcast(void* buf, size_t bytes, int root) {
  char* p = (char*)buf;
  while (bytes > 0) {
    size_t n = bytes < MP_CHUNK ? bytes : MP_CHUNK;
    mp_req rq;
    if (M.Ibcast(p, (int)n, MP_BYTE, root, g_comm, &rq) || quiet_wait(&rq)) return PA_FAIL("MPI_Ibcast failed");
    p += n; bytes -= n;
  }
  return 0;
}
int pa_mpi_send(const void* buf, size_t bytes, int dest, int tag) {
  const char* p = (const char*)buf;
  do {
    size_t n = bytes < MP_CHUNK ? bytes : MP_CHUNK;
    if (M.Send(p, (int)n, MP_BYTE, dest, tag, g_comm)) return PA_FAIL("MPI_Send failed");
    p += n; bytes -= n;
  } while (bytes > 0);
  return 0;
}
int pa_mpi_recv(void* buf, size_t bytes, int src, int tag) {
  char* p = (char*)buf;
  do {
    size_t n = bytes < MP_CHUNK ? bytes : MP_CHUNK;
    mp_req rq;
    if (M.Irecv(p, (int)n, MP_BYTE, src, tag, g_comm, &rq) || quiet_wait(&rq)) return PA_FAIL("MPI_Irecv failed");
    p += n; bytes -= n;
  } while (bytes > 0);
  return 0;
}
int pa_mpi_min_int(int* v) {
  if (M.Allreduce(MP_IN_PLACE, v, 1, MP_INT, MP_MIN, g_comm)) return PA_FAIL("MPI_Allreduce failed");
  return 0;
}
int pa_mpi_max_int(int* v) {
  if (M.Allreduce(MP_IN_PLACE, v, 1, MP_INT, MP_MAX, g_comm)) return PA_FAIL("MPI_Allreduce failed");
  return 0;
}
int pa_mpi_barrier(void) { return M.Barrier(g_comm) ? PA_FAIL("MPI_Barrier failed") : 0; }

/* Every rank tells every other which of ITS rows it needs: want[] lists global row ids grouped by
 * owner (want_cnt[g] of them for rank g); on return *asked holds the ids the others want from us,
 * grouped by asking rank (asked_cnt[g]).  The owner's send list is what the receiver asked for,
 * in the receiver's order (utils/cplm_v0/cplm_v0_matmult_v2.c:184-192 ships whole panels instead). */
int pa_mpi_swap_lists(const int* want, const int* want_cnt, int** asked, int* asked_cnt) {
  int s = g_msize;
  int* sd = (int*)malloc((size_t)s * sizeof(int));
  int* rd = (int*)malloc((size_t)s * sizeof(int));
  if (!sd || !rd) { free(sd); free(rd); return PA_FAIL("out of host memory"); }
  if (M.Alltoall(want_cnt, 1, MP_INT, asked_cnt, 1, MP_INT, g_comm)) { free(sd); free(rd); return PA_FAIL("MPI_Alltoall failed"); }
  long long tot = 0;
  for (int g = 0; g < s; ++g) { sd[g] = g ? sd[g - 1] + want_cnt[g - 1] : 0; rd[g] = (int)tot; tot += asked_cnt[g]; }
  int* got = (int*)malloc((size_t)(tot ? tot : 1) * sizeof(int));
  if (!got) { free(sd); free(rd); return PA_FAIL("out of host memory"); }
  int rc = M.Alltoallv(want, want_cnt, sd, MP_INT, got, asked_cnt, rd, MP_INT, g_comm);
  free(sd); free(rd);
  if (rc) { free(got); return PA_FAIL("MPI_Alltoallv failed"); }
  *asked = got;
  return 0;
}

/* ---- the two hooks staged through the host ------------------------------------------------ */
static double* g_pin = NULL; static size_t g_pin_cap = 0;     /* pinned: all-reduce, and send | recv */
static int pin_reserve(size_t doubles) {
  if (doubles <= g_pin_cap) return 0;
  pa_rt_host_free(g_pin);
  g_pin_cap = doubles * 2 > 4096 ? doubles * 2 : 4096;
  g_pin = (double*)pa_rt_host_alloc(g_pin_cap * sizeof(double));
  if (!g_pin) { g_pin_cap = 0; return 1; }
  return 0;
}
static int host_allreduce(void* ctx, double* dev_buf, int count) {
  (void)ctx;
  if (pin_reserve((size_t)count)) return 1;
  if (pa_rt_d2h(g_pin, dev_buf, (size_t)count * sizeof(double))) return 1;     /* (waits for the stream) */
  if (M.Allreduce(MP_IN_PLACE, g_pin, count, MP_DOUBLE, MP_SUM, g_comm)) return 1;
  return pa_rt_h2d(dev_buf, g_pin, (size_t)count * sizeof(double));
}
static int host_exchange(void* ctx, const double* dev_send, const int* send_counts, double* dev_recv,
                         const int* recv_counts, const int* peers, int npeers) {
  (void)ctx;
  size_t ns = 0, nr = 0;
  for (int i = 0; i < npeers; ++i) { ns += (size_t)send_counts[i]; nr += (size_t)recv_counts[i]; }
  if (pin_reserve(ns + nr)) return 1;
  double* hs = g_pin; double* hr = g_pin + ns;
  if (pa_rt_d2h(hs, dev_send, ns * sizeof(double))) return 1;
  mp_req* rq = (mp_req*)malloc((size_t)(2 * npeers + 1) * sizeof(mp_req));
  if (!rq) return 1;
  int nq = 0, rc = 0;
  size_t so = 0, ro = 0;
  for (int i = 0; i < npeers && !rc; ++i) {
    if (recv_counts[i] > 0) rc = M.Irecv(hr + ro, recv_counts[i], MP_DOUBLE, peers[i], 77, g_comm, &rq[nq++]);
    if (!rc && send_counts[i] > 0) rc = M.Isend(hs + so, send_counts[i], MP_DOUBLE, peers[i], 77, g_comm, &rq[nq++]);
    so += (size_t)send_counts[i]; ro += (size_t)recv_counts[i];
  }
  if (!rc && nq) rc = M.Waitall(nq, rq, MP_STATUS_IGNORE);
  free(rq);
  if (rc) return 1;
  return pa_rt_h2d(dev_recv, hr, nr * sizeof(double));
}

/* Position of this rank among the ranks of its node (what picks the device). */
static int local_rank(int* lrank, int* lsize) {
  mp_comm node;
  if (M.Comm_split_type(g_comm, MP_COMM_TYPE_SHARED, g_mrank, MP_INFO_NULL, &node)) return 1;
  int rc = M.Comm_rank(node, lrank) || M.Comm_size(node, lsize);
  M.Comm_free(&node);
  return rc;
}

/* Device and hooks for the ranks of the attached communicator; collective, done once.
 * PREALPS_COMM = rccl | mpi | auto (default): auto takes RCCL when no two ranks share a device
 * and librccl loads on every rank. */
int pa_mpi_bind(void) {
  if (g_bound) return 0;
  int lr = 0, ls = 1;
  if (local_rank(&lr, &ls)) return PA_FAIL("MPI_Comm_split_type failed");
  int ndev = pa_rt_device_count();
  int bad = 0;
  if (!pa_rt_ready()) {
    if (ndev < 1) bad = PA_FAIL("HIP device unavailable: %s", "no device is visible to this rank; this library has no CPU path");
    else bad = preAlps_hip_init(lr % ndev);
  }
  if (pa_mpi_agree(bad)) return bad ? 1 : PA_FAIL("another rank found no usable device");
  const char* how = getenv("PREALPS_COMM");
  const char* alt = getenv("PREALPS_RCCL_LIB");      /* (a stand-in library may serve several ranks of one device) */
  int want_rccl = !(how && !strcmp(how, "mpi"));
  int can = want_rccl && (ls <= ndev || (alt && *alt)) && pa_rccl_available() == 0;
  if (pa_mpi_min_int(&can)) return 1;
  if (how && !strcmp(how, "rccl") && !can)
    return PA_FAIL("PREALPS_COMM=rccl, but %s", ls > ndev ? "several ranks share a device (RCCL refuses that)" : pa_rccl_error());
  if (can) {
    char id[128];
    memset(id, 0, sizeof(id));
    bad = (g_mrank == 0) ? preAlps_hip_rccl_unique_id(id) : 0;
    if (pa_mpi_agree(bad)) return bad ? 1 : PA_FAIL("rank 0 could not create the RCCL id");
    if (pa_mpi_bcast(id, sizeof(id), 0)) return 1;
    bad = preAlps_hip_rccl_init(id, g_mrank, g_msize);
    if (pa_mpi_agree(bad)) return bad ? 1 : PA_FAIL("another rank could not join the RCCL communicator");
    g_binding = "rccl";
  } else {
    if (preAlps_hip_set_world(g_mrank, g_msize)) return 1;
    if (preAlps_hip_set_comm(host_allreduce, host_exchange, NULL)) return 1;
    g_binding = "mpi-host-staged";
  }
  if (preAlps_hip_comm_selftest()) return 1;
  g_bound = 1;
  if (getenv("PREALPS_SETUP_TRACE"))
    fprintf(stderr, "[setup] rank %d of %d: local rank %d of %d, %d device(s) visible, hooks: %s\n", g_mrank, g_msize, lr, ls,
            ndev, g_binding);
  return 0;
}

void pa_mpi_release(void) {
  pa_rt_host_free(g_pin);
  g_pin = NULL; g_pin_cap = 0;
}
