/*
 * pa_host.h -- internal declarations shared by the C host sources of
 * libprealps_hip.so (context.c, operator.c, block_jacobi.c, ecg.c).
 */
#ifndef PA_HOST_H
#define PA_HOST_H

#include <stddef.h>
#include "preAlps_hip.h"
#include "pa_device.h"

/* ---- errors (CPLM_Abort / CPLM_ASSERT, utils/cplm_core/cplm_utils.h:31-37) */
int pa_fail_at(const char* func, const char* fmt, ...);
#define PA_FAIL(...) pa_fail_at(__func__, __VA_ARGS__)
#define PA_CHECK(call)                                                       \
  do {                                                                       \
    if ((call) != 0) return PA_FAIL("%s failed: %s", #call, pa_rt_error());  \
  } while (0)
#define PA_REQUIRE_GPU()                                                     \
  do {                                                                       \
    if (!pa_rt_ready() && preAlps_hip_init(pa_default_device()) != 0)        \
      return PA_FAIL("HIP device unavailable: %s", pa_rt_error());           \
  } while (0)

int pa_default_device(void);
double pa_wtime(void);
int pa_host_threads(void);            /* OpenMP threads worth starting here (cgroup quota, affinity) */

/* ---- process group ------------------------------------------------------ */
int pa_world_rank(void);
int pa_world_size(void);
int pa_comm_is_loopback(void);
/* sum a device buffer over the processes (no-op for one process) */
int pa_allreduce(double* dev_buf, int count);
int pa_exchange(const double* dev_send, const int* send_counts, double* dev_recv,
                const int* recv_counts, const int* peers, int npeers);

/* ---- an MPI launcher around us (mpi_glue.c): resolved at run time, MPICH ABI -------------- */
int pa_mpi_attach(MPI_Comm comm, int* rank, int* size);   /* 1: take rank / size from comm (> 1 ranks); 0: none; -1: an MPI this library cannot use */
void pa_mpi_abort(void);                                  /* MPI_Abort(MPI_COMM_WORLD, 1) when an MPI is attached */
int pa_mpi_agree(int rc);                                 /* collective: nonzero everywhere when nonzero anywhere */
int pa_mpi_active(void);
const char* pa_mpi_binding(void);                         /* "rccl", "mpi-host-staged", "none" */
int pa_mpi_bind(void);                                    /* device of this rank + the two hooks; collective */
int pa_mpi_bcast(void* buf, size_t bytes, int root);      /* waiting ranks sleep between polls */
int pa_mpi_send(const void* buf, size_t bytes, int dest, int tag);
int pa_mpi_recv(void* buf, size_t bytes, int src, int tag);
int pa_mpi_min_int(int* v);
int pa_mpi_max_int(int* v);
int pa_mpi_barrier(void);
int pa_mpi_swap_lists(const int* want, const int* want_cnt, int** asked, int* asked_cnt);
void pa_mpi_release(void);
void pa_host_solo(int on);            /* 1: the other ranks of the node sleep: pa_host_threads() = the whole CPU share */

/* ---- phase timing ------------------------------------------------------- */
enum { PA_T_OPERATOR, PA_T_PRECOND, PA_T_GRAM, PA_T_TRSM, PA_T_UPDATE, PA_T_SMALL, PA_T_COMM, PA_T_COUNT };
int pa_timing_enabled(void);
void pa_time_begin(int key);
double pa_time_end(int key);   /* device seconds of the closed outermost region, else -1 */

/* ---- operator state shared with block_jacobi.c / ecg.c ------------------ */
typedef struct {
  int built;
  int N;            /* global rows */
  int nparts;       /* subdomains (the reference's ranks) */
  int part0, part1; /* parts owned by this process: [part0, part1) */
  int row_off;      /* first global row owned */
  int m;            /* local rows */
  int* rowPos;      /* nparts + 1, global */
  int* perm;        /* N: perm[new] = old */
  CPLM_Mat_CSR_t A; /* local row panel, global column ids (host) */
  int halo;         /* halo rows */
} pa_operator_info_t;
const pa_operator_info_t* pa_operator_info(void);
/* Workgroups of the SpMM at panel stride ts when it can leave the ECG Gram block behind
 * (pa_k_spmm_gram_arm: run plan, 4 columns); builds the plan if need be.  0: it cannot. */
int pa_operator_gram_blocks(int ts);
/* 1: the caller's kernel packs the send rows of the panel X it is about to write (row r into the slots
 * pk_slot[pk_off[r] .. pk_off[r + 1]) of sendbuf, ts doubles each); the next preAlps_BlockOperator(X, .) skips its pack */
int pa_operator_pack_hint(int ts, const double* X, const int** pk_off, const int** pk_slot, double** sendbuf);

int pa_panel_stride(int enlFac);
static inline int pa_desc_stride(const CPLM_Mat_Dense_t* A) { return A->info.lda; }
void pa_set_desc(CPLM_Mat_Dense_t* A, int M, int N, int m, int n, int ts);

/* ---- nested dissection of a diagonal block (partition.c), for the sparse block solve (nd.c) */
typedef struct {
  int nsn;          /* supernodes (leaves and separators) in postorder: children before parents */
  int* first;       /* nsn + 1: rows [first[s], first[s+1]) of the new order form supernode s */
  int* parent;      /* nsn: parent supernode, -1 for a root */
  int* perm;        /* n: perm[new] = old (block-local row) */
} pa_nd_tree_t;
int pa_nd_order(int n, const int* rp, const int* ci, int leaf_rows, pa_nd_tree_t* t);
void pa_nd_tree_free(pa_nd_tree_t* t);

/* sparse block solve for large blocks (nd.c); returns 0, 1 (error reported) or 2 (*fail_row = local
 * panel row of a non-positive pivot) */
int pa_nd_create(const CPLM_Mat_CSR_t* A, int nblk, const int* blocks, const int* row0, const int* nrows,
                 const int* grow0, int m_local, int* fail_row);
int pa_nd_apply(int ts, const double* in, double* out);
void pa_nd_free(void);
double pa_nd_factor_bytes(void);
double pa_nd_inverse_deviation(void);   /* largest |T (I + Lhat) - I| over the fronts of the last pa_nd_create */
int pa_nd_active(void);

double pa_bj_factor_bytes(void);
int pa_bj_max_bandwidth(void);
double pa_bj_setup_seconds(int which);   /* 0: ordering + band Cholesky, 1: sweep layouts + upload */
int pa_bj_nparts(void);
int pa_bj_nd_blocks(void);
int pa_bj_gram_blocks(void);       /* blocks of an apply that can leave [in | prev]^T out behind (0: cannot) */
double pa_bj_g4_bytes(void);        /* bytes of the one-copy records of bj_g4.hip, 0 if absent */
double pa_bj_pairs_bytes(void);     /* bytes of the paired sweep records (both sweeps), 0 if absent */

#endif
