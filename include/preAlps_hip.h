/*
 * preAlps_hip.h -- entry points of libprealps_hip.so that have no counterpart
 * in the reference: device/context management, the process-group hooks, the
 * in-memory operator builder and host<->HBM panel transfers.  Everything a
 * reference driver needs is in preAlps_abi.h; this header is what a new
 * binding (ctypes, cgo, JNI, Fortran ISO_C) adds on top.  Plain C ABI: only
 * pointers, ints, doubles.
 */
#ifndef PREALPS_HIP_H
#define PREALPS_HIP_H

#include "preAlps_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- context ----------------------------------------------------------- */
/* Select the HIP device of this process and create the library's stream.
 * Fails (non-zero) when no gfx950 device is usable: there is no CPU fallback. */
int preAlps_hip_init(int device);
void preAlps_hip_shutdown(void);
/* Run every kernel on the caller's stream (a hipStream_t), e.g. the stream a
 * host framework issues its collectives on.  NULL restores the own stream. */
int preAlps_hip_set_stream(void* hip_stream);
void* preAlps_hip_get_stream(void);
int preAlps_hip_sync(void);
/* 1 (default): errors print "ABORTING from <fn> : [Proc: r] ..." and abort(),
 * like CPLM_Abort (utils/cplm_core/cplm_utils.c:17-58).  0: entry points
 * return non-zero and preAlps_hip_last_error() holds the message. */
void preAlps_hip_set_abort_mode(int abort_on_error);
const char* preAlps_hip_last_error(void);
/* Row stride (in doubles) of a device panel for a given enlarging factor. */
int preAlps_hip_panel_stride(int enlFac);

/* ---- process group ----------------------------------------------------- */
/* One process per GPU.  rank/size describe the processes that share one
 * operator; the library itself never opens a socket: sums and halo rows go
 * through the two hooks, which the host binds to RCCL (torch.distributed
 * backend "nccl", MPI, ...).  Buffers handed to the hooks are device memory
 * and all work queued on preAlps_hip_get_stream() must be ordered before the
 * hook's own communication (the hook is responsible for that ordering). */
typedef int (*preAlps_allreduce_fn)(void* ctx, double* dev_buf, int count);
/* peers[i] sends recv_counts[i] doubles to us and gets send_counts[i] doubles
 * from us; both buffers are packed peer after peer in the order of `peers`. */
typedef int (*preAlps_exchange_fn)(void* ctx, const double* dev_send,
                                   const int* send_counts, double* dev_recv,
                                   const int* recv_counts, const int* peers,
                                   int npeers);
int preAlps_hip_set_world(int rank, int size);
int preAlps_hip_set_comm(preAlps_allreduce_fn allreduce,
                         preAlps_exchange_fn exchange, void* ctx);

/* Built-in hooks: RCCL (ncclAllReduce / grouped ncclSend+ncclRecv over xGMI) on the
 * library stream.  Rank 0 creates the 128-byte id, the launcher broadcasts it, every
 * rank calls preAlps_hip_rccl_init (which also does set_world + set_comm). */
int preAlps_hip_rccl_available(void);   /* 0: librccl.so loads; ranks vote on this before the collective init */
int preAlps_hip_rccl_unique_id(char* id128);
int preAlps_hip_rccl_init(const char* id128, int rank, int size);
/* One shard of a `size`-process run rehearsed in a single process: rank / size as given, the
 * all-reduce hook is the identity and halo rows arrive as zeros (the process then iterates on its own
 * diagonal block of the partitioned matrix with the launches of a real rank).  For measurements. */
int preAlps_hip_loopback(int rank, int size);
/* Collective: checks the installed hooks with one all-reduce and one ring exchange. */
int preAlps_hip_comm_selftest(void);

/* ---- operator from memory ---------------------------------------------- */
/* Same pipeline as preAlps_OperatorBuild (utils/operator.c:38-134) with the
 * matrix taken from memory instead of a MatrixMarket file and an explicit
 * partition vector instead of METIS:  [scale: A <- D A D]  ->  rows grouped
 * part by part (original order inside a part)  ->  symmetric permutation  ->
 * row panel of this process (parts [rank*nparts/size, (rank+1)*nparts/size)).
 * part == NULL selects part[r] = floor(r*nparts/N).  Every process passes the
 * whole matrix (global CSR, 0-based, full symmetric pattern). */
int preAlps_OperatorBuildFromCSR(int N, const int* rowPtr, const int* colInd,
                                 const double* val, int nparts, const int* part,
                                 int scale);
/* k-way partition of the adjacency graph of a square matrix with a structurally symmetric
 * pattern (0-based CSR, diagonal stored) into nparts compact, balanced parts:
 * part[i] in [0, nparts) for every row i.  This is what the library calls in place of the
 * reference's METIS_PartGraphKway (utils/cplm_core/cplm_matcsr_core.c:394-457); consecutive
 * part ids are neighbouring regions, so a process that owns a range of ids owns a compact
 * piece of the domain.  Host code: needs no GPU. */
int preAlps_hip_partition_kway(int N, const int* rowPtr, const int* colInd, int nparts, int* part);
/* Host-side self check of the sparse factorisation that large diagonal blocks get (nested
 * dissection + multifrontal Cholesky, nd.c), on one SPD matrix taken as a single block; no GPU
 * needed.  stats[0..7] = supernodes, doubles of one panel copy, rows of the largest front, tree
 * height, ||L L^T x - A x|| / ||A x||, largest relative mismatch between the two panel copies,
 * largest deviation of the selective-inversion panels [T ; -G] from T (I + Lhat_11) = I and
 * G (I + Lhat_11) = Lhat_21, pivot columns of the widest supernode. */
int preAlps_hip_nd_selfcheck(int n, const int* rowPtr, const int* colInd, const double* val, int leaf_rows,
                             double* stats);
/* Cut the SpMM plan (slices, LDS staging lists) for this enlarging factor now rather than
 * inside the first preAlps_BlockOperator call; optional. */
int preAlps_hip_prepare_operator(int enlFac);
/* Plan-only mode: build the sharding and halo lists on the host without a GPU
 * (used by the multi-process CPU tests); preAlps_BlockOperator is refused. */
void preAlps_hip_plan_only(int on);
/* The halo plan of this process (library-owned arrays): peers[i] gets
 * send_rows[i] of our rows (local row ids in send_idx, peer after peer) and
 * owns recv_rows[i] of our halo slots (their global rows in halo_cols). */
int preAlps_OperatorGetHaloPlan(int* npeers, int** peers, int** send_rows, int** recv_rows,
                                int** send_idx, int* nsend, int** halo_cols, int* nhalo);
/* The inverse of send_idx, as the solver's update kernel uses it to pack the send buffer itself (several processes,
 * DESIGN section 5): local row r goes into the send-buffer slots slot_out[off_out[r] .. off_out[r + 1]).
 * off_out: m + 1 ints, slot_out: nsend ints (caller's arrays).  Works in plan-only mode. */
int preAlps_hip_pack_map(int* off_out, int* slot_out);
/* perm[new] = old, of the whole problem (library-owned, N ints). */
int preAlps_OperatorGetPermPtr(int** perm, int* n);
int preAlps_hip_nparts(void);

/* ---- helpers for drivers and tests ------------------------------------- */
/* The rhs of examples/test_ecg_prealps_op.c:172-184 for the local rows, as a
 * run of the reference with np = nparts ranks would build it. */
int preAlps_hip_reference_rhs(double* rhs_local);
/* The driver loop of examples/test_ecg_prealps_op.c:203-223 (fused variant:
 * examples/test_ecg_bench_fused.c:243-259). res_hist may be NULL. */
int preAlps_ECGSolve(preAlps_ECG_t* ecg, double* rhs, double* sol,
                     double* res_hist, int* bs_hist, int max_hist, int* n_hist);
/* 1 / 0: the two driver loops above and below replay each half of an iteration from a HIP graph
 * captured on its first passes (default: off, or PREALPS_ECG_GRAPH; plain launches measured faster). */
void preAlps_hip_graphs(int on);
/* The same loop advanced by nsteps full iterations from the current RCI state,
 * restarting from rhs when the stopping test fires (counts go to the optional
 * out-parameters).  Used for timing a fixed number of iterations. */
int preAlps_ECGAdvance(preAlps_ECG_t* ecg, double* rhs, int* rci_request, int nsteps, int* restarts,
                       int* last_iters, double* last_res);
/* Device panel <-> host column-major array (ld >= m). */
int preAlps_hip_panel_alloc(CPLM_Mat_Dense_t* A, int M, int N, int m, int n, int enlFac);
void preAlps_hip_panel_free(CPLM_Mat_Dense_t* A);
int preAlps_hip_panel_to_host(const CPLM_Mat_Dense_t* A, int enlFac, double* host, int ld);
int preAlps_hip_panel_from_host(CPLM_Mat_Dense_t* A, int enlFac, const double* host, int ld);
/* The tall-skinny panel kernels of the iteration on their own (what a kernel-level check calls;
 * the solver launches the same kernels): host_out (ld_out >= rows) = [A0 | A1]^T B, column major
 * (the dgemm of ecg.c:311,330,347,425,438,510);  Z -= [V0 | V1] beta with beta on the host,
 * column major (ecg.c:354,517);  P <- P U^-1, AP <- AP U^-1, X += P alpha, R -= AP alpha and
 * *host_res2 = sum of squares of the new R (ecg.c:324-338,434-435,500-501,250), U upper
 * triangular t x t, alpha t x X.n, both on the host, column major.  A1 / V1 may be NULL. */
int preAlps_hip_panel_gram(const CPLM_Mat_Dense_t* A0, const CPLM_Mat_Dense_t* A1, const CPLM_Mat_Dense_t* B,
                           double* host_out, int ld_out);
int preAlps_hip_panel_update(CPLM_Mat_Dense_t* Z, const CPLM_Mat_Dense_t* V0, const CPLM_Mat_Dense_t* V1,
                             const double* host_beta, int ldb);
int preAlps_hip_panel_trsm_update(CPLM_Mat_Dense_t* P, CPLM_Mat_Dense_t* AP, CPLM_Mat_Dense_t* X,
                                  CPLM_Mat_Dense_t* R, const double* host_U, const double* host_alpha,
                                  double* host_res2);
/* BF-Omin's second half on its own (ecg.c:358-393): P(:, c) = Z(:, piv[c]), c < Z.n, then the leading t columns
 * times U^-1 (t x t upper triangular, host, column major, piv 0-based on the host); one_pass = the solver's single
 * kernel, 0 = the three kernels it replaces (same bits). */
int preAlps_hip_panel_permute_solve(const CPLM_Mat_Dense_t* Z, CPLM_Mat_Dense_t* P, const int* host_piv, int t,
                                    const double* host_U, int one_pass);
/* Numeric facts about the built operator / preconditioner, by name:
 * "nnz_local", "rows_local", "halo_rows", "spmm_blocks", "bj_factor_bytes",
 * "bj_max_bandwidth", "bj_parts_local", "bj_nd_blocks" (blocks with the sparse factor),
 * "bj_nd_inverse_dev" (largest deviation of its inverted pivot triangles), ...  Returns non-zero
 * for unknown keys. */
int preAlps_hip_get_stat(const char* key, double* value);
/* A stopwatch made of two hipEvents on the library stream: start records the
 * first, stop records the second, waits for it and returns the device time
 * between them (what bench.py uses to time a batch of launches). */
int preAlps_hip_timer_start(void);
int preAlps_hip_timer_stop(double* seconds);
/* Streaming ceilings of the device from two calibration kernels (plain copy, plain read) over
 * `bytes` of freshly allocated HBM, `reps` launches each, timed with the stopwatch above. */
int preAlps_hip_hbm_probe(size_t bytes, int reps, double* copy_GBs, double* read_GBs);
/* Per-phase device time in seconds accumulated since the last reset, from
 * hipEvents on the library stream when timing is enabled (it adds a stream
 * sync per call, so it is off by default).  Keys: "operator", "precond",
 * "gram", "trsm", "update", "small", "comm". */
void preAlps_hip_timing(int enable);
void preAlps_hip_timing_reset(void);
int preAlps_hip_get_time(const char* key, double* seconds);

#ifdef __cplusplus
}
#endif
#endif
