/*
 * ecg.c -- Enlarged Conjugate Gradient, reverse-communication state machine,
 * with every panel resident in HBM and every O(m) operation a HIP kernel.
 *
 * Reference behaviour kept (src/solvers/ecg.c under /root/reference):
 *   _preAlps_ECGMalloc          :41-96     one pool, V | AV | Z | R | X | small
 *   _preAlps_ECGReset / Split   :98-171, :201-221
 *   preAlps_ECGInitialize       :173-199   (size >= enlFac check on nparts)
 *   preAlps_ECGStoppingCriterion:223-271   Frobenius norm of the residual block
 *   _preAlps_ECGIterateOmin     :289-400   incl. BF-Omin (dpstrf / dlapmt)
 *   _preAlps_ECGIterateOdir     :402-530   incl. D-Odir (dgesvd / dgeqrf / dormqr)
 *   _preAlps_ECGIterateOdirFused:532-658   one reduction per iteration
 *   preAlps_ECGFinalize/WrapUp/Free :660-692, preAlps_ECGPrint :694-728
 *
 * MI355X design instead of the reference's:
 *   - panels are row-interleaved [m][ts] so one CSR gather touches one 8*ts-byte
 *     chunk; descriptors say ROW_MAJOR with lda = ts and hold device pointers;
 *   - without block-size reduction the three mkl_domatcopy per iteration
 *     (ecg.c:521-523) are pointer rotations among the P / P_prev / Z buffers;
 *   - t x t blocks stay on the device (Gram partials -> fixed-order finish ->
 *     all-reduce hook -> one-wave Cholesky); the host only reads the residual
 *     norm once per iteration, inside preAlps_ECGStoppingCriterion.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pa_host.h"
#include "smalldense.h"

#define PA_ECG_MAGIC 0x45434731u

/* 1: the driver loops of this library (preAlps_ECGSolve / preAlps_ECGAdvance) replay the launches of
 * an iteration from HIP graphs captured on the first passes (see seg_begin below); -1 = decide from
 * PREALPS_ECG_GRAPH (default 0) at the next reset. */
static int g_graphs = -1;
/* > 0 inside preAlps_ECGSolve / preAlps_ECGAdvance.  Only there does the library ask its SpMM and its block
 * solve for the Gram blocks of the next half-iteration: between preAlps_BlockOperator / BlockJacobiApply and
 * preAlps_ECGIterate nothing else touches AP or Z in those loops, whereas a caller that drives the RCI
 * protocol itself may (a shifted operator, a second preconditioner stage), and the blocks would be stale.
 * PREALPS_RCI_FUSE=1: also for RCI callers who only call the two library routines. */
static int g_own_loop = 0;
static int rci_fuse(void) { static int v = -1; if (v < 0) { const char* e = getenv("PREALPS_RCI_FUSE"); v = e ? atoi(e) : 0; } return v; }
void preAlps_hip_graphs(int on) { g_graphs = on; }

/* The reference brackets every BLAS / MPI call with MPI_Wtime (ecg.c:316-320 ...).  Launches
 * are asynchronous here, so the host clock only sees the enqueue; with preAlps_hip_timing(1)
 * the phase's hipEvent pair is read instead (one stream sync per phase) and the timer fields
 * of preAlps_ECG_t hold device time, which is what preAlps_ECGPrint then reports. */
#define TIC(key) do { pa_time_begin(key); t0 = pa_wtime(); } while (0)
#define TAC(key, field) \
  do { double d_ = pa_time_end(key); ecg->field += d_ >= 0.0 ? d_ : pa_wtime() - t0; } while (0)

typedef struct {
  unsigned magic;
  int ts, T, m;
  size_t pool_doubles;
  /* panel buffers (device) */
  double* buf_v[2];   /* P slot, P_prev slot */
  double* buf_av[2];  /* AP slot, AP_prev slot */
  double* buf_z;
  double* d_R; double* d_X;
  /* small blocks (device): F = [alpha T^2 | beta 2T^2 | mu T^2 | rtr T^2] */
  double* d_F; double* d_alpha; double* d_beta; double* d_mu; double* d_rtr;
  double* d_res2; double* d_q;
  double* d_partials; double* d_rtr_part;
  double* d_bj_parts; int bj_cap;       /* Gram blocks left behind by the block solve (pa_k_bj_gram_arm), 32 doubles per block */
  double* d_spmm_parts; int spmm_cap;   /* Gram blocks left behind by the SpMM (pa_k_spmm_gram_arm), 32 doubles each */
  int rtr_nblk, rtr_valid;
  int* d_info; int* d_piv;
  double* h_pin;      /* pinned: [0] res2, [1..] scratch */
  int* h_pin_i;
  int rotate;         /* NO_BS_RED: rotate pointers instead of copying */
  void* ev_res;       /* recorded after the residual norm has been copied to the host */
  int fuse;           /* NO_BS_RED: two-pass first half (PREALPS_ECG_FUSE=0 keeps the four-pass one) */
  /* Lazy normalisation (Orthodir, NO_BS_RED; PREALPS_ECG_LAZY_NORM=0 switches it off):
   * P <- P U^-1 and AP <- AP U^-1 (ecg.c:434-435 of the reference) are never written.  X and R get the same update
   * from rows normalised in registers; the block solve runs on AP_raw; the Gram blocks are formed on the raw
   * panels; and the update kernel of the second half applies U^-1 (this iteration's for P and Z, the previous
   * one's for P_prev) where the reference's panels would carry it -- the same algebra, two panel writes and
   * their 66 MB per iteration less on the headline problem.  d_uu: the two factors, uu_cur: this iteration's. */
  int lazy_norm, uu_cur;
  /* D-Odir rides the same path while every direction is live (ecg->P->info.n == enlFac); the first reduction
   * writes the normalised panels (P, AP with this iteration's factor, P_prev, AP_prev with the previous one's)
   * and ends it for the rest of the solve.  z_ready: the library's own loops queue the block solve Z = M^-1 AP
   * (AP is the raw panel, which nothing rewrites) BEFORE the host waits for alpha and decides about a reduction,
   * so the wait and the SVD are hidden behind it; the loop then skips its own apply.  ev_alpha: alpha is on the host. */
  int z_ready;
  int zz_nblk;        /* partial blocks of Z^T Z the last update kernel left in d_partials (BF-Omin), 0 = none */
  void* ev_alpha;
  double* d_uu;
  int poll;           /* the host polls the word a kernel writes behind the norm instead of waiting for an event */
  double seq, sent_seq, wait_seq;   /* last number handed out / the one travelling with the current norm / awaited */
  int lazy_stop;      /* several processes: the residual norm rides on the beta all-reduce (see below) */
  double* lazy_ptr;   /* where the first half left [res2, potrf status] for that */
  /* graphs (driver loops of this library): an iteration is two segments -- 0: the first half up to
   * the residual norm, 1: preconditioner apply + second half + product -- and the panel pointers
   * rotate with period 6 (three P slots x two AP slots), so there are 2 x 6 graphs */
  int use_graphs;
  int phase;          /* iterations since the last reset, mod 6 */
  void* graph[2][6];
  unsigned char seen[2][6];
} ecg_priv_t;

static ecg_priv_t* priv_of(preAlps_ECG_t* ecg) {
  if (!ecg || !ecg->iwork) return NULL;
  int T = ecg->enlFac;
  int skip = (T + 3) & ~3; /* iwork[0..T) stays the pivot array of the reference */
  ecg_priv_t* p = (ecg_priv_t*)(ecg->iwork + skip);
  return p->magic == PA_ECG_MAGIC ? p : NULL;
}

static void publish_pointers(preAlps_ECG_t* ecg, ecg_priv_t* pv) {
  ecg->V->val = pv->buf_v[0];
  ecg->AV->val = pv->buf_av[0];
  ecg->P->val = pv->buf_v[0];
  ecg->AP->val = pv->buf_av[0];
  ecg->Z->val = pv->buf_z;
  ecg->R->val = pv->d_R;
  ecg->X->val = pv->d_X;
  ecg->alpha->val = pv->d_alpha;
  ecg->beta->val = pv->d_beta;
  ecg->P_p = pv->buf_v[0];
  ecg->AP_p = pv->buf_av[0];
  ecg->R_p = pv->d_R;
  ecg->Z_p = pv->buf_z;
}

static int pa_env_flag(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

/* ------------------------------------------------------------- malloc ---- */
int _preAlps_ECGMalloc(preAlps_ECG_t* ecg) {
  PA_REQUIRE_GPU();
  int m = ecg->locPbSize, T = ecg->enlFac;
  if (T < 1 || T > 16) return PA_FAIL("enlarging factor %d outside the supported range 1..16", T);
  int ts = pa_panel_stride(T);
  ecg->X = (CPLM_Mat_Dense_t*)calloc(1, sizeof(CPLM_Mat_Dense_t));
  ecg->R = (CPLM_Mat_Dense_t*)calloc(1, sizeof(CPLM_Mat_Dense_t));
  ecg->V = (CPLM_Mat_Dense_t*)calloc(1, sizeof(CPLM_Mat_Dense_t));
  ecg->AV = (CPLM_Mat_Dense_t*)calloc(1, sizeof(CPLM_Mat_Dense_t));
  ecg->Z = (CPLM_Mat_Dense_t*)calloc(1, sizeof(CPLM_Mat_Dense_t));
  ecg->alpha = (CPLM_Mat_Dense_t*)calloc(1, sizeof(CPLM_Mat_Dense_t));
  ecg->beta = (CPLM_Mat_Dense_t*)calloc(1, sizeof(CPLM_Mat_Dense_t));
  ecg->P = (CPLM_Mat_Dense_t*)calloc(1, sizeof(CPLM_Mat_Dense_t));
  ecg->AP = (CPLM_Mat_Dense_t*)calloc(1, sizeof(CPLM_Mat_Dense_t));
  int skip = (T + 3) & ~3;
  ecg->iwork = (int*)calloc(1, skip * sizeof(int) + sizeof(ecg_priv_t));
  ecg_priv_t* pv = (ecg_priv_t*)(ecg->iwork + skip);
  pv->magic = PA_ECG_MAGIC; pv->ts = ts; pv->T = T; pv->m = m;
  /* pool: same order as the reference (V, AV, Z, R, X, then the small blocks);
   * Orthomin needs no P_prev / AP_prev slots (ecg.c:55-61) */
  size_t panel = (size_t)(m > 0 ? m : 1) * ts;
  int nv = (ecg->ortho_alg == ORTHOMIN) ? 1 : 2;
  size_t small = 5 * (size_t)T * T + 2 * (size_t)T * T /* q, scratch */ + 8 + 2 * (size_t)T * T /* uu */;
  size_t parts = (size_t)pa_gram_max_blocks() * (2 * (size_t)ts * ts + ts);
  /* T = 4: the library's SpMM can form [AP | R]^T P while it computes AP (one block per workgroup) */
  pv->spmm_cap = (T == 4 && ts == 4) ? pa_operator_gram_blocks(ts) : 0;
  if (pv->spmm_cap) parts += ((size_t)pv->spmm_cap + pa_finish32_scratch_blocks()) * 32;
  /* ... and the library's block solve beta = [AP | AP_prev]^T Z while Z is still in its registers (Orthodir) */
  pv->bj_cap = (T == 4 && ts == 4 && ecg->ortho_alg == ORTHODIR && pa_env_flag("PREALPS_BJ_GRAM", 1)) ? pa_bj_gram_blocks() : 0;
  if (pv->bj_cap) parts += ((size_t)pv->bj_cap + pa_finish32_scratch_blocks()) * 32;
  pv->pool_doubles = (2 * nv + 3) * panel + small + parts;
  ecg->work = (double*)pa_rt_malloc(pv->pool_doubles * sizeof(double));
  if (!ecg->work) return PA_FAIL("device pool of %zu doubles: %s", pv->pool_doubles, pa_rt_error());
  double* w = ecg->work;
  pv->buf_v[0] = w; w += panel;
  pv->buf_v[1] = (nv == 2) ? w : NULL; if (nv == 2) w += panel;
  pv->buf_av[0] = w; w += panel;
  pv->buf_av[1] = (nv == 2) ? w : NULL; if (nv == 2) w += panel;
  pv->buf_z = w; w += panel;
  pv->d_R = w; w += panel;
  pv->d_X = w; w += panel;
  pv->d_F = w; pv->d_alpha = w; pv->d_beta = w + (size_t)T * T; pv->d_mu = w + 3 * (size_t)T * T;
  pv->d_rtr = w + 4 * (size_t)T * T; w += 5 * (size_t)T * T;
  pv->d_q = w; w += 2 * (size_t)T * T;
  pv->d_res2 = w; w += 8;
  pv->d_uu = w; w += 2 * (size_t)T * T;
  pv->d_partials = w; w += (size_t)pa_gram_max_blocks() * 2 * ts * ts;
  pv->d_rtr_part = w; w += (size_t)pa_gram_max_blocks() * ts;
  pv->d_spmm_parts = pv->spmm_cap ? w : NULL;
  if (pv->spmm_cap) w += ((size_t)pv->spmm_cap + pa_finish32_scratch_blocks()) * 32;
  pv->d_bj_parts = pv->bj_cap ? w : NULL;
  pv->d_info = (int*)pa_rt_malloc((8 + T) * sizeof(int));
  if (!pv->d_info) return PA_FAIL("device allocation failed: %s", pa_rt_error());
  pv->d_piv = pv->d_info + 8;
  pv->h_pin = (double*)pa_rt_host_alloc_coherent((16 + 4 * (size_t)T * T) * sizeof(double));
  pv->h_pin_i = (int*)pa_rt_host_alloc((8 + T) * sizeof(int));
  if (!pv->h_pin || !pv->h_pin_i) return PA_FAIL("pinned allocation failed: %s", pa_rt_error());
  pv->ev_res = pa_rt_event_create();
  pv->ev_alpha = pa_rt_event_create();
  if (!pv->ev_res || !pv->ev_alpha) return PA_FAIL("hipEventCreate failed");
  publish_pointers(ecg, pv);
  return 0;
}

/* The first half of the next iteration starts with [AP | R]^T P (fused_first_half): when the
 * product P -> AP is the library's SpMM on 4-column panels, ask it to leave that block behind.
 * Called whenever the P / AP pointers have been published; anything else ends the request. */
static void request_gram_from_spmm(preAlps_ECG_t* ecg, ecg_priv_t* pv) {
  if ((g_own_loop > 0 || rci_fuse()) && pv->spmm_cap > 0 && pv->fuse && ecg->bs_red == NO_BS_RED &&
      ecg->ortho_alg != ORTHODIR_FUSED && ecg->enlFac == 4 && ecg->P->info.n == 4)
    pa_k_spmm_gram_arm(ecg->P->val, ecg->AP->val, pv->d_R, pv->d_spmm_parts, pv->spmm_cap);
  else if (pv->d_spmm_parts)
    pa_k_spmm_gram_disarm(pv->d_spmm_parts);
}

/* ------------------------------------------------------------- reset ---- */
int _preAlps_ECGReset(preAlps_ECG_t* ecg, double* rhs, int* rci_request) {
  ecg_priv_t* pv = priv_of(ecg);
  if (!pv) return PA_FAIL("solver memory not allocated (_preAlps_ECGMalloc)");
  const pa_operator_info_t* op = pa_operator_info();
  if (!op) return PA_FAIL("the operator must be built before the solver");
  if (ecg->locPbSize != op->m) return PA_FAIL("locPbSize %d differs from the operator's %d rows", ecg->locPbSize, op->m);
  ecg->tot_t = ecg->comm_t = ecg->trsm_t = ecg->gemm_t = ecg->potrf_t = ecg->pstrf_t = 0.0;
  ecg->lapmt_t = ecg->gesvd_t = ecg->geqrf_t = ecg->ormqr_t = ecg->copy_t = 0.0;
  int M = ecg->globPbSize, m = ecg->locPbSize, t = ecg->enlFac, ts = pv->ts;
  /* pointer order back to the initial one */
  {
    size_t panel = (size_t)(m > 0 ? m : 1) * ts;
    double* w = ecg->work;
    int nv = (ecg->ortho_alg == ORTHOMIN) ? 1 : 2;
    pv->buf_v[0] = w; w += panel; if (nv == 2) { pv->buf_v[1] = w; w += panel; }
    pv->buf_av[0] = w; w += panel; if (nv == 2) { pv->buf_av[1] = w; w += panel; }
    pv->buf_z = w;
  }
  publish_pointers(ecg, pv);
  pv->rotate = (ecg->bs_red == NO_BS_RED);
  { const char* f = getenv("PREALPS_ECG_FUSE"); pv->fuse = f ? atoi(f) : 1; }
  pv->lazy_norm = pv->fuse && ecg->ortho_alg == ORTHODIR && pa_env_flag("PREALPS_ECG_LAZY_NORM", 1);
  pv->uu_cur = 0;
  pv->z_ready = 0;
  /* With more than one process every collective costs tens of microseconds.  The norm of the
   * new residual is only needed for the stopping decision, so the driver loops of this library
   * (preAlps_ECGSolve / ECGAdvance) let it travel with the beta all-reduce of the same
   * iteration and decide one half-step later; the last half-step is then simply unused.  The
   * RCI entry preAlps_ECGStoppingCriterion keeps working (it reduces the norm by itself).
   * One process: the same order saves the launch that sums the norm (it is summed with beta):
   * 2-3 us per iteration (in-process A/B), so it is the default there too; PREALPS_ECG_LAZY_STOP=0
   * decides right after the update, as the reference does. */
  { const char* f = getenv("PREALPS_ECG_LAZY_STOP");
    pv->lazy_stop = pv->fuse && ecg->bs_red == NO_BS_RED && ecg->ortho_alg != ORTHODIR_FUSED && ecg->enlFac >= 2 &&
                    (f ? atoi(f) : 1);
    pv->lazy_ptr = NULL; }
  /* graphs (opt-in: preAlps_hip_graphs(1) or PREALPS_ECG_GRAPH=1): one process, or the one-shard
   * rehearsal of preAlps_hip_loopback, whose sums are free, so the stopping test need not ride on one;
   * =2 forces them for any process group (the hooks must then be capturable: RCCL is, host-staged ones
   * are not).  Off by default because they measure SLOWER on this stack: 436.5 us per iteration against
   * 423.2 us with plain launches on the headline problem (same box, 200 iterations each; ROCm 7.2 puts
   * more idle time between the nodes of a graph than between launches queued on a stream) */
  {
    const char* ge = getenv("PREALPS_ECG_GRAPH");
    int want = g_graphs >= 0 ? g_graphs : (ge ? atoi(ge) : 0);
    int group_ok = pa_world_size() == 1 || pa_comm_is_loopback() || want == 2;
    pv->use_graphs = want && group_ok && pv->rotate && pv->fuse && ecg->ortho_alg != ORTHODIR_FUSED &&
                     !pa_timing_enabled();
    if (pv->use_graphs) pv->lazy_stop = 0;
    pv->phase = 0;
  }
  /* the residual norm reaches the host through two pinned words a kernel writes; a third word behind
   * them (a sequence number) lets the host poll for them instead of waiting for an event recorded in the
   * stream.  Default with several processes, where it measures 5 us per iteration faster (one-shard
   * rehearsal: 112.5 against 117.8 us); one process: no difference (the 6 us bubble the event leaves
   * behind k_trace_finish reappears behind the block solve), so the event stays.  PREALPS_ECG_POLL=0 / 1
   * forces; graphs replay fixed arguments and keep the event. */
  { const char* f = getenv("PREALPS_ECG_POLL"); pv->poll = (f ? atoi(f) : pa_world_size() > 1) && !pv->use_graphs; }
  pv->sent_seq = pv->wait_seq = 0.0;
  if (pv->h_pin) pv->h_pin[2] = 0.0;
  pa_set_desc(ecg->X, M, t, m, t, ts);
  pa_set_desc(ecg->R, M, t, m, t, ts);
  pa_set_desc(ecg->Z, M, t, m, t, ts);
  if (ecg->ortho_alg == ORTHOMIN) {
    pa_set_desc(ecg->V, M, t, m, t, ts);
    pa_set_desc(ecg->AV, M, t, m, t, ts);
    CPLM_MatDenseSetInfo(ecg->alpha, t, t, t, t, COL_MAJOR);
    CPLM_MatDenseSetInfo(ecg->beta, t, t, t, t, COL_MAJOR);
  } else {
    pa_set_desc(ecg->V, M, 2 * t, m, 2 * t, ts);
    pa_set_desc(ecg->AV, M, 2 * t, m, 2 * t, ts);
    CPLM_MatDenseSetInfo(ecg->alpha, t, t, t, t, COL_MAJOR);
    CPLM_MatDenseSetInfo(ecg->beta, 2 * t, t, 2 * t, t, COL_MAJOR);
  }
  pa_set_desc(ecg->P, M, t, m, t, ts);
  pa_set_desc(ecg->AP, M, t, m, t, ts);
  PA_CHECK(pa_rt_memset(ecg->work, 0, pv->pool_doubles * sizeof(double)));
  if (pv->lazy_norm) {        /* (the first update meets an empty P_prev: any factor will do, the identity is one) */
    double eye[2 * 16 * 16];
    memset(eye, 0, sizeof(eye));
    for (int k = 0; k < 2; ++k) for (int i = 0; i < t; ++i) eye[(size_t)k * t * t + i + (size_t)t * i] = 1.0;
    PA_CHECK(pa_rt_h2d(pv->d_uu, eye, 2 * (size_t)t * t * sizeof(double)));
  }
  /* normb and R0: column (rank % t) of every reference rank = part */
  double nb2 = 0.0;
  double* r0 = (double*)calloc((size_t)(m > 0 ? m : 1) * ts, sizeof(double));
  if (!r0) return PA_FAIL("out of host memory");
  for (int p = op->part0; p < op->part1; ++p) {
    int base = op->rowPos[p] - op->row_off, l = op->rowPos[p + 1] - op->rowPos[p];
    int col = p % t;
    double s = 0.0;
    for (int i = 0; i < l; ++i) { double v = rhs[base + i]; s += v * v; r0[(size_t)(base + i) * ts + col] = v; }
    nb2 += s;
  }
  int rc = pa_rt_h2d(pv->d_R, r0, (size_t)m * ts * sizeof(double));
  free(r0);
  if (rc) return PA_FAIL("%s", pa_rt_error());
  if (pa_world_size() > 1) {
    double t0;
    TIC(PA_T_COMM);
    PA_CHECK(pa_rt_h2d(pv->d_res2, &nb2, sizeof(double)));
    if (pa_allreduce(pv->d_res2, 1)) return 1;
    PA_CHECK(pa_rt_d2h(&nb2, pv->d_res2, sizeof(double)));
    TAC(PA_T_COMM, comm_t);
  }
  ecg->normb = sqrt(nb2);
  ecg->res = 1.0; ecg->iter = 0; ecg->bs = t; ecg->kbs = ecg->V->info.n;
  pv->rtr_valid = 0;
  *rci_request = 0;
  request_gram_from_spmm(ecg, pv);
  return 0;
}

int preAlps_ECGInitialize(preAlps_ECG_t* ecg, double* rhs, int* rci_request) {
  const pa_operator_info_t* op = pa_operator_info();
  if (!op) return PA_FAIL("the operator must be built before the solver");
  /* ecg.c:178-183 with the reference's "processors" = subdomains */
  if (op->nparts < ecg->enlFac)
    return PA_FAIL("Enlarging factor must be lower than the number of processors"
                   " in the MPI communicator! size: %d ; enlarging factor: %d", op->nparts, ecg->enlFac);
  int rc = _preAlps_ECGMalloc(ecg);
  if (rc) return rc;
  return _preAlps_ECGReset(ecg, rhs, rci_request);
}

int _preAlps_ECGSplit(double* x, CPLM_Mat_Dense_t* XSplit, int colIndex) {
  if (!XSplit || !XSplit->val || !x) return PA_FAIL(" wrong test 'XSplit->val != NULL && x != NULL'");
  int m = XSplit->info.m, ts = pa_desc_stride(XSplit);
  double* tmp = (double*)malloc((size_t)(m > 0 ? m : 1) * ts * sizeof(double));
  if (!tmp) return PA_FAIL("out of host memory");
  int rc = pa_rt_d2h(tmp, XSplit->val, (size_t)m * ts * sizeof(double));
  for (int i = 0; i < m; ++i) tmp[(size_t)i * ts + colIndex] = x[i];
  rc = rc || pa_rt_h2d(XSplit->val, tmp, (size_t)m * ts * sizeof(double));
  free(tmp);
  if (rc) return PA_FAIL("%s", pa_rt_error());
  return 0;
}

/* ---------------------------------------------------------- stopping ---- */
/* The stopping test in two halves: `begin` queues the reduction of the residual
 * norm and its copy to pinned host memory and marks that point with an event;
 * `end` waits for the event only -- not for the stream -- so work queued after
 * `begin` (the preconditioner apply of the same iteration) keeps the GPU busy
 * while the host reads the norm. */
static int stopping_end(preAlps_ECG_t* ecg, ecg_priv_t* pv, int* stop);
static int stopping_queue(preAlps_ECG_t* ecg, ecg_priv_t* pv) {
  int T = ecg->enlFac;
  int single = pa_world_size() == 1;
  if (pv->rtr_valid < 2) {
    if (!pv->rtr_valid) {
      PA_CHECK(pa_k_colnorm2(pv->m, pv->ts, pv->d_R, pv->d_rtr_part, &pv->rtr_nblk));
    }
    if (single && pv->rtr_valid == 1) {
      /* column sums left by the update kernel (lazy stopping test), one process: the sum goes straight to
       * the pinned words, as it does from the update kernel's own launch without the lazy test */
      pv->sent_seq = 0.0;
      if (pv->poll) { pv->sent_seq = (pv->seq += 1.0); pa_k_note_seq(pv->sent_seq); }
      PA_CHECK(pa_k_trace_finish(pv->d_rtr_part, pv->rtr_nblk, pv->ts, T, pv->d_res2, pv->d_info, pv->h_pin));
      pv->rtr_valid = 2;
    } else {
      PA_CHECK(pa_k_trace_finish(pv->d_rtr_part, pv->rtr_nblk, pv->ts, T, pv->d_res2, pv->d_info, NULL));
      pv->rtr_valid = 3;   /* summed, but not on the host yet */
    }
  }
  /* rtr_valid == 2: the update kernel summed the norm itself and, in a single-process run,
   * already wrote it to the pinned words the host reads */
  pv->wait_seq = 0.0;
  if (!single || pv->rtr_valid == 3) {
    double* src = (pv->rtr_valid == 2 && pv->lazy_ptr) ? pv->lazy_ptr : pv->d_res2;
    double t0;
    TIC(PA_T_COMM);
    if (pa_allreduce(src, 1)) return 1;
    TAC(PA_T_COMM, comm_t);
    PA_CHECK(pa_rt_d2h_async(pv->h_pin, src, 2 * sizeof(double)));
  } else {
    pv->wait_seq = pv->sent_seq;     /* the update kernel's launch writes the words and this number behind them */
  }
  pv->sent_seq = 0.0;
  pv->rtr_valid = 0;
  return 0;
}
static int stopping_begin(preAlps_ECG_t* ecg, ecg_priv_t* pv) {
  if (stopping_queue(ecg, pv)) return 1;
  if (pv->wait_seq == 0.0) PA_CHECK(pa_rt_event_record(pv->ev_res));
  return 0;
}
/* Wait for the sequence number behind the two pinned words (see _preAlps_ECGReset). */
static int wait_for_note(ecg_priv_t* pv) {
  volatile double* h = pv->h_pin;
  double t0 = pa_wtime();
  unsigned long spins = 0;
  int idle_seen = 0;
  while (h[2] != pv->wait_seq) {
    __builtin_ia32_pause();
    if ((++spins & 0x3fff) == 0) {
      int st = pa_rt_stream_state();
      if (st < 0) return PA_FAIL("%s", pa_rt_error());
      if (st == 1 && idle_seen++ > 2) return PA_FAIL("the residual norm never reached the host (stream idle)");
      if (pa_wtime() - t0 > 60.0) return PA_FAIL("timed out waiting for the residual norm");
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  pv->wait_seq = 0.0;
  return 0;
}

/* ---- graph segments ------------------------------------------------------------------------
 * seg_begin .. seg_end bracket the host code of one segment.  First pass of a (segment, phase): the
 * code runs as it is.  Second pass: the stream is captured while it runs (nothing executes), the
 * graph is instantiated and launched.  From then on the host code still runs -- it rotates
 * pointers and counts -- with every launch suppressed (pa_rt_skip), and the graph is launched in
 * its place: one launch instead of three to six per segment. */
static int seg_begin(ecg_priv_t* pv, int seg) {
  if (!pv->use_graphs || pa_timing_enabled()) return 0;    /* (phase timers put events between the launches) */
  if (pv->graph[seg][pv->phase]) { pa_rt_skip(1); return 2; }
  if (pv->seen[seg][pv->phase]++ == 0) return 0;
  if (pa_rt_capture_begin()) { pv->use_graphs = 0; return 0; }   /* (nothing queued yet: go on without graphs) */
  return 1;
}
static int seg_end(ecg_priv_t* pv, int seg, int mode, int body_rc) {
  if (mode == 2) pa_rt_skip(0);
  if (mode == 1) {
    void* exec = NULL;
    int rc = pa_rt_capture_end(&exec);
    if (rc || body_rc) { pa_rt_graph_free(exec); return PA_FAIL("capturing an iteration segment failed: %s", pa_rt_error()); }
    pv->graph[seg][pv->phase] = exec;
  }
  if (body_rc) return 1;
  if (mode && pa_rt_graph_launch(pv->graph[seg][pv->phase])) return PA_FAIL("%s", pa_rt_error());
  return 0;
}
/* One full iteration from the state "AP = A P is there" (rci 0) to the same state: the order of
 * examples/test_ecg_prealps_op.c:208-221 with the apply, the second half and the product queued
 * before the host looks at the residual norm (they do not touch X or R and are unused after a stop). */
static int graph_iteration(preAlps_ECG_t* ecg, ecg_priv_t* pv, int* rci_request, int* stop) {
  int mode = seg_begin(pv, 0);
  int rc = preAlps_ECGIterate(ecg, rci_request) || stopping_queue(ecg, pv);
  if (seg_end(pv, 0, mode, rc)) return 1;
  PA_CHECK(pa_rt_event_record(pv->ev_res));
  mode = seg_begin(pv, 1);
  rc = (ecg->ortho_alg == ORTHOMIN ? preAlps_BlockJacobiApply(ecg->R, ecg->Z) : preAlps_BlockJacobiApply(ecg->AP, ecg->Z)) ||
       preAlps_ECGIterate(ecg, rci_request) || preAlps_BlockOperator(ecg->P, ecg->AP);
  if (seg_end(pv, 1, mode, rc)) return 1;
  pv->phase = (pv->phase + 1) % 6;
  return stopping_end(ecg, pv, stop);
}

static int stopping_end(preAlps_ECG_t* ecg, ecg_priv_t* pv, int* stop) {
  if (pv->wait_seq != 0.0) { if (wait_for_note(pv)) return 1; }
  else PA_CHECK(pa_rt_event_wait(pv->ev_res));
  double res2 = pv->h_pin[0];
  int info = (int)pv->h_pin[1];
  if (info != 0 && ecg->ortho_alg == ORTHOMIN) return PA_FAIL("ACHQR: dpotrf:\n ERROR: P^tAP is not spd!");
  ecg->res = sqrt(res2);
  /* !(a > b) also stops on NaN, like the reference's comparison */
  if (ecg->res > ecg->normb * ecg->tol && ecg->iter < ecg->maxIter && ecg->bs > 0) *stop = 0;
  else *stop = 1;
  return 0;
}

int preAlps_ECGStoppingCriterion(preAlps_ECG_t* ecg, int* stop) {
  ecg_priv_t* pv = priv_of(ecg);
  if (!pv) return PA_FAIL("solver not initialised");
  if (!stop) return PA_FAIL(" wrong test 'stop != NULL'");
  double tg = pa_wtime();
  if (stopping_begin(ecg, pv)) return 1;
  if (stopping_end(ecg, pv, stop)) return 1;
  ecg->tot_t += pa_wtime() - tg;
  return 0;
}

/* ------------------------------------------------------ shared pieces ---- */
/* W = AP^T P -> all-reduce -> U^T U ; P <- P U^-1 ; AP <- AP U^-1 ;
 * alpha = P^T R -> all-reduce   (ecg.c:311-333, :425-443) */
static int a_orthonormalise_and_alpha(preAlps_ECG_t* ecg, ecg_priv_t* pv, int t) {
  int m = pv->m, ts = pv->ts;
  double t0;
  TIC(PA_T_GRAM);
  PA_CHECK(pa_k_gram_finish(m, ts, ecg->AP->val, NULL, ecg->P->val, pv->d_partials, t, 0, t, pv->d_mu, t,
                            0, 0, NULL, NULL, NULL));
  TAC(PA_T_GRAM, gemm_t);
  TIC(PA_T_COMM);
  if (pa_allreduce(pv->d_mu, t * t)) return 1;
  TAC(PA_T_COMM, comm_t);
  TIC(PA_T_SMALL);
  PA_CHECK(pa_k_potrf(pv->d_mu, t, pv->d_info));
  TAC(PA_T_SMALL, potrf_t);
  TIC(PA_T_TRSM);
  PA_CHECK(pa_k_trsm(m, ts, t, pv->d_mu, ecg->P->val, ecg->AP->val));
  TAC(PA_T_TRSM, trsm_t);
  TIC(PA_T_GRAM);
  PA_CHECK(pa_k_gram_finish(m, ts, ecg->P->val, NULL, ecg->R->val, pv->d_partials, ecg->alpha->info.m, 0,
                            ecg->alpha->info.n, pv->d_alpha, ecg->alpha->info.lda, 0, 0, NULL, NULL, NULL));
  TAC(PA_T_GRAM, gemm_t);
  TIC(PA_T_COMM);
  if (pa_allreduce(pv->d_alpha, ecg->alpha->info.lda * ecg->alpha->info.n)) return 1;
  TAC(PA_T_COMM, comm_t);
  return 0;
}

/* The rci == 0 half without block-size reduction, in two passes over the panels
 * instead of four: [W ; G^T] = [AP | R]^T P (one Gram kernel, one all-reduce),
 * then U = chol(W), alpha = U^-T G on one wave, then P U^-1, AP U^-1, X += P alpha,
 * R -= AP alpha and the residual column norms in a single kernel.  Algebraically
 * ecg.c:425-443 + :500-501; alpha is formed from the Gram of the un-normalised P. */
static int fused_update(preAlps_ECG_t* ecg, ecg_priv_t* pv, int t, const double* gram);
static int fused_gram(preAlps_ECG_t* ecg, ecg_priv_t* pv, int t, int factor_now);
static int fused_first_half(preAlps_ECG_t* ecg, ecg_priv_t* pv, int t) {
  if (fused_gram(ecg, pv, t, 0)) return 1;
  return fused_update(ecg, pv, t, pa_world_size() == 1 ? NULL : pv->d_q);
}
/* first pass: [W ; G^T] summed over the processes in d_q; one process (or factor_now): U = chol(W) in d_mu and
 * alpha = U^-T G in d_alpha as well, so that the host can look at alpha before the panels are touched (D-Odir) */
static int fused_gram(preAlps_ECG_t* ecg, ecg_priv_t* pv, int t, int factor_now) {
  int m = pv->m, ts = pv->ts, T = ecg->enlFac;
  int single = pa_world_size() == 1;
  double* buf = pv->d_q; /* (t+T) x t */
  double t0;
  /* blocks the SpMM left behind while it formed AP (request_gram_from_spmm): only their sum is left to do */
  int from_spmm = pv->spmm_cap > 0 ? pa_k_spmm_gram_take(ecg->P->val, ecg->AP->val) : 0;
  TIC(PA_T_GRAM);
  if (single) {
    /* nothing to reduce across processes: the launch that sums the partial blocks factors them
     * right away */
    if (from_spmm) PA_CHECK(pa_k_finish32(pv->d_spmm_parts, from_spmm, pv->d_spmm_parts + (size_t)pv->spmm_cap * 32, t, T, buf,
                                         pv->d_mu, pv->d_alpha, pv->d_info));
    else PA_CHECK(pa_k_gram_finish(m, ts, ecg->AP->val, pv->d_R, ecg->P->val, pv->d_partials, t, T, t, buf, t + T,
                                   t, T, pv->d_mu, pv->d_alpha, pv->d_info));
    TAC(PA_T_GRAM, gemm_t);
  } else {
    if (from_spmm) PA_CHECK(pa_k_finish32(pv->d_spmm_parts, from_spmm, pv->d_spmm_parts + (size_t)pv->spmm_cap * 32, 0, 0, buf,
                                         NULL, NULL, NULL));
    else PA_CHECK(pa_k_gram_finish(m, ts, ecg->AP->val, pv->d_R, ecg->P->val, pv->d_partials, t, T, t, buf, t + T,
                                   0, 0, NULL, NULL, NULL));
    TAC(PA_T_GRAM, gemm_t);
    TIC(PA_T_COMM);
    if (pa_allreduce(buf, (t + T) * t)) return 1;
    TAC(PA_T_COMM, comm_t);
    /* (the factorisation of the reduced block and alpha: in the update kernel's prologue) */
    if (factor_now) {
      TIC(PA_T_SMALL);
      PA_CHECK(pa_k_potrf_alpha(buf, t, T, pv->d_mu, pv->d_alpha, pv->d_info));
      TAC(PA_T_SMALL, potrf_t);
    }
  }
  return 0;
}
/* second pass: P U^-1, AP U^-1, X += P alpha, R -= AP alpha and the residual column norms; gram != NULL: the
 * kernel's prologue factors the summed block first */
static int fused_update(preAlps_ECG_t* ecg, ecg_priv_t* pv, int t, const double* gram) {
  int nb = 0, m = pv->m, ts = pv->ts, T = ecg->enlFac;
  int single = pa_world_size() == 1;
  double t0;
  TIC(PA_T_UPDATE);
  /* the slot right behind beta: free once the kernel has read U from it (Odir: d_mu) */
  pv->lazy_ptr = pv->lazy_stop ? pv->d_beta + (size_t)ecg->beta->info.lda * ecg->beta->info.n : NULL;
  /* with the lazy stopping test the column sums of R^2 are added up by the launch that sums beta
   * (orthogonalise_z), next to which the norm travels: no launch of its own here */
  int defer = pv->lazy_ptr != NULL;
  pv->sent_seq = 0.0;
  if (single && pv->poll && !defer) { pv->sent_seq = (pv->seq += 1.0); pa_k_note_seq(pv->sent_seq); }
  PA_CHECK(pa_k_trsm_update(m, ts, t, ecg->X->info.n, pv->d_mu, pv->d_alpha, ecg->P->val, ecg->AP->val,
                            pv->d_X, pv->d_R, pv->d_rtr_part, &nb, defer ? 0 : T,
                            pv->lazy_ptr ? pv->lazy_ptr : pv->d_res2, pv->d_info, single ? pv->h_pin : NULL,
                            gram, pv->lazy_norm ? pv->d_uu + (size_t)pv->uu_cur * T * T : NULL));
  pv->rtr_nblk = nb;
  TAC(PA_T_UPDATE, trsm_t);
  pv->rtr_valid = defer ? 1 : 2;
  return 0;
}

/* X += P alpha ; R -= AP alpha (ecg.c:337-338, :500-501) + residual norms */
static int update_iterate(preAlps_ECG_t* ecg, ecg_priv_t* pv) {
  double t0;
  TIC(PA_T_UPDATE);
  pv->sent_seq = 0.0;
  if (pa_world_size() == 1 && pv->poll) { pv->sent_seq = (pv->seq += 1.0); pa_k_note_seq(pv->sent_seq); }
  PA_CHECK(pa_k_update_xr(pv->m, pv->ts, ecg->P->info.n, ecg->X->info.n, pv->d_alpha, ecg->P->val,
                          ecg->AP->val, pv->d_X, pv->d_R, pv->d_rtr_part, &pv->rtr_nblk, ecg->enlFac,
                          pv->d_res2, pv->d_info, pa_world_size() == 1 ? pv->h_pin : NULL));
  TAC(PA_T_UPDATE, gemm_t);
  pv->rtr_valid = 2;
  pv->lazy_ptr = NULL;
  return 0;
}

/* The reference's "Swapping time" (ecg.c:521-523 / :358): without block-size
 * reduction all columns are live, so the copies become pointer rotations. */
static int shift_directions(preAlps_ECG_t* ecg, ecg_priv_t* pv, int ncopy) {
  double t0 = pa_wtime();
  if (ecg->ortho_alg == ORTHOMIN) {
    if (pv->rotate) { double* p = pv->buf_v[0]; pv->buf_v[0] = pv->buf_z; pv->buf_z = p; }
    else PA_CHECK(pa_k_copy_cols(pv->m, pv->ts, ncopy, pv->buf_z, pv->buf_v[0]));
  } else if (pv->rotate || ncopy == ecg->enlFac) {
    /* (with block-size reduction too while every column is still live: the columns a reduction rotates away
     * stay behind in P / AP, and from then on ncopy < enlFac and the copies below keep them where they are) */
    double* oldprev = pv->buf_v[1];
    pv->buf_v[1] = pv->buf_v[0]; pv->buf_v[0] = pv->buf_z; pv->buf_z = oldprev;
    double* oldaprev = pv->buf_av[1];
    pv->buf_av[1] = pv->buf_av[0]; pv->buf_av[0] = oldaprev;
  } else {
    PA_CHECK(pa_k_copy_cols(pv->m, pv->ts, ncopy, pv->buf_v[0], pv->buf_v[1]));
    PA_CHECK(pa_k_copy_cols(pv->m, pv->ts, ncopy, pv->buf_av[0], pv->buf_av[1]));
    PA_CHECK(pa_k_copy_cols(pv->m, pv->ts, ncopy, pv->buf_z, pv->buf_v[0]));
  }
  publish_pointers(ecg, pv);
  ecg->copy_t += pa_wtime() - t0;
  return 0;
}

/* beta = AV^T Z over kbs columns of [slot 0 | slot 1] -> all-reduce ;
 * Z -= V beta (ecg.c:510-517) */
static int orthogonalise_z(preAlps_ECG_t* ecg, ecg_priv_t* pv) {
  int T = ecg->enlFac;
  int kb = ecg->beta->info.m; /* rows of beta = columns of V in use */
  int a_lo = kb < T ? kb : T, a_hi = kb - a_lo;
  double t0;
  int cnt = ecg->beta->info.lda * ecg->beta->info.n;
  const double* note = NULL;     /* device words the update kernel copies to pinned memory */
  TIC(PA_T_GRAM);
  int from_bj = pv->bj_cap > 0 ? pa_k_bj_gram_take(pv->buf_av[0], pv->buf_z) : 0;
  if (from_bj > 0 && !(a_lo == T && a_hi == T && T == 4 && ecg->beta->info.n == 4 && ecg->beta->info.lda == 8)) from_bj = 0;
  if (from_bj > 0) {
    /* the block solve left one 8 x 4 block per subdomain behind: only their sum is left to do (and, with the
     * lazy stopping test, the residual norm next to it) */
    double* scratch = pv->d_bj_parts + (size_t)pv->bj_cap * 32;
    if (pv->rtr_valid == 1 && pv->lazy_ptr && pv->lazy_ptr == pv->d_beta + cnt) {
      PA_CHECK(pa_k_finish32_trace(pv->d_bj_parts, from_bj, scratch, pv->d_beta, pv->d_rtr_part, pv->rtr_nblk, pv->ts, T,
                                   pv->lazy_ptr, pv->d_info));
      pv->rtr_valid = 2;
    } else {
      if (pv->rtr_valid == 1 && pv->lazy_ptr) {
        PA_CHECK(pa_k_trace_finish(pv->d_rtr_part, pv->rtr_nblk, pv->ts, T, pv->lazy_ptr, pv->d_info, NULL));
        pv->rtr_valid = 2;
      }
      PA_CHECK(pa_k_finish32(pv->d_bj_parts, from_bj, scratch, 0, 0, pv->d_beta, NULL, NULL, NULL));
    }
  } else if (pv->rtr_valid == 1 && pv->lazy_ptr && pv->lazy_ptr == pv->d_beta + cnt) {
    /* the column sums fused_first_half left: summed by the same launch, into the slot behind beta */
    PA_CHECK(pa_k_gram_finish_trace(pv->m, pv->ts, pv->buf_av[0], a_hi > 0 ? pv->buf_av[1] : NULL, pv->buf_z,
                                    pv->d_partials, a_lo, a_hi, ecg->beta->info.n, pv->d_beta, ecg->beta->info.lda,
                                    pv->d_rtr_part, pv->rtr_nblk, T, pv->lazy_ptr, pv->d_info));
    pv->rtr_valid = 2;
  } else {
    if (pv->rtr_valid == 1 && pv->lazy_ptr) {     /* (beta not where expected: the norm by itself) */
      PA_CHECK(pa_k_trace_finish(pv->d_rtr_part, pv->rtr_nblk, pv->ts, T, pv->lazy_ptr, pv->d_info, NULL));
      pv->rtr_valid = 2;
    }
    PA_CHECK(pa_k_gram_finish(pv->m, pv->ts, pv->buf_av[0], a_hi > 0 ? pv->buf_av[1] : NULL, pv->buf_z,
                              pv->d_partials, a_lo, a_hi, ecg->beta->info.n, pv->d_beta, ecg->beta->info.lda,
                              0, 0, NULL, NULL, NULL));
  }
  TAC(PA_T_GRAM, gemm_t);
  TIC(PA_T_COMM);
  {
    if (pv->rtr_valid == 2 && pv->lazy_ptr == pv->d_beta + cnt) {
      /* the norm of the new residual (and the Cholesky status) ride along; the update kernel
       * below passes them on to the pinned words the host reads (a copy here costs a launch and
       * 8 us of idle stream behind it) */
      if (pa_allreduce(pv->d_beta, cnt + 2)) return 1;
      note = pv->lazy_ptr;
      pv->rtr_valid = 0;
    } else {
      if (pa_allreduce(pv->d_beta, cnt)) return 1;
      if (pv->lazy_ptr && pv->rtr_valid == 2) {   /* (beta not where expected: reduce the norm by itself) */
        if (pa_allreduce(pv->lazy_ptr, 1)) return 1;
        PA_CHECK(pa_rt_d2h_async(pv->h_pin, pv->lazy_ptr, 2 * sizeof(double)));
        PA_CHECK(pa_rt_event_record(pv->ev_res));
        pv->wait_seq = 0.0;
        pv->rtr_valid = 0;
      }
    }
  }
  TAC(PA_T_COMM, comm_t);
  TIC(PA_T_UPDATE);
  int vn = ecg->V->info.n;
  int v_lo = vn < T ? vn : T, v_hi = vn - v_lo;
  int polled = note && pv->poll && ecg->Z->info.n > 0;
  if (polled) { pv->wait_seq = (pv->seq += 1.0); pa_k_note_seq(pv->wait_seq); }
  /* several processes: the Z written here is the X of the product that follows (shift_directions rotates it into P);
   * the rows the neighbours need are packed from the kernel's registers and the operator skips its pack launch
   * (when P turns out to be another buffer -- copies instead of rotation, BF-Omin's permuted panel -- it packs) */
  if ((g_own_loop > 0 || rci_fuse()) && pv->ts <= 4 && ecg->Z->info.n > 0 && !pv->use_graphs && pa_world_size() > 1) {
    const int* pk_off = NULL; const int* pk_slot = NULL; double* sendbuf = NULL;
    if (pa_operator_pack_hint(pv->ts, pv->buf_z, &pk_off, &pk_slot, &sendbuf) &&
        !pa_k_update_z_pack(pv->ts, pk_off, pk_slot, sendbuf))
      pa_operator_pack_hint(pv->ts, NULL, &pk_off, &pk_slot, &sendbuf);      /* (not taken: withdraw) */
  }
  PA_CHECK(pa_k_update_z(pv->m, pv->ts, v_lo, v_hi, ecg->Z->info.n, pv->d_beta, ecg->beta->info.lda,
                         pv->buf_v[0], pv->buf_v[1], pv->buf_z, note, note ? pv->h_pin : NULL,
                         pv->lazy_norm ? pv->d_uu + (size_t)pv->uu_cur * T * T : NULL,
                         pv->lazy_norm ? pv->d_uu + (size_t)(1 - pv->uu_cur) * T * T : NULL,
                         /* BF-Omin forms Z^T Z next (8 / 16 columns: the update kernel leaves its partial blocks) */
                         (ecg->ortho_alg == ORTHOMIN && ecg->bs_red == ADAPT_BS && pv->fuse) ? pv->d_partials : NULL,
                         T, &pv->zz_nblk));
  if (pv->lazy_norm) pv->uu_cur ^= 1;
  if (note && ecg->Z->info.n <= 0) PA_CHECK(pa_rt_d2h_async(pv->h_pin, note, 2 * sizeof(double)));   /* (no launch above) */
  if (note && !polled) { pv->wait_seq = 0.0; PA_CHECK(pa_rt_event_record(pv->ev_res)); }
  TAC(PA_T_UPDATE, gemm_t);
  return 0;
}

/* ---- D-Odir: reduction of the search directions (ecg.c:445-497, :593-637) -- */
static int fetch_alpha(preAlps_ECG_t* ecg, ecg_priv_t* pv) {
  PA_CHECK(pa_rt_d2h_async(pv->h_pin + 16, pv->d_alpha, (size_t)ecg->P->info.n * ecg->enlFac * sizeof(double)));
  PA_CHECK(pa_rt_event_record(pv->ev_alpha));
  return 0;
}
static int reduce_directions_odir(preAlps_ECG_t* ecg, ecg_priv_t* pv, int with_Z, int* pending_trsm, int fetched) {
  int M = ecg->globPbSize, m = pv->m, ts = pv->ts, nrhs = ecg->enlFac;
  int t = ecg->P->info.n, t1 = 0;
  double tol = ecg->tol * ecg->normb / sqrt((double)nrhs), t0;
  double* ha = pv->h_pin + 16;                 /* t x nrhs, ld t */
  double* hq = ha + (size_t)nrhs * nrhs;       /* t x t */
  double sig[16];
  t0 = pa_wtime();
  if (fetched) PA_CHECK(pa_rt_event_wait(pv->ev_alpha));     /* (fetch_alpha queued the copy; more work may be queued behind it) */
  else {
    PA_CHECK(pa_rt_d2h_async(ha, pv->d_alpha, (size_t)t * nrhs * sizeof(double)));
    PA_CHECK(pa_rt_sync());
  }
  ecg->copy_t += pa_wtime() - t0;
  t0 = pa_wtime();
  pa_sd_left_singular(t, nrhs, ha, t, hq, sig); /* hq = U, sig decreasing */
  ecg->gesvd_t += pa_wtime() - t0;
  for (int i = 0; i < t; ++i) { if (sig[i] > tol) t1++; else break; }
  if (t1 > 0 && t1 < nrhs && t1 < t) {
    t0 = pa_wtime();
    pa_sd_qr_q(t, hq);                         /* hq <- Q of the Householder QR of U */
    ecg->geqrf_t += pa_wtime() - t0;
    TIC(PA_T_UPDATE);
    pa_sd_qt_times(t, nrhs, hq, ha);           /* alpha <- Q^T alpha */
    /* keep the first t1 rows, leading dimension t1 (mkl_dimatcopy, ecg.c:483) */
    double* packed = hq + (size_t)nrhs * nrhs;
    for (int j = 0; j < nrhs; ++j) for (int i = 0; i < t1; ++i) packed[i + (size_t)t1 * j] = ha[i + (size_t)t * j];
    PA_CHECK(pa_rt_h2d(pv->d_alpha, packed, (size_t)t1 * nrhs * sizeof(double)));
    PA_CHECK(pa_rt_h2d(pv->d_q, hq, (size_t)t * t * sizeof(double)));
    if (pending_trsm && *pending_trsm) {       /* the caller held P U^-1, AP U^-1 back for its fused update */
      PA_CHECK(pa_k_trsm(m, ts, t, pv->d_mu, ecg->P->val, ecg->AP->val));
      if (pv->lazy_norm) {
        /* the panels were never normalised so far: the previous directions now, with their own factor, and the
         * block solve's result if it is already there (Z = M^-1 AP_raw: Z U^-1 is what the reference holds) */
        PA_CHECK(pa_k_trsm(m, ts, nrhs, pv->d_uu + (size_t)(1 - pv->uu_cur) * nrhs * nrhs, pv->buf_v[1], pv->buf_av[1]));
        if (pv->z_ready) PA_CHECK(pa_k_trsm(m, ts, t, pv->d_mu, ecg->Z->val, NULL));
        pv->lazy_norm = 0;
      }
      *pending_trsm = 0;
    }
    PA_CHECK(pa_k_right_mult(m, ts, t, pv->d_q, ecg->P->val));
    PA_CHECK(pa_k_right_mult(m, ts, t, pv->d_q, ecg->AP->val));
    if (with_Z || pv->z_ready) PA_CHECK(pa_k_right_mult(m, ts, t, pv->d_q, ecg->Z->val));
    TAC(PA_T_UPDATE, ormqr_t);
    CPLM_MatDenseSetInfo(ecg->alpha, t1, nrhs, t1, nrhs, COL_MAJOR);
    pa_set_desc(ecg->P, M, t1, m, t1, ts);
    pa_set_desc(ecg->AP, M, t1, m, t1, ts);
    pa_set_desc(ecg->Z, M, t1, m, t1, ts);
    ecg->bs = t1;
    ecg->kbs = t + nrhs;
  }
  CPLM_MatDenseSetInfo(ecg->beta, ecg->kbs, t1, ecg->kbs, t1, COL_MAJOR);
  pa_set_desc(ecg->V, M, ecg->kbs, m, ecg->kbs, ts);
  pa_set_desc(ecg->AV, M, ecg->kbs, m, ecg->kbs, ts);
  return 0;
}

/* ------------------------------------------------------------ Orthodir ---- */
int _preAlps_ECGIterateOdir(preAlps_ECG_t* ecg, int* rci_request) {
  ecg_priv_t* pv = priv_of(ecg);
  if (!pv) return PA_FAIL("solver not initialised");
  int t = ecg->P->info.n;
  if (*rci_request == 0) {
    if (ecg->bs_red == NO_BS_RED && pv->fuse) {
      if (fused_first_half(ecg, pv, t)) return 1;
      /* next: Z = M^-1 AP by the caller, then beta = [AP | AP_prev]^T Z here: ask the block solve for it */
      if ((g_own_loop > 0 || rci_fuse()) && pv->bj_cap > 0 && t == 4 && ecg->beta->info.m == 8 &&
          ecg->beta->info.lda == 8 && ecg->beta->info.n == 4)
        pa_k_bj_gram_arm(pv->buf_av[0], pv->buf_z, pv->buf_av[1], pv->d_bj_parts, pv->bj_cap);
      else if (pv->d_bj_parts) pa_k_bj_gram_disarm(pv->d_bj_parts);
    } else if (pv->fuse && t == ecg->enlFac) {
      /* D-Odir while every direction is still live: the same two passes, with the look at alpha in between
       * (alpha = U^-T G needs no normalised panel); a reduction decided there goes the four-pass way */
      int pending = 1;
      if (fused_gram(ecg, pv, t, 1)) return 1;
      if (fetch_alpha(ecg, pv)) return 1;
      if (g_own_loop > 0 && pv->lazy_norm) {
        if (preAlps_BlockJacobiApply(ecg->AP, ecg->Z)) return 1;
        pv->z_ready = 1;
      }
      if (reduce_directions_odir(ecg, pv, 0, &pending, 1)) return 1;
      if (pending ? fused_update(ecg, pv, t, NULL) : update_iterate(ecg, pv)) return 1;
    } else {
      if (a_orthonormalise_and_alpha(ecg, pv, t)) return 1;
      if (ecg->bs_red == ADAPT_BS && reduce_directions_odir(ecg, pv, 0, NULL, 0)) return 1;
      if (update_iterate(ecg, pv)) return 1;
    }
    ecg->iter++;
    *rci_request = 1;
  } else if (*rci_request == 1) {
    if (orthogonalise_z(ecg, pv)) return 1;
    if (shift_directions(ecg, pv, t)) return 1;
    request_gram_from_spmm(ecg, pv);
    *rci_request = 0;
  }
  return 0;
}

/* ------------------------------------------------------------ Orthomin ---- */
int _preAlps_ECGIterateOmin(preAlps_ECG_t* ecg, int* rci_request) {
  ecg_priv_t* pv = priv_of(ecg);
  if (!pv) return PA_FAIL("solver not initialised");
  int M = ecg->globPbSize, m = pv->m, ts = pv->ts, nrhs = ecg->enlFac;
  int t = ecg->P->info.n;
  if (*rci_request == 0) {
    if (pv->fuse && (ecg->bs_red == NO_BS_RED || t == nrhs)) {
      if (fused_first_half(ecg, pv, t)) return 1;   /* (BF-Omin: while the panel still has its full rank) */
    } else {
      if (a_orthonormalise_and_alpha(ecg, pv, t)) return 1;
      if (update_iterate(ecg, pv)) return 1;
    }
    ecg->iter++;
    *rci_request = 1;
  } else if (*rci_request == 1) {
    if (orthogonalise_z(ecg, pv)) return 1;
    /* BF-Omin, two passes instead of four: the copy Z -> P (ecg.c:358) waits for the pivots and is done by the
     * kernel that permutes and solves; the Gram block is formed on Z, the very numbers P would hold */
    const int one_pass = ecg->bs_red == ADAPT_BS && pv->fuse;
    if (!one_pass && shift_directions(ecg, pv, nrhs)) return 1;
    if (ecg->bs_red == ADAPT_BS) {
      /* BF-Omin: G = P^T P -> pivoted Cholesky -> permute, P <- P U^-1 (ecg.c:361-393) */
      int nb = 0, rank = 0;
      double t0;
      const double* newp = one_pass ? pv->buf_z : ecg->P->val;
      TIC(PA_T_GRAM);
      if (one_pass && pv->zz_nblk > 0) nb = pv->zz_nblk;        /* (the update kernel formed the blocks of Z^T Z) */
      else PA_CHECK(pa_k_gram(m, ts, newp, NULL, newp, pv->d_partials, &nb));
      PA_CHECK(pa_k_finish(pv->d_partials, nb, 1, ts, nrhs, 0, nrhs, pv->d_mu, nrhs));
      TAC(PA_T_GRAM, gemm_t);
      TIC(PA_T_COMM);
      if (pa_allreduce(pv->d_mu, nrhs * nrhs)) return 1;
      TAC(PA_T_COMM, comm_t);
      double* hg = pv->h_pin + 16;
      PA_CHECK(pa_rt_d2h_async(hg, pv->d_mu, (size_t)nrhs * nrhs * sizeof(double)));
      PA_CHECK(pa_rt_sync());
      t0 = pa_wtime();
      pa_sd_pstrf_upper(nrhs, hg, nrhs, ecg->iwork, &rank, -1.0);
      ecg->pstrf_t += pa_wtime() - t0;
      TIC(PA_T_UPDATE);
      for (int j = 0; j < nrhs; ++j) pv->h_pin_i[j] = ecg->iwork[j] - 1;
      /* (both small copies from pinned memory without a host wait: the words are next written behind the next
       * iteration's wait for its Gram block) */
      PA_CHECK(pa_rt_h2d_async(pv->d_piv, pv->h_pin_i, nrhs * sizeof(int)));
      if (!one_pass) PA_CHECK(pa_k_permute_cols(m, ts, nrhs, pv->d_piv, ecg->P->val));
      TAC(PA_T_UPDATE, lapmt_t);
      /* leading rank x rank block, leading dimension rank for the kernel */
      double* hu = hg + (size_t)nrhs * nrhs;
      for (int j = 0; j < rank; ++j) for (int i = 0; i < rank; ++i) hu[i + (size_t)rank * j] = hg[i + (size_t)nrhs * j];
      PA_CHECK(pa_rt_h2d_async(pv->d_q, hu, (size_t)rank * rank * sizeof(double)));
      TIC(PA_T_TRSM);
      if (one_pass) {
        double tc = pa_wtime();
        PA_CHECK(pa_k_permute_trsm(m, ts, nrhs, pv->d_piv, rank, pv->d_q, pv->buf_z, pv->buf_v[0]));
        publish_pointers(ecg, pv);
        ecg->copy_t += pa_wtime() - tc;
      } else {
        PA_CHECK(pa_k_trsm(m, ts, rank, pv->d_q, ecg->P->val, NULL));
      }
      TAC(PA_T_TRSM, trsm_t);
      t = rank;
      pa_set_desc(ecg->P, M, t, m, t, ts);
      pa_set_desc(ecg->AP, M, t, m, t, ts);
      CPLM_MatDenseSetInfo(ecg->alpha, t, nrhs, t, nrhs, COL_MAJOR);
      CPLM_MatDenseSetInfo(ecg->beta, t, nrhs, t, nrhs, COL_MAJOR);
      ecg->bs = t;
    }
    request_gram_from_spmm(ecg, pv);
    *rci_request = 0;
  }
  return 0;
}

/* ------------------------------------------------------ fused Orthodir ---- */
int _preAlps_ECGIterateOdirFused(preAlps_ECG_t* ecg, int* rci_request) {
  ecg_priv_t* pv = priv_of(ecg);
  if (!pv) return PA_FAIL("solver not initialised");
  int m = pv->m, ts = pv->ts, nrhs = ecg->enlFac;
  int t = ecg->P->info.n;
  double t0;
  /* the four local Gram blocks, stacked [alpha | beta | mu | RtR] (ecg.c:554-560) */
  TIC(PA_T_GRAM);
  PA_CHECK(pa_k_gram_finish(m, ts, ecg->P->val, NULL, ecg->R->val, pv->d_partials, ecg->alpha->info.m, 0,
                            ecg->alpha->info.n, pv->d_alpha, ecg->alpha->info.lda, 0, 0, NULL, NULL, NULL));
  {
    int kb = ecg->beta->info.m, a_lo = kb < nrhs ? kb : nrhs, a_hi = kb - a_lo;
    PA_CHECK(pa_k_gram_finish(m, ts, pv->buf_av[0], a_hi > 0 ? pv->buf_av[1] : NULL, pv->buf_z, pv->d_partials,
                              a_lo, a_hi, ecg->beta->info.n, pv->d_beta, ecg->beta->info.lda, 0, 0, NULL, NULL,
                              NULL));
  }
  PA_CHECK(pa_k_gram_finish(m, ts, ecg->AP->val, NULL, ecg->P->val, pv->d_partials, t, 0, t, pv->d_mu, t, 0, 0,
                            NULL, NULL, NULL));
  /* R has not changed since the previous call's update: its column norms are
   * already there, except on the first call */
  if (!pv->rtr_valid) PA_CHECK(pa_k_colnorm2(m, ts, pv->d_R, pv->d_rtr_part, &pv->rtr_nblk));
  PA_CHECK(pa_k_trace_finish(pv->d_rtr_part, pv->rtr_nblk, ts, nrhs, pv->d_rtr, NULL, NULL));
  TAC(PA_T_GRAM, gemm_t);
  TIC(PA_T_COMM);
  if (pa_allreduce(pv->d_F, 5 * nrhs * nrhs)) return 1; /* the single reduction (ecg.c:563) */
  TAC(PA_T_COMM, comm_t);
  PA_CHECK(pa_rt_d2h_async(pv->h_pin, pv->d_rtr, sizeof(double)));
  PA_CHECK(pa_rt_sync());
  ecg->res = sqrt(pv->h_pin[0]);
  if (ecg->res < ecg->tol * ecg->normb || ecg->iter > ecg->maxIter) *rci_request = 1;
  else *rci_request = 0;
  TIC(PA_T_SMALL);
  /* mu = U^T U ; beta <- beta U^-1 ; alpha <- U^-T alpha ; beta(0:t,0:t) <- U^-T beta */
  PA_CHECK(pa_k_fused_small(pv->d_mu, t, nrhs, ecg->beta->info.m, ecg->beta->info.n, ecg->kbs,
                            pv->d_alpha, pv->d_beta, pv->d_info));
  TAC(PA_T_SMALL, potrf_t);
  TIC(PA_T_TRSM);
  PA_CHECK(pa_k_trsm(m, ts, t, pv->d_mu, ecg->P->val, ecg->AP->val));
  PA_CHECK(pa_k_trsm(m, ts, ecg->Z->info.n, pv->d_mu, ecg->Z->val, NULL));
  TAC(PA_T_TRSM, trsm_t);
  TIC(PA_T_UPDATE);
  {
    int vn = ecg->V->info.n, v_lo = vn < nrhs ? vn : nrhs, v_hi = vn - v_lo;
    PA_CHECK(pa_k_update_z(m, ts, v_lo, v_hi, ecg->Z->info.n, pv->d_beta, ecg->beta->info.lda,
                           pv->buf_v[0], pv->buf_v[1], pv->buf_z, NULL, NULL, NULL, NULL, NULL, 0, NULL));
  }
  TAC(PA_T_UPDATE, gemm_t);
  if (ecg->bs_red == ADAPT_BS && reduce_directions_odir(ecg, pv, 1, NULL, 0)) return 1;
  if (update_iterate(ecg, pv)) return 1;
  ecg->iter++;
  return shift_directions(ecg, pv, ecg->bs);
}

int preAlps_ECGIterate(preAlps_ECG_t* ecg, int* rci_request) {
  double t0 = pa_wtime();
  int rc = 0;
  if (ecg->ortho_alg == ORTHOMIN) rc = _preAlps_ECGIterateOmin(ecg, rci_request);
  else if (ecg->ortho_alg == ORTHODIR) rc = _preAlps_ECGIterateOdir(ecg, rci_request);
  else if (ecg->ortho_alg == ORTHODIR_FUSED) rc = _preAlps_ECGIterateOdirFused(ecg, rci_request);
  ecg->tot_t += pa_wtime() - t0;
  return rc;
}

/* ------------------------------------------------------------ wrap up ---- */
int _preAlps_ECGWrapUp(preAlps_ECG_t* ecg, double* solution) {
  ecg_priv_t* pv = priv_of(ecg);
  if (!pv) return PA_FAIL("solver not initialised");
  /* x = X * ones (ecg.c:674); the partial sums go through the free Z buffer */
  double* d_sol = pv->d_partials;
  size_t room = (size_t)pa_gram_max_blocks() * 2 * pv->ts * pv->ts;
  double* tmp = NULL;
  if ((size_t)pv->m > room) { tmp = (double*)pa_rt_malloc((size_t)pv->m * sizeof(double)); if (!tmp) return PA_FAIL("%s", pa_rt_error()); d_sol = tmp; }
  int rc = pa_k_rowsum(pv->m, pv->ts, ecg->X->info.n, pv->d_X, d_sol) ||
           pa_rt_d2h(solution, d_sol, (size_t)pv->m * sizeof(double));
  pa_rt_free(tmp);
  if (rc) return PA_FAIL("%s", pa_rt_error());
  return 0;
}

void _preAlps_ECGFree(preAlps_ECG_t* ecg) {
  ecg_priv_t* pv = priv_of(ecg);
  if (pv) {
    /* (only this object's requests: another solver's may be armed) */
    if (pv->d_spmm_parts) pa_k_spmm_gram_disarm(pv->d_spmm_parts);
    if (pv->d_bj_parts) pa_k_bj_gram_disarm(pv->d_bj_parts);
    pa_rt_sync();
    pa_rt_free(pv->d_info);
    pa_rt_host_free(pv->h_pin);
    pa_rt_host_free(pv->h_pin_i);
    pa_rt_event_destroy(pv->ev_res);
    pa_rt_event_destroy(pv->ev_alpha);
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 6; ++b) pa_rt_graph_free(pv->graph[a][b]);
    pv->magic = 0;
  }
  free(ecg->X); free(ecg->R); free(ecg->V); free(ecg->AV); free(ecg->alpha); free(ecg->beta);
  free(ecg->Z); free(ecg->P); free(ecg->AP);
  ecg->X = ecg->R = ecg->V = ecg->AV = ecg->alpha = ecg->beta = ecg->Z = ecg->P = ecg->AP = NULL;
  pa_rt_free(ecg->work); ecg->work = NULL;
  free(ecg->iwork); ecg->iwork = NULL;
}

int preAlps_ECGFinalize(preAlps_ECG_t* ecg, double* solution) {
  int rc = _preAlps_ECGWrapUp(ecg, solution);
  _preAlps_ECGFree(ecg);
  return rc;
}

void preAlps_ECGPrint(preAlps_ECG_t* ecg, int verbosity) {
  int rank = pa_world_rank();
  printf("[%d] prints ECG_t...\n", rank);
  printf("=== Summary ===\n");
  printf("\titer: %d\n\tres : %e\n\tbs  : %1d\n", ecg->iter, ecg->res, ecg->bs);
  printf("=== Timings ===\n");
  printf("\ttot_t  : %e s\n", ecg->tot_t);
  printf("\tcomm_t : %e s\n", ecg->comm_t);
  printf("\ttrsm_t : %e s\n", ecg->trsm_t);
  printf("\tgemm_t : %e s\n", ecg->gemm_t);
  printf("\tpotrf_t: %e s\n", ecg->potrf_t);
  printf("\tpstrf_t: %e s\n", ecg->pstrf_t);
  printf("\tlapmt_t: %e s\n", ecg->lapmt_t);
  printf("\tgesvd_t: %e s\n", ecg->gesvd_t);
  printf("\tgeqrf_t: %e s\n", ecg->geqrf_t);
  printf("\tormqr_t: %e s\n", ecg->ormqr_t);
  printf("\tcopy_t : %e s\n", ecg->copy_t);
  if (verbosity > 1 && ecg->X) {
    const char* names[9] = {"X", "R", "V", "AV", "P", "AP", "Z", "alpha", "beta"};
    CPLM_Mat_Dense_t* d[9] = {ecg->X, ecg->R, ecg->V, ecg->AV, ecg->P, ecg->AP, ecg->Z, ecg->alpha, ecg->beta};
    printf("=== Memory consumption ===\n");
    for (int i = 0; i < 9; ++i)
      printf("%s\n\tM=%d N=%d m=%d n=%d lda=%d nval=%d (HBM, %s)\n", names[i], d[i]->info.M, d[i]->info.N,
             d[i]->info.m, d[i]->info.n, d[i]->info.lda, d[i]->info.nval,
             d[i]->info.stor_type == ROW_MAJOR ? "row-interleaved" : "column major");
    printf("\n");
  }
  printf("[%d] ends printing ECG_t!\n", rank);
}

/* The driver loop of examples/test_ecg_prealps_op.c:203-223 (fused:
 * examples/test_ecg_bench_fused.c:243-259). */
static int ecg_solve_loop(preAlps_ECG_t* ecg, double* rhs, double* sol, double* res_hist, int* bs_hist,
                          int max_hist, int* n_hist);
static void leave_own_loop(void) {
  if (--g_own_loop == 0) { pa_k_spmm_gram_disarm(NULL); pa_k_bj_gram_disarm(NULL); }
}
int preAlps_ECGSolve(preAlps_ECG_t* ecg, double* rhs, double* sol, double* res_hist, int* bs_hist,
                     int max_hist, int* n_hist) {
  ++g_own_loop;
  int rc = ecg_solve_loop(ecg, rhs, sol, res_hist, bs_hist, max_hist, n_hist);
  leave_own_loop();
  return rc;
}
static int ecg_solve_loop(preAlps_ECG_t* ecg, double* rhs, double* sol, double* res_hist, int* bs_hist,
                          int max_hist, int* n_hist) {
  int rci = 0, stop = 0, nh = 0;
  if (preAlps_ECGInitialize(ecg, rhs, &rci)) return 1;
  if (preAlps_BlockJacobiApply(ecg->R, ecg->P)) return 1;
  ecg_priv_t* pvg = priv_of(ecg);
  if (pvg && pvg->use_graphs) {
    if (preAlps_BlockOperator(ecg->P, ecg->AP)) return 1;
    while (stop != 1) {
      if (graph_iteration(ecg, pvg, &rci, &stop)) return 1;
      if (res_hist && nh < max_hist) { res_hist[nh] = ecg->res; if (bs_hist) bs_hist[nh] = ecg->bs; }
      ++nh;
    }
  } else if (ecg->ortho_alg != ORTHODIR_FUSED) {
    if (preAlps_BlockOperator(ecg->P, ecg->AP)) return 1;
    while (stop != 1) {
      if (preAlps_ECGIterate(ecg, &rci)) return 1;
      if (rci == 0) {
        if (preAlps_BlockOperator(ecg->P, ecg->AP)) return 1;
      } else if (rci == 1) {
        /* same calls as the reference loop; the apply is queued before the host waits for
         * the residual norm (it does not depend on it and is simply unused after a stop) */
        ecg_priv_t* pv = priv_of(ecg);
        if (!pv) return PA_FAIL("solver not initialised");
        if (pv->z_ready) {
          /* D-Odir at full width: the first half already queued the block solve (behind which the host looked at
           * alpha); the second half and the product go out before the host waits for the norm, as below */
          pv->z_ready = 0;
          if (stopping_begin(ecg, pv)) return 1;
          if (preAlps_ECGIterate(ecg, &rci)) return 1;
          if (preAlps_BlockOperator(ecg->P, ecg->AP)) return 1;
        } else if (pv->lazy_stop && pv->lazy_ptr) {
          /* several processes: the norm travels with the beta all-reduce of the second half;
           * the half-step and the product queued before the decision are unused after a stop */
          if (ecg->ortho_alg == ORTHOMIN) { if (preAlps_BlockJacobiApply(ecg->R, ecg->Z)) return 1; }
          else if (preAlps_BlockJacobiApply(ecg->AP, ecg->Z)) return 1;
          if (preAlps_ECGIterate(ecg, &rci)) return 1;
          if (preAlps_BlockOperator(ecg->P, ecg->AP)) return 1;
        } else {
          if (stopping_begin(ecg, pv)) return 1;
          if (ecg->ortho_alg == ORTHOMIN) { if (preAlps_BlockJacobiApply(ecg->R, ecg->Z)) return 1; }
          else if (preAlps_BlockJacobiApply(ecg->AP, ecg->Z)) return 1;
        }
        if (stopping_end(ecg, pv, &stop)) return 1;
        if (res_hist && nh < max_hist) { res_hist[nh] = ecg->res; if (bs_hist) bs_hist[nh] = ecg->bs; }
        ++nh;
        if (stop == 1) break;
      }
    }
  } else {
    while (rci != 1) {
      if (preAlps_BlockOperator(ecg->P, ecg->AP)) return 1;
      if (preAlps_BlockJacobiApply(ecg->AP, ecg->Z)) return 1;
      if (preAlps_ECGIterate(ecg, &rci)) return 1;
      if (res_hist && nh < max_hist) { res_hist[nh] = ecg->res; if (bs_hist) bs_hist[nh] = ecg->bs; }
      ++nh;
    }
  }
  if (n_hist) *n_hist = nh < max_hist ? nh : max_hist;
  if (sol) return preAlps_ECGFinalize(ecg, sol);
  return 0;
}

/* Advance the driver loop (examples/test_ecg_prealps_op.c:208-221) by nsteps full
 * iterations, restarting from the same rhs (as after preAlps_ECGInitialize) whenever the
 * stopping test fires; what bench.py times. */
static int ecg_advance_loop(preAlps_ECG_t* ecg, double* rhs, int* rci_request, int nsteps, int* restarts,
                            int* last_iters, double* last_res);
int preAlps_ECGAdvance(preAlps_ECG_t* ecg, double* rhs, int* rci_request, int nsteps, int* restarts,
                       int* last_iters, double* last_res) {
  ++g_own_loop;
  int rc = ecg_advance_loop(ecg, rhs, rci_request, nsteps, restarts, last_iters, last_res);
  leave_own_loop();
  return rc;
}
static int ecg_advance_loop(preAlps_ECG_t* ecg, double* rhs, int* rci_request, int nsteps, int* restarts,
                            int* last_iters, double* last_res) {
  int stop = 0, done = 0;
  if (ecg->ortho_alg == ORTHODIR_FUSED) {
    /* the loop of examples/test_ecg_bench_fused.c:252-259: one reduction per iteration; *rci_request
     * becomes 1 when converged (then: restart from the same rhs) */
    while (done < nsteps) {
      if (preAlps_BlockOperator(ecg->P, ecg->AP)) return 1;
      if (preAlps_BlockJacobiApply(ecg->AP, ecg->Z)) return 1;
      if (preAlps_ECGIterate(ecg, rci_request)) return 1;
      ++done;
      if (*rci_request == 1) {
        if (restarts) ++*restarts;
        if (last_iters) *last_iters = ecg->iter;
        if (last_res) *last_res = ecg->res;
        if (_preAlps_ECGReset(ecg, rhs, rci_request)) return 1;
        if (preAlps_BlockJacobiApply(ecg->R, ecg->P)) return 1;
      }
    }
    return 0;
  }
  {
    ecg_priv_t* pvg = priv_of(ecg);
    if (pvg && pvg->use_graphs && *rci_request == 0) {
      while (done < nsteps) {
        if (graph_iteration(ecg, pvg, rci_request, &stop)) return 1;
        ++done;
        if (stop == 1) {
          if (restarts) ++*restarts;
          if (last_iters) *last_iters = ecg->iter;
          if (last_res) *last_res = ecg->res;
          if (_preAlps_ECGReset(ecg, rhs, rci_request)) return 1;
          if (preAlps_BlockJacobiApply(ecg->R, ecg->P)) return 1;
          if (preAlps_BlockOperator(ecg->P, ecg->AP)) return 1;
        }
      }
      return 0;
    }
  }
  while (done < nsteps) {
    if (preAlps_ECGIterate(ecg, rci_request)) return 1;
    if (*rci_request == 0) {
      if (preAlps_BlockOperator(ecg->P, ecg->AP)) return 1;
      ++done;
    } else {
      /* queue the stopping test, then the preconditioner apply of this iteration, and only
       * then wait for the norm: the apply does not depend on it and is discarded on stop */
      ecg_priv_t* pv = priv_of(ecg);
      if (!pv) return PA_FAIL("solver not initialised");
      int early = pv->z_ready;
      int lazy = early || (pv->lazy_stop && pv->lazy_ptr);
      if (lazy) {   /* see preAlps_ECGSolve */
        if (early) { pv->z_ready = 0; if (stopping_begin(ecg, pv)) return 1; }
        else if (ecg->ortho_alg == ORTHOMIN) { if (preAlps_BlockJacobiApply(ecg->R, ecg->Z)) return 1; }
        else if (preAlps_BlockJacobiApply(ecg->AP, ecg->Z)) return 1;
        if (preAlps_ECGIterate(ecg, rci_request)) return 1;
        if (preAlps_BlockOperator(ecg->P, ecg->AP)) return 1;
      } else {
        if (stopping_begin(ecg, pv)) return 1;
        if (ecg->ortho_alg == ORTHOMIN) { if (preAlps_BlockJacobiApply(ecg->R, ecg->Z)) return 1; }
        else if (preAlps_BlockJacobiApply(ecg->AP, ecg->Z)) return 1;
      }
      if (stopping_end(ecg, pv, &stop)) return 1;
      if (lazy && stop != 1) ++done;
      if (stop == 1) {
        if (restarts) ++*restarts;
        if (last_iters) *last_iters = ecg->iter;
        if (last_res) *last_res = ecg->res;
        if (_preAlps_ECGReset(ecg, rhs, rci_request)) return 1;
        if (preAlps_BlockJacobiApply(ecg->R, ecg->P)) return 1;
        if (preAlps_BlockOperator(ecg->P, ecg->AP)) return 1;
        ++done;
        continue;
      }
    }
  }
  return 0;
}
