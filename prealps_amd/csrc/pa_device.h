/*
 * pa_device.h -- the thin C-ABI shim between the C host code (ecg.c,
 * operator.c, block_jacobi.c) and the HIP side (runtime.hip, kernels.hip).
 * Host code never includes a HIP header; device code never sees a preAlps
 * struct.  Panels are row-interleaved: element (i, j) at p[i * ts + j].
 */
#ifndef PA_DEVICE_H
#define PA_DEVICE_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- runtime (runtime.hip) --------------------------------------------- */
int pa_rt_init(int device);
void pa_rt_shutdown(void);
int pa_rt_ready(void);
void pa_rt_set_stream(void* s);
void* pa_rt_stream(void);
int pa_rt_sync(void);
const char* pa_rt_error(void);
void* pa_rt_malloc(size_t bytes);
void pa_rt_free(void* d);
void* pa_rt_host_alloc(size_t bytes);
void* pa_rt_host_alloc_coherent(size_t bytes);   /* fine-grained: a kernel writes, the host polls */
int pa_rt_stream_state(void);                     /* 0: work pending, 1: idle, -1: error */
void pa_rt_host_free(void* h);
int pa_rt_memset(void* d, int v, size_t bytes);
int pa_rt_h2d(void* d, const void* h, size_t bytes);
int pa_rt_d2h(void* h, const void* d, size_t bytes);
int pa_rt_d2d(void* dst, const void* src, size_t bytes);
int pa_rt_h2d_async(void* d, const void* pinned, size_t bytes);
int pa_rt_d2h_async(void* pinned, const void* d, size_t bytes);
void* pa_rt_side_stream(void);
int pa_rt_stream_wait_event(void* stream, void* event);
int pa_rt_event_record_on(void* event, void* stream);
void* pa_rt_event_create(void);
void pa_rt_event_destroy(void* e);
int pa_rt_event_record(void* e);
int pa_rt_event_wait(void* e);
double pa_rt_event_elapsed_s(void* a, void* b);
int pa_rt_num_cus(void);
int pa_rt_device_count(void);
/* graphs (runtime.hip): capture the work queued on the current stream, replay it later */
int pa_rt_capture_begin(void);
int pa_rt_capture_end(void** exec_out);
int pa_rt_graph_launch(void* exec);
void pa_rt_graph_free(void* exec);
void pa_rt_skip(int on);      /* 1: asynchronous device operations become no-ops (a graph queues them) */
int pa_rt_skipping(void);

/* ---- native RCCL hooks (comm_rccl.hip) --------------------------------- */
const char* pa_rccl_error(void);
int pa_rccl_available(void);
int pa_rccl_unique_id(char* id128);
int pa_rccl_init(const char* id128, int rank, int size);
void pa_rccl_shutdown(void);
int pa_rccl_allreduce(void* ctx, double* buf, int count);
int pa_rccl_exchange(void* ctx, const double* send, const int* send_counts, double* recv,
                     const int* recv_counts, const int* peers, int npeers);

/* ---- SpMM (utils/cplm_v0/cplm_v0_matmult_v2.c:108-343, K1) ------------- */
/* The local row panel in sliced-ELL form (SELL-64): rows are cut into slices
 * of 64 consecutive rows (slices never cross a subdomain), each slice padded to
 * its longest row and stored entry-major, so lane r of a wavefront reads
 * val[off + k*64 + r]: every load of the 12 B/nonzero stream is coalesced.
 * Padding entries carry value 0 and the row's own (diagonal) column. */
typedef struct {
  int m;                    /* local rows */
  int nslices;
  const long long* sl_off;  /* nslices+1: offset of the slice in val / col */
  const int* sl_len;        /* nslices: padded row length */
  const int* sl_row0;       /* nslices: first local row */
  const int* sl_nrows;      /* nslices: rows in the slice (<= 64) */
  const int* col;           /* local column: < m own row, >= m halo slot */
  const double* val;
  int nblk;                 /* workgroup blocks: consecutive slices of one subdomain */
  const int* blk_slice;     /* nblk+1: first slice of each block */
  const int* blk_win;       /* 2*nblk: [w0, w1) local X rows staged in LDS */
  const int* order;         /* block ids: interior blocks first, then blocks that read halo rows */
  int n_interior;
  int win_cap;              /* rows of the largest window */
  /* staged variant (preferred): every X row a block touches is copied to LDS
   * first -- its own row range, then the listed external rows -- and `col16`
   * holds the LDS slot of each entry (2 B per nonzero instead of 4). */
  int staged;
  const unsigned short* col16;
  const int* blk_ext_off;   /* nblk+1: range of the block in ext_rows */
  const int* ext_rows;      /* local row ids (>= m: halo slot) staged after the own rows */
  int stage_cap;            /* rows of the largest staging area */
  /* run variant of the staged plan: `col16` holds one LDS slot per run of three consecutive
   * slots and `val` three values per run ([k][3][lane], sl_off / sl_len count runs); the
   * staging area is [external rows below the own range | own rows | external rows above |
   * two zero rows], blk_nlow = how many of a block's external rows lie below its own range. */
  int runs;
  const int* blk_nlow;
  int runs_cols;            /* columns a workgroup of the run plan stages and computes (= the panel stride, or 8
                             * when a 16-column panel is split between two workgroups) */
} pa_spmm_plan_t;
/* phase 0: interior blocks, 1: halo-reading blocks, 2: all */
int pa_k_spmm(const pa_spmm_plan_t* pl, int ts, const double* X, const double* Xhalo,
              double* Y, int phase);
/* Ask the next product X -> Y of pa_k_spmm (4-column panels, run plan) to leave the partial
 * blocks of [Y | R]^T X (8 x 4 each, the layout of pa_k_gram with two panels) in `partials`
 * (room for cap blocks); pa_k_spmm_gram_take returns how many there are (0: the product ran
 * without them) and ends the request; pa_k_finish32 sums them (t > 0: and factors, as
 * pa_k_gram_finish does). */
void pa_k_spmm_gram_arm(const double* X, const double* Y, const double* R, double* partials, int cap);
void pa_k_spmm_gram_disarm(const double* owner);   /* owner: the request's partials (only that request ends); NULL: any */
long long pa_k_spmm_gram_launches(void);   /* launches of the SpMM with the Gram block so far */
int pa_k_spmm_gram_take(const double* X, const double* Y);
int pa_k_finish32(const double* partials, int nblk, double* scratch, int t, int T, double* out, double* mu,
                  double* alpha, int* info);       /* scratch: pa_finish32_scratch_blocks() x 32 doubles */
int pa_finish32_scratch_blocks(void);
/* the same sum, followed by the residual norm from the update kernel's column sums (res2[0], res2[1] = *info) */
int pa_k_finish32_trace(const double* partials, int nblk, double* scratch, double* out, const double* rtr_partials,
                        int rtr_nblk, int ts, int nc, double* res2, int* info);
/* Ask the next block solve in -> out on a 4-column panel (every block through bj_g4.hip) to leave the partial
 * blocks of [in | prev]^T out behind, one per block; pa_k_bj_gram_take: how many there are (0: none). */
void pa_k_bj_gram_arm(const double* in, const double* out, const double* prev, double* partials, int cap);
void pa_k_bj_gram_disarm(const double* owner);
long long pa_k_bj_gram_applies(void);        /* block solves that left the Gram block behind so far */
int pa_k_bj_gram_take(const double* in, const double* out);
/* sendbuf[i*ts + c] = X[idx[i]*ts + c] */
int pa_k_pack_rows(int n, int ts, const int* idx, const double* X, double* sendbuf);

/* HBM calibration: which = 0 copy src -> dst, 1 read src (dst receives nothing). */
int pa_k_probe(int which, size_t bytes, const double* src, double* dst);
int pa_k_spacer(int us);

/* ---- tall-skinny kernels (ecg.c:250,311,330,347,425,438,510 K2; K3; K4) -- */
/* Partial Gram blocks C = [A0 | A1]^T B over the local rows, one (npan*ts) x ts
 * column-major block per workgroup into `partials`; returns the number of
 * workgroups through nblk.  A1 may be NULL (npan = 1). */
int pa_k_gram(int m, int ts, const double* A0, const double* A1, const double* B,
              double* partials, int* nblk);
/* pa_k_gram + pa_k_finish.  With t > 0 (out = [W ; G^T], a_lo = nb = t, a_hi = T,
 * ld_out = t + T) the finishing launch goes on like pa_k_potrf_alpha on `out`. */
int pa_k_gram_finish(int m, int ts, const double* A0, const double* A1, const double* B,
                     double* partials, int a_lo, int a_hi, int nb, double* out, int ld_out, int t,
                     int T, double* mu, double* alpha, int* info);
/* pa_k_gram + one launch that sums the block (as pa_k_finish) and the residual norm (as
 * pa_k_trace_finish: res2[0] = norm^2, res2[1] = *info). */
int pa_k_gram_finish_trace(int m, int ts, const double* A0, const double* A1, const double* B, double* partials,
                           int a_lo, int a_hi, int nb, double* out, int ld_out, const double* rtr_partials,
                           int rtr_nblk, int nc, double* res2, const int* info);
int pa_gram_max_blocks(void);
/* out[i + ld_out*j] = sum over blocks, for rows i < a_lo (panel 0) and
 * a_lo <= i < a_lo + a_hi (panel 1, column i - a_lo), j < nb. */
int pa_k_finish(const double* partials, int nblk, int npan, int ts, int a_lo, int a_hi,
                int nb, double* out, int ld_out);
/* In-place upper Cholesky of the t x t column-major W (LAPACKE_dpotrf 'U',
 * ecg.c:318,431,577); *info (device int) = 0 or failing column + 1. */
int pa_k_potrf(double* W, int t, int* info);
/* The t x t part of one fused Orthodir step (ecg.c:577-587): Cholesky of mu,
 * beta <- beta U^-1, alpha <- U^-T alpha, beta(0:t,0:t) <- U^-T beta(0:t,0:t). */
int pa_k_fused_small(double* mu, int t, int nrhs, int bm, int bn, int ldb, double* alpha,
                     double* beta, int* info);
/* P <- P U^-1 and AP <- AP U^-1 on the first t columns (cblas_dtrsm Right,
 * Upper, ecg.c:324-327,434-435).  AP may be NULL. */
int pa_k_trsm(int m, int ts, int t, const double* U, double* P, double* AP);
/* X += P alpha, R -= AP alpha (ecg.c:337-338,500-501), alpha is t x nc with
 * leading dimension t; also the per-workgroup sums of R(:,c)^2 for the
 * stopping test (ecg.c:250) into rtr_partials[blk*ts + c]. */
int pa_k_update_xr(int m, int ts, int t, int nc, const double* alpha, const double* P,
                   const double* AP, double* X, double* R, double* rtr_partials, int* nblk,
                   int trace_nc, double* res2, const int* info, double* host);
/* trace_nc > 0: followed by pa_k_trace_finish over trace_nc columns into res2 and, when
 * `host` (pinned, device-visible) is given, into host[0..1] as well (no copy needed). */
/* buf = [W ; G^T] ((t+T) x t, leading dimension t+T) with W = AP^T P and G = P^T R of the
 * un-normalised P  ->  mu = chol(W) (upper, t x t), alpha = U^-T G (t x T, ld t). */
int pa_k_potrf_alpha(const double* buf, int t, int T, double* mu, double* alpha, int* info);
/* pa_k_trsm followed by pa_k_update_xr in one pass over P, AP, X, R.  gram != NULL: U and alpha
 * are outputs -- every workgroup computes them from gram = [W ; G^T] as pa_k_potrf_alpha does
 * (nc = T) and the first one stores them and *info. */
int pa_k_trsm_update(int m, int ts, int t, int nc, double* U, double* alpha, double* P,
                     double* AP, double* X, double* R, double* rtr_partials, int* nblk, int trace_nc,
                     double* res2, int* info, double* host, const double* gram, double* ukeep);
/* ukeep != NULL (lazy normalisation, panels of up to 4 columns): P and AP are NOT overwritten -- X and R get the
 * same update from rows normalised in registers -- and the t x t factor U is stored in ukeep for pa_k_update_z. */
/* Standalone sums of R(:,c)^2 (same layout as above). */
int pa_k_colnorm2(int m, int ts, const double* R, double* rtr_partials, int* nblk);
/* res2[0] = sum over blocks and columns c < nc; res2[1] = *info (0 if info is NULL). */
int pa_k_trace_finish(const double* rtr_partials, int nblk, int ts, int nc, double* res2,
                      const int* info, double* host);    /* host: pinned words that receive the same two values (may be NULL) */
/* Z(:, :nc) -= [V0(:, :a_lo) | V1(:, :a_hi)] beta, beta is (a_lo+a_hi) x nc,
 * leading dimension ldb (ecg.c:354,517). */
int pa_k_update_z(int m, int ts, int a_lo, int a_hi, int nc, const double* beta, int ldb,
                  const double* V0, const double* V1, double* Z, const double* note_src, double* note_host,
                  const double* ucur, const double* uprev, double* zz_part, int zz_cols, int* zz_nblk);
/* One-shot request to the next pa_k_update_z (panels of up to 4 columns): also copy the new row r of Z into the slots
 * pk_slot[pk_off[r] .. pk_off[r + 1]) of sendbuf (ts doubles each) -- the halo pack of the product that follows.
 * 0: not taken (wider panels; the caller must not tell the operator the rows are packed). */
int pa_k_update_z_pack(int ts, const int* pk_off, const int* pk_slot, double* sendbuf);
/* ucur != NULL (lazy normalisation): V0 / Z belong to the factor ucur, V1 to uprev, beta holds the raw Gram
 * blocks [V0-side ; V1-side]^T Z; the kernel applies U^-1 where the reference's panels would carry it. */
/* note_host != NULL: note_src[0..1] (device) are also written to note_host[0..1] (pinned, device-visible) */
/* The next launch that writes two words to pinned host memory (pa_k_trsm_update / pa_k_update_xr with
 * `host`, pa_k_update_z with note_host) also writes host[2] = seq behind them (seq != 0; one-shot), for
 * a host that polls that word instead of waiting for an event. */
void pa_k_note_seq(double seq);
/* dst(:, :nc) = src(:, :nc) (mkl_domatcopy, ecg.c:358,521-523). */
int pa_k_copy_cols(int m, int ts, int nc, const double* src, double* dst);
/* A(:, :t) <- A(:, :t) Q, Q is t x t column-major (the effect of LAPACKE_dormqr
 * 'R','N' at ecg.c:476-479 with Q formed explicitly). */
int pa_k_right_mult(int m, int ts, int t, const double* Q, double* A);
/* A(:, j) <- A(:, piv[j]) for j < n (LAPACKE_dlapmt forward, ecg.c:380); piv 0-based, device. */
int pa_k_permute_trsm(int m, int ts, int n, const int* piv, int t, const double* U, const double* src, double* dst);
int pa_k_permute_cols(int m, int ts, int n, const int* piv, double* A);
/* sol[i] = sum_j X[i][j], j < nc (ecg.c:674). */
int pa_k_rowsum(int m, int ts, int nc, const double* X, double* sol);

/* ---- block-Jacobi apply (block_jacobi.c:93-109, K8) --------------------- */
typedef struct {
  int nparts;              /* blocks owned by this process */
  const int* row0;         /* nparts: first local row of each block */
  const int* nrows;        /* nparts */
  const int* bw;           /* nparts: bandwidth w of the block's factor */
  const long long* off;    /* nparts: offset (doubles, even) of the block in Lf / Lb */
  const int* map_f;        /* local row visited at forward step j (m entries) */
  const int* map_b;        /* local row visited at backward step j */
  /* Narrow bands: one record of wr = (w+1 rounded up to even) doubles per step:
   * forward  Lf[off + j*wr + d-1] = L(j+d, j) / L(j,j),                 d = 1..w, then a zero
   * backward Lb[off + j*wr + d-1] = L(b-1-j, b-1-j-d) / L(b-1-j, b-1-j).
   * Wide bands: records of W = bjw_window(w) doubles, the value for target row i at i mod W.
   * Arrays are padded by 2 KiB at the end. */
  const double* Lf;
  const double* Lb;
  /* optional (NULL: absent): the same records in pairs for the classes R = 2, 3 (k_bj_pairs), block p at
   * off2[p]: per chunk of 8 steps two sub-blocks of [2 pivot pairs][w + 4 target rows] double2 */
  const double* Lf2; const double* Lb2; const long long* off2;
  /* optional (NULL: absent): ONE copy of the factor for both sweeps of panels of up to 4 columns, in
   * selective-inversion form by groups of four pivots (bj_g4.hip), block p at off2[p]: per group
   * (w + 4) rows x 4 pivots of M = [T - I ; -G], eligible classes flagged in class_g4 */
  const double* Lg4;
  const int* class_g4;    /* host array, nclass: 1 = every block of the class has a record in Lg4 */
  const int* class_bmax;   /* host array, nclass: most rows of a block in the class */
  const double* invd_f;    /* 1 / L(j,j) in forward step order (m entries) */
  const double* invd_b;    /* ... in backward step order */
  int nclass;              /* parts grouped by register sets R = ceil((w+64)/64) of the one-wavefront kernel */
  const int* class_R;      /* host array, nclass: register sets per lane; < 0: wide class (workgroup per block) */
  const int* class_count;  /* host array */
  const int* class_wmax;   /* host array: widest band in the class (sizes the LDS chunks) */
  const int* const* class_list; /* host array of device pointers to part ids */
} pa_bj_plan_t;
int pa_k_bj_g4(const pa_bj_plan_t* pl, const int* list, int count, int wmax, int bmax, int xs, int ncol,
                const double* in, double* out);
/* One-shot: the next pa_k_bj_g4 on a 4-column panel also leaves the 8 x 4 block [in | prev]^T out of every
 * block in part (32 doubles per block, in the order of its list; the layout of pa_k_gram with two panels). */
void pa_k_bj_g4_gram(const double* prev, double* part);
int pa_bj_max_R(void);
int pa_k_bj_pairs(const int* list, int count, const int* nrows, const int* bw, const long long* off,
                  const long long* off2, const double* L, double* L2);
/* bj_g4.hip: the records of Lg4 from the plain forward records, and the block solve that reads them
 * (blocks of at most pa_bj_g4_max_rows() rows, bands up to pa_bj_g4_max_band()); `in` / `out` point
 * at the first of up to four columns of panels with row stride xs. */
int pa_bj_g4_max_rows(void);
int pa_bj_g4_max_band(void);
int pa_bj_g4_max_band8(void);
int pa_k_bj_g4_setup(const int* list, int count, const int* nrows, const int* bw, const long long* off,
                      const long long* off2, const double* L, double* Lg4);
/* Band Cholesky on the device for blocks with bandwidth <= pa_bj_factor_wmax(): `band` holds
 * each listed block's rows in factor order, (w+1) doubles per row (A(i, i-d) at d), at offset
 * boff[part]; writes the forward / backward sweep records and 1/L(j,j) straight into Lf, Lb
 * (pre-zeroed), invd_f, invd_b.  *fail (device, pre-zeroed) = 1 + local position of the first
 * non-positive pivot seen. */
int pa_bj_factor_wmax(void);
/* dst[off[e]] = val[e], e < n */
int pa_k_scatter(size_t n, const long long* off, const double* val, double* dst);
/* Wider bands (up to 4032): `band` diagonal-major per block (A(i, i-d) at boff + d*nrows + i),
 * factored in place (blocked, one workgroup per block), then laid out into Lf / Lb / invd
 * (records in window-slot order for bands above wide_from, else [d = 1..w | 0]). */
int pa_k_bj_factor_big(const int* list, int count, int wmax, int wide_from, const int* row0, const int* nrows,
                       const int* bw, const long long* off, const long long* boff, double* band, double* Lf, double* Lb,
                       double* invd_f, double* invd_b, int* fail);
int pa_k_bj_factor(const int* list, int count, int wmax, const int* row0, const int* nrows, const int* bw,
                   const long long* off, const long long* boff, const double* band, double* Lf, double* Lb,
                   double* invd_f, double* invd_b, int* fail);
int pa_k_bj_apply(const pa_bj_plan_t* pl, int ts, const double* in, double* out);

/* ---- sparse block solve for large diagonal blocks (nd.c) --------------------------------- */
/* Supernodes of a nested-dissection Cholesky factor, all blocks of the process in one numbering.
 * Supernode s: n[s] pivot columns, m[s] rows below, panel P = [T ; -G] (see kernels.hip) twice:
 * F + offF[s] column major with leading dimension ld[s] >= n + m, B + offB[s] row major with row
 * length PA_ND_LD(n); only the entries below the diagonal are read.  rows[rows_off[s] + r] =
 * local panel row of front row r; src[2 (rows_off[s] + r) + c] = where front row r finds the
 * contribution of child c (row of its vector, -1: none), that vector starting at row
 * ccoff[2 s + c] of `contrib`; this supernode's own contribution starts at row coff[s].
 * dinv[local row] = 1 / L(row, row).  Y: scratch panel for the forward result (local rows). */
/* leading dimensions of both copies: multiples of 128 bytes, so that the 512-byte segment a
 * wavefront loads never straddles an extra cache line */
#define PA_ND_LD(x) (((x) + 15) & ~15)
typedef struct {
  const int* n; const int* m; const int* ld; const long long* offF; const long long* offB; const int* rows_off;
  const int* coff; const int* ccoff; const int* rows; const int* src; const double* dinv;
  const double* F; const double* B; double* contrib; double* Y;
  int nlevel;                  /* levels of the forest, bottom-up; host arrays of device pointers: */
  const int* f_count; const int* const* f_front; const int* const* f_row0;   /* forward: (front, first front row) per workgroup */
  const int* b_count; const int* const* b_front; const int* const* b_col0;   /* backward: (front, first pivot column) per workgroup */
} pa_nd_plan_t;
/* ---- numeric factorisation of those blocks on the device (nd_factor.hip) ------------------- */
/* n, m, ld, offF, offB, rows_off, rows, src, F, B, dinv as in pa_nd_plan_t.  front[s] = device
 * address of the dense front of supernode s while its level (and its parent's) is being worked on:
 * column major, leading dimension ld[s], lower triangle.  child[2 s + c] = supernode id of child c
 * (-1: none); newrow[rows_off[s] + r] = index of front row r in its block's elimination order;
 * the block's own entries of the pivot columns of s: columns acp[acol0[s] + j] .. of (ari, acv),
 * rows in elimination order, lower triangle.  fail: smallest (supernode << 32 | pivot column) whose
 * pivot was not positive, ~0 if none; fail[1]: see pa_k_ndf_check. */
typedef struct {
  const int* n; const int* m; const int* ld; const long long* offF; const long long* offB; const int* rows_off;
  const int* rows; const int* src; const int* child; const int* newrow;
  const long long* acol0; const long long* acp; const int* ari; const double* acv;
  const unsigned long long* front; const int* ldf;
  double* F; double* B; double* dinv; unsigned long long* fail;
} pa_ndf_args_t;
/* tiles = (front, tile row, tile column) of 64 x 64 entries; chunks = (front, first row) of
 * pa_nd_chunk_rows() rows; jb = first pivot of the step (a multiple of 64) */
int pa_k_ndf_assemble(const pa_ndf_args_t* a, const int* tf, const int* ti, const int* tj, int ntiles,
                      const int* fronts, int nfronts);
int pa_k_ndf_potrf(const pa_ndf_args_t* a, const int* fronts, int nfronts, int jb);
int pa_k_ndf_trsm(const pa_ndf_args_t* a, const int* cfront, const int* crow0, int nchunks, int jb, int inverse);
int pa_k_ndf_update(const pa_ndf_args_t* a, const int* tf, const int* ti, const int* tj, int ntiles, int jb,
                    int inverse);
int pa_k_ndf_pinit(const pa_ndf_args_t* a, const int* cfront, const int* crow0, int nchunks);
int pa_k_ndf_finalize(const pa_ndf_args_t* a, const int* tf, const int* ti, const int* tj, int ntiles);
/* after finalize: fail[1] = max over the fronts of |L_11 (L_11^-1 1) - 1| with the inverse taken from the
 * finished panels (bit pattern of a non-negative double) */
int pa_k_ndf_check(const pa_ndf_args_t* a, const int* fronts, int nfronts, int nmax);

int pa_nd_chunk_rows(void);           /* front rows per forward workgroup */
int pa_nd_block_cols(void);           /* pivot columns per backward workgroup */
int pa_k_nd_apply(const pa_nd_plan_t* pl, int ts, const double* in, double* out);

#ifdef __cplusplus
}
#endif
#endif
