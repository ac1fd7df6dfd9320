/*
 * operator.c -- the global-static linear operator of the ECG drivers, resident
 * in HBM.  Host side of the SpMM hot path.
 *
 * Reference behaviour kept (all under /root/reference):
 *   preAlps_OperatorBuild      utils/operator.c:38-134    load -> scale ->
 *                              partition -> permute -> row panels
 *   MatrixMarket reader        utils/cplm_light/cplm_matcsr.c:96-243
 *   SymRACScaling              utils/cplm_light/cplm_matcsr.c:1461-1554
 *   ordering from a partition  utils/cplm_v0/cplm_v0_metis_utils.c:22-43,197-222
 *   symmetric permutation      utils/cplm_v0/cplm_v0_matcsr.c:941-1022
 *   preAlps_BlockOperator      utils/operator.c:334-351 ->
 *                              utils/cplm_v0/cplm_v0_matmult_v2.c:108-343
 *   getters                    utils/operator.c:353-393
 *
 * MI355X design instead of the reference's: the number of subdomains is
 * `nparts` (not the process count); a process owns a contiguous range of
 * parts; only boundary rows travel between processes (the reference ships
 * whole panels); the O(m*P) colPos table is replaced by row blocks with an
 * interior / halo-reading split so the exchange overlaps the interior SpMM.
 */
#include <ctype.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <sys/mman.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

#include "pa_host.h"

typedef struct {
  pa_operator_info_t info;
  /* device matrix: SELL-64 slices with local column ids */
  long long* d_sl_off; int* d_sl_len; int* d_sl_row0; int* d_sl_nrows;
  int* d_col; double* d_val;
  int lnnz;
  double sell_entries;   /* stored entries including padding */
  /* SpMM plan */
  pa_spmm_plan_t plan;
  int* d_blk_slice; int* d_blk_win; int* d_order;
  unsigned short* d_col16; int* d_blk_ext_off; int* d_ext_rows; int* d_blk_nlow;
  int* lcol;             /* host: local column ids of the panel (own rows < m <= halo slots) */
  int plan_ts;           /* panel stride the current SpMM plan was cut for (0: none) */
  double stream_bytes;   /* bytes of matrix data one SpMM streams */
  /* halo exchange */
  int npeers;
  int* peers;        /* process ids */
  int* send_rows;    /* per peer, rows */
  int* recv_rows;
  int* send_cnt;     /* scratch: elements for the current stride */
  int* recv_cnt;
  int nsend;         /* total rows packed */
  int* send_idx;     /* local rows packed for the peers, peer after peer */
  int* halo_cols;    /* global row of every halo slot, ascending */
  int* d_send_idx;
  int* d_pk_off; int* d_pk_slot;   /* the inverse of send_idx: row r is packed into slots pk_slot[pk_off[r] .. pk_off[r + 1]) */
  const double* prepacked;         /* the panel whose send rows the solver's update kernel has already packed (pa_operator_pack_hint) */
  double* d_sendbuf; double* d_halo;
  int buf_ts;        /* stride the buffers are sized for */
  void* ev_packed;   /* send buffer is packed (main stream) */
  void* ev_halo;     /* halo rows have arrived (side stream) */
  int* colPos_dummy;
} pa_operator_t;

static pa_operator_t g_op;
static double g_setup_build_s, g_setup_plan_s;   /* host seconds: scale/permute/halo lists, SpMM plan */

const pa_operator_info_t* pa_operator_info(void) { return g_op.info.built ? &g_op.info : NULL; }

/* ------------------------------------------------------------------ utils */
/* Large host arrays (hundreds of MB, written once): 2 MiB alignment + a transparent-huge-page
 * hint, so that filling them is not dominated by 4 KiB page faults.  Release with free(). */
static void* big_alloc(size_t bytes) {
  void* p = NULL;
  if (bytes < ((size_t)8 << 20)) return malloc(bytes ? bytes : 1);
  if (posix_memalign(&p, (size_t)2 << 20, bytes)) return NULL;
  (void)madvise(p, bytes, MADV_HUGEPAGE);
  return p;
}

/* Rows per SpMM workgroup.  The measured optimum on a full GPU is 256 (192 at 8 columns); a process that
 * owns few rows (one shard of a multi-GPU run: 130 k rows are 519 such blocks on 256 CUs) gets smaller
 * blocks, so that every CU has at least four workgroups to hide the staging and streaming latencies. */
static int env_int(const char* name, int dflt);
static size_t g_dbg_val_bytes, g_dbg_slot_bytes;
static int spmm_block_rows(int m, int dflt) {
  int cus = pa_rt_num_cus() > 0 ? pa_rt_num_cus() : 256;
  int rows = dflt;
  while (rows > 64 && (long long)m / rows < 4LL * cus) rows -= 64;
  return rows;
}

static int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}

typedef struct { int c; double v; } cv_t;
static int cmp_cv(const void* a, const void* b) {
  int ca = ((const cv_t*)a)->c, cb = ((const cv_t*)b)->c;
  return (ca > cb) - (ca < cb);
}

static int owner_of_part(int p, int nparts, int size) {
  /* process g owns parts [g*nparts/size, (g+1)*nparts/size) */
  int g = (int)(((long long)p * size) / nparts);
  while (g + 1 < size && (long long)(g + 1) * nparts / size <= p) ++g;
  while (g > 0 && (long long)g * nparts / size > p) --g;
  return g;
}

static int part_of_row(const int* rowPos, int nparts, int row) {
  int lo = 0, hi = nparts; /* rowPos[lo] <= row < rowPos[hi] */
  while (hi - lo > 1) {
    int mid = (lo + hi) / 2;
    if (rowPos[mid] <= row) lo = mid; else hi = mid;
  }
  return lo;
}

/* --------------------------------------------------------------- free ---- */
static void free_plan(pa_operator_t* o) {
  pa_rt_free(o->d_sl_off); pa_rt_free(o->d_sl_len); pa_rt_free(o->d_sl_row0); pa_rt_free(o->d_sl_nrows);
  pa_rt_free(o->d_col); pa_rt_free(o->d_val);
  pa_rt_free(o->d_blk_slice); pa_rt_free(o->d_blk_win); pa_rt_free(o->d_order);
  pa_rt_free(o->d_col16); pa_rt_free(o->d_blk_ext_off); pa_rt_free(o->d_ext_rows); pa_rt_free(o->d_blk_nlow);
  o->d_blk_nlow = NULL;
  o->d_sl_off = NULL; o->d_sl_len = o->d_sl_row0 = o->d_sl_nrows = o->d_col = NULL; o->d_val = NULL;
  o->d_blk_slice = o->d_blk_win = o->d_order = NULL; o->d_col16 = NULL; o->d_blk_ext_off = o->d_ext_rows = NULL;
  memset(&o->plan, 0, sizeof(o->plan));
  o->plan_ts = 0;
}

void preAlps_OperatorFree(void) {
  pa_operator_t* o = &g_op;
  free_plan(o);
  free(o->lcol);
  free(o->info.rowPos); free(o->info.perm);
  free(o->info.A.rowPtr); free(o->info.A.colInd); free(o->info.A.val);
  free(o->peers); free(o->send_rows); free(o->recv_rows); free(o->send_cnt); free(o->recv_cnt);
  free(o->send_idx); free(o->halo_cols);
  pa_rt_event_destroy(o->ev_packed); pa_rt_event_destroy(o->ev_halo);
  pa_rt_free(o->d_send_idx); pa_rt_free(o->d_sendbuf); pa_rt_free(o->d_halo);
  pa_rt_free(o->d_pk_off); pa_rt_free(o->d_pk_slot);
  free(o->colPos_dummy);
  memset(o, 0, sizeof(*o));
}

/* ------------------------------------------------------------ the plan ---- */
/* Local CSR (local column ids) -> SELL-64 slices + workgroup blocks.  A block
 * is a run of slices of one subdomain (at most spmm_block_rows() rows);
 * its LDS window is the subdomain's own row range, or the 256 rows around the
 * block when the subdomain is larger than that (1024-row windows measured slower). */
static int build_plan_staged(pa_operator_t* o, int ts);
static int build_plan_runs(pa_operator_t* o, int ts);

static int build_plan(pa_operator_t* o, int ts) {
  const pa_operator_info_t* in = &o->info;
  const int* rowptr = in->A.rowPtr;
  const int* colind = o->lcol;
  const double* val = in->A.val;
  int m = in->m;
  free_plan(o);
  /* -1 (default): stage when the external rows a block copies are small next to its matrix
   * slice (long rows: elasticity); short rows (7-point stencils) gather through L2 instead */
  int want = env_int("PREALPS_SPMM_STAGED", -1);
  if (want != 0 && env_int("PREALPS_SPMM_RUNS", 1)) {
    /* rows whose nonzeros come in runs of consecutive columns (vector problems: 3 dofs per
     * node) share one LDS slot per run of three: 8.67 B per nonzero instead of 10 */
    int rc = build_plan_runs(o, ts);
    if (rc < 0) return 1;
    if (rc == 0) { o->plan_ts = ts; return 0; }
    free_plan(o);
  }
  if (want < 0 && ts >= 16) want = 0; /* wide panels: the 128-B X rows gather well from L2 (measured) */
  if (want != 0) {
    int rc = build_plan_staged(o, ts);
    if (rc < 0) return 1;
    if (rc == 0) {
      double ext_bytes = (o->stream_bytes - 10.0 * o->sell_entries) / 4.0 * ts * 8.0;
      /* round 4: at up to 4 columns the staged kernel (batched staging, its Gram block in the epilogue) is worth
       * it up to external rows of half the matrix slice: 7-point Poisson 100^3 (38 %) 226.6 -> 221.6 us per
       * iteration against the window kernel, although the plain product alone is 3 us slower
       * (tools/probe/r4_poisson_plan_ab.py); wider panels keep the quarter */
      const double thr = ts <= 4 ? 0.5 : 0.25;
      if (want > 0 || ext_bytes < thr * 10.0 * o->sell_entries) { o->plan_ts = ts; return 0; }
    }
    free_plan(o); /* too many rows to stage, or not worth it: use the general kernel */
  }
  int win_cap = 256;
  int blk_rows = spmm_block_rows(m, 256);
  if (blk_rows < 64) blk_rows = 64;
  blk_rows &= ~63;
  int nslices = 0;
  for (int p = in->part0; p < in->part1; ++p) nslices += (in->rowPos[p + 1] - in->rowPos[p] + 63) / 64;
  long long* sl_off = (long long*)malloc(((size_t)nslices + 1) * sizeof(long long));
  int* sl_len = (int*)malloc((nslices ? nslices : 1) * sizeof(int));
  int* sl_row0 = (int*)malloc((nslices ? nslices : 1) * sizeof(int));
  int* sl_nrows = (int*)malloc((nslices ? nslices : 1) * sizeof(int));
  int* sl_part = (int*)malloc((nslices ? nslices : 1) * sizeof(int));
  int s = 0;
  sl_off[0] = 0;
  for (int p = in->part0; p < in->part1; ++p) {
    int pr0 = in->rowPos[p] - in->row_off, pr1 = in->rowPos[p + 1] - in->row_off;
    for (int r = pr0; r < pr1; r += 64, ++s) {
      int nr = pr1 - r < 64 ? pr1 - r : 64, len = 0;
      for (int i = 0; i < nr; ++i) { int l = rowptr[r + i + 1] - rowptr[r + i]; if (l > len) len = l; }
      sl_len[s] = len; sl_row0[s] = r; sl_nrows[s] = nr; sl_part[s] = p;
      sl_off[s + 1] = sl_off[s] + (long long)len * 64;
    }
  }
  size_t tot = (size_t)sl_off[nslices];
  o->sell_entries = (double)tot;
  int* scol = (int*)malloc((tot + 64) * sizeof(int));
  double* sval = (double*)calloc(tot + 64, sizeof(double));
  if (!scol || !sval) return PA_FAIL("out of host memory for %zu SELL entries", tot);
  for (int q = 0; q < nslices; ++q) {
    int r = sl_row0[q], nr = sl_nrows[q], len = sl_len[q];
    int* c = scol + sl_off[q];
    double* v = sval + sl_off[q];
    for (int i = 0; i < 64; ++i) {
      int row = i < nr ? r + i : r;          /* unused lanes mirror the first row */
      int l = i < nr ? rowptr[row + 1] - rowptr[row] : 0;
      for (int k = 0; k < len; ++k) {
        if (k < l) { c[(size_t)k * 64 + i] = colind[rowptr[row] + k]; v[(size_t)k * 64 + i] = val[rowptr[row] + k]; }
        else { c[(size_t)k * 64 + i] = row; v[(size_t)k * 64 + i] = 0.0; } /* padding: 0 * x[row] */
      }
    }
  }
  for (size_t k = tot; k < tot + 64; ++k) scol[k] = 0;
  /* blocks */
  int cap_blocks = nslices > 0 ? nslices : 1, nblk = 0;
  int* blk_slice = (int*)malloc(((size_t)cap_blocks + 1) * sizeof(int));
  int* blk_win = (int*)malloc((size_t)2 * cap_blocks * sizeof(int));
  char* needs_halo = (char*)malloc(cap_blocks);
  int q = 0, max_win = 0;
  while (q < nslices) {
    int p = sl_part[q], q1 = q, rows = 0;
    while (q1 < nslices && sl_part[q1] == p && rows + 64 <= blk_rows) { rows += 64; ++q1; }
    int pr0 = in->rowPos[p] - in->row_off, pr1 = in->rowPos[p + 1] - in->row_off;
    int r0 = sl_row0[q], r1 = sl_row0[q1 - 1] + sl_nrows[q1 - 1];
    int w0, w1;
    if (win_cap <= 0) { w0 = w1 = r0; }
    else if (pr1 - pr0 <= win_cap) { w0 = pr0; w1 = pr1; }
    else {
      int c = (r0 + r1) / 2;
      w0 = c - win_cap / 2;
      if (w0 < pr0) w0 = pr0;
      w1 = w0 + win_cap;
      if (w1 > pr1) { w1 = pr1; w0 = w1 - win_cap; }
    }
    if (w1 - w0 > max_win) max_win = w1 - w0;
    char h = 0;
    for (int k = rowptr[r0]; k < rowptr[r1] && !h; ++k) h = colind[k] >= m;
    blk_slice[nblk] = q; blk_win[2 * nblk] = w0; blk_win[2 * nblk + 1] = w1; needs_halo[nblk] = h;
    ++nblk;
    q = q1;
  }
  blk_slice[nblk] = nslices;
  int* order = (int*)malloc((nblk > 0 ? nblk : 1) * sizeof(int));
  int ni = 0;
  for (int b = 0; b < nblk; ++b) if (!needs_halo[b]) order[ni++] = b;
  int k2 = ni;
  for (int b = 0; b < nblk; ++b) if (needs_halo[b]) order[k2++] = b;
  o->d_sl_off = (long long*)pa_rt_malloc(((size_t)nslices + 1) * sizeof(long long));
  o->d_sl_len = (int*)pa_rt_malloc((nslices ? nslices : 1) * sizeof(int));
  o->d_sl_row0 = (int*)pa_rt_malloc((nslices ? nslices : 1) * sizeof(int));
  o->d_sl_nrows = (int*)pa_rt_malloc((nslices ? nslices : 1) * sizeof(int));
  o->d_col = (int*)pa_rt_malloc((tot + 64) * sizeof(int));
  o->d_val = (double*)pa_rt_malloc((tot + 64) * sizeof(double));
  o->d_blk_slice = (int*)pa_rt_malloc(((size_t)nblk + 1) * sizeof(int));
  o->d_blk_win = (int*)pa_rt_malloc((size_t)2 * (nblk > 0 ? nblk : 1) * sizeof(int));
  o->d_order = (int*)pa_rt_malloc((nblk > 0 ? nblk : 1) * sizeof(int));
  int rc = (!o->d_sl_off || !o->d_sl_len || !o->d_sl_row0 || !o->d_sl_nrows || !o->d_col || !o->d_val ||
            !o->d_blk_slice || !o->d_blk_win || !o->d_order);
  rc = rc || pa_rt_h2d(o->d_sl_off, sl_off, ((size_t)nslices + 1) * sizeof(long long));
  rc = rc || pa_rt_h2d(o->d_sl_len, sl_len, nslices * sizeof(int));
  rc = rc || pa_rt_h2d(o->d_sl_row0, sl_row0, nslices * sizeof(int));
  rc = rc || pa_rt_h2d(o->d_sl_nrows, sl_nrows, nslices * sizeof(int));
  rc = rc || pa_rt_h2d(o->d_col, scol, (tot + 64) * sizeof(int));
  rc = rc || pa_rt_h2d(o->d_val, sval, (tot + 64) * sizeof(double));
  rc = rc || pa_rt_h2d(o->d_blk_slice, blk_slice, ((size_t)nblk + 1) * sizeof(int));
  rc = rc || pa_rt_h2d(o->d_blk_win, blk_win, (size_t)2 * nblk * sizeof(int));
  rc = rc || pa_rt_h2d(o->d_order, order, nblk * sizeof(int));
  free(sl_off); free(sl_len); free(sl_row0); free(sl_nrows); free(sl_part); free(scol); free(sval);
  free(blk_slice); free(blk_win); free(needs_halo); free(order);
  if (rc) return PA_FAIL("uploading the SpMM plan failed: %s", pa_rt_error());
  pa_spmm_plan_t* pl = &o->plan;
  pl->m = m; pl->nslices = nslices; pl->sl_off = o->d_sl_off; pl->sl_len = o->d_sl_len;
  pl->sl_row0 = o->d_sl_row0; pl->sl_nrows = o->d_sl_nrows; pl->col = o->d_col; pl->val = o->d_val;
  pl->nblk = nblk; pl->blk_slice = o->d_blk_slice; pl->blk_win = o->d_blk_win; pl->order = o->d_order;
  pl->n_interior = ni; pl->win_cap = max_win;
  o->stream_bytes = 12.0 * (double)tot;
  o->plan_ts = ts;
  return 0;
}

/* Staged plan for panel stride ts.  Returns 0 on success, -1 on error, 1 when
 * some 64-row slice references more rows than fit the LDS staging area. */
static int build_plan_staged(pa_operator_t* o, int ts) {
  const pa_operator_info_t* in = &o->info;
  const int* rowptr = in->A.rowPtr;
  const int* colind = o->lcol;
  const double* val = in->A.val;
  int m = in->m, ncols = m + in->halo;
  /* LDS budget of a block: 32 KiB at ts <= 4, 64 KiB at ts = 8 (two workgroups per CU) */
  int cap_rows = (ts <= 4 ? 32768 : 49152) / (ts * 8);
  if (cap_rows > 65535) cap_rows = 65535;
  /* 8-column panels: 192 rows and 48 KiB of staging (three workgroups per CU) measured 7 % faster */
  int blk_rows = spmm_block_rows(m, ts <= 4 ? 256 : 192);
  if (blk_rows < 64) blk_rows = 64;
  blk_rows &= ~63;
  if (blk_rows > cap_rows) blk_rows = cap_rows & ~63;
  if (blk_rows < 64) return 1;
  int nslices = 0;
  for (int p = in->part0; p < in->part1; ++p) nslices += (in->rowPos[p + 1] - in->rowPos[p] + 63) / 64;
  long long* sl_off = (long long*)malloc(((size_t)nslices + 1) * sizeof(long long));
  int* sl_len = (int*)malloc((nslices ? nslices : 1) * sizeof(int));
  int* sl_row0 = (int*)malloc((nslices ? nslices : 1) * sizeof(int));
  int* sl_nrows = (int*)malloc((nslices ? nslices : 1) * sizeof(int));
  int* sl_part = (int*)malloc((nslices ? nslices : 1) * sizeof(int));
  int s = 0;
  sl_off[0] = 0;
  for (int p = in->part0; p < in->part1; ++p) {
    int pr0 = in->rowPos[p] - in->row_off, pr1 = in->rowPos[p + 1] - in->row_off;
    for (int r = pr0; r < pr1; r += 64, ++s) {
      int nr = pr1 - r < 64 ? pr1 - r : 64, len = 0;
      for (int i = 0; i < nr; ++i) { int l = rowptr[r + i + 1] - rowptr[r + i]; if (l > len) len = l; }
      sl_len[s] = len; sl_row0[s] = r; sl_nrows[s] = nr; sl_part[s] = p;
      sl_off[s + 1] = sl_off[s] + (long long)len * 64;
    }
  }
  size_t tot = (size_t)sl_off[nslices];
  /* blocks: runs of slices of one subdomain; shrink until everything they touch fits */
  int* blk_slice = (int*)malloc(((size_t)nslices + 1) * sizeof(int));
  int* blk_ext_off = (int*)malloc(((size_t)nslices + 1) * sizeof(int));
  char* needs_halo = (char*)malloc(nslices ? nslices : 1);
  int* stamp = (int*)calloc(ncols ? ncols : 1, sizeof(int));   /* block id + 1 that last saw the column */
  int* slot_of = (int*)malloc((ncols ? ncols : 1) * sizeof(int));
  size_t ext_cap = 1024, next_tot = 0;
  int* ext_rows = (int*)malloc(ext_cap * sizeof(int));
  unsigned short* c16 = (unsigned short*)malloc((tot + 64) * sizeof(unsigned short));
  double* sval = (double*)calloc(tot + 64, sizeof(double));
  int nblk = 0, q = 0, max_stage = 0, overflow = 0, gen = 0;
  (void)sl_part;
  if (!c16 || !sval || !stamp || !slot_of) { overflow = -1; }
  while (q < nslices && !overflow) {
    /* staged blocks may span consecutive subdomains: fewer external rows per row */
    int nsl = 0;
    while (q + nsl < nslices && (nsl + 1) * 64 <= blk_rows) ++nsl;
    for (;;) {
      int r0 = sl_row0[q], r1 = sl_row0[q + nsl - 1] + sl_nrows[q + nsl - 1];
      int nown = r1 - r0, next = 0;
      size_t mark0 = next_tot;
      char h = 0;
      ++gen;
      for (int k = rowptr[r0]; k < rowptr[r1]; ++k) {
        int c = colind[k];
        if (c >= r0 && c < r1) continue;
        if (stamp[c] != gen) {
          stamp[c] = gen;
          if (next_tot == ext_cap) { ext_cap *= 2; ext_rows = (int*)realloc(ext_rows, ext_cap * sizeof(int)); }
          ext_rows[next_tot++] = c; ++next;
          if (c >= m) h = 1;
        }
      }
      if (nown + next > cap_rows) {
        next_tot = mark0;
        if (nsl == 1) { overflow = 1; break; }
        nsl = (nsl + 1) / 2;
        continue;
      }
      /* external rows in ascending order (neighbouring rows end up adjacent in LDS and in L2) */
      int* er = ext_rows + mark0;
      for (int a = 1; a < next; ++a) { int v = er[a], b2 = a; while (b2 > 0 && er[b2 - 1] > v) { er[b2] = er[b2 - 1]; --b2; } er[b2] = v; }
      for (int a = 0; a < next; ++a) slot_of[er[a]] = nown + a;
      for (int sq = q; sq < q + nsl; ++sq) {
        int r = sl_row0[sq], nr = sl_nrows[sq], len = sl_len[sq];
        unsigned short* cc = c16 + sl_off[sq];
        double* vv = sval + sl_off[sq];
        for (int i = 0; i < 64; ++i) {
          int row = i < nr ? r + i : r;
          int l = i < nr ? rowptr[row + 1] - rowptr[row] : 0;
          for (int k = 0; k < len; ++k) {
            if (k < l) {
              int c = colind[rowptr[row] + k];
              cc[(size_t)k * 64 + i] = (unsigned short)((c >= r0 && c < r1) ? c - r0 : slot_of[c]);
              vv[(size_t)k * 64 + i] = val[rowptr[row] + k];
            } else { cc[(size_t)k * 64 + i] = (unsigned short)(row - r0); vv[(size_t)k * 64 + i] = 0.0; }
          }
        }
      }
      if (nown + next > max_stage) max_stage = nown + next;
      blk_slice[nblk] = q; blk_ext_off[nblk] = (int)mark0; needs_halo[nblk] = h;
      ++nblk;
      q += nsl;
      break;
    }
  }
  int rc = overflow;
  if (!rc) {
    blk_slice[nblk] = nslices; blk_ext_off[nblk] = (int)next_tot;
    for (size_t k = tot; k < tot + 64; ++k) c16[k] = 0;
    int* order = (int*)malloc((nblk > 0 ? nblk : 1) * sizeof(int));
    int ni = 0;
    for (int b = 0; b < nblk; ++b) if (!needs_halo[b]) order[ni++] = b;
    int k2 = ni;
    for (int b = 0; b < nblk; ++b) if (needs_halo[b]) order[k2++] = b;
    o->d_sl_off = (long long*)pa_rt_malloc(((size_t)nslices + 1) * sizeof(long long));
    o->d_sl_len = (int*)pa_rt_malloc((nslices ? nslices : 1) * sizeof(int));
    o->d_sl_row0 = (int*)pa_rt_malloc((nslices ? nslices : 1) * sizeof(int));
    o->d_sl_nrows = (int*)pa_rt_malloc((nslices ? nslices : 1) * sizeof(int));
    o->d_col16 = (unsigned short*)pa_rt_malloc((tot + 64) * sizeof(unsigned short));
    o->d_val = (double*)pa_rt_malloc((tot + 64) * sizeof(double));
    o->d_blk_slice = (int*)pa_rt_malloc(((size_t)nblk + 1) * sizeof(int));
    o->d_blk_ext_off = (int*)pa_rt_malloc(((size_t)nblk + 1) * sizeof(int));
    o->d_ext_rows = (int*)pa_rt_malloc((next_tot + 1) * sizeof(int));   /* one spare entry: k_spmm_runs reads ids unconditionally */
    o->d_order = (int*)pa_rt_malloc((nblk > 0 ? nblk : 1) * sizeof(int));
    int bad = (!o->d_sl_off || !o->d_sl_len || !o->d_sl_row0 || !o->d_sl_nrows || !o->d_col16 || !o->d_val ||
               !o->d_blk_slice || !o->d_blk_ext_off || !o->d_ext_rows || !o->d_order);
    bad = bad || pa_rt_h2d(o->d_sl_off, sl_off, ((size_t)nslices + 1) * sizeof(long long));
    bad = bad || pa_rt_h2d(o->d_sl_len, sl_len, nslices * sizeof(int));
    bad = bad || pa_rt_h2d(o->d_sl_row0, sl_row0, nslices * sizeof(int));
    bad = bad || pa_rt_h2d(o->d_sl_nrows, sl_nrows, nslices * sizeof(int));
    bad = bad || pa_rt_h2d(o->d_col16, c16, (tot + 64) * sizeof(unsigned short));
    bad = bad || pa_rt_h2d(o->d_val, sval, (tot + 64) * sizeof(double));
    bad = bad || pa_rt_h2d(o->d_blk_slice, blk_slice, ((size_t)nblk + 1) * sizeof(int));
    bad = bad || pa_rt_h2d(o->d_blk_ext_off, blk_ext_off, ((size_t)nblk + 1) * sizeof(int));
    bad = bad || pa_rt_h2d(o->d_ext_rows, ext_rows, next_tot * sizeof(int));
    bad = bad || pa_rt_h2d(o->d_order, order, nblk * sizeof(int));
    free(order);
    if (bad) { PA_FAIL("uploading the SpMM plan failed: %s", pa_rt_error()); rc = -1; }
    else {
      pa_spmm_plan_t* pl = &o->plan;
      pl->m = m; pl->nslices = nslices; pl->sl_off = o->d_sl_off; pl->sl_len = o->d_sl_len;
      pl->sl_row0 = o->d_sl_row0; pl->sl_nrows = o->d_sl_nrows; pl->val = o->d_val;
      pl->nblk = nblk; pl->blk_slice = o->d_blk_slice; pl->order = o->d_order; pl->n_interior = ni;
      pl->staged = 1; pl->col16 = o->d_col16; pl->blk_ext_off = o->d_blk_ext_off; pl->ext_rows = o->d_ext_rows;
      pl->stage_cap = max_stage;
      o->sell_entries = (double)tot;
      o->stream_bytes = 10.0 * (double)tot + 4.0 * (double)next_tot;
    }
  } else if (rc < 0) {
    PA_FAIL("out of host memory for the SpMM plan");
  }
  free(sl_off); free(sl_len); free(sl_row0); free(sl_nrows); free(sl_part);
  free(blk_slice); free(blk_ext_off); free(needs_halo); free(stamp); free(slot_of); free(ext_rows);
  free(c16); free(sval);
  return rc;
}

/* -------------------------------------------------------------- build ---- */
static int g_plan_only = 0;
/* Host-side planning without touching the GPU (sharding, halo lists): what the
 * multi-process CPU tests exercise.  BlockOperator is unavailable in this mode. */
void preAlps_hip_plan_only(int on) { g_plan_only = on ? 1 : 0; }

/* ---- the build, in three steps ------------------------------------------------------------
 * (1) order_and_scale: what needs the whole matrix -- the diagonal check, the scaling vector of
 *     SymRACScaling and the ordering that groups rows part by part (rank 0 only when the set-up is
 *     distributed);
 * (2) build_panel: this process's row panel from its raw rows (scaled, renumbered, sorted), the
 *     halo slots and who owns them;
 * (3) the send lists: from the whole matrix when every process holds it, else by asking the
 *     owners (pa_mpi_swap_lists);  then the upload. */
#define TRACE_DECL double t_tr = pa_wtime(); const int tr = getenv("PREALPS_SETUP_TRACE") != NULL
#define TRACE(what) do { if (tr) { double n_ = pa_wtime(); fprintf(stderr, "[setup] %-28s %.3f s\n", what, n_ - t_tr); t_tr = n_; } } while (0)

static int order_and_scale(int N, const int* rowPtr, const int* colInd, const double* val, int nparts,
                           const int* part, int scale, pa_operator_info_t* in, double** d_out, int** iperm_out) {
  TRACE_DECL;
  {
    int bad_row = -1;
#pragma omp parallel for num_threads(pa_host_threads()) schedule(static)
    for (int i = 0; i < N; ++i) {
      int seen = 0;
      for (int k = rowPtr[i]; k < rowPtr[i + 1] && !seen; ++k) seen = colInd[k] == i;
      if (!seen) {
#pragma omp critical
        { if (bad_row < 0 || i < bad_row) bad_row = i; }
      }
    }
    if (bad_row >= 0) return PA_FAIL("Diagonal is not set correctly (row %d)", bad_row);
  }
  TRACE("diagonal check");
  /* SymRACScaling: d_i = 1/sqrt(max_j |a_ij|) */
  double* d = NULL;
  if (scale) {
    d = (double*)malloc((size_t)N * sizeof(double));
    if (!d) return PA_FAIL("out of host memory");
    int zero_row = 0;
#pragma omp parallel for num_threads(pa_host_threads()) schedule(static) reduction(|| : zero_row)
    for (int i = 0; i < N; ++i) {
      double mx = 0.0;
      for (int k = rowPtr[i]; k < rowPtr[i + 1]; ++k) { double a = fabs(val[k]); if (a > mx) mx = a; }
      if (mx == 0.0) zero_row = 1;
      d[i] = mx > 0.0 ? sqrt(1.0 / mx) : 0.0;
    }
    if (zero_row) { free(d); return PA_FAIL("Impossible to scale the matrix, rcmin=0"); }
  }
  TRACE("scaling");
  /* ordering: rows grouped part by part, original order inside a part */
  in->N = N; in->nparts = nparts;
  in->rowPos = (int*)calloc((size_t)nparts + 1, sizeof(int));
  in->perm = (int*)malloc((size_t)N * sizeof(int));
  int* iperm = (int*)malloc((size_t)N * sizeof(int));
  if (!in->rowPos || !in->perm || !iperm) { free(d); free(iperm); return PA_FAIL("out of host memory"); }
  for (int i = 0; i < N; ++i) {
    int p = part ? part[i] : (int)(((long long)i * nparts) / N);
    if (p < 0 || p >= nparts) { free(d); free(iperm); return PA_FAIL("partition entry %d of row %d out of range", p, i); }
    in->rowPos[p + 1]++;
  }
  for (int p = 0; p < nparts; ++p) {
    if (in->rowPos[p + 1] == 0) { free(d); free(iperm); return PA_FAIL("part %d is empty", p); }
    in->rowPos[p + 1] += in->rowPos[p];
  }
  {
    int* fill = (int*)malloc((size_t)nparts * sizeof(int));
    memcpy(fill, in->rowPos, (size_t)nparts * sizeof(int));
    for (int i = 0; i < N; ++i) {
      int p = part ? part[i] : (int)(((long long)i * nparts) / N);
      in->perm[fill[p]] = i;
      iperm[i] = fill[p]++;
    }
    free(fill);
  }
  TRACE("ordering");
  *d_out = d; *iperm_out = iperm;
  return 0;
}

/* Rows [row_off, row_off + m) of the permuted, scaled matrix from their raw form: row i of the panel
 * is row (whole ? perm[row_off + i] : i) of (rp, ci, v), columns in the ORIGINAL numbering.  Leaves the
 * panel (global permuted column ids, sorted), the halo slots (halo_cols ascending, lcol) and the
 * receive counts per process in `o`; `mark_out` (N ints: halo slot + 1) is handed to the caller. */
static int build_panel(pa_operator_t* o, const int* rp, const int* ci, const double* v, int whole,
                       const double* d, const int* iperm, long long nnz_global, int** recv_by_proc_out) {
  pa_operator_info_t* in = &o->info;
  TRACE_DECL;
  const int N = in->N, nparts = in->nparts, rank = pa_world_rank(), size = pa_world_size();
  in->part0 = (int)((long long)rank * nparts / size);
  in->part1 = (int)((long long)(rank + 1) * nparts / size);
  in->row_off = in->rowPos[in->part0];
  int m = in->rowPos[in->part1] - in->row_off;
  in->m = m;
#define SRC(i) (whole ? in->perm[in->row_off + (i)] : (i))
  size_t lnnz = 0;
  for (int i = 0; i < m; ++i) { int r = SRC(i); lnnz += (size_t)(rp[r + 1] - rp[r]); }
  if (lnnz > 2147483000u) return PA_FAIL("local panel has too many nonzeros for int32 indices");
  CPLM_Mat_CSR_t* A = &in->A;
  A->rowPtr = (int*)malloc((size_t)(m + 1) * sizeof(int));
  A->colInd = (int*)big_alloc((lnnz ? lnnz : 1) * sizeof(int));
  A->val = (double*)big_alloc((lnnz ? lnnz : 1) * sizeof(double));
  if (!A->rowPtr || !A->colInd || !A->val) return PA_FAIL("out of host memory for the row panel (%zu entries)", lnnz);
  A->rowPtr[0] = 0;
  {
    int maxlen = 0;
    for (int i = 0; i < m; ++i) {
      int r = SRC(i); int l = rp[r + 1] - rp[r];
      if (l > maxlen) maxlen = l;
      A->rowPtr[i + 1] = A->rowPtr[i] + l;
    }
    /* rows are independent: scale, renumber the columns and sort each one (threads as available) */
    int dup_row = -1;   /* (benign race: any offending row will do for the message) */
#pragma omp parallel num_threads(pa_host_threads())
    {
      cv_t* buf = (cv_t*)malloc((maxlen ? maxlen : 1) * sizeof(cv_t));
#pragma omp for schedule(static)
      for (int i = 0; i < m; ++i) {
        int old = in->perm[in->row_off + i], r = whole ? old : i;
        int l = 0, sorted = 1;
        for (int k = rp[r]; k < rp[r + 1]; ++k, ++l) {
          buf[l].c = iperm[ci[k]];
          buf[l].v = d ? d[old] * v[k] * d[ci[k]] : v[k];
          if (l > 0 && buf[l].c < buf[l - 1].c) sorted = 0;
        }
        if (!sorted) qsort(buf, l, sizeof(cv_t), cmp_cv);
        for (int q = 1; q < l; ++q) if (buf[q].c == buf[q - 1].c) dup_row = old;
        int base = A->rowPtr[i];
        for (int q = 0; q < l; ++q) { A->colInd[base + q] = buf[q].c; A->val[base + q] = buf[q].v; }
      }
      free(buf);
    }
    if (dup_row >= 0)
      return PA_FAIL("row %d holds the same column twice: sum duplicate entries before building the operator", dup_row);
  }
#undef SRC
  TRACE("permute + sort rows");
  A->info.M = N; A->info.N = N; A->info.nnz = (int)(nnz_global > 2147483647LL ? 2147483647LL : nnz_global);
  A->info.m = m; A->info.n = N;
  A->info.lnnz = (int)lnnz; A->info.blockSize = 1; A->info.format = FORMAT_CSR;
  A->info.structure = UNSYMMETRIC;
  o->lnnz = (int)lnnz;
  /* halo: off-process columns, grouped by owner (ascending global index) */
  int lo = in->row_off, hi = in->row_off + m;
  int* mark = (int*)calloc((size_t)N, sizeof(int)); /* halo slot + 1 */
  if (!mark) return PA_FAIL("out of host memory");
  int halo = 0;
  for (size_t k = 0; k < lnnz; ++k) { int c = A->colInd[k]; if ((c < lo || c >= hi) && !mark[c]) { mark[c] = 1; ++halo; } }
  int* halo_cols = (int*)malloc((halo ? halo : 1) * sizeof(int));
  { int q = 0; for (int c = 0; c < N; ++c) if (mark[c]) { halo_cols[q] = c; mark[c] = ++q; } }
  in->halo = halo;
  TRACE("halo marks");
  o->peers = (int*)malloc((size > 0 ? size : 1) * sizeof(int));
  o->recv_rows = (int*)calloc(size, sizeof(int));
  o->send_rows = (int*)calloc(size, sizeof(int));
  o->send_cnt = (int*)calloc(size, sizeof(int));
  o->recv_cnt = (int*)calloc(size, sizeof(int));
  int* recv_by_proc = (int*)calloc(size, sizeof(int));
  for (int q = 0; q < halo; ++q)
    recv_by_proc[owner_of_part(part_of_row(in->rowPos, nparts, halo_cols[q]), nparts, size)]++;
  /* device CSR with local column ids */
  int* lcol = (int*)big_alloc((lnnz + 8) * sizeof(int));
  if (!lcol) { free(mark); free(recv_by_proc); return PA_FAIL("out of host memory"); }
#pragma omp parallel for num_threads(pa_host_threads()) schedule(static)
  for (long long k = 0; k < (long long)lnnz; ++k) { int c = A->colInd[k]; lcol[k] = (c >= lo && c < hi) ? c - lo : m + mark[c] - 1; }
  for (size_t k = lnnz; k < lnnz + 8; ++k) lcol[k] = 0;
  free(mark);
  o->halo_cols = halo_cols;
  o->lcol = lcol;
  TRACE("local columns");
  *recv_by_proc_out = recv_by_proc;
  return 0;
}

/* peers / send_rows / recv_rows / send_idx from need-lists: asked[] = for every process g (in rank
 * order) the local rows of ours it reads, asked_cnt[g] of them, ascending. */
static int finish_peers(pa_operator_t* o, const int* recv_by_proc, int* asked_local, const int* asked_cnt) {
  int size = pa_world_size();
  o->nsend = 0; o->npeers = 0;
  for (int g = 0; g < size; ++g) {
    if (asked_cnt[g] || recv_by_proc[g]) {
      o->peers[o->npeers] = g; o->recv_rows[o->npeers] = recv_by_proc[g]; o->send_rows[o->npeers] = asked_cnt[g];
      o->npeers++;
    }
    o->nsend += asked_cnt[g];
  }
  o->send_idx = asked_local;
  return 0;
}

static int upload_operator(pa_operator_t* o, double t_build0) {
  pa_operator_info_t* in = &o->info;
  if (g_plan_only) { in->built = 1; return 0; }
  int rc = 0;
  if (o->nsend > 0) {
    o->d_send_idx = (int*)pa_rt_malloc((size_t)o->nsend * sizeof(int));
    rc = !o->d_send_idx || pa_rt_h2d(o->d_send_idx, o->send_idx, (size_t)o->nsend * sizeof(int));
  }
  if (rc) return PA_FAIL("uploading the operator failed: %s", pa_rt_error());
  in->built = 1;
  g_setup_build_s = pa_wtime() - t_build0;
  return 0;
}

int preAlps_OperatorBuildFromCSR(int N, const int* rowPtr, const int* colInd, const double* val,
                                 int nparts, const int* part, int scale) {
  if (!g_plan_only) PA_REQUIRE_GPU();
  if (g_op.info.built) preAlps_OperatorFree();
  double t_build0 = pa_wtime();
  pa_operator_t* o = &g_op;
  pa_operator_info_t* in = &o->info;
  int rank = pa_world_rank(), size = pa_world_size();
  if (N < 1 || nparts < 1 || nparts > N) return PA_FAIL("invalid sizes N=%d nparts=%d", N, nparts);
  if (nparts < size)
    return PA_FAIL("Each process needs at least one block (nparts = %d < %d = processes)", nparts, size);
  double* d = NULL; int* iperm = NULL; int* recv_by_proc = NULL;
  if (order_and_scale(N, rowPtr, colInd, val, nparts, part, scale, in, &d, &iperm)) return 1;
  if (build_panel(o, rowPtr, colInd, val, 1, d, iperm, (long long)rowPtr[N], &recv_by_proc)) { free(d); free(iperm); return 1; }
  free(d);
  TRACE_DECL;
  int m = in->m, lo = in->row_off, hi = in->row_off + m;
  /* Send lists, exactly: process g receives from us the rows of ours that occur as columns in
   * ITS rows (ascending) -- which is how g numbers its halo slots.  Every process holds the
   * whole matrix here, so this needs no communication and, unlike taking "our rows that touch a
   * column of g", it stays right for a pattern that is not structurally symmetric (a `general`
   * .mtx with explicit zeros dropped on one side). */
  unsigned char* need = NULL;   /* need[g * m + i]: process g reads our local row i */
  if (size > 1) {
    need = (unsigned char*)calloc((size_t)size * (m ? m : 1), 1);
    if (!need) { free(iperm); free(recv_by_proc); return PA_FAIL("out of host memory for the halo plan"); }
#pragma omp parallel for num_threads(pa_host_threads()) schedule(dynamic, 4096)
    for (int i = 0; i < N; ++i) {
      int r = iperm[i];
      if (r >= lo && r < hi) continue;
      unsigned char* ng = need + (size_t)owner_of_part(part_of_row(in->rowPos, nparts, r), nparts, size) * m;
      for (int k = rowPtr[i]; k < rowPtr[i + 1]; ++k) {
        int c = iperm[colInd[k]];
        if (c >= lo && c < hi) ng[c - lo] = 1;   /* (all writers store the same value) */
      }
    }
  }
  free(iperm);
  size_t send_cap = 1024, nsend = 0;
  int* send_idx = (int*)malloc(send_cap * sizeof(int));
  int* asked_cnt = (int*)calloc(size, sizeof(int));
  for (int g = 0; g < size; ++g) {
    if (!need || g == rank) continue;
    for (int i = 0; i < m; ++i) if (need[(size_t)g * m + i]) {
      if (nsend + 1 > send_cap) { send_cap *= 2; send_idx = (int*)realloc(send_idx, send_cap * sizeof(int)); }
      send_idx[nsend++] = i; asked_cnt[g]++;
    }
  }
  free(need);
  finish_peers(o, recv_by_proc, send_idx, asked_cnt);
  free(recv_by_proc); free(asked_cnt);
  TRACE("peer lists");
  return upload_operator(o, t_build0);
}

/* ---- MatrixMarket reader (utils/cplm_light/cplm_matcsr.c:96-243) ----------------------------------
 * coordinate real general|symmetric; 0-based files are detected from the first entry like the reference
 * does; symmetric files are expanded; repeated (i, j) entries are summed (in file order).  The file is
 * mapped and the host threads parse it in line-aligned pieces (an 81 M-nonzero matrix is 1.3 GB of text:
 * minutes through fscanf, seconds this way); values go through strtod, so they are the doubles
 * fscanf("%lf") would have produced. */
typedef struct { int c; double v; } mm_ent_t;

static void mm_sort_row(mm_ent_t* e, int n, mm_ent_t* tmp) {   /* stable, by column */
  if (n <= 96) {
    for (int a = 1; a < n; ++a) { mm_ent_t x = e[a]; int q = a; while (q > 0 && e[q - 1].c > x.c) { e[q] = e[q - 1]; --q; } e[q] = x; }
    return;
  }
  int h = n / 2;
  mm_sort_row(e, h, tmp); mm_sort_row(e + h, n - h, tmp);
  memcpy(tmp, e, (size_t)h * sizeof(mm_ent_t));
  int i = 0, j = h, k = 0;
  while (i < h && j < n) e[k++] = (e[j].c < tmp[i].c) ? e[j++] : tmp[i++];
  while (i < h) e[k++] = tmp[i++];
}

/* one "i j v" line starting at p (p < end); returns the position behind the line, NULL on a bad line */
static const char* mm_parse_line(const char* p, const char* end, int* I, int* J, double* V) {
  long val[2];
  for (int f = 0; f < 2; ++f) {
    while (p < end && (*p == ' ' || *p == '\t')) ++p;
    int neg = 0;
    if (p < end && (*p == '-' || *p == '+')) { neg = *p == '-'; ++p; }
    if (p >= end || *p < '0' || *p > '9') return NULL;
    long x = 0;
    while (p < end && *p >= '0' && *p <= '9') { x = 10 * x + (*p - '0'); ++p; }
    val[f] = neg ? -x : x;
  }
  while (p < end && (*p == ' ' || *p == '\t')) ++p;
  /* the number's characters, NUL terminated for strtod (never reads past the mapping) */
  char buf[64];
  int l = 0;
  while (p + l < end && l < 63 && p[l] != '\n' && p[l] != '\r' && p[l] != ' ' && p[l] != '\t') { buf[l] = p[l]; ++l; }
  buf[l] = 0;
  if (l == 0) return NULL;
  char* stop = NULL;
  *V = strtod(buf, &stop);
  if (stop == buf) return NULL;
  p += l;
  while (p < end && *p != '\n') ++p;
  *I = (int)val[0]; *J = (int)val[1];
  return p < end ? p + 1 : end;
}

static int load_mtx(const char* file, int* N_out, int** rp_out, int** ci_out, double** v_out) {
  FILE* fd = fopen(file, "r");
  if (!fd) return PA_FAIL("Impossible to open the file %s", file);
  char line[1025], banner[64], mtx[64], crd[64], dt[64], sym[64];
  if (!fgets(line, sizeof(line), fd) ||
      sscanf(line, "%63s %63s %63s %63s %63s", banner, mtx, crd, dt, sym) != 5 ||
      strcasecmp(banner, "%%MatrixMarket") || strcasecmp(mtx, "matrix") ||
      strcasecmp(crd, "coordinate") || strcasecmp(dt, "real") ||
      (strcasecmp(sym, "general") && strcasecmp(sym, "symmetric"))) {
    fclose(fd);
    return PA_FAIL("Only sparse real < symmetric | general > matrix are currently supported (%s)", file);
  }
  int is_sym = !strcasecmp(sym, "symmetric");
  do { if (!fgets(line, sizeof(line), fd)) { fclose(fd); return PA_FAIL("truncated file %s", file); } } while (line[0] == '%');
  int M = 0, Nc = 0; long long nz = 0;
  if (sscanf(line, "%d %d %lld", &M, &Nc, &nz) != 3 || M < 1 || Nc != M || nz < 1) {
    fclose(fd);
    return PA_FAIL("[LoadMatrixMarket] Error: Invalid matrix dimensions in %s", file);
  }
  long body = ftell(fd);
  fseek(fd, 0, SEEK_END);
  long fsize = ftell(fd);
  if (body < 0 || fsize <= body) { fclose(fd); return PA_FAIL("truncated file %s", file); }
  const char* map = (const char*)mmap(NULL, (size_t)fsize, PROT_READ, MAP_PRIVATE, fileno(fd), 0);
  fclose(fd);
  if (map == MAP_FAILED) return PA_FAIL("cannot map %s", file);
  (void)madvise((void*)map, (size_t)fsize, MADV_SEQUENTIAL);
  const char* beg = map + body;
  const char* end = map + fsize;
  TRACE_DECL;
  int T = pa_host_threads();
  if ((long long)T > nz / 4096 + 1) T = (int)(nz / 4096 + 1);
  if (T < 1) T = 1;
  /* pieces that start behind a newline */
  const char** cut = (const char**)malloc(((size_t)T + 1) * sizeof(char*));
  long long* first = (long long*)calloc((size_t)T + 1, sizeof(long long));
  int* I = (int*)malloc((size_t)nz * sizeof(int));
  int* J = (int*)malloc((size_t)nz * sizeof(int));
  double* V = (double*)malloc((size_t)nz * sizeof(double));
  if (!cut || !first || !I || !J || !V) { free(cut); free(first); free(I); free(J); free(V); munmap((void*)map, (size_t)fsize); return PA_FAIL("out of host memory for %lld entries", nz); }
  cut[0] = beg; cut[T] = end;
  for (int t = 1; t < T; ++t) {
    const char* p = beg + (size_t)((double)(end - beg) * t / T);
    if (p < cut[t - 1]) p = cut[t - 1];
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    cut[t] = nl ? nl + 1 : end;
  }
  /* lines per piece (blank lines do not count) */
#pragma omp parallel for num_threads(T) schedule(static, 1)
  for (int t = 0; t < T; ++t) {
    long long n = 0;
    const char* p = cut[t];
    while (p < cut[t + 1]) {
      const char* nl = (const char*)memchr(p, '\n', (size_t)(cut[t + 1] - p));
      const char* le = nl ? nl : cut[t + 1];
      const char* q = p;
      while (q < le && (*q == ' ' || *q == '\t' || *q == '\r')) ++q;
      if (q < le && *q != '%') ++n;           /* (blank lines and comment lines inside the body do not count) */
      p = nl ? nl + 1 : cut[t + 1];
    }
    first[t + 1] = n;
  }
  for (int t = 0; t < T; ++t) first[t + 1] += first[t];
  TRACE("  mtx: count lines");
  int bad = first[T] < nz;     /* (extra lines behind the nz announced are ignored, like fscanf would) */
  long long bad_line = -1;
  if (!bad) {
#pragma omp parallel for num_threads(T) schedule(static, 1)
    for (int t = 0; t < T; ++t) {
      long long k = first[t];
      const char* p = cut[t];
      while (p && p < cut[t + 1] && k < nz) {
        const char* q = p;
        while (q < cut[t + 1] && (*q == ' ' || *q == '\t' || *q == '\r')) ++q;
        if (q < cut[t + 1] && *q == '\n') { p = q + 1; continue; }
        if (q >= cut[t + 1]) break;
        if (*q == '%') {                         /* a comment line in the body: skipped */
          const char* nl = (const char*)memchr(q, '\n', (size_t)(cut[t + 1] - q));
          p = nl ? nl + 1 : cut[t + 1];
          continue;
        }
        p = mm_parse_line(q, cut[t + 1], &I[k], &J[k], &V[k]);
        if (!p) {
#pragma omp critical
          { if (bad_line < 0 || k < bad_line) bad_line = k; }
          break;
        }
        ++k;
      }
    }
  }
  munmap((void*)map, (size_t)fsize);
  free(cut);
  TRACE("  mtx: parse");
  if (bad || bad_line >= 0) {
    long long k = bad ? first[T] : bad_line;
    free(first); free(I); free(J); free(V);
    return PA_FAIL("bad entry %lld in %s", k, file);
  }
  free(first);
  int base = (I[0] == 0 || J[0] == 0) ? 0 : 1;
  /* row histograms per thread (thread t owns entries [t nz / T, (t+1) nz / T)): positions that keep
   * the file order inside every row, without atomics */
  int oob = 0;
#pragma omp parallel for num_threads(pa_host_threads()) schedule(static) reduction(|| : oob)
  for (long long k = 0; k < nz; ++k) {
    I[k] -= base; J[k] -= base;
    if (I[k] < 0 || I[k] >= M || J[k] < 0 || J[k] >= M) oob = 1;
  }
  if (oob) { free(I); free(J); free(V); return PA_FAIL("index out of range in %s", file); }
  int TH = pa_host_threads();
  while (TH > 1 && (size_t)TH * (size_t)M * sizeof(int) > ((size_t)1 << 30)) TH /= 2;
  int* hist = (int*)calloc((size_t)TH * (size_t)M, sizeof(int));
  int* rp = (int*)malloc(((size_t)M + 1) * sizeof(int));
  if (!hist || !rp) { free(hist); free(rp); free(I); free(J); free(V); return PA_FAIL("out of host memory"); }
#pragma omp parallel for num_threads(TH) schedule(static, 1)
  for (int t = 0; t < TH; ++t) {
    int* h = hist + (size_t)t * M;
    long long k0 = nz * t / TH, k1 = nz * (t + 1) / TH;
    for (long long k = k0; k < k1; ++k) { h[I[k]]++; if (is_sym && I[k] != J[k]) h[J[k]]++; }
  }
  long long tot = 0;
  for (int i = 0; i < M; ++i) {
    rp[i] = (int)tot;
    for (int t = 0; t < TH; ++t) { int c = hist[(size_t)t * M + i]; hist[(size_t)t * M + i] = (int)tot; tot += c; }
    if (tot > 2147483000LL) { free(hist); free(rp); free(I); free(J); free(V); return PA_FAIL("%s has too many nonzeros for int32 indices", file); }
  }
  rp[M] = (int)tot;
  mm_ent_t* ent = (mm_ent_t*)big_alloc((size_t)(tot ? tot : 1) * sizeof(mm_ent_t));
  if (!ent) { free(hist); free(rp); free(I); free(J); free(V); return PA_FAIL("out of host memory for %lld entries", tot); }
#pragma omp parallel for num_threads(TH) schedule(static, 1)
  for (int t = 0; t < TH; ++t) {
    int* h = hist + (size_t)t * M;
    long long k0 = nz * t / TH, k1 = nz * (t + 1) / TH;
    for (long long k = k0; k < k1; ++k) {
      ent[h[I[k]]].c = J[k]; ent[h[I[k]]++].v = V[k];
      if (is_sym && I[k] != J[k]) { ent[h[J[k]]].c = I[k]; ent[h[J[k]]++].v = V[k]; }
    }
  }
  free(hist); free(I); free(J); free(V);
  TRACE("  mtx: rows (histograms, scatter)");
  /* sort every row by column (stable), sum repeated entries, compact */
  int* len = (int*)malloc((size_t)M * sizeof(int));
  if (!len) { free(ent); free(rp); return PA_FAIL("out of host memory"); }
  int no_mem = 0;        /* a thread could not get its sort buffer: the rows it met stay unsorted, the build fails */
#pragma omp parallel num_threads(pa_host_threads())
  {
    int cap = 256;
    mm_ent_t* tmp = (mm_ent_t*)malloc((size_t)cap * sizeof(mm_ent_t));
#pragma omp for schedule(dynamic, 1024)
    for (int i = 0; i < M; ++i) {
      int k0 = rp[i], n = rp[i + 1] - k0;
      if (n / 2 + 1 > cap || !tmp) { cap = n > cap ? n : cap; free(tmp); tmp = (mm_ent_t*)malloc((size_t)cap * sizeof(mm_ent_t)); }
      int sorted = 1;
      for (int k = 1; k < n && sorted; ++k) sorted = ent[k0 + k - 1].c <= ent[k0 + k].c;
      if (!sorted && !tmp) {
#pragma omp atomic write
        no_mem = 1;
      }
      if (!sorted && tmp) mm_sort_row(ent + k0, n, tmp);
      int out = 0;
      for (int k = 0; k < n; ++k) {
        if (out > 0 && ent[k0 + k].c == ent[k0 + out - 1].c) ent[k0 + out - 1].v += ent[k0 + k].v;
        else ent[k0 + out++] = ent[k0 + k];
      }
      len[i] = out;
    }
    free(tmp);
  }
  if (no_mem) { free(len); free(ent); free(rp); return PA_FAIL("out of host memory while sorting the rows of %s", file); }
  int* rp2 = (int*)malloc(((size_t)M + 1) * sizeof(int));
  long long out = 0;
  for (int i = 0; rp2 && i < M; ++i) { rp2[i] = (int)out; out += len[i]; }
  int* ci = (int*)big_alloc((size_t)(out ? out : 1) * sizeof(int));
  double* vv = (double*)big_alloc((size_t)(out ? out : 1) * sizeof(double));
  if (!rp2 || !ci || !vv) { free(rp2); free(ci); free(vv); free(len); free(ent); free(rp); return PA_FAIL("out of host memory"); }
  rp2[M] = (int)out;
#pragma omp parallel for num_threads(pa_host_threads()) schedule(static)
  for (int i = 0; i < M; ++i)
    for (int k = 0; k < len[i]; ++k) { ci[rp2[i] + k] = ent[rp[i] + k].c; vv[rp2[i] + k] = ent[rp[i] + k].v; }
  free(len); free(ent); free(rp);
  TRACE("  mtx: sort + compact");
  *N_out = M; *rp_out = rp2; *ci_out = ci; *v_out = vv;
  return 0;
}

/* The partition vector of the whole matrix, as preAlps_OperatorBuild picks it: from
 * PREALPS_PARTITION_FILE (one part id per line, e.g. METIS output), contiguous row blocks with
 * PREALPS_PARTITION=contiguous (*part_out = NULL), else the library's k-way graph partitioner
 * (partition.c), which stands where the reference calls METIS_PartGraphKway
 * (utils/operator.c:77-97 -> utils/cplm_core/cplm_matcsr_core.c:394-457). */
static int choose_partition(int N, const int* rp, const int* ci, int nparts, int** part_out) {
  int* part = NULL;
  *part_out = NULL;
  const char* pf = getenv("PREALPS_PARTITION_FILE");
  if (pf && *pf) {
    FILE* f = fopen(pf, "r");
    if (!f) return PA_FAIL("Impossible to open the file %s", pf);
    part = (int*)malloc((size_t)N * sizeof(int));
    for (int i = 0; part && i < N; ++i)
      if (fscanf(f, "%d", &part[i]) != 1) { fclose(f); free(part); return PA_FAIL("partition file %s is too short", pf); }
    fclose(f);
    if (!part) return PA_FAIL("out of host memory");
  } else {
    const char* how = getenv("PREALPS_PARTITION");
    if (!(how && !strcmp(how, "contiguous")) && nparts > 1) {
      if (nparts > N) return PA_FAIL("invalid sizes N=%d nparts=%d", N, nparts);
      part = (int*)malloc((size_t)N * sizeof(int));
      if (!part || preAlps_hip_partition_kway(N, rp, ci, nparts, part)) { free(part); return 1; }
    }
  }
  *part_out = part;
  return 0;
}

/* One MPI rank per GPU (the reference's own mode: utils/operator.c:38-134).  Rank 0 reads, scales
 * and partitions the matrix -- with the whole CPU share of the node, the other ranks sleep --
 * broadcasts the ordering and the scaling vector (12 bytes per row) and sends every rank the raw
 * rows of its panel (CPLM_MatCSRGetRowPanel + CPLM_MatCSRSend, utils/operator.c:99-108;
 * utils/cplm_light/cplm_matcsr.c:382-497); each rank then scales, renumbers and sorts its own rows,
 * finds its halo and asks the owners for exactly those rows.  No rank but 0 ever holds more than
 * its panel. */
static int build_distributed(const char* file, int rank, int size) {
  double t_build0 = pa_wtime();
  pa_operator_t* o = &g_op;
  pa_operator_info_t* in = &o->info;
  TRACE_DECL;
  int N = 0; int* rp = NULL; int* ci = NULL; double* v = NULL;
  double* d = NULL; int* iperm = NULL;
  long long hdr[4] = {0, 0, 0, 0};          /* rc, N, nparts, nnz */
  const int abort_mode_rc = 1;
  if (rank == 0) {
    pa_host_solo(1);
    int rc = load_mtx(file, &N, &rp, &ci, &v);
    TRACE("rank 0: read the matrix");
    int nparts = env_int("PREALPS_NPARTS", size);
    int* part = NULL;
    if (!rc && (nparts < size || nparts > N))
      rc = PA_FAIL("Each process needs at least one block (nparts = %d, %d processes, %d rows)", nparts, size, N);
    rc = rc || choose_partition(N, rp, ci, nparts, &part);
    TRACE("rank 0: partition");
    rc = rc || order_and_scale(N, rp, ci, v, nparts, part, 1, in, &d, &iperm);
    free(part);
    pa_host_solo(0);
    hdr[0] = rc ? abort_mode_rc : 0; hdr[1] = N; hdr[2] = nparts; hdr[3] = rc ? 0 : rp[N];
  }
  if (pa_mpi_bcast(hdr, sizeof(hdr), 0)) return 1;
  if (hdr[0]) {
    free(rp); free(ci); free(v); free(d); free(iperm);
    return rank == 0 ? 1 : PA_FAIL("rank 0 could not read or partition %s", file);
  }
  N = (int)hdr[1];
  int nparts = (int)hdr[2];
  if (rank != 0) {
    in->N = N; in->nparts = nparts;
    in->rowPos = (int*)malloc(((size_t)nparts + 1) * sizeof(int));
    in->perm = (int*)malloc((size_t)N * sizeof(int));
    iperm = (int*)malloc((size_t)N * sizeof(int));
    d = (double*)malloc((size_t)N * sizeof(double));
  }
  {
    /* (a rank that cannot hold the vectors must not leave the others inside the broadcasts) */
    const int mine = rank != 0 && (!in->rowPos || !in->perm || !iperm || !d);
    if (pa_mpi_agree(mine)) return mine ? PA_FAIL("out of host memory") : PA_FAIL("another rank ran out of host memory");
  }
  if (pa_mpi_bcast(in->rowPos, ((size_t)nparts + 1) * sizeof(int), 0) || pa_mpi_bcast(in->perm, (size_t)N * sizeof(int), 0) ||
      pa_mpi_bcast(d, (size_t)N * sizeof(double), 0))
    return 1;
  if (rank != 0)
    for (int k = 0; k < N; ++k) iperm[in->perm[k]] = k;
  TRACE("ordering + scaling vector (broadcast)");
  int* recv_by_proc = NULL;
  int prc = 0;
  /* entries of every rank's panel, announced before the rows travel: every buffer of the exchange exists (and
   * the ranks have agreed that it does) before the first message */
  long long* nzs = (long long*)calloc((size_t)size, sizeof(long long));
  if (pa_mpi_agree(nzs == NULL)) { free(nzs); return PA_FAIL("out of host memory"); }
  if (rank == 0) {
    /* raw rows of every other panel: row lengths, column ids, values -- three messages per rank */
    /* buffers for the largest panel, made before the first message: the ranks that wait for their rows
     * hear about a failure here (pa_mpi_agree below) instead of waiting for ever */
    size_t cap = 0, capm = 0;
    for (int g = 1; g < size; ++g) {
      int p0 = (int)((long long)g * nparts / size), p1 = (int)((long long)(g + 1) * nparts / size);
      int r0 = in->rowPos[p0], mg = in->rowPos[p1] - r0;
      size_t nz = 0;
      for (int i = 0; i < mg; ++i) { int old = in->perm[r0 + i]; nz += (size_t)(rp[old + 1] - rp[old]); }
      if (nz > cap) cap = nz;
      if ((size_t)mg > capm) capm = (size_t)mg;
      nzs[g] = (long long)nz;
    }
    if (pa_mpi_bcast(nzs, (size_t)size * sizeof(long long), 0)) { free(nzs); return 1; }
    int* bl = (int*)malloc((capm + 1) * sizeof(int));
    int* bc = (int*)big_alloc((cap + 16) * sizeof(int));
    double* bv = (double*)big_alloc((cap + 16) * sizeof(double));
    if (pa_mpi_agree(!bl || !bc || !bv)) { free(bl); free(bc); free(bv); free(nzs); return PA_FAIL("out of host memory for the row panels"); }
    for (int g = 1; g < size; ++g) {
      int p0 = (int)((long long)g * nparts / size), p1 = (int)((long long)(g + 1) * nparts / size);
      int r0 = in->rowPos[p0], mg = in->rowPos[p1] - r0;
      size_t nz = 0;
      for (int i = 0; i < mg; ++i) { int old = in->perm[r0 + i]; nz += (size_t)(rp[old + 1] - rp[old]); }
      bl[0] = 0;
      for (int i = 0; i < mg; ++i) { int old = in->perm[r0 + i]; bl[i + 1] = bl[i] + (rp[old + 1] - rp[old]); }
#pragma omp parallel for num_threads(pa_host_threads()) schedule(static)
      for (int i = 0; i < mg; ++i) {
        int old = in->perm[r0 + i];
        memcpy(bc + bl[i], ci + rp[old], (size_t)(rp[old + 1] - rp[old]) * sizeof(int));
        memcpy(bv + bl[i], v + rp[old], (size_t)(rp[old + 1] - rp[old]) * sizeof(double));
      }
      if (pa_mpi_send(bl, ((size_t)mg + 1) * sizeof(int), g, 41) || pa_mpi_send(bc, nz * sizeof(int), g, 42) ||
          pa_mpi_send(bv, nz * sizeof(double), g, 43))
        return 1;
    }
    free(bl); free(bc); free(bv);
    TRACE("rank 0: panels sent");
    prc = build_panel(o, rp, ci, v, 1, d, iperm, hdr[3], &recv_by_proc);
    free(rp); free(ci); free(v);
  } else {
    int p0 = (int)((long long)rank * nparts / size), p1 = (int)((long long)(rank + 1) * nparts / size);
    int mg = in->rowPos[p1] - in->rowPos[p0];
    if (pa_mpi_bcast(nzs, (size_t)size * sizeof(long long), 0)) { free(nzs); return PA_FAIL("receiving the panel sizes failed"); }
    size_t nz = (size_t)nzs[rank];
    int* lrp = (int*)malloc(((size_t)mg + 1) * sizeof(int));
    int* lci = (int*)big_alloc((nz ? nz : 1) * sizeof(int));
    double* lv = (double*)big_alloc((nz ? nz : 1) * sizeof(double));
    if (pa_mpi_agree(!lrp || !lci || !lv)) {      /* (pairs with rank 0's) */
      free(lrp); free(lci); free(lv); free(nzs);
      return PA_FAIL("out of host memory for the row panels (%zu entries here)", nz);
    }
    if (pa_mpi_recv(lrp, ((size_t)mg + 1) * sizeof(int), 0, 41) || (size_t)lrp[mg] != nz) return PA_FAIL("receiving the panel failed");
    if (pa_mpi_recv(lci, nz * sizeof(int), 0, 42) || pa_mpi_recv(lv, nz * sizeof(double), 0, 43)) return 1;
    TRACE("panel received");
    int rc = build_panel(o, lrp, lci, lv, 0, d, iperm, hdr[3], &recv_by_proc);
    free(lrp); free(lci); free(lv);
    prc = rc;
  }
  free(nzs);
  if (pa_mpi_agree(prc)) return prc ? 1 : PA_FAIL("another rank could not build its row panel");
  free(d);
  /* send lists: every rank asks the owners for the rows behind its halo slots (halo_cols is
   * ascending, hence grouped by owner); what a rank is asked for, in that order, is what it packs */
  int* asked = NULL;
  int* asked_cnt = (int*)calloc((size_t)size, sizeof(int));
  if (pa_mpi_agree(asked_cnt == NULL)) { free(asked_cnt); return PA_FAIL("out of host memory"); }
  if (pa_mpi_swap_lists(o->halo_cols, recv_by_proc, &asked, asked_cnt)) return 1;
  {
    long long tot = 0;
    for (int g = 0; g < size; ++g) tot += asked_cnt[g];
    for (long long k = 0; k < tot; ++k) {
      int r = asked[k] - in->row_off;
      if (r < 0 || r >= in->m) { prc = PA_FAIL("a neighbour asked for row %d, which this rank does not own", asked[k]); break; }
      asked[k] = r;
    }
  }
  free(iperm);
  if (!prc) finish_peers(o, recv_by_proc, asked, asked_cnt);
  free(recv_by_proc); free(asked_cnt);
  TRACE("peer lists (exchanged)");
  if (!prc) prc = upload_operator(o, t_build0);
  /* every rank leaves with the same answer: the solver's collectives come next */
  if (pa_mpi_agree(prc)) return prc ? 1 : PA_FAIL("another rank could not finish its part of the operator");
  return 0;
}

/* The number of subdomains is PREALPS_NPARTS (default: the process count, as in the
 * reference).  Started by an MPI launcher (the reference driver: MPI_Init, then
 * preAlps_OperatorBuild(file, MPI_COMM_WORLD)), rank and size come from `comm`, the device from
 * the rank's position on its node, the process-group hooks bind to RCCL (or to MPI through the
 * host when ranks share a device) and the set-up is distributed from rank 0: no caller change
 * (mpi_glue.c).  PREALPS_PLAN_ONLY=1: plan the sharding without touching a GPU. */
int preAlps_OperatorBuild(const char* matrixFilename, MPI_Comm comm) {
  size_t len = strlen(matrixFilename);
  if (len < 3 || strcmp(matrixFilename + len - 3, "mtx") != 0)
    return PA_FAIL("Only MatrixMarket (.mtx) files are supported: %s", matrixFilename);
  if (env_int("PREALPS_PLAN_ONLY", 0)) g_plan_only = 1;
  int mrank = 0, msize = 1;
  const int att = pa_mpi_attach(comm, &mrank, &msize);
  if (att < 0) return 1;       /* (an MPI this library cannot speak to: never go on as a lone process) */
  if (att) {
    if (g_op.info.built) preAlps_OperatorFree();
    if (preAlps_hip_set_world(mrank, msize)) return 1;
    if (!g_plan_only && pa_mpi_bind()) return 1;
    return build_distributed(matrixFilename, mrank, msize);
  }
  int N = 0; int* rp = NULL; int* ci = NULL; double* v = NULL;
  TRACE_DECL;
  int rc = load_mtx(matrixFilename, &N, &rp, &ci, &v);
  if (rc) return rc;
  TRACE("read the matrix");
  int nparts = env_int("PREALPS_NPARTS", pa_world_size());
  int* part = NULL;
  if (choose_partition(N, rp, ci, nparts, &part)) { free(rp); free(ci); free(v); return 1; }
  rc = preAlps_OperatorBuildFromCSR(N, rp, ci, v, nparts, part, 1);
  free(part); free(rp); free(ci); free(v);
  return rc;
}

/* -------------------------------------------------------------- getters ---- */
int preAlps_OperatorGetSizes(int* M, int* m) {
  if (!g_op.info.built) return PA_FAIL("operator not built");
  *M = g_op.info.N; *m = g_op.info.m;
  return 0;
}
int preAlps_OperatorGetA(CPLM_Mat_CSR_t* A) {
  if (!g_op.info.built) return PA_FAIL("operator not built");
  *A = g_op.info.A; /* shallow view, library-owned (operator.c:353-359) */
  return 0;
}
int preAlps_OperatorGetRowPosPtr(int** rowPos, int* sizeRowPos) {
  if (!g_op.info.built) return PA_FAIL("operator not built");
  *rowPos = g_op.info.rowPos; *sizeRowPos = g_op.info.nparts + 1;
  return 0;
}
/* The reference's colPos is an O(m * P) table that only its own SpMM and
 * GetDiagBlock consume; here the row blocks carry that information, so the
 * getter hands back a one-entry placeholder that BlockJacobiCreate ignores. */
int preAlps_OperatorGetColPosPtr(int** colPos, int* sizeColPos) {
  if (!g_op.info.built) return PA_FAIL("operator not built");
  if (!g_op.colPos_dummy) g_op.colPos_dummy = (int*)calloc(1, sizeof(int));
  *colPos = g_op.colPos_dummy; *sizeColPos = 1;
  return 0;
}
int preAlps_OperatorGetDepPtr(int** dep, int* sizeDep) {
  if (!g_op.info.built) return PA_FAIL("operator not built");
  *dep = g_op.peers; *sizeDep = g_op.npeers;
  return 0;
}
/* The halo plan of this process: peers[i] receives send_rows[i] of our rows
 * (local indices in send_idx, peer after peer) and owns recv_rows[i] of our
 * halo slots (global rows in halo_cols, ascending, peer after peer). */
int preAlps_OperatorGetHaloPlan(int* npeers, int** peers, int** send_rows, int** recv_rows,
                                int** send_idx, int* nsend, int** halo_cols, int* nhalo) {
  if (!g_op.info.built) return PA_FAIL("operator not built");
  *npeers = g_op.npeers; *peers = g_op.peers; *send_rows = g_op.send_rows; *recv_rows = g_op.recv_rows;
  *send_idx = g_op.send_idx; *nsend = g_op.nsend; *halo_cols = g_op.halo_cols; *nhalo = g_op.info.halo;
  return 0;
}
int preAlps_OperatorGetPermPtr(int** perm, int* n) {
  if (!g_op.info.built) return PA_FAIL("operator not built");
  *perm = g_op.info.perm; *n = g_op.info.N;
  return 0;
}
int preAlps_hip_nparts(void) { return g_op.info.built ? g_op.info.nparts : 0; }

void preAlps_OperatorPrint(int rank) {
  const pa_operator_info_t* in = &g_op.info;
  if (!in->built || rank != pa_world_rank()) return;
  printf("[%d] operator: N = %d, parts = %d (own %d..%d), local rows = %d, local nnz = %d, halo rows = %d, "
         "neighbours = %d, SpMM row blocks = %d (%d interior)\n",
         rank, in->N, in->nparts, in->part0, in->part1 - 1, in->m, g_op.lnnz, in->halo, g_op.npeers,
         g_op.plan.nblk, g_op.plan.n_interior);
}

/* ----------------------------------------------------------------- SpMM ---- */
static int ensure_halo_buffers(pa_operator_t* o, int ts) {
  if (o->info.halo == 0 && o->nsend == 0) return 0;
  if (o->buf_ts >= ts) return 0;
  pa_rt_free(o->d_sendbuf); pa_rt_free(o->d_halo);
  o->d_sendbuf = (double*)pa_rt_malloc((size_t)(o->nsend ? o->nsend : 1) * ts * sizeof(double));
  o->d_halo = (double*)pa_rt_malloc((size_t)(o->info.halo ? o->info.halo : 1) * ts * sizeof(double));
  if (!o->d_sendbuf || !o->d_halo) return PA_FAIL("halo buffers: %s", pa_rt_error());
  o->buf_ts = ts;
  return 0;
}

/* The kernel that WRITES the panel X the next preAlps_BlockOperator call will multiply (the second half of an ECG
 * iteration: the new search directions) can pack the rows the neighbours need while it has them in registers -- one
 * launch less per iteration (k_pack_rows: 5 us of the 83 us a one-of-eight shard of the headline problem takes).
 * Returns 1 and the inverse send list (row -> slots of the send buffer, rows of `ts` doubles) when this process has
 * neighbours; the operator then skips its own pack for exactly that panel pointer, once.  The caller promises that
 * nothing else writes X in between (the library's own loops, PREALPS_RCI_FUSE=1). */
/* off (m + 2 ints, zero on entry; m + 1 used on return), slot (nsend ints): row r is packed into the send-buffer
 * slots slot[off[r] .. off[r + 1]), ascending (a row goes to every neighbour that reads it) */
static void inverse_send_list(int m, int nsend, const int* send_idx, int* off, int* slot) {
  for (int i = 0; i < nsend; ++i) ++off[send_idx[i] + 2];
  for (int r = 0; r < m; ++r) off[r + 2] += off[r + 1];       /* off[r + 1] = first slot entry of row r */
  for (int i = 0; i < nsend; ++i) slot[off[send_idx[i] + 1]++] = i;   /* ... now off[r + 1] = end of row r */
}
/* The same list on the host, for tests without a GPU (plan-only mode): off_out[m + 1], slot_out[nsend]. */
int preAlps_hip_pack_map(int* off_out, int* slot_out) {
  pa_operator_t* o = &g_op;
  if (!o->info.built) return PA_FAIL("operator not built");
  int m = o->info.m;
  int* off = (int*)calloc((size_t)m + 2, sizeof(int));
  if (!off) return PA_FAIL("out of host memory");
  inverse_send_list(m, o->nsend, o->send_idx, off, slot_out);
  memcpy(off_out, off, ((size_t)m + 1) * sizeof(int));
  free(off);
  return 0;
}
static long long g_packs_fused = 0;       /* products whose send rows the solver's update kernel had packed */
int pa_operator_pack_hint(int ts, const double* X, const int** pk_off, const int** pk_slot, double** sendbuf) {
  pa_operator_t* o = &g_op;
  o->prepacked = NULL;
  if (!o->info.built || g_plan_only || pa_world_size() <= 1 || o->npeers <= 0 || o->nsend <= 0 || !X) return 0;
  if (!env_int("PREALPS_PACK_FUSE", 1)) return 0;
  if (ensure_halo_buffers(o, ts)) return 0;
  if (o->buf_ts != ts) return 0;            /* (buffers sized for a wider panel: slots would be ts_buf apart) */
  if (!o->d_pk_off) {
    int m = o->info.m;
    int* off = (int*)calloc((size_t)m + 2, sizeof(int));
    int* slot = (int*)malloc((size_t)o->nsend * sizeof(int));
    int ok = off && slot;
    if (ok) {
      inverse_send_list(m, o->nsend, o->send_idx, off, slot);
      o->d_pk_off = (int*)pa_rt_malloc(((size_t)m + 1) * sizeof(int));
      o->d_pk_slot = (int*)pa_rt_malloc((size_t)o->nsend * sizeof(int));
      ok = o->d_pk_off && o->d_pk_slot && !pa_rt_h2d(o->d_pk_off, off, ((size_t)m + 1) * sizeof(int)) &&
           !pa_rt_h2d(o->d_pk_slot, slot, (size_t)o->nsend * sizeof(int));
      if (!ok) { pa_rt_free(o->d_pk_off); pa_rt_free(o->d_pk_slot); o->d_pk_off = o->d_pk_slot = NULL; }
    }
    free(off); free(slot);
    if (!ok) return 0;
  }
  *pk_off = o->d_pk_off; *pk_slot = o->d_pk_slot; *sendbuf = o->d_sendbuf;
  o->prepacked = X;
  return 1;
}

/* Cut the SpMM plan for a given enlarging factor now instead of at the first
 * preAlps_BlockOperator call (it is host work proportional to the local nonzeros). */
int preAlps_hip_prepare_operator(int enlFac) {
  pa_operator_t* o = &g_op;
  if (!o->info.built) return PA_FAIL("operator not built");
  if (g_plan_only) return 0;
  int ts = pa_panel_stride(enlFac);
  double t0 = pa_wtime();
  if (o->plan_ts != ts && build_plan(o, ts)) return 1;
  g_setup_plan_s = pa_wtime() - t0;
  return 0;
}

/* Run-to-run spread study: move the matrix values (which & 1) and / or the slot array
 * (which & 2) of the run plan to freshly allocated device memory; the old arrays stay allocated, so the
 * new ones land on other physical pages. */
static size_t g_dbg_val_bytes = 0, g_dbg_slot_bytes = 0;
int preAlps_hip_debug_move_plan(int which) {
  pa_operator_t* o = &g_op;
  if (!o->plan.runs || !g_dbg_val_bytes) return PA_FAIL("no run plan");
  if (which & 1) {
    double* nv = (double*)pa_rt_malloc(g_dbg_val_bytes);
    if (!nv || pa_rt_d2d(nv, o->d_val, g_dbg_val_bytes) || pa_rt_sync()) return PA_FAIL("%s", pa_rt_error());
    o->d_val = nv; o->plan.val = nv;
  }
  if (which & 2) {
    unsigned short* nc = (unsigned short*)pa_rt_malloc(g_dbg_slot_bytes);
    if (!nc || pa_rt_d2d(nc, o->d_col16, g_dbg_slot_bytes) || pa_rt_sync()) return PA_FAIL("%s", pa_rt_error());
    o->d_col16 = nc; o->plan.col16 = nc;
  }
  return 0;
}

int pa_operator_gram_blocks(int ts) {
  pa_operator_t* o = &g_op;
  if (!o->info.built || g_plan_only || ts != 4) return 0;
  if (o->plan_ts != ts && build_plan(o, ts)) return 0;
  /* Default: on, from absolute times of 800-iteration solves alternating in one process.  The window kernel
   * (k_spmm_gram; Poisson 100^3): 245.3 -> 236.9 us per iteration with the block (round 3, tools/probe/abs_ab.py).
   * The run kernel (k_spmm_runs_gram; elasticity 70^3): no gain in round 3 (397.0 -> 399.0 us: its epilogue and
   * the rows of R cost what k_gram and its sum cost); with round 4's kernel -- the matrix groups double-buffered,
   * so the epilogue of one wavefront runs beside the stream of the others -- 407.5 -> 398.1 us
   * (tools/probe/r4_solve_env_ab.py, profiles/r04_spmm_gram_ab.txt).  PREALPS_SPMM_GRAM=0 / 1 forces. */
  int dflt = 1;
  if (!env_int("PREALPS_SPMM_GRAM", dflt)) return 0;
  return o->plan.nblk;
}

/* AX = A X for the X->info.n current columns (operator.c:334-351). */
int preAlps_BlockOperator(CPLM_Mat_Dense_t* X, CPLM_Mat_Dense_t* AX) {
  pa_operator_t* o = &g_op;
  if (!o->info.built) return PA_FAIL("operator not built");
  if (g_plan_only) return PA_FAIL("the operator was built in plan-only mode (no GPU)");
  if (!X || !AX || !X->val || !AX->val) return PA_FAIL(" wrong test 'X->val != NULL && AX->val != NULL'");
  int ts = pa_desc_stride(X);
  if (pa_desc_stride(AX) != ts || X->info.m != o->info.m)
    return PA_FAIL("panel shapes do not match the operator (m %d vs %d, stride %d vs %d)", X->info.m,
                   o->info.m, ts, pa_desc_stride(AX));
  if (o->plan_ts != ts && build_plan(o, ts)) return 1;
  pa_time_begin(PA_T_OPERATOR);
  if (pa_world_size() > 1 && o->npeers > 0) {
    int rc = ensure_halo_buffers(o, ts);
    if (rc) return rc;
    for (int i = 0; i < o->npeers; ++i) { o->send_cnt[i] = o->send_rows[i] * ts; o->recv_cnt[i] = o->recv_rows[i] * ts; }
    /* Exchange beside the interior blocks only when those keep the device busy for longer than the
     * two cross-stream hand-overs cost (8 + 12 us of idle main stream measured around a 10 us interior
     * launch in the one-shard rehearsal): from about 12 M interior nonzeros (25 us); below that the
     * exchange runs on the main stream and one launch covers all blocks.  PREALPS_HALO_OVERLAP=0 / 1 forces. */
    static int overlap_env = -2;
    if (overlap_env == -2) overlap_env = env_int("PREALPS_HALO_OVERLAP", -1);
    int overlap = overlap_env >= 0 ? overlap_env
                                   : (double)o->lnnz * o->plan.n_interior / (o->plan.nblk > 0 ? o->plan.nblk : 1) >= 12e6;
    /* (the solver's update kernel has packed exactly this panel: pa_operator_pack_hint) */
    int packed = o->prepacked && o->prepacked == X->val && o->buf_ts == ts;
    o->prepacked = NULL;
    g_packs_fused += packed;
    if (!overlap) {
      if (!packed) PA_CHECK(pa_k_pack_rows(o->nsend, ts, o->d_send_idx, X->val, o->d_sendbuf));
      rc = pa_exchange(o->d_sendbuf, o->send_cnt, o->d_halo, o->recv_cnt, o->peers, o->npeers);
      if (rc) return rc;
      PA_CHECK(pa_k_spmm(&o->plan, ts, X->val, o->d_halo, AX->val, 2));
      pa_time_end(PA_T_OPERATOR);
      return 0;
    }
    if (!o->ev_packed) { o->ev_packed = pa_rt_event_create(); o->ev_halo = pa_rt_event_create(); }
    if (!o->ev_packed || !o->ev_halo) return PA_FAIL("hipEventCreate failed");
    /* pack on the main stream; the exchange runs on the side stream while the main stream
     * computes the blocks that read no halo row; the halo-reading blocks wait for it */
    void* main_stream = pa_rt_stream();
    void* side = pa_rt_side_stream();
    if (!packed) PA_CHECK(pa_k_pack_rows(o->nsend, ts, o->d_send_idx, X->val, o->d_sendbuf));
    PA_CHECK(pa_rt_event_record_on(o->ev_packed, main_stream));
    PA_CHECK(pa_rt_stream_wait_event(side, o->ev_packed));
    pa_rt_set_stream(side);
    rc = pa_exchange(o->d_sendbuf, o->send_cnt, o->d_halo, o->recv_cnt, o->peers, o->npeers);
    int rc2 = rc ? 0 : pa_rt_event_record_on(o->ev_halo, side);
    pa_rt_set_stream(main_stream);
    if (rc) return rc;
    if (rc2) return PA_FAIL("%s", pa_rt_error());
    PA_CHECK(pa_k_spmm(&o->plan, ts, X->val, o->d_halo, AX->val, 0));
    PA_CHECK(pa_rt_stream_wait_event(main_stream, o->ev_halo));
    PA_CHECK(pa_k_spmm(&o->plan, ts, X->val, o->d_halo, AX->val, 1));
  } else {
    PA_CHECK(pa_k_spmm(&o->plan, ts, X->val, NULL, AX->val, 2));
  }
  pa_time_end(PA_T_OPERATOR);
  return 0;
}

int preAlps_hip_get_stat(const char* key, double* value) {
  const pa_operator_t* o = &g_op;
  if (!strcmp(key, "nnz_local")) *value = o->lnnz;
  else if (!strcmp(key, "rows_local")) *value = o->info.m;
  else if (!strcmp(key, "halo_rows")) *value = o->info.halo;
  else if (!strcmp(key, "send_rows")) *value = o->nsend;
  else if (!strcmp(key, "spmm_blocks")) *value = o->plan.nblk;
  else if (!strcmp(key, "spmm_gram_launches")) *value = (double)pa_k_spmm_gram_launches();
  else if (!strcmp(key, "bj_gram_applies")) *value = (double)pa_k_bj_gram_applies();
  else if (!strcmp(key, "packs_fused")) *value = (double)g_packs_fused;
  else if (!strcmp(key, "spmm_slices")) *value = o->plan.nslices;
  else if (!strcmp(key, "spmm_stored_entries")) *value = o->sell_entries;
  else if (!strcmp(key, "spmm_stream_bytes")) *value = o->stream_bytes;
  else if (!strcmp(key, "spmm_val_address")) *value = (double)(uintptr_t)o->d_val;      /* (for the run-to-run spread study) */
  else if (!strcmp(key, "spmm_slot_address")) *value = (double)(uintptr_t)o->d_col16;
  else if (!strcmp(key, "spmm_staged")) *value = o->plan.staged;
  else if (!strcmp(key, "spmm_runs")) *value = o->plan.runs;
  else if (!strcmp(key, "spmm_stage_rows")) *value = o->plan.stage_cap;
  else if (!strcmp(key, "spmm_interior_blocks")) *value = o->plan.n_interior;
  else if (!strcmp(key, "setup_build_s")) *value = g_setup_build_s;
  else if (!strcmp(key, "setup_plan_s")) *value = g_setup_plan_s;
  else if (!strcmp(key, "setup_bj_factor_s")) *value = pa_bj_setup_seconds(0);
  else if (!strcmp(key, "setup_bj_layout_s")) *value = pa_bj_setup_seconds(1);
  else if (!strcmp(key, "bj_factor_bytes")) *value = pa_bj_factor_bytes();
  else if (!strcmp(key, "bj_max_bandwidth")) *value = pa_bj_max_bandwidth();
  else if (!strcmp(key, "bj_parts_local")) *value = pa_bj_nparts();
  else if (!strcmp(key, "bj_nd_blocks")) *value = pa_bj_nd_blocks();
  else if (!strcmp(key, "bj_pairs_bytes")) *value = pa_bj_pairs_bytes();
  else if (!strcmp(key, "bj_g4_bytes")) *value = pa_bj_g4_bytes();
  else if (!strcmp(key, "bj_nd_inverse_dev")) *value = pa_nd_inverse_deviation();
  else return 1;
  return 0;
}

/* rhs of examples/test_ecg_prealps_op.c:172-184 as np = nparts ranks build it:
 * every rank restarts the generator (srand(0)), the 2-norm is global, and
 * element 0 of every rank is left unscaled. */
int preAlps_hip_reference_rhs(double* rhs_local) {
  const pa_operator_info_t* in = &g_op.info;
  if (!in->built) return PA_FAIL("operator not built");
  int mmax = 0;
  for (int p = 0; p < in->nparts; ++p) { int l = in->rowPos[p + 1] - in->rowPos[p]; if (l > mmax) mmax = l; }
  double* stream = (double*)malloc((size_t)mmax * sizeof(double));
  srand(0);
  for (int i = 0; i < mmax; ++i) stream[i] = ((double)rand() / (double)RAND_MAX);
  double normb = 0.0;
  for (int p = 0; p < in->nparts; ++p) {
    int l = in->rowPos[p + 1] - in->rowPos[p];
    double s = 0.0;
    for (int i = 0; i < l; ++i) s += stream[i] * stream[i];
    normb += s;
  }
  normb = sqrt(normb);
  for (int p = in->part0; p < in->part1; ++p) {
    int base = in->rowPos[p] - in->row_off, l = in->rowPos[p + 1] - in->rowPos[p];
    rhs_local[base] = stream[0];
    for (int i = 1; i < l; ++i) rhs_local[base + i] = stream[i] / normb;
  }
  free(stream);
  return 0;
}

/* Staged plan with one LDS slot per run of up to three consecutive columns.  The staging
 * area lists the external rows below the block's own range, the own range, the external
 * rows above it and two zero rows, so slots ascend with the column and a run never breaks
 * at the edge of the own range.  A run covers slots [s, s+2]; entries of the row that fall
 * inside it fill its three values, the rest are zeros.  Returns 0 on success, 1 when the
 * plan does not pay (zero fill above 6 % of the plain SELL storage, external rows above a
 * quarter of the matrix stream, or a slice that does not fit the staging area), -1 on error. */
static int build_plan_runs(pa_operator_t* o, int ts) {
  const pa_operator_info_t* in = &o->info;
  const int* rowptr = in->A.rowPtr;
  const int* colind = o->lcol;
  const double* val = in->A.val;
  int m = in->m, ncols = m + in->halo;
  /* panels of 16 columns: the plan is cut for 8 columns and a block is worked twice, once per half of the
   * panel (spmm.hip; staging area and pay-off test of stride 8).  One workgroup staging all 16 columns was
   * measured slower: 628 against 371 us, the LDS reads of 128 B per nonzero and lane dominate. */
  if (ts >= 16) ts /= 2;
  /* 8 columns: the rows of the staging area are 80 bytes apart (spmm.hip: spmm_row, no LDS bank conflicts) and a
   * workgroup may stage 80 KiB, two workgroups per CU.  Block rows / staging budget measured with the padded rows
   * (70^3, iterations/s at 8 | 16 columns): 192 / 60 KiB 1834 | 1038, 256 / 64 KiB 1776 | 977, **256 / 80 KiB
   * 1849 | 1094**, 320 / 80 KiB 1763 | 998, 320 / 100 KiB 1679 | 912 (before the padding: 192 rows / 48 KiB,
   * three workgroups per CU, 1832 | 1025). */
  int cap_rows = (ts <= 4 ? 32768 : 81920) / (ts == 8 ? 80 : ts * 8) - 2;
  if (cap_rows > 65533) cap_rows = 65533;
  int blk_rows = spmm_block_rows(m, 256);
  if (blk_rows < 64) blk_rows = 64;
  blk_rows &= ~63;
  if (blk_rows > cap_rows) blk_rows = cap_rows & ~63;
  if (blk_rows < 64) return 1;
  int nslices = 0;
  double plain = 0.0;
  for (int p = in->part0; p < in->part1; ++p) nslices += (in->rowPos[p + 1] - in->rowPos[p] + 63) / 64;
  size_t ns1 = (size_t)(nslices ? nslices : 1);
  long long* sl_off = (long long*)malloc((ns1 + 1) * sizeof(long long));
  int* sl_len = (int*)malloc(ns1 * sizeof(int));
  int* sl_row0 = (int*)malloc(ns1 * sizeof(int));
  int* sl_nrows = (int*)malloc(ns1 * sizeof(int));
  int s = 0;
  for (int p = in->part0; p < in->part1; ++p) {
    int pr0 = in->rowPos[p] - in->row_off, pr1 = in->rowPos[p + 1] - in->row_off;
    for (int r = pr0; r < pr1; r += 64, ++s) {
      int nr = pr1 - r < 64 ? pr1 - r : 64, len = 0;
      for (int i = 0; i < nr; ++i) { int l = rowptr[r + i + 1] - rowptr[r + i]; if (l > len) len = l; }
      sl_row0[s] = r; sl_nrows[s] = nr;
      plain += 64.0 * len;
    }
  }
  int* blk_slice = (int*)malloc((ns1 + 1) * sizeof(int));
  int* blk_ext_off = (int*)malloc((ns1 + 1) * sizeof(int));
  int* blk_nlow = (int*)malloc(ns1 * sizeof(int));
  char* needs_halo = (char*)malloc(ns1);
  int* stamp = (int*)calloc(ncols ? ncols : 1, sizeof(int));
  int* slot_of = (int*)malloc((ncols ? ncols : 1) * sizeof(int));
  size_t ext_cap = 1024, next_tot = 0;
  int* ext_rows = (int*)malloc(ext_cap * sizeof(int));
  size_t run_cap = (size_t)(plain / 3.0 * 1.1) + 4096, nruns = 0;   /* stored runs (64 per step of a slice) */
  unsigned short* c16 = (unsigned short*)big_alloc(run_cap * sizeof(unsigned short));
  double* sval = (double*)big_alloc(run_cap * 3 * sizeof(double));
  int nblk = 0, q = 0, max_stage = 0, rc = 0, gen = 0;
  if (!c16 || !sval || !stamp || !slot_of || !ext_rows) rc = -1;
  sl_off[0] = 0;
  while (q < nslices && !rc) {
    int nsl = 0;
    while (q + nsl < nslices && (nsl + 1) * 64 <= blk_rows) ++nsl;
    for (;;) {
      int r0 = sl_row0[q], r1 = sl_row0[q + nsl - 1] + sl_nrows[q + nsl - 1];
      int nown = r1 - r0, next = 0, nlow = 0;
      size_t mark0 = next_tot;
      char h = 0;
      ++gen;
      for (int k = rowptr[r0]; k < rowptr[r1]; ++k) {
        int c = colind[k];
        if (c >= r0 && c < r1) continue;
        if (stamp[c] != gen) {
          stamp[c] = gen;
          if (next_tot == ext_cap) { ext_cap *= 2; ext_rows = (int*)realloc(ext_rows, ext_cap * sizeof(int)); }
          ext_rows[next_tot++] = c; ++next;
          if (c < r0) ++nlow;
          if (c >= m) h = 1;
        }
      }
      if (nown + next > cap_rows) {
        next_tot = mark0;
        if (nsl == 1) { rc = 1; break; }
        nsl = (nsl + 1) / 2;
        continue;
      }
      int* er = ext_rows + mark0;
      for (int a = 1; a < next; ++a) { int v = er[a], b2 = a; while (b2 > 0 && er[b2 - 1] > v) { er[b2] = er[b2 - 1]; --b2; } er[b2] = v; }
      for (int a = 0; a < next; ++a) slot_of[er[a]] = a < nlow ? a : nown + a;
      for (int sq = q; sq < q + nsl; ++sq) {
        int r = sl_row0[sq], nr = sl_nrows[sq], len3 = 0;
        /* pass 1: runs per row */
        for (int i = 0; i < nr; ++i) {
          int row = r + i, nrun = 0, last = -4;
          for (int k = rowptr[row]; k < rowptr[row + 1]; ++k) {
            int c = colind[k], sl = (c >= r0 && c < r1) ? nlow + c - r0 : slot_of[c];
            if (sl > last + 2 || sl < last) { last = sl; ++nrun; }
          }
          if (nrun > len3) len3 = nrun;
        }
        if (nruns + (size_t)len3 * 64 > run_cap) {
          run_cap = (run_cap + (size_t)len3 * 64) * 3 / 2;
          c16 = (unsigned short*)realloc(c16, run_cap * sizeof(unsigned short));
          sval = (double*)realloc(sval, run_cap * 3 * sizeof(double));
          if (!c16 || !sval) { rc = -1; break; }
        }
        unsigned short* cc = c16 + nruns;
        double* vv = sval + 3 * nruns;
        for (int i = 0; i < 64; ++i) {
          int row = i < nr ? r + i : r, own = nlow + row - r0, nrun = 0, last = -4;
          if (i < nr)
            for (int k = rowptr[row]; k < rowptr[row + 1]; ++k) {
              int c = colind[k], sl = (c >= r0 && c < r1) ? nlow + c - r0 : slot_of[c];
              if (sl > last + 2 || sl < last) {
                last = sl;
                cc[(size_t)nrun * 64 + i] = (unsigned short)sl;
                vv[((size_t)3 * nrun + 0) * 64 + i] = 0.0; vv[((size_t)3 * nrun + 1) * 64 + i] = 0.0;
                vv[((size_t)3 * nrun + 2) * 64 + i] = 0.0;
                ++nrun;
              }
              vv[((size_t)3 * (nrun - 1) + (sl - last)) * 64 + i] = val[k];
            }
          for (int k = nrun; k < len3; ++k) {     /* padding: zeros against the row's own slot */
            cc[(size_t)k * 64 + i] = (unsigned short)own;
            vv[((size_t)3 * k + 0) * 64 + i] = 0.0; vv[((size_t)3 * k + 1) * 64 + i] = 0.0;
            vv[((size_t)3 * k + 2) * 64 + i] = 0.0;
          }
        }
        sl_len[sq] = len3;
        nruns += (size_t)len3 * 64;
        sl_off[sq + 1] = (long long)nruns;
      }
      if (rc) break;
      if (nown + next + 2 > max_stage) max_stage = nown + next + 2;
      blk_slice[nblk] = q; blk_ext_off[nblk] = (int)mark0; blk_nlow[nblk] = nlow; needs_halo[nblk] = h;
      ++nblk;
      q += nsl;
      break;
    }
  }
  if (!rc) {
    double stream = 26.0 * (double)nruns + 4.0 * (double)next_tot;
    double ext_bytes = (double)next_tot * ts * 8.0;
    /* PREALPS_SPMM_RUNS=2 forces the plan (tests on irregular patterns) */
    if (env_int("PREALPS_SPMM_RUNS", 1) < 2 && (3.0 * (double)nruns > 1.06 * plain || ext_bytes >= 0.25 * 10.0 * plain)) rc = 1;
    else o->stream_bytes = stream;
  }
  if (!rc) {
    blk_slice[nblk] = nslices; blk_ext_off[nblk] = (int)next_tot;
    int* order = (int*)malloc((nblk > 0 ? nblk : 1) * sizeof(int));
    int ni = 0;
    for (int b = 0; b < nblk; ++b) if (!needs_halo[b]) order[ni++] = b;
    int k2 = ni;
    for (int b = 0; b < nblk; ++b) if (needs_halo[b]) order[k2++] = b;
    size_t nr1 = nruns ? nruns : 1;
    o->d_sl_off = (long long*)pa_rt_malloc((ns1 + 1) * sizeof(long long));
    o->d_sl_len = (int*)pa_rt_malloc(ns1 * sizeof(int));
    o->d_sl_row0 = (int*)pa_rt_malloc(ns1 * sizeof(int));
    o->d_sl_nrows = (int*)pa_rt_malloc(ns1 * sizeof(int));
    o->d_col16 = (unsigned short*)pa_rt_malloc(nr1 * sizeof(unsigned short));
    o->d_val = (double*)pa_rt_malloc(nr1 * 3 * sizeof(double));
    g_dbg_val_bytes = nr1 * 3 * sizeof(double); g_dbg_slot_bytes = nr1 * sizeof(unsigned short);
    o->d_blk_slice = (int*)pa_rt_malloc(((size_t)nblk + 1) * sizeof(int));
    o->d_blk_ext_off = (int*)pa_rt_malloc(((size_t)nblk + 1) * sizeof(int));
    o->d_blk_nlow = (int*)pa_rt_malloc((size_t)(nblk > 0 ? nblk : 1) * sizeof(int));
    o->d_ext_rows = (int*)pa_rt_malloc((next_tot + 1) * sizeof(int));   /* one spare entry: k_spmm_runs reads ids unconditionally */
    o->d_order = (int*)pa_rt_malloc((nblk > 0 ? nblk : 1) * sizeof(int));
    int bad = (!o->d_sl_off || !o->d_sl_len || !o->d_sl_row0 || !o->d_sl_nrows || !o->d_col16 || !o->d_val ||
               !o->d_blk_slice || !o->d_blk_ext_off || !o->d_blk_nlow || !o->d_ext_rows || !o->d_order);
    bad = bad || pa_rt_h2d(o->d_sl_off, sl_off, ((size_t)nslices + 1) * sizeof(long long));
    bad = bad || pa_rt_h2d(o->d_sl_len, sl_len, nslices * sizeof(int));
    bad = bad || pa_rt_h2d(o->d_sl_row0, sl_row0, nslices * sizeof(int));
    bad = bad || pa_rt_h2d(o->d_sl_nrows, sl_nrows, nslices * sizeof(int));
    bad = bad || pa_rt_h2d(o->d_col16, c16, nruns * sizeof(unsigned short));
    bad = bad || pa_rt_h2d(o->d_val, sval, nruns * 3 * sizeof(double));
    bad = bad || pa_rt_h2d(o->d_blk_slice, blk_slice, ((size_t)nblk + 1) * sizeof(int));
    bad = bad || pa_rt_h2d(o->d_blk_ext_off, blk_ext_off, ((size_t)nblk + 1) * sizeof(int));
    bad = bad || pa_rt_h2d(o->d_blk_nlow, blk_nlow, (size_t)nblk * sizeof(int));
    bad = bad || pa_rt_h2d(o->d_ext_rows, ext_rows, next_tot * sizeof(int));
    bad = bad || pa_rt_h2d(o->d_order, order, nblk * sizeof(int));
    free(order);
    if (bad) { PA_FAIL("uploading the SpMM plan failed: %s", pa_rt_error()); rc = -1; }
    else {
      pa_spmm_plan_t* pl = &o->plan;
      pl->m = m; pl->nslices = nslices; pl->sl_off = o->d_sl_off; pl->sl_len = o->d_sl_len;
      pl->sl_row0 = o->d_sl_row0; pl->sl_nrows = o->d_sl_nrows; pl->val = o->d_val;
      pl->nblk = nblk; pl->blk_slice = o->d_blk_slice; pl->order = o->d_order; pl->n_interior = ni;
      pl->staged = 1; pl->runs = 1; pl->runs_cols = ts; pl->col16 = o->d_col16; pl->blk_ext_off = o->d_blk_ext_off;
      pl->blk_nlow = o->d_blk_nlow; pl->ext_rows = o->d_ext_rows;
      pl->stage_cap = max_stage;
      o->sell_entries = 3.0 * (double)nruns;
    }
  } else if (rc < 0) {
    PA_FAIL("out of host memory for the SpMM plan");
  }
  free(sl_off); free(sl_len); free(sl_row0); free(sl_nrows);
  free(blk_slice); free(blk_ext_off); free(blk_nlow); free(needs_halo); free(stamp); free(slot_of); free(ext_rows);
  free(c16); free(sval);
  return rc;
}
