/*
 * block_jacobi.c -- block-Jacobi preconditioner Z = blockdiag(A)^-1 X with one
 * exact SPD solve per subdomain, factors resident in HBM.
 *
 * Reference behaviour kept (under /root/reference):
 *   preAlps_BlockJacobiCreate  src/preconditioners/block_jacobi.c:26-63
 *       diagonal block of the local row panel = entries whose column lies in
 *       the part's own row range (utils/cplm_v0/cplm_v0_matcsr.c:287-463),
 *       Cholesky factorisation (PARDISO phase 12, cplm_kernels.c:741-784)
 *   preAlps_BlockJacobiApply   block_jacobi.c:93-109 -> PARDISO phase 33 with
 *       nrhs = A_in->info.n (cplm_kernels.c:790-853)
 *   preAlps_BlockJacobiFree    block_jacobi.c:111-118
 *
 * MI355X design: a sparse direct solver with supernodes and pivoting queues
 * does not map onto 64-wide wavefronts; instead every block is reordered by
 * reverse Cuthill-McKee, factored once as a dense-band Cholesky L L^T (exact:
 * no fill leaves the band) and stored twice, column-wise for the forward sweep
 * and row-wise (reversed) for the backward sweep, so both sweeps stream their
 * band with coalesced loads (kernels.hip: k_bj_apply).  One process owns
 * nparts/size blocks; one wavefront solves one block.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pa_host.h"

typedef struct {
  int created;
  int np;                 /* local blocks */
  int m;
  /* device */
  int* d_row0; int* d_nrows; int* d_bw; long long* d_off;
  int* d_map_f; int* d_map_b;
  double* d_Lf; double* d_Lb; double* d_invd_f; double* d_invd_b;
  double* d_Lf2; double* d_Lb2; long long* d_off2;     /* paired records of the narrow classes (optional) */
  double pairs_bytes;
  double* d_Lg4; double g4_bytes;                      /* one-copy records of bj_g4.hip (optional) */
  int class_g4[16]; int class_bmax[16];
  /* classes by register sets */
  int nclass; int class_R[16]; int class_count[16]; int class_wmax[16]; int* class_list[16];
  const int* class_list_c[16];
  pa_bj_plan_t plan;
  double factor_bytes; int max_bw; int nd_blocks;
} pa_bj_t;

/* window of a wide block = record length (kernels.hip: bjw_window) */
static int bj_wide_window(int w) {
  int W = (w + 64 + 63) & ~63;
  if (W <= 1024) return W;
  W = (w + 64 + 127) & ~127;
  if (W <= 2048) return W;
  return (w + 64 + 255) & ~255;
}

/* block that holds local (factor-order) position `pos` */
static int part_of_local_row(const int* row0, const int* nrows, int np, int pos) {
  for (int q = 0; q < np; ++q) if (pos >= row0[q] && pos < row0[q] + nrows[q]) return q;
  return 0;
}

static pa_bj_t g_bj;
static double g_bj_setup_s[2];
double pa_bj_setup_seconds(int which) { return g_bj_setup_s[which ? 1 : 0]; }

double pa_bj_factor_bytes(void) { return g_bj.created ? g_bj.factor_bytes : 0.0; }
int pa_bj_max_bandwidth(void) { return g_bj.created ? g_bj.max_bw : 0; }
int pa_bj_nparts(void) { return g_bj.created ? g_bj.np : 0; }
int pa_bj_nd_blocks(void) { return g_bj.created ? g_bj.nd_blocks : 0; }
double pa_bj_pairs_bytes(void) { return g_bj.created ? g_bj.pairs_bytes : 0.0; }
double pa_bj_g4_bytes(void) { return g_bj.created ? g_bj.g4_bytes : 0.0; }
/* blocks of the apply when it can leave the ECG Gram block [in | prev]^T out behind (pa_k_bj_gram_arm): every
 * local block in one class served by bj_g4.hip, no sparse-factored blocks; 0: it cannot */
int pa_bj_gram_blocks(void) {
  const pa_bj_t* s = &g_bj;
  if (!s->created || s->nd_blocks > 0 || !s->d_Lg4 || s->nclass != 1 || !s->class_g4[0] || s->class_count[0] != s->np) return 0;
  return s->np;
}

void preAlps_BlockJacobiFree(void) {
  pa_bj_t* s = &g_bj;
  pa_rt_free(s->d_row0); pa_rt_free(s->d_nrows); pa_rt_free(s->d_bw); pa_rt_free(s->d_off);
  pa_rt_free(s->d_map_f); pa_rt_free(s->d_map_b);
  pa_rt_free(s->d_Lf); pa_rt_free(s->d_Lb); pa_rt_free(s->d_invd_f); pa_rt_free(s->d_invd_b);
  pa_rt_free(s->d_Lf2); pa_rt_free(s->d_Lb2); pa_rt_free(s->d_off2); pa_rt_free(s->d_Lg4);
  for (int c = 0; c < 16; ++c) pa_rt_free(s->class_list[c]);
  pa_nd_free();
  memset(s, 0, sizeof(*s));
}

/* ---- reverse Cuthill-McKee of one block ---------------------------------- */
typedef struct { int* xadj; int* adj; int* deg; int* order; int* pos; int* queue; int* level; } rcm_ws_t;

static int bfs_levels(int b, const int* xadj, const int* adj, int root, int* level, int* queue,
                      int stamp_unvisited, int* last_out) {
  /* level[] must hold stamp_unvisited for nodes of this component not yet seen */
  int head = 0, tail = 0, maxl = 0;
  (void)b;
  queue[tail++] = root; level[root] = 0;
  while (head < tail) {
    int u = queue[head++];
    for (int k = xadj[u]; k < xadj[u + 1]; ++k) {
      int v = adj[k];
      if (level[v] == stamp_unvisited) { level[v] = level[u] + 1; if (level[v] > maxl) maxl = level[v]; queue[tail++] = v; }
    }
  }
  *last_out = tail; /* nodes reached */
  return maxl;
}

static void rcm_order(int b, const int* xadj, const int* adj, const int* deg, int* order, int* pos,
                      int* queue, int* level) {
  const int UNSEEN = -1, DONE = -2;
  int placed = 0;
  for (int i = 0; i < b; ++i) pos[i] = UNSEEN; /* pos doubles as "placed" marker */
  while (placed < b) {
    /* seed: unplaced node of minimum degree */
    int seed = -1;
    for (int i = 0; i < b; ++i) if (pos[i] == UNSEEN && (seed < 0 || deg[i] < deg[seed])) seed = i;
    /* pseudo-peripheral root: repeat BFS from the min-degree node of the last level */
    int root = seed, ecc = -1;
    for (int it = 0; it < 6; ++it) {
      for (int i = 0; i < b; ++i) level[i] = (pos[i] == UNSEEN) ? UNSEEN : DONE;
      int reached = 0;
      int e = bfs_levels(b, xadj, adj, root, level, queue, UNSEEN, &reached);
      if (e <= ecc) break;
      ecc = e;
      int cand = -1;
      for (int q = 0; q < reached; ++q) { int u = queue[q]; if (level[u] == e && (cand < 0 || deg[u] < deg[cand])) cand = u; }
      if (cand < 0 || cand == root) break;
      root = cand;
    }
    /* Cuthill-McKee from root, neighbours by increasing degree */
    int head = placed, tail = placed;
    order[tail++] = root; pos[root] = DONE;
    while (head < tail) {
      int u = order[head++];
      int s0 = tail;
      for (int k = xadj[u]; k < xadj[u + 1]; ++k) { int v = adj[k]; if (pos[v] == UNSEEN) { pos[v] = DONE; order[tail++] = v; } }
      for (int a = s0 + 1; a < tail; ++a) { /* insertion sort by degree */
        int v = order[a], c = a;
        while (c > s0 && deg[order[c - 1]] > deg[v]) { order[c] = order[c - 1]; --c; }
        order[c] = v;
      }
    }
    placed = tail;
  }
  for (int i = 0; i < b / 2; ++i) { int t = order[i]; order[i] = order[b - 1 - i]; order[b - 1 - i] = t; }
  for (int i = 0; i < b; ++i) pos[order[i]] = i;
}

int preAlps_BlockJacobiCreate(CPLM_Mat_CSR_t* A, int* rowPos, int sizeRowPos, int* colPos,
                              int sizeColPos) {
  (void)colPos; (void)sizeColPos;
  PA_REQUIRE_GPU();
  const pa_operator_info_t* op = pa_operator_info();
  if (!op) return PA_FAIL("the operator must be built before the preconditioner");
  if (!A || !A->rowPtr || !rowPos || sizeRowPos != op->nparts + 1)
    return PA_FAIL(" wrong test 'A != NULL && sizeRowPos == nparts + 1'");
  if (g_bj.created) preAlps_BlockJacobiFree();
  pa_bj_t* s = &g_bj;
  int np = op->part1 - op->part0, m = op->m, row_off = op->row_off;
  s->np = np; s->m = m;
  int* row0 = (int*)malloc(np * sizeof(int));
  int* nrows = (int*)malloc(np * sizeof(int));
  int* bw = (int*)calloc(np, sizeof(int));
  long long* off = (long long*)malloc((np + 1) * sizeof(long long));
  int* map_f = (int*)malloc((size_t)(m ? m : 1) * sizeof(int));
  int* map_b = (int*)malloc((size_t)(m ? m : 1) * sizeof(int));
  double* invd_f = (double*)malloc((size_t)(m ? m : 1) * sizeof(double));
  double* invd_b = (double*)malloc((size_t)(m ? m : 1) * sizeof(double));
  double** bands = (double**)calloc(np, sizeof(double*));
  long long** coo_off = (long long**)calloc(np, sizeof(long long*));   /* wide blocks, device path */
  double** coo_val = (double**)calloc(np, sizeof(double*));
  size_t* coo_n = (size_t*)calloc(np, sizeof(size_t));
  int fail_row = -1;
  for (int q = 0; q < np; ++q) { row0[q] = rowPos[op->part0 + q] - row_off; nrows[q] = rowPos[op->part0 + q + 1] - rowPos[op->part0 + q]; }
  /* Large blocks (few subdomains of thousands of rows, the reference's own regime) get a sparse
   * nested-dissection factor instead of a band (nd.c).  PREALPS_BJ_ND: 0 never, 1 (default) for
   * blocks of at least PREALPS_BJ_ND_ROWS (2048) rows whose band exceeds 256, 2 for every block of
   * at least PREALPS_BJ_ND_ROWS rows.  (Measured on elasticity 70^3, per apply: blocks of 17.5 k rows
   * 2.3 ms against 15.3 ms with the band kernels; 2187 rows 1.35 against 1.46 ms; 648 rows 0.90
   * against 0.60 ms: small blocks stay with the band.) */
  char* is_nd = (char*)calloc(np ? np : 1, 1);
  const int nd_mode = getenv("PREALPS_BJ_ND") ? atoi(getenv("PREALPS_BJ_ND")) : 1;
  const int nd_rows = getenv("PREALPS_BJ_ND_ROWS") ? atoi(getenv("PREALPS_BJ_ND_ROWS")) : 2048;

  double t_setup0 = pa_wtime();
  /* PREALPS_BJ_FACTOR=host keeps every factorisation on the host threads */
  const char* fenv = getenv("PREALPS_BJ_FACTOR");
  const int dev_factor = !(fenv && !strcmp(fenv, "host")), dev_wmax = pa_bj_factor_wmax();
  /* pass 1 (parallel over blocks): RCM order, bandwidth, band assembly, host band Cholesky */
#pragma omp parallel for num_threads(pa_host_threads()) schedule(dynamic, 1)
  for (int q = 0; q < np; ++q) {
    int r0 = row0[q], b = nrows[q];
    int g0 = rowPos[op->part0 + q], g1 = g0 + b;
    int nadj = 0;
    for (int i = 0; i < b; ++i)
      for (int k = A->rowPtr[r0 + i]; k < A->rowPtr[r0 + i + 1]; ++k) { int c = A->colInd[k]; if (c >= g0 && c < g1 && c != g0 + i) ++nadj; }
    int* xadj = (int*)malloc((b + 1) * sizeof(int));
    int* adj = (int*)malloc((nadj ? nadj : 1) * sizeof(int));
    int* deg = (int*)malloc(b * sizeof(int));
    int* order = (int*)malloc(b * sizeof(int));
    int* pos = (int*)malloc(b * sizeof(int));
    int* queue = (int*)malloc(b * sizeof(int));
    int* level = (int*)malloc(b * sizeof(int));
    xadj[0] = 0;
    for (int i = 0; i < b; ++i) {
      int e = xadj[i];
      for (int k = A->rowPtr[r0 + i]; k < A->rowPtr[r0 + i + 1]; ++k) { int c = A->colInd[k]; if (c >= g0 && c < g1 && c != g0 + i) adj[e++] = c - g0; }
      xadj[i + 1] = e; deg[i] = e - xadj[i];
    }
    rcm_order(b, xadj, adj, deg, order, pos, queue, level);
    int w = 0, wnat = 0;
    for (int i = 0; i < b; ++i)
      for (int k = xadj[i]; k < xadj[i + 1]; ++k) {
        int dd = pos[i] - pos[adj[k]]; if (dd < 0) dd = -dd; if (dd > w) w = dd;
        int dn = i - adj[k]; if (dn < 0) dn = -dn; if (dn > wnat) wnat = dn;
      }
    /* RCM is a heuristic: on elongated boxes of a structured grid the given row order (long
     * axis slowest) can have the narrower band -- keep whichever is narrower */
    if (wnat <= w) { w = wnat; for (int i = 0; i < b; ++i) { order[i] = i; pos[i] = i; } }
    bw[q] = w;
    if (nd_mode && b >= nd_rows && (nd_mode >= 2 || w > 256)) {
      is_nd[q] = 1;
      for (int j = 0; j < b; ++j) { map_f[r0 + j] = j; map_b[r0 + j] = b - 1 - j; }
      free(xadj); free(adj); free(deg); free(order); free(pos); free(queue); free(level);
      continue;
    }
    /* band[i*(w+1) + d] = A(new i, new i-d); blocks that go to the blocked device
     * factorisation (k_bj_factor_big) are assembled diagonal-major instead: band[d*b + i] */
    int big_dev = dev_factor && w > dev_wmax;
    double* band = NULL;
    if (big_dev) {
      /* only the entries travel: (offset in the block's diagonal-major band, value) */
      size_t cnt = 0;
      for (int i = 0; i < b; ++i)
        for (int k = A->rowPtr[r0 + i]; k < A->rowPtr[r0 + i + 1]; ++k) {
          int c = A->colInd[k];
          if (c >= g0 && c < g1 && pos[c - g0] <= pos[i]) ++cnt;
        }
      long long* co = (long long*)malloc((cnt ? cnt : 1) * sizeof(long long));
      double* cv = (double*)malloc((cnt ? cnt : 1) * sizeof(double));
      cnt = 0;
      for (int i = 0; i < b; ++i) {
        int ni = pos[i];
        for (int k = A->rowPtr[r0 + i]; k < A->rowPtr[r0 + i + 1]; ++k) {
          int c = A->colInd[k];
          if (c < g0 || c >= g1) continue;
          int nj = pos[c - g0];
          if (nj > ni) continue;
          co[cnt] = (long long)(ni - nj) * b + ni; cv[cnt] = A->val[k]; ++cnt;
        }
      }
      coo_off[q] = co; coo_val[q] = cv; coo_n[q] = cnt;
    } else {
      band = (double*)calloc((size_t)b * (w + 1), sizeof(double));
      for (int i = 0; i < b; ++i) {
        int ni = pos[i];
        for (int k = A->rowPtr[r0 + i]; k < A->rowPtr[r0 + i + 1]; ++k) {
          int c = A->colInd[k];
          if (c < g0 || c >= g1) continue;
          int nj = pos[c - g0];
          if (nj <= ni) band[(size_t)ni * (w + 1) + (ni - nj)] = A->val[k];
        }
      }
    }
    size_t ld = (size_t)w + 1;
    /* the device factors every band (k_bj_factor / k_bj_factor_big) unless PREALPS_BJ_FACTOR=host */
    for (int i = 0; i < b && !dev_factor; ++i) {
      double* Li = band + (size_t)i * ld; /* Li[d] = L(i, i-d) */
      int jlo = i - w > 0 ? i - w : 0;
      for (int j = jlo; j < i; ++j) {
        const double* Lj = band + (size_t)j * ld;
        int klo = j - w > jlo ? j - w : jlo;
        double sum = Li[i - j];
        for (int k = klo; k < j; ++k) sum -= Li[i - k] * Lj[j - k];
        Li[i - j] = sum / Lj[0];
      }
      double dsum = Li[0];
      for (int k = jlo; k < i; ++k) dsum -= Li[i - k] * Li[i - k];
      if (!(dsum > 0.0)) {
#pragma omp critical
        { if (fail_row < 0) fail_row = row_off + r0 + order[i]; }
        dsum = NAN;
      }
      Li[0] = sqrt(dsum);
    }
    bands[q] = band;
    for (int j = 0; j < b; ++j) {
      map_f[r0 + j] = order[j];
      map_b[r0 + j] = order[b - 1 - j];
    }
    free(xadj); free(adj); free(deg); free(order); free(pos); free(queue); free(level);
  }
  int rc = 0;
  g_bj_setup_s[0] = pa_wtime() - t_setup0;
  t_setup0 = pa_wtime();
  if (fail_row >= 0) rc = PA_FAIL("diagonal block is not SPD (global row %d)", fail_row);
  /* pass 2: sweep layouts */
  off[0] = 0;
  int maxw = 0;
  /* narrow bands (one wavefront per block): one record of wr doubles per step,
   * [L(j+1..j+w, j) / L(j,j) | 0], wr = w + 1 rounded up to even.  Wide bands (one workgroup per
   * block): records of W = bj_wide_window(w) >= w + 64 doubles in window-slot order, the value
   * for target row i at column i mod W. */
  int maxR = pa_bj_max_R();
  /* Bands above `wide_from` get one workgroup per block (k_bj_wide) instead of one wavefront
   * (k_bj_apply / k_bj_mfma): always above 448, where the wavefront's registers end, and from
   * 97 on when the blocks are too few to give every SIMD a wavefront -- a lone wavefront per
   * SIMD is latency bound (measured on Poisson 100^3 with 512 blocks, w = 133: 1.6 ms). */
  int wide_from = np < 1024 ? pa_bj_factor_wmax() : 64 * maxR - 64;
  {
    /* PREALPS_BJ_WIDE_FROM = first band that is NOT given to a single wavefront, minus one
     * (tests use it to reach both dispatches on small problems).  The device factorisation lays
     * out window-slot records only for bands above pa_bj_factor_wmax(), and a wavefront's
     * registers end at 64 * maxR - 64. */
    const char* e = getenv("PREALPS_BJ_WIDE_FROM");
    if (e && *e) {
      wide_from = atoi(e);
      if (wide_from < pa_bj_factor_wmax()) wide_from = pa_bj_factor_wmax();
      if (wide_from > 64 * maxR - 64) wide_from = 64 * maxR - 64;
    }
  }
  int maxw_all = 0;
  for (int q = 0; q < np; ++q) {
    int wide = bw[q] > wide_from;
    long long reclen = wide ? bj_wide_window(bw[q]) : ((bw[q] + 2) & ~1);
    if (bw[q] > maxw_all) maxw_all = bw[q];
    if (is_nd[q]) { off[q + 1] = off[q]; continue; }     /* no band records: sparse factor */
    off[q + 1] = off[q] + (long long)nrows[q] * reclen;
    if (bw[q] > maxw) maxw = bw[q];
  }
  s->max_bw = maxw_all;
  if (!rc && bj_wide_window(maxw) > 4096)
    rc = PA_FAIL("block-Jacobi: a diagonal block has bandwidth %d after reordering; the workgroup-resident "
                 "solve supports up to 4032 -- use more (smaller) subdomains", maxw);
  size_t tot = (size_t)off[np];
  const size_t pad = 256; /* the last LDS-DMA piece of a chunk may read up to 1 KiB past it */
  /* device arrays first: the factors are written in place, by the factorisation kernel for
   * the narrow blocks and by per-block uploads for the ones factored on the host */
  if (!rc) {
    s->d_invd_f = (double*)pa_rt_malloc((size_t)(m ? m : 1) * sizeof(double));
    s->d_invd_b = (double*)pa_rt_malloc((size_t)(m ? m : 1) * sizeof(double));
    s->d_Lf = (double*)pa_rt_malloc((tot + pad) * sizeof(double));
    s->d_Lb = (double*)pa_rt_malloc((tot + pad) * sizeof(double));
    if (!s->d_Lf || !s->d_Lb || !s->d_invd_f || !s->d_invd_b ||
        pa_rt_memset(s->d_Lf, 0, (tot + pad) * sizeof(double)) || pa_rt_memset(s->d_Lb, 0, (tot + pad) * sizeof(double)))
      rc = PA_FAIL("allocating %zu factor entries on the device failed: %s", tot, pa_rt_error());
  }
  int ndev = 0, nnd = 0;
  for (int q = 0; q < np; ++q) { if (is_nd[q]) ++nnd; else if (dev_factor) ++ndev; }
  if (!rc && ndev + nnd < np) {
    /* host-factored blocks: runs of consecutive blocks (up to 64 MiB of records) are laid out
     * by the host threads into a staging buffer (256 MiB, or one block if larger) and go to
     * the device in one copy each */
    const size_t cap = (size_t)32 << 20;                   /* doubles: 256 MiB per staging buffer */
    size_t sf_cap = 0;
    double* sf = NULL; double* sg = NULL;
    int q = 0;
    while (q < np && !rc) {
      if (dev_factor || is_nd[q]) { ++q; continue; }
      int q1 = q;
      while (q1 < np && !dev_factor && !is_nd[q1] && (q1 == q || (size_t)(off[q1 + 1] - off[q]) <= cap)) ++q1;
      size_t len = (size_t)(off[q1] - off[q]);
      if (len > sf_cap) {
        sf_cap = len;
        sf = (double*)realloc(sf, sf_cap * sizeof(double));
        sg = (double*)realloc(sg, sf_cap * sizeof(double));
        if (!sf || !sg) { rc = PA_FAIL("out of host memory for %zu factor entries", sf_cap); break; }
      }
      memset(sf, 0, len * sizeof(double));
      memset(sg, 0, len * sizeof(double));
      /* work items = slabs of 256 steps of one block, so that a run of few large blocks
       * still keeps every host thread busy */
      int nitem = 0;
      for (int x = q; x < q1; ++x) nitem += (nrows[x] + 255) / 256;
      int* item_part = (int*)malloc((nitem ? nitem : 1) * sizeof(int));
      int* item_j0 = (int*)malloc((nitem ? nitem : 1) * sizeof(int));
      nitem = 0;
      for (int x = q; x < q1; ++x)
        for (int j0 = 0; j0 < nrows[x]; j0 += 256) { item_part[nitem] = x; item_j0[nitem++] = j0; }
#pragma omp parallel for num_threads(pa_host_threads()) schedule(dynamic, 4)
      for (int it = 0; it < nitem; ++it) {
        int x = item_part[it];
        int b = nrows[x], w = bw[x], r0 = row0[x];
        size_t ld = (size_t)w + 1;
        int wide = w > wide_from;
        size_t reclen = wide ? (size_t)bj_wide_window(w) : (size_t)((w + 2) & ~1);
        const double* band = bands[x];
        double* f = sf + (off[x] - off[q]);
        double* g = sg + (off[x] - off[q]);
        int j1 = item_j0[it] + 256 < b ? item_j0[it] + 256 : b;
        for (int j = item_j0[it]; j < j1; ++j) {
          int jr = b - 1 - j;
          invd_f[r0 + j] = 1.0 / band[(size_t)j * ld];
          invd_b[r0 + j] = 1.0 / band[(size_t)jr * ld];
          for (int dd = 1; dd <= w; ++dd) {
            if (wide) { /* window-slot order, pre-divided by the pivot */
              if (j + dd < b) f[(size_t)j * reclen + (size_t)(j + dd) % reclen] = band[(size_t)(j + dd) * ld + dd] * invd_f[r0 + j];
              if (jr - dd >= 0) g[(size_t)j * reclen + (size_t)(j + dd) % reclen] = band[(size_t)jr * ld + dd] * invd_b[r0 + j];
            } else {    /* [L(j+1..j+w, j) / L(j,j) | 0] (see kernels.hip: bj_block) */
              f[(size_t)j * reclen + dd - 1] = (j + dd < b) ? band[(size_t)(j + dd) * ld + dd] * invd_f[r0 + j] : 0.0;
              g[(size_t)j * reclen + dd - 1] = (jr - dd >= 0) ? band[(size_t)jr * ld + dd] * invd_b[r0 + j] : 0.0;
            }
          }
        }
      }
      free(item_part); free(item_j0);
      if (pa_rt_h2d(s->d_Lf + off[q], sf, len * sizeof(double)) || pa_rt_h2d(s->d_Lb + off[q], sg, len * sizeof(double)))
        rc = PA_FAIL("uploading the block factors failed: %s", pa_rt_error());
      q = q1;
    }
    free(sf); free(sg);
  }
  if (!rc && (pa_rt_h2d(s->d_invd_f, invd_f, (size_t)m * sizeof(double)) ||
              pa_rt_h2d(s->d_invd_b, invd_b, (size_t)m * sizeof(double))))
    rc = PA_FAIL("uploading the block factors failed: %s", pa_rt_error());
  /* classes */
  if (!rc) {
    int* cls = (int*)malloc(np * sizeof(int));
    s->nclass = 0;
    for (int q = 0; q < np; ++q) {
      int R = (bw[q] + 127) / 64, c;
      cls[q] = -1;
      if (is_nd[q]) continue;
      if (bw[q] > wide_from) { /* wide classes: -(register sets per lane) */
        int W = bj_wide_window(bw[q]);
        R = W <= 1024 ? -1 : (W <= 2048 ? -2 : -4);
      }
      for (c = 0; c < s->nclass; ++c) if (s->class_R[c] == R) break;
      if (c == s->nclass) { s->class_R[c] = R; s->class_count[c] = 0; s->class_wmax[c] = 0; s->class_bmax[c] = 0; s->nclass++; }
      cls[q] = c; s->class_count[c]++;
      if (bw[q] > s->class_wmax[c]) s->class_wmax[c] = bw[q];
      if (nrows[q] > s->class_bmax[c]) s->class_bmax[c] = nrows[q];
    }
    for (int c = 0; c < s->nclass && !rc; ++c) {
      int* list = (int*)malloc(s->class_count[c] * sizeof(int));
      int n = 0;
      for (int q = 0; q < np; ++q) if (cls[q] == c) list[n++] = q;
      s->class_list[c] = (int*)pa_rt_malloc(n * sizeof(int));
      rc = !s->class_list[c] || pa_rt_h2d(s->class_list[c], list, n * sizeof(int));
      s->class_list_c[c] = s->class_list[c];
      free(list);
    }
    free(cls);
    if (rc) rc = PA_FAIL("uploading block lists failed: %s", pa_rt_error());
  }
  if (!rc) {
    s->d_row0 = (int*)pa_rt_malloc(np * sizeof(int));
    s->d_nrows = (int*)pa_rt_malloc(np * sizeof(int));
    s->d_bw = (int*)pa_rt_malloc(np * sizeof(int));
    s->d_off = (long long*)pa_rt_malloc((np + 1) * sizeof(long long));
    s->d_map_f = (int*)pa_rt_malloc((size_t)(m ? m : 1) * sizeof(int));
    s->d_map_b = (int*)pa_rt_malloc((size_t)(m ? m : 1) * sizeof(int));
    int bad = !s->d_row0 || !s->d_nrows || !s->d_bw || !s->d_off || !s->d_map_f || !s->d_map_b;
    bad = bad || pa_rt_h2d(s->d_row0, row0, np * sizeof(int)) || pa_rt_h2d(s->d_nrows, nrows, np * sizeof(int)) ||
          pa_rt_h2d(s->d_bw, bw, np * sizeof(int)) || pa_rt_h2d(s->d_off, off, (np + 1) * sizeof(long long)) ||
          pa_rt_h2d(s->d_map_f, map_f, (size_t)m * sizeof(int)) || pa_rt_h2d(s->d_map_b, map_b, (size_t)m * sizeof(int));
    if (bad) rc = PA_FAIL("uploading the block factors failed: %s", pa_rt_error());
  }
  if (!rc && ndev > 0) {
    /* ship the assembled bands, factor and lay out on the device: bands up to dev_wmax with
     * the LDS-window kernel (row-major band), wider ones with the blocked kernel
     * (diagonal-major band, factored in place) */
    long long* boff = (long long*)malloc((np + 1) * sizeof(long long));
    int* slist = (int*)malloc(np * sizeof(int));
    int* blist = (int*)malloc(np * sizeof(int));
    size_t btot = 0;
    int ns = 0, nbig = 0, wsmall = 0, wbig = 0;
    for (int q = 0; q < np; ++q) {
      boff[q] = (long long)btot;
      if (is_nd[q]) continue;
      btot += (size_t)nrows[q] * (bw[q] + 1);
      if (bw[q] <= dev_wmax) { slist[ns++] = q; if (bw[q] > wsmall) wsmall = bw[q]; }
      else { blist[nbig++] = q; if (bw[q] > wbig) wbig = bw[q]; }
    }
    boff[np] = (long long)btot;
    double* d_band = (double*)pa_rt_malloc((btot ? btot : 1) * sizeof(double));
    long long* d_boff = (long long*)pa_rt_malloc((np + 1) * sizeof(long long));
    int* d_slist = (int*)pa_rt_malloc((ns ? ns : 1) * sizeof(int));
    int* d_blist = (int*)pa_rt_malloc((nbig ? nbig : 1) * sizeof(int));
    int* d_fail = (int*)pa_rt_malloc(sizeof(int));
    int fail = 0;
    if (!d_band || !d_boff || !d_slist || !d_blist || !d_fail)
      rc = PA_FAIL("allocating %zu band entries on the device failed: %s", btot, pa_rt_error());
    const int tr = getenv("PREALPS_SETUP_TRACE") != NULL;
    double t_tr = pa_wtime();
    const size_t cap = (size_t)32 << 20;
    double* hb = NULL;
    size_t hb_cap = 0;
    /* narrow bands: runs of consecutive blocks go up in one copy of at most 256 MiB; wide
     * bands: zero on the device, then only their entries are shipped and scattered */
    if (!rc && nbig > 0 && pa_rt_memset(d_band, 0, btot * sizeof(double))) rc = PA_FAIL("%s", pa_rt_error());
    for (int q = 0; q < np && !rc; ) {
      if (!bands[q]) { ++q; continue; }
      int q1 = q + 1;
      while (q1 < np && bands[q1] && (size_t)(boff[q1 + 1] - boff[q]) <= cap) ++q1;
      size_t len = (size_t)(boff[q1] - boff[q]);
      if (len > hb_cap) { hb_cap = len; hb = (double*)realloc(hb, hb_cap * sizeof(double)); }
      if (!hb) { rc = PA_FAIL("out of host memory for %zu band entries", len); break; }
#pragma omp parallel for num_threads(pa_host_threads()) schedule(dynamic, 16)
      for (int x = q; x < q1; ++x)
        memcpy(hb + (boff[x] - boff[q]), bands[x], (size_t)nrows[x] * (bw[x] + 1) * sizeof(double));
      if (pa_rt_h2d(d_band + boff[q], hb, len * sizeof(double))) rc = PA_FAIL("uploading the bands failed: %s", pa_rt_error());
      for (int x = q; x < q1; ++x) { free(bands[x]); bands[x] = NULL; }
      q = q1;
    }
    if (!rc && nbig > 0) {
      size_t ntot = 0;
      size_t* cbase = (size_t*)malloc((np + 1) * sizeof(size_t));
      for (int q = 0; q < np; ++q) { cbase[q] = ntot; ntot += coo_n[q]; }
      long long* go = (long long*)malloc((ntot ? ntot : 1) * sizeof(long long));
      double* gv = (double*)malloc((ntot ? ntot : 1) * sizeof(double));
      long long* d_go = (long long*)pa_rt_malloc((ntot ? ntot : 1) * sizeof(long long));
      double* d_gv = (double*)pa_rt_malloc((ntot ? ntot : 1) * sizeof(double));
      if (!go || !gv || !d_go || !d_gv) rc = PA_FAIL("out of memory for %zu band entries", ntot);
      if (!rc) {
#pragma omp parallel for num_threads(pa_host_threads()) schedule(dynamic, 1)
        for (int q = 0; q < np; ++q)
          for (size_t e = 0; e < coo_n[q]; ++e) { go[cbase[q] + e] = boff[q] + coo_off[q][e]; gv[cbase[q] + e] = coo_val[q][e]; }
        if (pa_rt_h2d(d_go, go, ntot * sizeof(long long)) || pa_rt_h2d(d_gv, gv, ntot * sizeof(double)) ||
            pa_k_scatter(ntot, d_go, d_gv, d_band) || pa_rt_sync())
          rc = PA_FAIL("assembling the wide bands on the device failed: %s", pa_rt_error());
      }
      pa_rt_free(d_go); pa_rt_free(d_gv); free(go); free(gv); free(cbase);
    }
    free(hb);
    if (tr) { fprintf(stderr, "[setup] band upload (%.1f GB)        %.3f s\n", 8e-9 * (double)btot, pa_wtime() - t_tr); t_tr = pa_wtime(); }
    if (!rc) {
      if (pa_rt_h2d(d_boff, boff, (np + 1) * sizeof(long long)) || pa_rt_h2d(d_slist, slist, ns * sizeof(int)) ||
          pa_rt_h2d(d_blist, blist, nbig * sizeof(int)) || pa_rt_h2d(d_fail, &fail, sizeof(int)) ||
          pa_k_bj_factor(d_slist, ns, wsmall, s->d_row0, s->d_nrows, s->d_bw, s->d_off, d_boff, d_band, s->d_Lf,
                         s->d_Lb, s->d_invd_f, s->d_invd_b, d_fail) ||
          pa_k_bj_factor_big(d_blist, nbig, wbig, wide_from, s->d_row0, s->d_nrows, s->d_bw, s->d_off, d_boff,
                             d_band, s->d_Lf, s->d_Lb, s->d_invd_f, s->d_invd_b, d_fail) ||
          pa_rt_d2h(&fail, d_fail, sizeof(int)))
        rc = PA_FAIL("factorising the diagonal blocks on the device failed: %s", pa_rt_error());
      else if (fail > 0)
        rc = PA_FAIL("diagonal block is not SPD (global row %d)", row_off + map_f[fail - 1] +
                     row0[part_of_local_row(row0, nrows, np, fail - 1)]);
    }
    if (tr) fprintf(stderr, "[setup] device factorisation + layout  %.3f s\n", pa_wtime() - t_tr);
    pa_rt_free(d_band); pa_rt_free(d_boff); pa_rt_free(d_slist); pa_rt_free(d_blist); pa_rt_free(d_fail);
    free(boff); free(slist); free(blist);
  }
  if (!rc && nnd > 0) {
    int* ndl = (int*)malloc((size_t)nnd * sizeof(int));
    int* grow0 = (int*)malloc((size_t)np * sizeof(int));
    int x = 0, nd_fail = -1;
    for (int q = 0; q < np; ++q) { grow0[q] = rowPos[op->part0 + q]; if (is_nd[q]) ndl[x++] = q; }
    int r2 = pa_nd_create(A, nnd, ndl, row0, nrows, grow0, m, &nd_fail);
    if (r2 == 2) rc = PA_FAIL("diagonal block is not SPD (global row %d)", row_off + nd_fail);
    else if (r2) rc = 1;
    free(ndl); free(grow0);
  }
  /* Second layouts of the narrow classes for panels of up to 4 columns, made on the device from the
   * plain forward records:
   *  - bj_g4.hip (default, PREALPS_BJ_G4=0 turns it off): ONE copy for both sweeps in selective-
   *    inversion form by groups of four pivots, for classes whose blocks have at most
   *    pa_bj_g4_max_rows() rows and bands up to pa_bj_g4_max_band();
   *  - k_bj_pairs (classes R = 2, 3 that bj_g4 does not take; PREALPS_BJ_PAIRS=0 turns it off): both
   *    sweeps' records in pairs of steps for k_bj_apply_pairs.
   * Same size per block either way: 8 ceil(b / 8) (w + 4) doubles per copy.  The plain records stay for
   * the 8- and 16-column kernels. */
  if (!rc) {
    const int want_g4 = !(getenv("PREALPS_BJ_G4") && atoi(getenv("PREALPS_BJ_G4")) == 0) && (long long)m * 16 < 2147483647LL;
    const int want_pairs = !(getenv("PREALPS_BJ_PAIRS") && atoi(getenv("PREALPS_BJ_PAIRS")) == 0);
    int any_g4 = 0, any_pairs = 0, cls_pairs[16];
    for (int c = 0; c < s->nclass; ++c) {
      s->class_g4[c] = want_g4 && s->class_R[c] > 0 && s->class_wmax[c] <= pa_bj_g4_max_band() &&
                        s->class_bmax[c] <= pa_bj_g4_max_rows();
      cls_pairs[c] = want_pairs && !s->class_g4[c] && (s->class_R[c] == 2 || s->class_R[c] == 3);
      any_g4 |= s->class_g4[c]; any_pairs |= cls_pairs[c];
    }
    long long* off2 = (long long*)calloc((size_t)np + 1, sizeof(long long));
    long long tot2 = 0;
    if (off2 && (any_g4 || any_pairs)) {
      for (int q = 0; q < np; ++q) {
        if (is_nd[q] || bw[q] > wide_from) continue;
        off2[q] = tot2;                                   /* (blocks of the other classes: unused) */
        tot2 += 8LL * ((nrows[q] + 7) / 8) * (bw[q] + 4);
      }
      s->d_off2 = (long long*)pa_rt_malloc(((size_t)np + 1) * sizeof(long long));
      if (!s->d_off2 || pa_rt_h2d(s->d_off2, off2, ((size_t)np + 1) * sizeof(long long)))
        rc = PA_FAIL("allocating the second sweep records failed: %s", pa_rt_error());
      if (!rc && any_g4) {
        s->d_Lg4 = (double*)pa_rt_malloc(((size_t)tot2 + 1024) * sizeof(double));
        if (!s->d_Lg4 || pa_rt_memset(s->d_Lg4, 0, ((size_t)tot2 + 1024) * sizeof(double)))
          rc = PA_FAIL("allocating the one-copy sweep records failed: %s", pa_rt_error());
        for (int c = 0; c < s->nclass && !rc; ++c)
          if (s->class_g4[c] &&
              pa_k_bj_g4_setup(s->class_list[c], s->class_count[c], s->d_nrows, s->d_bw, s->d_off, s->d_off2, s->d_Lf, s->d_Lg4))
            rc = PA_FAIL("k_bj_g4_setup failed");
        if (!rc) s->g4_bytes = 8.0 * (double)tot2;
      }
      if (!rc && any_pairs) {
        s->d_Lf2 = (double*)pa_rt_malloc(((size_t)tot2 + 1024) * sizeof(double));
        s->d_Lb2 = (double*)pa_rt_malloc(((size_t)tot2 + 1024) * sizeof(double));
        if (!s->d_Lf2 || !s->d_Lb2 ||
            pa_rt_memset(s->d_Lf2 + tot2, 0, 512 * sizeof(double)) || pa_rt_memset(s->d_Lb2 + tot2, 0, 512 * sizeof(double)))
          rc = PA_FAIL("allocating the paired sweep records failed: %s", pa_rt_error());
        for (int c = 0; c < s->nclass && !rc; ++c)
          if (cls_pairs[c])
            if (pa_k_bj_pairs(s->class_list[c], s->class_count[c], s->d_nrows, s->d_bw, s->d_off, s->d_off2, s->d_Lf, s->d_Lf2) ||
                pa_k_bj_pairs(s->class_list[c], s->class_count[c], s->d_nrows, s->d_bw, s->d_off, s->d_off2, s->d_Lb, s->d_Lb2))
              rc = PA_FAIL("k_bj_pairs failed");
        if (!rc) s->pairs_bytes = 2.0 * 8.0 * (double)tot2;
      }
      if (!rc && pa_rt_sync()) rc = PA_FAIL("%s", pa_rt_error());
    }
    free(off2);
  }
  for (int q = 0; q < np; ++q) { free(bands[q]); free(coo_off[q]); free(coo_val[q]); }   /* (NULL where already released) */
  free(bands); free(coo_off); free(coo_val); free(coo_n);
  free(row0); free(nrows); free(bw); free(off); free(map_f); free(map_b); free(invd_f); free(invd_b); free(is_nd);
  if (rc) { preAlps_BlockJacobiFree(); return rc; }
  s->factor_bytes = 2.0 * 8.0 * (double)tot + pa_nd_factor_bytes();
  s->nd_blocks = nnd;
  pa_bj_plan_t* pl = &s->plan;
  pl->nparts = np; pl->row0 = s->d_row0; pl->nrows = s->d_nrows; pl->bw = s->d_bw; pl->off = s->d_off;
  pl->map_f = s->d_map_f; pl->map_b = s->d_map_b; pl->Lf = s->d_Lf; pl->Lb = s->d_Lb;
  pl->invd_f = s->d_invd_f; pl->invd_b = s->d_invd_b;
  pl->Lf2 = s->d_Lf2; pl->Lb2 = s->d_Lb2; pl->off2 = s->d_off2;
  pl->Lg4 = s->d_Lg4; pl->class_g4 = s->class_g4; pl->class_bmax = s->class_bmax;
  pl->nclass = s->nclass; pl->class_R = s->class_R; pl->class_count = s->class_count;
  pl->class_wmax = s->class_wmax;
  pl->class_list = s->class_list_c;
  s->created = 1;
  g_bj_setup_s[1] = pa_wtime() - t_setup0;
  return 0;
}

/* B_out = M^-1 A_in on the A_in->info.n current columns; the output
 * descriptor takes the input's shape (cplm_kernels.c:819-828). */
int preAlps_BlockJacobiApply(CPLM_Mat_Dense_t* A_in, CPLM_Mat_Dense_t* B_out) {
  pa_bj_t* s = &g_bj;
  if (!s->created) return PA_FAIL("preconditioner not created");
  if (!A_in || !B_out || !A_in->val || !B_out->val) return PA_FAIL(" wrong test 'A_in->val != NULL && B_out->val != NULL'");
  int ts = pa_desc_stride(A_in);
  if (pa_desc_stride(B_out) != ts || A_in->info.m != s->m)
    return PA_FAIL("panel shapes do not match the preconditioner (m %d vs %d)", A_in->info.m, s->m);
  B_out->info.n = A_in->info.n; B_out->info.N = A_in->info.N;
  B_out->info.nval = B_out->info.m * B_out->info.n;
  pa_time_begin(PA_T_PRECOND);
  if (pa_k_bj_apply(&s->plan, ts, A_in->val, B_out->val)) return PA_FAIL("block-Jacobi kernel launch failed");
  if (s->nd_blocks > 0 && pa_nd_apply(ts, A_in->val, B_out->val)) return 1;
  pa_time_end(PA_T_PRECOND);
  return 0;
}

/* ---- generic handle (preAlps_preconditioner.c:20-76) ---------------------- */
int preAlps_PreconditionerCreate(PreAlps_preconditioner_t** precond, Prec_Type_t precond_type, void* data) {
  if (!precond) return PA_FAIL(" wrong test 'precond != NULL'");
  *precond = (PreAlps_preconditioner_t*)malloc(sizeof(PreAlps_preconditioner_t));
  if (!*precond) return PA_FAIL("Malloc fails for precond[].");
  (*precond)->side = LEFT_PREC;
  (*precond)->type = precond_type;
  (*precond)->data = data;
  return 0;
}

int preAlps_PreconditionerDestroy(PreAlps_preconditioner_t** precond) {
  if (precond && *precond) { free(*precond); *precond = NULL; }
  return 0;
}

int preAlps_PreconditionerMatApply(PreAlps_preconditioner_t* precond, CPLM_Mat_Dense_t* A_in,
                                   CPLM_Mat_Dense_t* B_out) {
  if (!precond) return PA_FAIL(" wrong test 'precond != NULL'");
  if (precond->type == PREALPS_BLOCKJACOBI) return preAlps_BlockJacobiApply(A_in, B_out);
  if (precond->type == PREALPS_NOPREC) {   /* CPLM_MatDenseCopy: B_out takes A_in's shape and values */
    if (!A_in || !B_out || !A_in->val || !B_out->val) return PA_FAIL(" wrong test 'A_in->val != NULL && B_out->val != NULL'");
    int ts = pa_desc_stride(A_in);
    if (pa_desc_stride(B_out) != ts || A_in->info.m != B_out->info.m)
      return PA_FAIL("panel shapes do not match (m %d vs %d)", A_in->info.m, B_out->info.m);
    B_out->info.n = A_in->info.n; B_out->info.N = A_in->info.N;
    B_out->info.nval = B_out->info.m * B_out->info.n;
    if (pa_rt_d2d(B_out->val, A_in->val, (size_t)A_in->info.m * ts * sizeof(double))) return PA_FAIL("%s", pa_rt_error());
    return 0;
  }
  return PA_FAIL("Unknown preconditioner: %d (LORASC / PRESC are not part of this library)", (int)precond->type);
}
