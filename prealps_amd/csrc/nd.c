/*
 * nd.c -- exact solves with LARGE diagonal blocks (few subdomains, thousands of rows each: the
 * reference's own regime, one block per rank, src/preconditioners/block_jacobi.c:26-63,93-109
 * with PARDISO's sparse Cholesky) by a sparse supernodal factorisation instead of a band:
 *
 *   ordering   nested dissection of the block's graph (partition.c: pa_nd_order); leaves and
 *              separators are the supernodes, in postorder;
 *   symbolic   the row structure of every supernode = the separator vertices of its ancestors
 *              that its subtree touches (merged up the tree);
 *   numeric    multifrontal Cholesky on the host threads (one block per thread, dense fronts,
 *              the update matrices handed from child to parent);
 *   storage    per supernode a dense trapezoid (n columns, n + m rows) in HBM in "selective
 *              inversion" form: the triangle holds T = (L_11 D^-1)^-1, the rows below
 *              -G = -(L_21 D^-1) T (D = diag L_11), so that the solves are products, not
 *              recurrences; twice -- column major for the forward sweep, row major for the
 *              backward sweep -- each sweep streams its copy once with coalesced loads;
 *   solve      level by level up the forest and down again (kernels.hip: k_nd_forward /
 *              k_nd_backward), one launch per level, t right-hand sides at once.
 *
 * Against the band factor of a 18^3-node elasticity block (17.5 k rows, band 1031: 281 MB in two
 * copies) this needs ~130 MB, and a level of the tree is thousands of independent workgroups
 * where the band solve ran on one CU per block.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pa_host.h"

typedef struct {
  int created;
  int nsn;                 /* supernodes of all ND blocks of this process */
  pa_nd_plan_t plan;
  /* device */
  int* d_n; int* d_m; int* d_ld; long long* d_offF; long long* d_offB; int* d_rows_off;
  int* d_coff; int* d_ccoff; int* d_rows; int* d_src; double* d_dinv; double* d_F; double* d_B;
  double* d_contrib; size_t contrib_rows; int contrib_ts;
  double* d_Y; int m_local; double inv_dev;
  int nlevel; int* f_count; int** f_front; int** f_row0; int* b_count; int** b_front; int** b_col0;
  double bytes;
} pa_nd_t;

static pa_nd_t g_nd;

double pa_nd_factor_bytes(void) { return g_nd.created ? g_nd.bytes : 0.0; }
int pa_nd_active(void) { return g_nd.created; }

void pa_nd_free(void) {
  pa_nd_t* s = &g_nd;
  pa_rt_free(s->d_n); pa_rt_free(s->d_m); pa_rt_free(s->d_ld); pa_rt_free(s->d_offF); pa_rt_free(s->d_offB);
  pa_rt_free(s->d_rows_off); pa_rt_free(s->d_coff); pa_rt_free(s->d_ccoff); pa_rt_free(s->d_rows); pa_rt_free(s->d_src);
  pa_rt_free(s->d_dinv); pa_rt_free(s->d_F); pa_rt_free(s->d_B); pa_rt_free(s->d_contrib);
  for (int i = 0; i < s->nlevel; ++i) {
    if (s->f_front) pa_rt_free(s->f_front[i]);
    if (s->f_row0) pa_rt_free(s->f_row0[i]);
    if (s->b_front) pa_rt_free(s->b_front[i]);
    if (s->b_col0) pa_rt_free(s->b_col0[i]);
  }
  free(s->f_count); free(s->f_front); free(s->f_row0); free(s->b_count); free(s->b_front); free(s->b_col0);
  pa_rt_free(s->d_Y);
  memset(s, 0, sizeof(*s));
}

/* ---- per-block symbolic + numeric work -------------------------------------------------------------- */
typedef struct {
  int b;                   /* rows of the block */
  int row0;                /* first local panel row of the block */
  pa_nd_tree_t tree;
  int* child;              /* 2 per supernode (-1: none) */
  int* height;
  int* m;                  /* rows below, per supernode */
  int** below;             /* sorted new indices of the rows below, per supernode */
  long long nF, nB;        /* doubles of the two panel copies of the whole block */
  long long rows_total;    /* sum of n + m */
  long long contrib_total; /* sum of m */
  /* CSC of the lower triangle in the new order */
  int* cp; int* ri; double* cv;
} nd_block_t;

static void nd_block_free(nd_block_t* B) {
  pa_nd_tree_free(&B->tree);
  if (B->below) for (int s = 0; s < B->tree.nsn; ++s) free(B->below[s]);
  free(B->below); free(B->child); free(B->height); free(B->m); free(B->cp); free(B->ri); free(B->cv);
  memset(B, 0, sizeof(*B));
}

static int cmp_int(const void* a, const void* b) {
  int x = *(const int*)a, y = *(const int*)b;
  return (x > y) - (x < y);
}

/* A = local row panel (global column ids), the block = rows [r0, r0 + b) x columns [g0, g0 + b) */
/* A supernode of ns > width columns is cut into pieces of equal width rounded up to a multiple of 64
 * (so a piece may exceed `width` by up to 63 columns); the count follows from the ROUNDED width, so
 * no piece is empty. */
static int chain_pieces(int ns, int width, int* w_out) {
  int np = (ns + width - 1) / width;
  int w = ((ns + np - 1) / np + 63) & ~63;
  if (w_out) *w_out = w;
  return (ns + w - 1) / w;
}

static int nd_symbolic(nd_block_t* B, const CPLM_Mat_CSR_t* A, int r0, int g0, int b, int leaf_rows) {
  memset(B, 0, sizeof(*B));
  B->b = b; B->row0 = r0;
  /* The diagonal block with a structurally symmetric pattern: entry (i, j) is taken from row i
   * where it is stored there, else from row j (a `general` MatrixMarket file may hold a symmetric
   * matrix with explicit zeros dropped on one side only; operator.c accepts those, and both the
   * dissection -- a separator must cut the graph in both directions -- and the lower triangle
   * below need the whole pattern). */
  int* lrp = (int*)calloc((size_t)b + 1, sizeof(int));
  if (!lrp) return 1;
  for (int i = 0; i < b; ++i)
    for (int k = A->rowPtr[r0 + i]; k < A->rowPtr[r0 + i + 1]; ++k) {
      int c = A->colInd[k];
      if (c >= g0 && c < g0 + b) { lrp[i + 1]++; if (c - g0 != i) lrp[c - g0 + 1]++; }
    }
  for (int i = 0; i < b; ++i) lrp[i + 1] += lrp[i];
  size_t cnt = (size_t)lrp[b];
  int* lci = (int*)malloc((cnt ? cnt : 1) * sizeof(int));
  double* lv = (double*)malloc((cnt ? cnt : 1) * sizeof(double));
  int* fillp = (int*)malloc(((size_t)b + 1) * sizeof(int));
  if (!lci || !lv || !fillp) { free(lrp); free(lci); free(lv); free(fillp); return 1; }
  memcpy(fillp, lrp, ((size_t)b + 1) * sizeof(int));
  /* own entries first (ascending columns), mirrored ones behind them: marked by ~column */
  for (int i = 0; i < b; ++i)
    for (int k = A->rowPtr[r0 + i]; k < A->rowPtr[r0 + i + 1]; ++k) {
      int c = A->colInd[k];
      if (c >= g0 && c < g0 + b) { lci[fillp[i]] = c - g0; lv[fillp[i]++] = A->val[k]; }
    }
  for (int i = 0; i < b; ++i)
    for (int k = A->rowPtr[r0 + i]; k < A->rowPtr[r0 + i + 1]; ++k) {
      int c = A->colInd[k];
      if (c >= g0 && c < g0 + b && c - g0 != i) { lci[fillp[c - g0]] = ~i; lv[fillp[c - g0]++] = A->val[k]; }
    }
  free(fillp);
  {
    /* per row: merge the two ascending runs, an own entry wins over its mirror image */
    int maxl = 0;
    for (int i = 0; i < b; ++i) if (lrp[i + 1] - lrp[i] > maxl) maxl = lrp[i + 1] - lrp[i];
    int* tc = (int*)malloc((size_t)(maxl ? maxl : 1) * sizeof(int));
    double* tv = (double*)malloc((size_t)(maxl ? maxl : 1) * sizeof(double));
    if (!tc || !tv) { free(lrp); free(lci); free(lv); free(tc); free(tv); return 1; }
    int out = 0, begin = 0;
    for (int i = 0; i < b; ++i) {
      int e = lrp[i + 1], p = begin, q = begin, l = 0;
      while (q < e && lci[q] >= 0) ++q;         /* [begin, q) own, [q, e) mirrored */
      int pe = q;
      while (p < pe || q < e) {
        int cp_ = p < pe ? lci[p] : 2147483647, cq = q < e ? ~lci[q] : 2147483647;
        if (cp_ <= cq) { tc[l] = cp_; tv[l++] = lv[p++]; if (cq == cp_) ++q; }
        else { tc[l] = cq; tv[l++] = lv[q++]; }
      }
      begin = e;
      lrp[i] = out;
      memcpy(lci + out, tc, (size_t)l * sizeof(int));
      memcpy(lv + out, tv, (size_t)l * sizeof(double));
      out += l;
    }
    lrp[b] = out;
    free(tc); free(tv);
  }
  int rc = pa_nd_order(b, lrp, lci, leaf_rows, &B->tree);
  if (rc) { free(lrp); free(lci); free(lv); return 1; }
#define ND_FREE_LOCAL() do { free(lrp); free(lci); free(lv); } while (0)
  /* Very wide supernodes (the top separators of blocks of 100 k rows and more) are cut into chains of
   * at most `width` columns (PREALPS_ND_WIDTH, 2048), each piece the only child of the next: it bounds
   * the work of inverting a pivot triangle and forming G (n^2 (n / 3 + m) per front) and the unused
   * square above the triangle in the panels.  Same entries, same arithmetic.  Measured on elasticity
   * 70^3, same box: 64 blocks of 17.5 k rows do not care (2.24 ms per apply at 512 and at 2048); 8
   * blocks of 128 k rows: 5.32 / 4.89 / 4.70 ms per apply and 0.59 / 0.60 / 0.63 s of numeric phase at
   * 512 / 1024 / 2048 (fewer levels = fewer launches), 4.5 % more memory at 2048. */
  {
    const char* we = getenv("PREALPS_ND_WIDTH");
    const int width = we ? atoi(we) : 2048;
    const int n0 = B->tree.nsn;
    int extra_tot = 0;
    int* base = (int*)malloc(((size_t)n0 + 1) * sizeof(int));
    if (!base) { ND_FREE_LOCAL(); return 1; }
    for (int s = 0; s < n0; ++s) {
      int ns = B->tree.first[s + 1] - B->tree.first[s];
      base[s] = s + extra_tot;
      if (width >= 64 && ns > width) extra_tot += chain_pieces(ns, width, NULL) - 1;
    }
    base[n0] = n0 + extra_tot;
    if (extra_tot > 0) {
      int n2 = n0 + extra_tot;
      int* first2 = (int*)malloc(((size_t)n2 + 1) * sizeof(int));
      int* parent2 = (int*)malloc((size_t)n2 * sizeof(int));
      if (!first2 || !parent2) { free(base); free(first2); free(parent2); ND_FREE_LOCAL(); return 1; }
      for (int s = 0; s < n0; ++s) {
        int c0 = B->tree.first[s], ns = B->tree.first[s + 1] - c0, np = base[s + 1] - base[s];
        int w = ns;                                            /* piece width: even shares, multiples of 64 */
        if (np > 1) chain_pieces(ns, width, &w);
        for (int k = 0; k < np; ++k) {
          int id = base[s] + k, lo = c0 + k * w;
          if (lo > c0 + ns) lo = c0 + ns;
          first2[id] = lo;
          parent2[id] = k + 1 < np ? id + 1 : (B->tree.parent[s] >= 0 ? base[B->tree.parent[s]] : -1);
        }
      }
      first2[n2] = B->tree.first[n0];
      free(B->tree.first); free(B->tree.parent);
      B->tree.first = first2; B->tree.parent = parent2; B->tree.nsn = n2;
    }
    free(base);
  }
  const int nsn = B->tree.nsn;
  const int* perm = B->tree.perm;
  int* ip = (int*)malloc((size_t)b * sizeof(int));
  int* sn_of = (int*)malloc((size_t)b * sizeof(int));
  B->child = (int*)malloc((size_t)2 * nsn * sizeof(int));
  B->height = (int*)calloc((size_t)nsn, sizeof(int));
  B->m = (int*)calloc((size_t)nsn, sizeof(int));
  B->below = (int**)calloc((size_t)nsn, sizeof(int*));
  B->cp = (int*)calloc((size_t)b + 1, sizeof(int));
  if (!ip || !sn_of || !B->child || !B->height || !B->m || !B->below || !B->cp) { free(ip); free(sn_of); ND_FREE_LOCAL(); return 1; }
  for (int i = 0; i < b; ++i) ip[perm[i]] = i;
  for (int s = 0; s < nsn; ++s) {
    B->child[2 * s] = B->child[2 * s + 1] = -1;
    for (int i = B->tree.first[s]; i < B->tree.first[s + 1]; ++i) sn_of[i] = s;
  }
  for (int s = 0; s < nsn; ++s) {
    int p = B->tree.parent[s];
    if (p < 0) continue;
    if (p <= s) { free(ip); free(sn_of); ND_FREE_LOCAL(); return 1; }       /* postorder violated */
    if (B->child[2 * p] < 0) B->child[2 * p] = s; else if (B->child[2 * p + 1] < 0) B->child[2 * p + 1] = s; else { free(ip); free(sn_of); ND_FREE_LOCAL(); return 1; }
    if (B->height[s] + 1 > B->height[p]) B->height[p] = B->height[s] + 1;
  }
  /* lower triangle by columns in the new order: column j = row perm[j] of the (symmetric) block */
  for (int j = 0; j < b; ++j) {
    int old = perm[j], c2 = 0;
    for (int k = lrp[old]; k < lrp[old + 1]; ++k) if (ip[lci[k]] >= j) ++c2;
    B->cp[j + 1] = B->cp[j] + c2;
  }
  B->ri = (int*)malloc((size_t)(B->cp[b] ? B->cp[b] : 1) * sizeof(int));
  B->cv = (double*)malloc((size_t)(B->cp[b] ? B->cp[b] : 1) * sizeof(double));
  if (!B->ri || !B->cv) { free(ip); free(sn_of); ND_FREE_LOCAL(); return 1; }
  {
    typedef struct { int r; double v; } rv_t;
    int maxc = 0;
    for (int j = 0; j < b; ++j) if (B->cp[j + 1] - B->cp[j] > maxc) maxc = B->cp[j + 1] - B->cp[j];
    rv_t* buf = (rv_t*)malloc((size_t)(maxc ? maxc : 1) * sizeof(rv_t));
    if (!buf) { free(ip); free(sn_of); ND_FREE_LOCAL(); return 1; }
    for (int j = 0; j < b; ++j) {
      int old = perm[j], l = 0;
      for (int k = lrp[old]; k < lrp[old + 1]; ++k)
        if (ip[lci[k]] >= j) { buf[l].r = ip[lci[k]]; buf[l].v = lv[k]; ++l; }
      for (int a = 1; a < l; ++a) { rv_t x = buf[a]; int q = a; while (q > 0 && buf[q - 1].r > x.r) { buf[q] = buf[q - 1]; --q; } buf[q] = x; }
      for (int a = 0; a < l; ++a) { B->ri[B->cp[j] + a] = buf[a].r; B->cv[B->cp[j] + a] = buf[a].v; }
    }
    free(buf);
  }
  ND_FREE_LOCAL();
  /* rows below every supernode: entries of its columns and of its children's lists beyond its last column */
  int* mark = (int*)malloc((size_t)b * sizeof(int));
  int* tmp = (int*)malloc((size_t)b * sizeof(int));
  if (!mark || !tmp) { free(ip); free(sn_of); free(mark); free(tmp); return 1; }
  for (int i = 0; i < b; ++i) mark[i] = -1;
  for (int s = 0; s < nsn; ++s) {
    int last = B->tree.first[s + 1], l = 0;
    for (int j = B->tree.first[s]; j < last; ++j)
      for (int k = B->cp[j]; k < B->cp[j + 1]; ++k) { int r = B->ri[k]; if (r >= last && mark[r] != s) { mark[r] = s; tmp[l++] = r; } }
    for (int c = 0; c < 2; ++c) {
      int ch = B->child[2 * s + c];
      if (ch < 0) continue;
      for (int k = 0; k < B->m[ch]; ++k) {
        int r = B->below[ch][k];
        if (r < B->tree.first[s]) { free(ip); free(sn_of); free(mark); free(tmp); return 2; }   /* not a separator tree */
        if (r >= last && mark[r] != s) { mark[r] = s; tmp[l++] = r; }
      }
    }
    qsort(tmp, l, sizeof(int), cmp_int);
    B->m[s] = l;
    B->below[s] = (int*)malloc((size_t)(l ? l : 1) * sizeof(int));
    if (!B->below[s]) { free(ip); free(sn_of); free(mark); free(tmp); return 1; }
    memcpy(B->below[s], tmp, (size_t)l * sizeof(int));
    int n = last - B->tree.first[s];
    int ld = PA_ND_LD(n + l), ldb = PA_ND_LD(n);
    B->nF += (long long)ld * n;
    B->nB += (long long)ldb * (n + l);
    B->rows_total += n + l;
    B->contrib_total += l;
  }
  free(ip); free(sn_of); free(mark); free(tmp);
  return 0;
}

/* Dense partial Cholesky of the first n columns of the f x f front (lower triangle, column
 * major, leading dimension f): L11 L11^T = F11, L21 = F21 L11^-T, F22 <- F22 - L21 L21^T.
 * Returns 0 or 1 + the failing column. */
__attribute__((target_clones("avx2,fma", "default")))
static int front_factor(int f, int n, double* F) {
  const int NB = 32;
  for (int jb = 0; jb < n; jb += NB) {
    int nb = n - jb < NB ? n - jb : NB;
    for (int j = jb; j < jb + nb; ++j) {
      double* cj = F + (size_t)j * f;
      for (int k = jb; k < j; ++k) {
        const double* ck = F + (size_t)k * f;
        const double l = ck[j];
        for (int i = j; i < f; ++i) cj[i] -= ck[i] * l;
      }
      if (!(cj[j] > 0.0)) return j + 1;
      const double d = sqrt(cj[j]), id = 1.0 / d;
      cj[j] = d;
      for (int i = j + 1; i < f; ++i) cj[i] *= id;
    }
    /* trailing columns c >= jb + nb: F[c:, c] -= sum_k F[c:, k] F[c, k] */
    for (int c = jb + nb; c < f; ++c) {
      double* cc = F + (size_t)c * f;
      for (int k = jb; k < jb + nb; ++k) {
        const double* ck = F + (size_t)k * f;
        const double l = ck[c];
        if (l == 0.0) continue;
        for (int i = c; i < f; ++i) cc[i] -= ck[i] * l;
      }
    }
  }
  return 0;
}

/* Selective inversion of a factored front, in place in the forward copy (column major, leading
 * dimension ld, columns already divided by their pivots: I + Lhat_11 on top, Lhat_21 below):
 *   triangle   <- strictly lower part of T = (I + Lhat_11)^-1 (unit lower triangular),
 *   rows below <- -G, G = Lhat_21 T.
 * With these the solves are products (kernels.hip): forward a = T w and contribution -= G w,
 * backward z_1 = T^T D^-1 y_1 - G^T z_2 -- exactly transposed operators, so the block solve stays
 * symmetric.  wk: n * n + n doubles.  Returns the largest entry of T (I + Lhat_11) - I. */
__attribute__((target_clones("avx2,fma", "default")))
static double nd_selinv(int n, int m, int ld, double* pf, double* wk) {
  double* Lc = wk;                 /* n x n copy of I + Lhat_11 (column major, ld n) */
  double* t = wk + (size_t)n * n;  /* one column of T */
  for (int j = 0; j < n; ++j) {
    for (int i = 0; i <= j; ++i) Lc[(size_t)j * n + i] = i == j ? 1.0 : 0.0;
    for (int i = j + 1; i < n; ++i) Lc[(size_t)j * n + i] = pf[(size_t)j * ld + i];
  }
  for (int j = 0; j < n; ++j) {
    for (int i = j + 1; i < n; ++i) t[i] = 0.0;
    t[j] = 1.0;
    for (int k = j; k < n; ++k) {
      const double tk = t[k];
      if (tk == 0.0) continue;
      const double* lk = Lc + (size_t)k * n;
      for (int i = k + 1; i < n; ++i) t[i] -= lk[i] * tk;
    }
    for (int i = j + 1; i < n; ++i) pf[(size_t)j * ld + i] = t[i];
  }
  /* T (I + Lhat) - I: T was built as a right inverse, column by column, so the product in the other
   * order is an independent check (the one in the building order cancels exactly) */
  double dev = 0.0;
  for (int j = 0; j < n; ++j)
    for (int i = j + 1; i < n; ++i) {
      double sm = Lc[(size_t)j * n + i];                       /* k = i: T(i,i) = 1 */
      for (int k = j + 1; k < i; ++k) sm += pf[(size_t)k * ld + i] * Lc[(size_t)j * n + k];
      sm += pf[(size_t)j * ld + i];                            /* k = j: Lc(j,j) = 1 */
      if (fabs(sm) > dev) dev = fabs(sm);
    }
  /* G(:, j) = Lhat_21(:, j) + sum_{k > j} T(k, j) Lhat_21(:, k): ascending j overwrites column j when no
   * later column needs it any more; four columns j at a time share the loads of column k */
  if (m > 0) {
    double* R = pf + n;            /* rows below: R[j * ld + i], i < m */
    for (int j0 = 0; j0 < n; j0 += 4) {
      const int nb = n - j0 < 4 ? n - j0 : 4;
      for (int c = 0; c < nb; ++c) {          /* the corner inside the group */
        double* gj = R + (size_t)(j0 + c) * ld;
        for (int k = j0 + c + 1; k < j0 + nb; ++k) {
          const double tk = pf[(size_t)(j0 + c) * ld + k];
          const double* rk = R + (size_t)k * ld;
          for (int i = 0; i < m; ++i) gj[i] += tk * rk[i];
        }
      }
      if (nb == 4) {
        double* restrict g0 = R + (size_t)j0 * ld; double* restrict g1 = g0 + ld;
        double* restrict g2 = g1 + ld; double* restrict g3 = g2 + ld;
        for (int i0 = 0; i0 < m; i0 += 512) {   /* row tiles: the four output tiles stay in L1 */
          const int i1 = i0 + 512 < m ? i0 + 512 : m;
          for (int k = j0 + 4; k < n; ++k) {
            const double t0 = pf[(size_t)j0 * ld + k], t1 = pf[(size_t)(j0 + 1) * ld + k];
            const double t2 = pf[(size_t)(j0 + 2) * ld + k], t3 = pf[(size_t)(j0 + 3) * ld + k];
            const double* restrict rk = R + (size_t)k * ld;
            for (int i = i0; i < i1; ++i) {
              const double v = rk[i];
              g0[i] += t0 * v; g1[i] += t1 * v; g2[i] += t2 * v; g3[i] += t3 * v;
            }
          }
        }
      } else {
        for (int c = 0; c < nb; ++c) {
          double* gj = R + (size_t)(j0 + c) * ld;
          for (int k = j0 + nb; k < n; ++k) {
            const double tk = pf[(size_t)(j0 + c) * ld + k];
            const double* rk = R + (size_t)k * ld;
            for (int i = 0; i < m; ++i) gj[i] += tk * rk[i];
          }
        }
      }
    }
    for (int j = 0; j < n; ++j) { double* gj = R + (size_t)j * ld; for (int i = 0; i < m; ++i) gj[i] = -gj[i]; }
  }
  return dev;
}

/* Multifrontal factorisation of one block into the two panel copies (host staging buffers hF, hB,
 * laid out supernode after supernode) and dinv (1 / L_jj per new index).  *fail = 1 + new index
 * of a non-positive pivot.  invert: 1 = panels in selective-inversion form (nd_selinv; what the
 * device kernels expect; *dev = largest deviation of an inverse), 0 = the plain factor (selfcheck). */
static double g_cpu_factor, g_cpu_selinv;   /* PREALPS_SETUP_TRACE: CPU seconds over all threads */
static int nd_numeric(const nd_block_t* B, double* hF, double* hB, double* dinv, int* fail, int invert, double* dev) {
  const int nsn = B->tree.nsn, b = B->b;
  double t_fac = 0.0, t_inv = 0.0;
  double** upd = (double**)calloc((size_t)nsn, sizeof(double*));     /* update matrices waiting for the parent */
  int* loc = (int*)malloc((size_t)b * sizeof(int));
  if (!upd || !loc) { free(upd); free(loc); return 1; }
  long long oF = 0, oB = 0;
  int rc = 0;
  *fail = 0;
  for (int s = 0; s < nsn && !rc; ++s) {
    const int c0 = B->tree.first[s], n = B->tree.first[s + 1] - c0, m = B->m[s], f = n + m;
    double* F = (double*)calloc((size_t)f * f + 1, sizeof(double));
    if (!F) { rc = 1; break; }
    for (int j = 0; j < n; ++j) loc[c0 + j] = j;
    for (int k = 0; k < m; ++k) loc[B->below[s][k]] = n + k;
    for (int j = 0; j < n; ++j)
      for (int k = B->cp[c0 + j]; k < B->cp[c0 + j + 1]; ++k) F[(size_t)j * f + loc[B->ri[k]]] = B->cv[k];
    for (int c = 0; c < 2; ++c) {
      int ch = B->child[2 * s + c];
      if (ch < 0) continue;
      const int mc = B->m[ch];
      const double* U = upd[ch];
      for (int a = 0; a < mc; ++a) {
        const int ja = loc[B->below[ch][a]];
        for (int r = a; r < mc; ++r) F[(size_t)ja * f + loc[B->below[ch][r]]] += U[(size_t)a * mc + r];   /* (lists are sorted: ja <= row) */
      }
      free(upd[ch]); upd[ch] = NULL;
    }
    double tt = pa_wtime();
    int bad = front_factor(f, n, F);
    t_fac += pa_wtime() - tt;
    if (bad) { *fail = c0 + bad; free(F); rc = 2; break; }
    /* panel copies: forward = columns divided by their pivot (unit diagonal implied), backward =
     * row major, the n pivot rows divided by THEIR pivot, the m rows below as they are */
    const int ld = PA_ND_LD(f), ldb = PA_ND_LD(n);
    double* pf = hF + oF; double* pb = hB + oB;
    memset(pf, 0, (size_t)ld * n * sizeof(double));
    memset(pb, 0, (size_t)ldb * f * sizeof(double));
    for (int j = 0; j < n; ++j) {
      const double* cj = F + (size_t)j * f;
      const double id = 1.0 / cj[j];
      dinv[c0 + j] = id;
      for (int i = j + 1; i < f; ++i) pf[(size_t)j * ld + i] = cj[i] * id;
    }
    for (int i = 0; i < f; ++i) {
      const double sc = i < n ? 1.0 / F[(size_t)i * f + i] : 1.0;
      const int kmax = i < n ? i : n;
      for (int k = 0; k < kmax; ++k) pb[(size_t)i * ldb + k] = F[(size_t)k * f + i] * sc;
    }
    if (invert) {
      double* wk = (double*)malloc(((size_t)n * n + n) * sizeof(double));
      if (!wk) { free(F); rc = 1; break; }
      tt = pa_wtime();
      const double d = nd_selinv(n, m, ld, pf, wk);
      t_inv += pa_wtime() - tt;
      if (dev && d > *dev) *dev = d;
      free(wk);
      for (int i = 0; i < f; ++i) {            /* the row-major copy: the same numbers */
        const int kmax = i < n ? i : n;
        for (int k = 0; k < kmax; ++k) pb[(size_t)i * ldb + k] = pf[(size_t)k * ld + i];
      }
    }
    oF += (long long)ld * n; oB += (long long)ldb * f;
    if (m > 0 && B->tree.parent[s] >= 0) {
      double* U = (double*)calloc((size_t)m * m, sizeof(double));
      if (!U) { free(F); rc = 1; break; }
      for (int a = 0; a < m; ++a) memcpy(U + (size_t)a * m + a, F + (size_t)(n + a) * f + n + a, (size_t)(m - a) * sizeof(double));
      upd[s] = U;
    }
    free(F);
  }
  for (int s = 0; s < nsn; ++s) free(upd[s]);
  free(upd); free(loc);
  if (getenv("PREALPS_SETUP_TRACE")) {
#pragma omp critical
    { g_cpu_factor += t_fac; g_cpu_selinv += t_inv; }
  }
  return rc;
}

/* ---- numeric phase on the device (nd_factor.hip) ---------------------------------------------------------- */
typedef struct { int* f; int* ti; int* tj; int n; } nd_tiles_t;

static void nd_tiles_free(nd_tiles_t* t) { pa_rt_free(t->f); pa_rt_free(t->ti); pa_rt_free(t->tj); memset(t, 0, sizeof(*t)); }

/* upload the (front, tile row, tile column) list of one level; lower = tiles of the f x f lower
 * triangle, otherwise the f x n panel */
static int nd_tiles_build(nd_tiles_t* t, const int* ids, int cnt, const int* h_n, const int* h_m, int lower) {
  long long tot = 0;
  for (int q = 0; q < cnt; ++q) {
    long long ft = (h_n[ids[q]] + h_m[ids[q]] + 63) / 64, nt = (h_n[ids[q]] + 63) / 64;
    tot += lower ? ft * (ft + 1) / 2 : ft * nt;
  }
  memset(t, 0, sizeof(*t));
  if (tot > 2147483000LL) return 1;
  int* hf = (int*)malloc((size_t)(tot ? tot : 1) * sizeof(int)); int* hi = (int*)malloc((size_t)(tot ? tot : 1) * sizeof(int));
  int* hj = (int*)malloc((size_t)(tot ? tot : 1) * sizeof(int));
  if (!hf || !hi || !hj) { free(hf); free(hi); free(hj); return 1; }
  long long x = 0;
  for (int q = 0; q < cnt; ++q) {
    int g = ids[q], ft = (h_n[g] + h_m[g] + 63) / 64, nt = (h_n[g] + 63) / 64;
    for (int ti = 0; ti < ft; ++ti)
      for (int tj = 0; tj < (lower ? ti + 1 : nt); ++tj) { hf[x] = g; hi[x] = ti; hj[x++] = tj; }
  }
  t->n = (int)tot;
  t->f = (int*)pa_rt_malloc((size_t)(tot ? tot : 1) * sizeof(int)); t->ti = (int*)pa_rt_malloc((size_t)(tot ? tot : 1) * sizeof(int));
  t->tj = (int*)pa_rt_malloc((size_t)(tot ? tot : 1) * sizeof(int));
  int bad = !t->f || !t->ti || !t->tj || pa_rt_h2d(t->f, hf, (size_t)tot * sizeof(int)) ||
            pa_rt_h2d(t->ti, hi, (size_t)tot * sizeof(int)) || pa_rt_h2d(t->tj, hj, (size_t)tot * sizeof(int));
  free(hf); free(hi); free(hj);
  if (bad) nd_tiles_free(t);
  return bad;
}

/* hipMalloc maps memory eagerly, tens of milliseconds per gigabyte: the level buffers of the fronts
 * are recycled through a small free list (best fit) instead of being returned and asked for again. */
typedef struct { void* p[64]; size_t bytes[64]; int n; } nd_pool_t;
static void* nd_pool_get(nd_pool_t* pool, size_t bytes, size_t* got) {
  int best = -1;
  for (int i = 0; i < pool->n; ++i)
    if (pool->bytes[i] >= bytes && (best < 0 || pool->bytes[i] < pool->bytes[best])) best = i;
  if (best >= 0) {
    void* p = pool->p[best]; *got = pool->bytes[best];
    pool->p[best] = pool->p[pool->n - 1]; pool->bytes[best] = pool->bytes[pool->n - 1]; --pool->n;
    return p;
  }
  *got = bytes;
  return pa_rt_malloc(bytes);
}
static void nd_pool_put(nd_pool_t* pool, void* p, size_t bytes) {
  if (!p) return;
  if (pool->n < 64) { pool->p[pool->n] = p; pool->bytes[pool->n++] = bytes; }
  else pa_rt_free(p);
}
static void nd_pool_drain(nd_pool_t* pool) {
  for (int i = 0; i < pool->n; ++i) pa_rt_free(pool->p[i]);
  pool->n = 0;
}

/* Factor all blocks level by level.  S holds the uploaded plan (n, m, ld, offsets, rows, src, the
 * forward chunk lists); B the symbolic structure of the blocks, sn0 their first supernode ids.
 * Returns 0, 1 (resources; PA_FAIL raised) or 2 (*fail_g / *fail_col: first non-positive pivot). */
static int nd_numeric_device(pa_nd_t* S, const nd_block_t* B, int nblk, const int* sn0, int nsn, const int* h_n,
                             const int* h_m, const int* h_ld, const int* h_rows_off, const int* h_height, int maxh,
                             long long totrows, int* fail_g, int* fail_col) {
  int rc = 0;
  const int trace = getenv("PREALPS_SETUP_TRACE") != NULL;
  double t0 = pa_wtime();
  if (pa_nd_chunk_rows() != 256) return PA_FAIL("block solve: chunk size and factor kernels disagree");
  int* h_child = (int*)malloc((size_t)2 * nsn * sizeof(int));
  int* h_parent_h = (int*)malloc((size_t)nsn * sizeof(int));        /* height of the parent, -1 for roots */
  int* h_newrow = (int*)malloc((size_t)(totrows ? totrows : 1) * sizeof(int));
  long long* h_acol0 = (long long*)malloc((size_t)nsn * sizeof(long long));
  unsigned long long* h_front = (unsigned long long*)calloc((size_t)nsn, sizeof(unsigned long long));
  long long ncp = 0, nnzA = 0;
  for (int x = 0; x < nblk; ++x) { ncp += B[x].b + 1; nnzA += B[x].cp[B[x].b]; }
  long long* h_acp = (long long*)malloc((size_t)ncp * sizeof(long long));
  if (!h_child || !h_parent_h || !h_newrow || !h_acol0 || !h_front || !h_acp) rc = PA_FAIL("out of host memory for the block factorisation");
  int *d_child = NULL, *d_newrow = NULL, *d_ari = NULL;
  long long *d_acol0 = NULL, *d_acp = NULL;
  double* d_acv = NULL;
  unsigned long long *d_front = NULL, *d_fail = NULL;
  nd_pool_t pool;
  pool.n = 0;
  size_t* lvl_bytes = (size_t*)calloc((size_t)maxh + 1, sizeof(size_t));
  void** lvl_buf = (void**)calloc((size_t)maxh + 1, sizeof(void*));
  int* lvl_maxpar = (int*)malloc(((size_t)maxh + 1) * sizeof(int));
  if (!lvl_buf || !lvl_maxpar || !lvl_bytes) rc = PA_FAIL("out of host memory for the block factorisation");
  if (!rc) {
    d_child = (int*)pa_rt_malloc((size_t)2 * nsn * sizeof(int)); d_newrow = (int*)pa_rt_malloc((size_t)(totrows ? totrows : 1) * sizeof(int));
    d_acol0 = (long long*)pa_rt_malloc((size_t)nsn * sizeof(long long)); d_acp = (long long*)pa_rt_malloc((size_t)ncp * sizeof(long long));
    d_ari = (int*)pa_rt_malloc((size_t)(nnzA ? nnzA : 1) * sizeof(int)); d_acv = (double*)pa_rt_malloc((size_t)(nnzA ? nnzA : 1) * sizeof(double));
    d_front = (unsigned long long*)pa_rt_malloc((size_t)nsn * sizeof(unsigned long long));
    d_fail = (unsigned long long*)pa_rt_malloc(2 * sizeof(unsigned long long));
    if (!d_child || !d_newrow || !d_acol0 || !d_acp || !d_ari || !d_acv || !d_front || !d_fail)
      rc = PA_FAIL("allocating the inputs of the block factorisation on the device failed: %s", pa_rt_error());
  }
  if (!rc) {
    long long cpo = 0, eo = 0;
    for (int h = 0; h <= maxh; ++h) lvl_maxpar[h] = -1;
    for (int x = 0; x < nblk && !rc; ++x) {
      const nd_block_t* Bx = &B[x];
      for (int j = 0; j <= Bx->b; ++j) h_acp[cpo + j] = eo + Bx->cp[j];
      for (int s = 0; s < Bx->tree.nsn; ++s) {
        int g = sn0[x] + s, c0 = Bx->tree.first[s], n = h_n[g], m = h_m[g];
        h_acol0[g] = cpo + c0;
        for (int c = 0; c < 2; ++c) h_child[2 * g + c] = Bx->child[2 * s + c] < 0 ? -1 : sn0[x] + Bx->child[2 * s + c];
        int par = Bx->tree.parent[s];
        h_parent_h[g] = par < 0 ? -1 : h_height[sn0[x] + par];
        if (h_parent_h[g] > lvl_maxpar[h_height[g]]) lvl_maxpar[h_height[g]] = h_parent_h[g];
        for (int j = 0; j < n; ++j) h_newrow[h_rows_off[g] + j] = c0 + j;
        for (int k = 0; k < m; ++k) h_newrow[h_rows_off[g] + n + k] = Bx->below[s][k];
      }
      long long ne = Bx->cp[Bx->b];
      if (ne > 0 && (pa_rt_h2d(d_ari + eo, Bx->ri, (size_t)ne * sizeof(int)) || pa_rt_h2d(d_acv + eo, Bx->cv, (size_t)ne * sizeof(double))))
        rc = PA_FAIL("uploading the blocks failed: %s", pa_rt_error());
      cpo += Bx->b + 1; eo += ne;
    }
    unsigned long long none[2] = {~0ULL, 0ULL};
    if (!rc && (pa_rt_h2d(d_child, h_child, (size_t)2 * nsn * sizeof(int)) || pa_rt_h2d(d_newrow, h_newrow, (size_t)totrows * sizeof(int)) ||
                pa_rt_h2d(d_acol0, h_acol0, (size_t)nsn * sizeof(long long)) || pa_rt_h2d(d_acp, h_acp, (size_t)ncp * sizeof(long long)) ||
                pa_rt_h2d(d_fail, none, sizeof(none))))
      rc = PA_FAIL("uploading the blocks failed: %s", pa_rt_error());
  }
  pa_ndf_args_t a;
  memset(&a, 0, sizeof(a));
  a.n = S->d_n; a.m = S->d_m; a.ld = S->d_ld; a.offF = S->d_offF; a.offB = S->d_offB; a.rows_off = S->d_rows_off;
  a.rows = S->d_rows; a.src = S->d_src; a.child = d_child; a.newrow = d_newrow; a.acol0 = d_acol0; a.acp = d_acp;
  a.ari = d_ari; a.acv = d_acv; a.front = d_front; a.ldf = S->d_ld; a.F = S->d_F; a.B = S->d_B; a.dinv = S->d_dinv; a.fail = d_fail;
  int* ids = (int*)malloc((size_t)(nsn ? nsn : 1) * sizeof(int));
  if (!ids && !rc) rc = PA_FAIL("out of host memory for the block factorisation");
  double front_gb_peak = 0.0, front_gb_now = 0.0, t_malloc = 0.0, t_free = 0.0, t_sync = 0.0;
  const double t_levels0 = pa_wtime();
  for (int h = 0; h <= maxh && !rc; ++h) {
    int cnt = 0, nmax = 0;
    size_t doubles = 0;
    for (int g = 0; g < nsn; ++g) if (h_height[g] == h) {
      ids[cnt++] = g;
      if (h_n[g] > nmax) nmax = h_n[g];
      doubles += (size_t)h_ld[g] * (size_t)(h_n[g] + h_m[g]);
    }
    if (!cnt) continue;
    double tm0 = pa_wtime();
    lvl_buf[h] = nd_pool_get(&pool, (doubles ? doubles : 1) * sizeof(double), &lvl_bytes[h]);
    t_malloc += pa_wtime() - tm0;
    if (!lvl_buf[h]) { rc = PA_FAIL("block factorisation: %.2f GB for the fronts of level %d: %s", 8e-9 * (double)doubles, h, pa_rt_error()); break; }
    front_gb_now += 8e-9 * (double)doubles;
    if (front_gb_now > front_gb_peak) front_gb_peak = front_gb_now;
    size_t off = 0;
    for (int q = 0; q < cnt; ++q) {
      h_front[ids[q]] = (unsigned long long)(size_t)((double*)lvl_buf[h] + off);
      off += (size_t)h_ld[ids[q]] * (size_t)(h_n[ids[q]] + h_m[ids[q]]);
    }
    int* d_ids = (int*)pa_rt_malloc((size_t)cnt * sizeof(int));
    nd_tiles_t ll, rc_t;
    memset(&ll, 0, sizeof(ll)); memset(&rc_t, 0, sizeof(rc_t));
    if (!d_ids || pa_rt_h2d(d_ids, ids, (size_t)cnt * sizeof(int)) || pa_rt_h2d(d_front, h_front, (size_t)nsn * sizeof(unsigned long long)) ||
        nd_tiles_build(&ll, ids, cnt, h_n, h_m, 1) || nd_tiles_build(&rc_t, ids, cnt, h_n, h_m, 0))
      rc = PA_FAIL("block factorisation: work lists of level %d: %s", h, pa_rt_error());
    const int* cf = S->f_front[h]; const int* cr = S->f_row0[h]; const int nch = S->f_count[h];
    if (!rc && pa_k_ndf_assemble(&a, ll.f, ll.ti, ll.tj, ll.n, d_ids, cnt)) rc = PA_FAIL("block factorisation: kernel launch failed");
    for (int jb = 0; jb < nmax && !rc; jb += 64)
      if (pa_k_ndf_potrf(&a, d_ids, cnt, jb) || pa_k_ndf_trsm(&a, cf, cr, nch, jb, 0) || pa_k_ndf_update(&a, ll.f, ll.ti, ll.tj, ll.n, jb, 0))
        rc = PA_FAIL("block factorisation: kernel launch failed");
    if (!rc && pa_k_ndf_pinit(&a, cf, cr, nch)) rc = PA_FAIL("block factorisation: kernel launch failed");
    for (int jb = ((nmax - 1) / 64) * 64; jb >= 0 && !rc; jb -= 64)
      if (pa_k_ndf_trsm(&a, cf, cr, nch, jb, 1) || pa_k_ndf_update(&a, rc_t.f, rc_t.ti, rc_t.tj, rc_t.n, jb, 1))
        rc = PA_FAIL("block factorisation: kernel launch failed");
    if (!rc && (pa_k_ndf_finalize(&a, rc_t.f, rc_t.ti, rc_t.tj, rc_t.n) || pa_k_ndf_check(&a, d_ids, cnt, nmax)))
      rc = PA_FAIL("block factorisation: kernel launch failed");
    tm0 = pa_wtime();
    if (!rc && pa_rt_sync()) rc = PA_FAIL("block factorisation of level %d failed: %s", h, pa_rt_error());
    t_sync += pa_wtime() - tm0;
    pa_rt_free(d_ids); nd_tiles_free(&ll); nd_tiles_free(&rc_t);
    /* fronts nobody above this level reads any more */
    for (int l = 0; l <= h; ++l)
      if (lvl_buf[l] && lvl_maxpar[l] <= h) {
        size_t dl = 0;
        for (int g = 0; g < nsn; ++g) if (h_height[g] == l) dl += (size_t)h_ld[g] * (size_t)(h_n[g] + h_m[g]);
        front_gb_now -= 8e-9 * (double)dl;
        tm0 = pa_wtime();
        nd_pool_put(&pool, lvl_buf[l], lvl_bytes[l]); lvl_buf[l] = NULL;
        t_free += pa_wtime() - tm0;
      }
  }
  if (!rc) {
    unsigned long long key[2] = {~0ULL, 0ULL};
    if (pa_rt_d2h(key, d_fail, sizeof(key))) rc = PA_FAIL("block factorisation: %s", pa_rt_error());
    else if (key[0] != ~0ULL && (key[0] & 0xffffffffULL) == 0xffffffffULL)
      rc = PA_FAIL("block factorisation: an entry of supernode %d has no row in its front (symbolic structure inconsistent)", (int)(key[0] >> 32));
    else if (key[0] != ~0ULL) { *fail_g = (int)(key[0] >> 32); *fail_col = (int)(key[0] & 0xffffffffULL); rc = 2; }
    else {
      double dv;
      memcpy(&dv, &key[1], sizeof(dv));
      S->inv_dev = dv;
      if (dv > 1e-6)
        fprintf(stderr, "[prealps_hip] warning: the pivot triangles of the sparse block factor are ill conditioned "
                        "(L (L^-1 1) off by %.1e); the block solve loses that much accuracy\n", dv);
    }
  }
  if (trace) fprintf(stderr, "[nd] numeric phase on the device: %.2f s (inputs %.2f s; levels: allocating fronts %.2f s, freeing %.2f s, "
                             "waiting for the kernels %.2f s), at most %.2f GB of fronts at a time\n",
                     pa_wtime() - t0, t_levels0 - t0, t_malloc, t_free, t_sync, front_gb_peak);
  for (int h = 0; lvl_buf && h <= maxh; ++h) pa_rt_free(lvl_buf[h]);
  nd_pool_drain(&pool);
  free(lvl_buf); free(lvl_bytes); free(lvl_maxpar); free(ids);
  pa_rt_free(d_child); pa_rt_free(d_newrow); pa_rt_free(d_acol0); pa_rt_free(d_acp); pa_rt_free(d_ari); pa_rt_free(d_acv);
  pa_rt_free(d_front); pa_rt_free(d_fail);
  free(h_child); free(h_parent_h); free(h_newrow); free(h_acol0); free(h_front); free(h_acp);
  return rc;
}

/* ---- build ------------------------------------------------------------------------------------------------- */
/* blocks: nblk local block ids q; row0[q] / nrows[q] local panel rows; grow0[q] global first row. */
int pa_nd_create(const CPLM_Mat_CSR_t* A, int nblk, const int* blocks, const int* row0, const int* nrows,
                 const int* grow0, int m_local, int* fail_row) {
  if (g_nd.created) pa_nd_free();
  if (nblk <= 0) return 0;
  pa_nd_t* S = &g_nd;
  const char* le = getenv("PREALPS_ND_LEAF");
  const int leaf_rows = le ? atoi(le) : 96;
  nd_block_t* B = (nd_block_t*)calloc((size_t)nblk, sizeof(nd_block_t));
  if (!B) return PA_FAIL("out of host memory");
  int rc = 0;
  *fail_row = -1;
  const int trace = getenv("PREALPS_SETUP_TRACE") != NULL;
  double t_phase = pa_wtime();
#pragma omp parallel for num_threads(pa_host_threads()) schedule(dynamic, 1)
  for (int x = 0; x < nblk; ++x) {
    int q = blocks[x];
    int r1 = nd_symbolic(&B[x], A, row0[q], grow0[q], nrows[q], leaf_rows);
    if (r1) {
#pragma omp critical
      { if (r1 > rc) rc = r1; }
    }
  }
  if (rc) { for (int x = 0; x < nblk; ++x) nd_block_free(&B[x]); free(B); return PA_FAIL("nested dissection of the diagonal blocks failed (%s)", rc == 2 ? "a separator does not separate: is the matrix structurally singular?" : "out of memory"); }
  if (trace) { fprintf(stderr, "[nd] ordering + symbolic phase of %d blocks: %.2f s\n", nblk, pa_wtime() - t_phase); t_phase = pa_wtime(); }
  /* global numbering of the supernodes and offsets */
  int nsn = 0, maxh = 0;
  long long totF = 0, totB = 0, totrows = 0, totc = 0;
  int* sn0 = (int*)malloc(((size_t)nblk + 1) * sizeof(int));
  long long* bF = (long long*)malloc(((size_t)nblk + 1) * sizeof(long long));
  long long* bB = (long long*)malloc(((size_t)nblk + 1) * sizeof(long long));
  for (int x = 0; x < nblk; ++x) {
    sn0[x] = nsn; bF[x] = totF; bB[x] = totB;
    nsn += B[x].tree.nsn; totF += B[x].nF; totB += B[x].nB; totrows += B[x].rows_total; totc += B[x].contrib_total;
    for (int s = 0; s < B[x].tree.nsn; ++s) if (B[x].height[s] > maxh) maxh = B[x].height[s];
  }
  sn0[nblk] = nsn; bF[nblk] = totF; bB[nblk] = totB;
  if (totrows > 2147483000LL || totc > 2147483000LL) rc = PA_FAIL("block solve: index space of the fronts exceeds int32");
  int* h_n = (int*)malloc((size_t)nsn * sizeof(int)); int* h_m = (int*)malloc((size_t)nsn * sizeof(int));
  int* h_ld = (int*)malloc((size_t)nsn * sizeof(int));
  long long* h_offF = (long long*)malloc((size_t)nsn * sizeof(long long));
  long long* h_offB = (long long*)malloc((size_t)nsn * sizeof(long long));
  int* h_rows_off = (int*)malloc((size_t)nsn * sizeof(int)); int* h_coff = (int*)malloc((size_t)nsn * sizeof(int));
  int* h_ccoff = (int*)malloc((size_t)2 * nsn * sizeof(int));
  int* h_rows = (int*)malloc((size_t)(totrows ? totrows : 1) * sizeof(int));
  int* h_src = (int*)malloc((size_t)2 * (totrows ? totrows : 1) * sizeof(int));
  int* h_height = (int*)malloc((size_t)nsn * sizeof(int));
  double* h_dinv = (double*)calloc((size_t)(m_local ? m_local : 1), sizeof(double));
  if (!h_n || !h_m || !h_ld || !h_offF || !h_offB || !h_rows_off || !h_coff || !h_ccoff || !h_rows || !h_src || !h_height || !h_dinv)
    rc = PA_FAIL("out of host memory for the block-solve plan");
  if (!rc) {
    long long ro = 0, co = 0;
    for (int x = 0; x < nblk; ++x) {
      const nd_block_t* Bx = &B[x];
      long long oF = bF[x], oB = bB[x];
      for (int s = 0; s < Bx->tree.nsn; ++s) {
        int g = sn0[x] + s, c0 = Bx->tree.first[s], n = Bx->tree.first[s + 1] - c0, m = Bx->m[s];
        h_n[g] = n; h_m[g] = m; h_ld[g] = PA_ND_LD(n + m); h_offF[g] = oF; h_offB[g] = oB;
        h_rows_off[g] = (int)ro; h_coff[g] = (int)co; h_height[g] = Bx->height[s];
        oF += (long long)h_ld[g] * n; oB += (long long)PA_ND_LD(n) * (n + m);
        for (int j = 0; j < n; ++j) h_rows[ro + j] = Bx->row0 + Bx->tree.perm[c0 + j];
        for (int k = 0; k < m; ++k) h_rows[ro + n + k] = Bx->row0 + Bx->tree.perm[Bx->below[s][k]];
        for (int r = 0; r < 2 * (n + m); ++r) h_src[2 * ro + r] = -1;
        ro += n + m; co += m;
      }
      /* where every front row finds its children's contributions (and their offsets) */
      for (int s = 0; s < Bx->tree.nsn; ++s) {
        int g = sn0[x] + s, c0 = Bx->tree.first[s], n = h_n[g];
        for (int c = 0; c < 2; ++c) {
          int ch = Bx->child[2 * s + c];
          h_ccoff[2 * g + c] = ch < 0 ? -1 : h_coff[sn0[x] + ch];
          if (ch < 0) continue;
          /* child's rows below are a subset of (our columns, our rows below): both sorted */
          int pos = 0;
          for (int k = 0; k < Bx->m[ch]; ++k) {
            int r = Bx->below[ch][k], lr;
            if (r < c0 + n) lr = r - c0;
            else { while (pos < Bx->m[s] && Bx->below[s][pos] < r) ++pos; lr = n + pos; }
            h_src[2 * ((long long)h_rows_off[g] + lr) + c] = k;
          }
        }
      }
    }
  }
  /* device arrays */
  if (!rc) {
    S->d_n = (int*)pa_rt_malloc((size_t)nsn * sizeof(int)); S->d_m = (int*)pa_rt_malloc((size_t)nsn * sizeof(int));
    S->d_ld = (int*)pa_rt_malloc((size_t)nsn * sizeof(int));
    S->d_offF = (long long*)pa_rt_malloc((size_t)nsn * sizeof(long long));
    S->d_offB = (long long*)pa_rt_malloc((size_t)nsn * sizeof(long long));
    S->d_rows_off = (int*)pa_rt_malloc((size_t)nsn * sizeof(int)); S->d_coff = (int*)pa_rt_malloc((size_t)nsn * sizeof(int));
    S->d_ccoff = (int*)pa_rt_malloc((size_t)2 * nsn * sizeof(int));
    S->d_rows = (int*)pa_rt_malloc((size_t)(totrows ? totrows : 1) * sizeof(int));
    S->d_src = (int*)pa_rt_malloc((size_t)2 * (totrows ? totrows : 1) * sizeof(int));
    S->d_dinv = (double*)pa_rt_malloc((size_t)(m_local ? m_local : 1) * sizeof(double));
    S->d_F = (double*)pa_rt_malloc((size_t)(totF + 64) * sizeof(double));
    S->d_B = (double*)pa_rt_malloc((size_t)(totB + 64) * sizeof(double));
    if (!S->d_n || !S->d_m || !S->d_ld || !S->d_offF || !S->d_offB || !S->d_rows_off || !S->d_coff || !S->d_ccoff ||
        !S->d_rows || !S->d_src || !S->d_dinv || !S->d_F || !S->d_B)
      rc = PA_FAIL("allocating %.2f GB of block factors on the device failed: %s", 8e-9 * (double)(totF + totB), pa_rt_error());
  }
  /* numeric factorisation: on the device (nd_factor.hip, after the plan is uploaded, below) or, with
   * PREALPS_ND_NUMERIC=host, block after block on the host threads, each block uploaded when done */
  const char* nume = getenv("PREALPS_ND_NUMERIC");
  const int numeric_on_host = nume && !strcmp(nume, "host");
  if (!rc && numeric_on_host) {
    int fail_new = 0, fail_blk = -1;
    double inv_dev = 0.0;
#pragma omp parallel for num_threads(pa_host_threads()) schedule(dynamic, 1)
    for (int x = 0; x < nblk; ++x) {
      if (rc) continue;
      double* hF = (double*)malloc((size_t)(B[x].nF ? B[x].nF : 1) * sizeof(double));
      double* hB = (double*)malloc((size_t)(B[x].nB ? B[x].nB : 1) * sizeof(double));
      double* di = (double*)malloc((size_t)B[x].b * sizeof(double));
      double dv = 0.0;
      int fail = 0, r2 = (!hF || !hB || !di) ? 1 : nd_numeric(&B[x], hF, hB, di, &fail, 1, &dv);
      if (!r2) {
        for (int i = 0; i < B[x].b; ++i) h_dinv[B[x].row0 + B[x].tree.perm[i]] = di[i];
#pragma omp critical
        {
          if (dv > inv_dev) inv_dev = dv;
          if (pa_rt_h2d(S->d_F + bF[x], hF, (size_t)B[x].nF * sizeof(double)) ||
              pa_rt_h2d(S->d_B + bB[x], hB, (size_t)B[x].nB * sizeof(double))) r2 = 3;
        }
      }
      if (r2) {
#pragma omp critical
        { if (!rc) { rc = r2; fail_new = fail; fail_blk = x; } }
      }
      free(hF); free(hB); free(di);
    }
    if (rc == 2) { *fail_row = B[fail_blk].row0 + B[fail_blk].tree.perm[fail_new - 1]; }
    else if (rc) rc = PA_FAIL("factorising the large diagonal blocks failed (%s)", rc == 3 ? pa_rt_error() : "out of host memory");
    S->inv_dev = inv_dev;
    if (trace) {
      fprintf(stderr, "[nd] numeric phase (factor, selective inversion, upload): %.2f s; CPU seconds over the threads: %.2f in the dense partial factorisations, %.2f in the inversions\n",
              pa_wtime() - t_phase, g_cpu_factor, g_cpu_selinv);
      g_cpu_factor = g_cpu_selinv = 0.0;
      t_phase = pa_wtime();
    }
    if (!rc && inv_dev > 1e-6)
      fprintf(stderr, "[prealps_hip] warning: the pivot triangles of the sparse block factor are ill conditioned "
                      "(inverse off by %.1e); the block solve loses that much accuracy\n", inv_dev);
  }
  if (!rc) {
    int bad = pa_rt_h2d(S->d_n, h_n, (size_t)nsn * sizeof(int)) || pa_rt_h2d(S->d_m, h_m, (size_t)nsn * sizeof(int)) ||
              pa_rt_h2d(S->d_ld, h_ld, (size_t)nsn * sizeof(int)) || pa_rt_h2d(S->d_offF, h_offF, (size_t)nsn * sizeof(long long)) ||
              pa_rt_h2d(S->d_offB, h_offB, (size_t)nsn * sizeof(long long)) ||
              pa_rt_h2d(S->d_rows_off, h_rows_off, (size_t)nsn * sizeof(int)) || pa_rt_h2d(S->d_coff, h_coff, (size_t)nsn * sizeof(int)) ||
              pa_rt_h2d(S->d_ccoff, h_ccoff, (size_t)2 * nsn * sizeof(int)) ||
              pa_rt_h2d(S->d_rows, h_rows, (size_t)totrows * sizeof(int)) || pa_rt_h2d(S->d_src, h_src, (size_t)2 * totrows * sizeof(int)) ||
              pa_rt_h2d(S->d_dinv, h_dinv, (size_t)m_local * sizeof(double));
    if (bad) rc = PA_FAIL("uploading the block-solve plan failed: %s", pa_rt_error());
  }
  /* launch lists, one level of the forest per launch: forward workgroups = (front, chunk of
   * pa_nd_chunk_rows() front rows), backward workgroups = (front, block of pa_nd_block_cols()
   * pivot columns) */
  if (!rc) {
    const int CH = pa_nd_chunk_rows(), CB = pa_nd_block_cols();
    S->nlevel = maxh + 1;
    S->f_count = (int*)calloc((size_t)S->nlevel, sizeof(int)); S->b_count = (int*)calloc((size_t)S->nlevel, sizeof(int));
    S->f_front = (int**)calloc((size_t)S->nlevel, sizeof(int*)); S->f_row0 = (int**)calloc((size_t)S->nlevel, sizeof(int*));
    S->b_front = (int**)calloc((size_t)S->nlevel, sizeof(int*)); S->b_col0 = (int**)calloc((size_t)S->nlevel, sizeof(int*));
    if (!S->f_count || !S->b_count || !S->f_front || !S->f_row0 || !S->b_front || !S->b_col0) rc = PA_FAIL("out of host memory for the block-solve plan");
    for (int h = 0; h <= maxh && !rc; ++h) {
      long long nf = 0, nb = 0;
      for (int g = 0; g < nsn; ++g) if (h_height[g] == h) { nf += (h_n[g] + h_m[g] + CH - 1) / CH; nb += (h_n[g] + CB - 1) / CB; }
      if (nf > 2147483000LL || nb > 2147483000LL) { rc = PA_FAIL("block solve: too many workgroups in one level"); break; }
      int* ff = (int*)malloc((size_t)(nf ? nf : 1) * sizeof(int)); int* fr = (int*)malloc((size_t)(nf ? nf : 1) * sizeof(int));
      int* bf = (int*)malloc((size_t)(nb ? nb : 1) * sizeof(int)); int* bc = (int*)malloc((size_t)(nb ? nb : 1) * sizeof(int));
      if (!ff || !fr || !bf || !bc) { free(ff); free(fr); free(bf); free(bc); rc = PA_FAIL("out of host memory for the block-solve plan"); break; }
      long long x = 0, y = 0;
      for (int g = 0; g < nsn; ++g) if (h_height[g] == h) {
        for (int r0 = 0; r0 < h_n[g] + h_m[g]; r0 += CH) { ff[x] = g; fr[x++] = r0; }
        for (int k0 = 0; k0 < h_n[g]; k0 += CB) { bf[y] = g; bc[y++] = k0; }
      }
      S->f_count[h] = (int)nf; S->b_count[h] = (int)nb;
      if (getenv("PREALPS_SETUP_TRACE")) {
        long long by = 0; int cnt = 0, nmx = 0, fmx = 0;
        for (int g = 0; g < nsn; ++g) if (h_height[g] == h) {
          ++cnt; by += (long long)h_ld[g] * h_n[g];
          if (h_n[g] > nmx) nmx = h_n[g];
          if (h_n[g] + h_m[g] > fmx) fmx = h_n[g] + h_m[g];
        }
        fprintf(stderr, "[nd] level %d: %d fronts (widest %d columns, tallest %d rows), %.1f MB per copy, %lld forward / %lld backward workgroups\n",
                h, cnt, nmx, fmx, 8e-6 * (double)by, nf, nb);
      }
      S->f_front[h] = (int*)pa_rt_malloc((size_t)(nf ? nf : 1) * sizeof(int)); S->f_row0[h] = (int*)pa_rt_malloc((size_t)(nf ? nf : 1) * sizeof(int));
      S->b_front[h] = (int*)pa_rt_malloc((size_t)(nb ? nb : 1) * sizeof(int)); S->b_col0[h] = (int*)pa_rt_malloc((size_t)(nb ? nb : 1) * sizeof(int));
      if (!S->f_front[h] || !S->f_row0[h] || !S->b_front[h] || !S->b_col0[h] ||
          pa_rt_h2d(S->f_front[h], ff, (size_t)nf * sizeof(int)) || pa_rt_h2d(S->f_row0[h], fr, (size_t)nf * sizeof(int)) ||
          pa_rt_h2d(S->b_front[h], bf, (size_t)nb * sizeof(int)) || pa_rt_h2d(S->b_col0[h], bc, (size_t)nb * sizeof(int)))
        rc = PA_FAIL("uploading the block-solve plan failed: %s", pa_rt_error());
      free(ff); free(fr); free(bf); free(bc);
    }
  }
  if (!rc && !numeric_on_host) {
    int fg = -1, fc = 0;
    rc = nd_numeric_device(S, B, nblk, sn0, nsn, h_n, h_m, h_ld, h_rows_off, h_height, maxh, totrows, &fg, &fc);
    if (rc == 2) {
      int x = 0;
      while (x + 1 < nblk && sn0[x + 1] <= fg) ++x;
      *fail_row = B[x].row0 + B[x].tree.perm[B[x].tree.first[fg - sn0[x]] + fc];
    }
  }
  free(h_n); free(h_m); free(h_ld); free(h_offF); free(h_offB); free(h_rows_off); free(h_coff); free(h_ccoff);
  free(h_rows); free(h_src); free(h_height); free(h_dinv); free(sn0); free(bF); free(bB);
  for (int x = 0; x < nblk; ++x) nd_block_free(&B[x]);
  free(B);
  if (rc) { pa_nd_free(); return rc == 2 ? 2 : 1; }
  S->nsn = nsn; S->contrib_rows = (size_t)totc; S->m_local = m_local; S->bytes = 8.0 * (double)(totF + totB);
  pa_nd_plan_t* pl = &S->plan;
  pl->n = S->d_n; pl->m = S->d_m; pl->ld = S->d_ld; pl->offF = S->d_offF; pl->offB = S->d_offB; pl->rows_off = S->d_rows_off;
  pl->coff = S->d_coff; pl->ccoff = S->d_ccoff; pl->rows = S->d_rows; pl->src = S->d_src; pl->dinv = S->d_dinv;
  pl->F = S->d_F; pl->B = S->d_B;
  pl->nlevel = S->nlevel; pl->f_count = S->f_count; pl->b_count = S->b_count;
  pl->f_front = (const int* const*)S->f_front; pl->f_row0 = (const int* const*)S->f_row0;
  pl->b_front = (const int* const*)S->b_front; pl->b_col0 = (const int* const*)S->b_col0;
  S->created = 1;
  return 0;
}

double pa_nd_inverse_deviation(void) { return g_nd.created ? g_nd.inv_dev : 0.0; }

/* out(rows of the ND blocks) = blockdiag^-1 in(...) for the ts-strided panels */
int pa_nd_apply(int ts, const double* in, double* out) {
  pa_nd_t* S = &g_nd;
  if (!S->created) return 0;
  if (S->contrib_ts < ts) {
    pa_rt_free(S->d_contrib);
    pa_rt_free(S->d_Y);
    S->d_contrib = (double*)pa_rt_malloc((S->contrib_rows ? S->contrib_rows : 1) * (size_t)ts * sizeof(double));
    S->d_Y = (double*)pa_rt_malloc((size_t)(S->m_local ? S->m_local : 1) * (size_t)ts * sizeof(double));
    if (!S->d_contrib || !S->d_Y) return PA_FAIL("block solve: scratch of %zu + %d rows: %s", S->contrib_rows, S->m_local, pa_rt_error());
    S->contrib_ts = ts;
  }
  S->plan.contrib = S->d_contrib;
  S->plan.Y = S->d_Y;
  if (pa_k_nd_apply(&S->plan, ts, in, out)) return PA_FAIL("block-solve kernel launch failed");
  return 0;
}

/* ---- host-side self check (no GPU): ordering, symbolic and numeric phases on one matrix -------------- */
/* Factors the n x n SPD matrix (CSR, full symmetric pattern, sorted or not) as ONE block exactly
 * like pa_nd_create does and reports: stats[0] supernodes, [1] doubles of one panel copy,
 * [2] rows of the largest front, [3] height of the tree, [4] ||L L^T x - A x|| / ||A x|| for a
 * fixed pseudo-random x with L rebuilt from the forward panels, [5] the largest relative
 * difference between the backward panels and the same entries of the forward panels, [6] the
 * largest entry of T (I + Lhat_11) - I and of G (I + Lhat_11) - Lhat_21 over all fronts (the
 * selective-inversion form the device kernels use, nd_selinv), [7] the widest supernode.
 * It multiplies with the factor, it does not solve: there is no CPU solve path in this library. */
int preAlps_hip_nd_selfcheck(int n, const int* rowPtr, const int* colInd, const double* val, int leaf_rows,
                             double* stats) {
  if (n < 1 || !rowPtr || !colInd || !val || !stats) return PA_FAIL("invalid arguments");
  CPLM_Mat_CSR_t A;
  memset(&A, 0, sizeof(A));
  A.rowPtr = (int*)rowPtr; A.colInd = (int*)colInd; A.val = (double*)val;
  nd_block_t B;
  int rc = nd_symbolic(&B, &A, 0, 0, n, leaf_rows > 0 ? leaf_rows : 96);
  if (rc) { nd_block_free(&B); return PA_FAIL("nested dissection failed (%s)", rc == 2 ? "a separator does not separate" : "out of memory"); }
  double* hF = (double*)malloc((size_t)(B.nF ? B.nF : 1) * sizeof(double));
  double* hB = (double*)malloc((size_t)(B.nB ? B.nB : 1) * sizeof(double));
  double* di = (double*)malloc((size_t)n * sizeof(double));
  double* x = (double*)malloc((size_t)n * sizeof(double));
  double* u = (double*)calloc((size_t)n, sizeof(double));
  double* w = (double*)calloc((size_t)n, sizeof(double));
  double* ax = (double*)calloc((size_t)n, sizeof(double));
  int fail = 0;
  if (!hF || !hB || !di || !x || !u || !w || !ax) rc = PA_FAIL("out of host memory");
  if (!rc) {
    rc = nd_numeric(&B, hF, hB, di, &fail, 0, NULL);
    if (rc == 2) rc = PA_FAIL("matrix is not SPD (row %d)", B.tree.perm[fail - 1]);
    else if (rc) rc = PA_FAIL("out of host memory");
  }
  if (!rc) {
    unsigned long long st = 88172645463325252ULL;
    for (int i = 0; i < n; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; x[i] = (double)(st >> 11) / 9007199254740992.0 - 0.5; }
    int maxf = 0, maxh = 0;
    long long oF = 0, oB = 0;
    double dmax = 0.0;
    for (int s = 0; s < B.tree.nsn; ++s) {       /* u = L^T x */
      int c0 = B.tree.first[s], ns = B.tree.first[s + 1] - c0, m = B.m[s], f = ns + m, ld = PA_ND_LD(f), ldb = PA_ND_LD(ns);
      if (f > maxf) maxf = f;
      if (B.height[s] > maxh) maxh = B.height[s];
      for (int j = 0; j < ns; ++j) {
        double ljj = 1.0 / di[c0 + j], acc = ljj * x[c0 + j];
        for (int i = j + 1; i < f; ++i) {
          int gi = i < ns ? c0 + i : B.below[s][i - ns];
          double lij = hF[oF + (size_t)j * ld + i] * ljj;
          acc += lij * x[gi];
          /* the same entry in the backward copy */
          double want = i < ns ? lij * di[c0 + i] : lij, got = hB[oB + (size_t)i * ldb + j];
          double d = fabs(got - want) / (fabs(want) + 1e-300);
          if (want != 0.0 && d > dmax) dmax = d;
        }
        u[c0 + j] = acc;
      }
      oF += (long long)ld * ns; oB += (long long)ldb * f;
    }
    oF = 0;
    for (int s = 0; s < B.tree.nsn; ++s) {       /* w = L u */
      int c0 = B.tree.first[s], ns = B.tree.first[s + 1] - c0, m = B.m[s], f = ns + m, ld = PA_ND_LD(f);
      for (int j = 0; j < ns; ++j) {
        double ljj = 1.0 / di[c0 + j];
        w[c0 + j] += ljj * u[c0 + j];
        for (int i = j + 1; i < f; ++i) {
          int gi = i < ns ? c0 + i : B.below[s][i - ns];
          w[gi] += hF[oF + (size_t)j * ld + i] * ljj * u[c0 + j];
        }
      }
      oF += (long long)ld * ns;
    }
    for (int j = 0; j < n; ++j)                  /* A x from the permuted lower triangle */
      for (int k = B.cp[j]; k < B.cp[j + 1]; ++k) {
        int i = B.ri[k];
        ax[i] += B.cv[k] * x[j];
        if (i != j) ax[j] += B.cv[k] * x[i];
      }
    double num = 0.0, den = 0.0;
    for (int i = 0; i < n; ++i) { num += (w[i] - ax[i]) * (w[i] - ax[i]); den += ax[i] * ax[i]; }
    if (getenv("PREALPS_SETUP_TRACE"))
      for (int s2 = B.tree.nsn - 1; s2 >= 0 && s2 >= B.tree.nsn - 40; --s2)
        if (B.height[s2] >= maxh - 2)
          fprintf(stderr, "[nd] supernode %d height %d: %d columns, %d rows below\n", s2, B.height[s2],
                  B.tree.first[s2 + 1] - B.tree.first[s2], B.m[s2]);
    stats[0] = B.tree.nsn; stats[1] = (double)B.nF; stats[2] = maxf; stats[3] = maxh;
    stats[4] = sqrt(num / (den > 0.0 ? den : 1.0)); stats[5] = dmax;
    /* the selective-inversion form the device works with: T must invert the pivot triangle, and
     * G (I + Lhat_11) must give Lhat_21 back */
    double dev = 0.0;
    int widest = 0;
    oF = 0;
    for (int s = 0; s < B.tree.nsn && !rc; ++s) {
      int ns = B.tree.first[s + 1] - B.tree.first[s], m = B.m[s], f = ns + m, ld = PA_ND_LD(f);
      double* pf = hF + oF;
      double* wk = (double*)malloc(((size_t)ns * ns + ns + (size_t)ns * (m ? m : 1)) * sizeof(double));
      if (!wk) { rc = PA_FAIL("out of host memory"); break; }
      double* L21 = wk + (size_t)ns * ns + ns;           /* Lhat_21 before it is overwritten */
      for (int j = 0; j < ns; ++j) memcpy(L21 + (size_t)j * m, pf + (size_t)j * ld + ns, (size_t)m * sizeof(double));
      double d = nd_selinv(ns, m, ld, pf, wk);
      if (d > dev) dev = d;
      /* wk[0 .. ns*ns) still holds I + Lhat_11 (column major, ld ns) */
      for (int j = 0; j < ns; ++j)
        for (int i = 0; i < m; ++i) {
          double sm = 0.0;                                /* (G (I + Lhat_11))(i, j) = sum_{k >= j} G(i,k) Lc(k,j) */
          for (int k = j; k < ns; ++k) sm += -pf[(size_t)k * ld + ns + i] * wk[(size_t)j * ns + k];
          sm -= L21[(size_t)j * m + i];
          if (fabs(sm) > dev) dev = fabs(sm);
        }
      free(wk);
      if (ns > widest) widest = ns;
      oF += (long long)ld * ns;
    }
    stats[6] = dev; stats[7] = widest;
  }
  free(hF); free(hB); free(di); free(x); free(u); free(w); free(ax);
  nd_block_free(&B);
  return rc;
}
