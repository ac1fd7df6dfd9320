/*
 * partition.c -- k-way partitioning of the adjacency graph of a sparse matrix into many
 * compact, equally heavy parts: the replacement for the reference's call of
 * METIS_PartGraphKway (utils/cplm_core/cplm_matcsr_core.c:394-457, driven from
 * utils/cplm_v0/cplm_v0_matcsr.c:114-167 and utils/operator.c:77-97).
 *
 * What the ECG path needs from the partition is not METIS' minimum edge cut but what the
 * block-Jacobi solve of this library lives on: thousands of small subdomains of equal size
 * (one wavefront each), compact so that their band after RCM is narrow and the iteration
 * count low, and numbered so that a contiguous range of part ids -- what one GPU owns -- is
 * itself a compact region (few halo rows).  Host code, no GPU needed.  Steps:
 *
 *   1. rows with identical column lists (the dofs of one node of a vector problem) are merged
 *      into one weighted vertex: the 3 dofs of a node are never split, the graph shrinks 3x;
 *   2. recursive bisection.  A sub-graph gets coordinates without any geometry: its hop
 *      distances to a handful of landmark vertices chosen far from each other (breadth-first
 *      searches).  The vertices are projected onto the principal axis of that point cloud --
 *      for a mesh-like graph the direction in which the piece is longest -- and cut at the
 *      weighted median, k/2 parts to one side, the rest to the other; a few boundary passes
 *      straighten the cut.  Balance is exact up to one vertex per cut;
 *   3. slivers a cut has separated from their part join the neighbour they touch most;
 *   4. parts are numbered in the order of the leaves of the bisection tree.
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pa_host.h"

typedef struct {
  int n;        /* vertices */
  int* xadj;    /* n + 1 */
  int* adj;     /* neighbours, no self loops, no duplicates */
  int* vw;      /* vertex weights (rows merged into the vertex) */
} graph_t;

static void graph_free(graph_t* g) { free(g->xadj); free(g->adj); free(g->vw); memset(g, 0, sizeof(*g)); }

static int cmp_int(const void* a, const void* b) {
  int x = *(const int*)a, y = *(const int*)b;
  return (x > y) - (x < y);
}

static inline uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

/* ---- 1. merge indistinguishable rows, build the vertex graph ------------------------------ */
/* cid[i] = vertex of row i.  With merge == 0 every row is its own vertex. */
static int build_graph(int N, const int* rp, const int* ci, int merge, int* cid, graph_t* g) {
  int* rep = (int*)malloc((size_t)N * sizeof(int));
  if (!rep) return 1;
  if (merge) {
    uint64_t* h = (uint64_t*)malloc((size_t)N * sizeof(uint64_t));
    int* stamp = (int*)malloc((size_t)N * sizeof(int));
    if (!h || !stamp) { free(rep); free(h); free(stamp); return 1; }
#pragma omp parallel for num_threads(pa_host_threads()) schedule(static)
    for (int i = 0; i < N; ++i) {
      uint64_t s = 0;
      for (int k = rp[i]; k < rp[i + 1]; ++k) s += mix64((uint64_t)ci[k] + 1);   /* order independent */
      h[i] = s; rep[i] = -1; stamp[i] = -1;
    }
    for (int i = 0; i < N; ++i) {
      if (rep[i] >= 0) continue;
      rep[i] = i;
      int len = rp[i + 1] - rp[i], marked = 0;
      for (int k = rp[i]; k < rp[i + 1]; ++k) {
        int j = ci[k];
        if (j <= i || rep[j] >= 0 || rp[j + 1] - rp[j] != len || h[j] != h[i]) continue;
        if (!marked) { for (int q = rp[i]; q < rp[i + 1]; ++q) stamp[ci[q]] = i; marked = 1; }
        int same = 1;
        for (int q = rp[j]; q < rp[j + 1] && same; ++q) same = stamp[ci[q]] == i;
        if (same) rep[j] = i;
      }
    }
    free(h); free(stamp);
  } else {
    for (int i = 0; i < N; ++i) rep[i] = i;
  }
  int n = 0;
  for (int i = 0; i < N; ++i) if (rep[i] == i) cid[i] = n++;
  for (int i = 0; i < N; ++i) cid[i] = cid[rep[i]];
  g->n = n;
  g->xadj = (int*)calloc((size_t)n + 1, sizeof(int));
  g->vw = (int*)calloc((size_t)n, sizeof(int));
  int* rows = (int*)malloc((size_t)n * sizeof(int));   /* representative row of each vertex */
  if (!g->xadj || !g->vw || !rows) { free(rep); free(rows); return 1; }
  for (int i = 0; i < N; ++i) { g->vw[cid[i]]++; if (rep[i] == i) rows[cid[i]] = i; }
  free(rep);
  int maxlen = 0;
  for (int v = 0; v < n; ++v) { int l = rp[rows[v] + 1] - rp[rows[v]]; if (l > maxlen) maxlen = l; }
  /* two passes (count, fill); per row: map, sort, unique */
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) {
      for (int v = 0; v < n; ++v) g->xadj[v + 1] += g->xadj[v];
      g->adj = (int*)malloc((size_t)(g->xadj[n] ? g->xadj[n] : 1) * sizeof(int));
      if (!g->adj) { free(rows); return 1; }
    }
#pragma omp parallel num_threads(pa_host_threads())
    {
      int* buf = (int*)malloc((size_t)(maxlen ? maxlen : 1) * sizeof(int));
#pragma omp for schedule(dynamic, 1024)
      for (int v = 0; v < n; ++v) {
        int r = rows[v], l = 0;
        int sorted = 1;
        for (int k = rp[r]; k < rp[r + 1]; ++k) {
          int c = cid[ci[k]];
          if (c == v) continue;
          if (l && c < buf[l - 1]) sorted = 0;
          buf[l++] = c;
        }
        /* (sorted columns + vertices numbered in row order: usually nothing to do) */
        if (!sorted) {
          if (l <= 64) {
            for (int a = 1; a < l; ++a) { int x = buf[a], b2 = a - 1; while (b2 >= 0 && buf[b2] > x) { buf[b2 + 1] = buf[b2]; --b2; } buf[b2 + 1] = x; }
          } else qsort(buf, l, sizeof(int), cmp_int);
        }
        int u = 0;
        for (int q = 0; q < l; ++q) if (q == 0 || buf[q] != buf[q - 1]) buf[u++] = buf[q];
        if (pass == 0) g->xadj[v + 1] = u;
        else memcpy(g->adj + g->xadj[v], buf, (size_t)u * sizeof(int));
      }
      free(buf);
    }
  }
  free(rows);
  return 0;
}

/* ---- 2. recursive bisection in a landmark embedding ---------------------------------------------- */
typedef struct { double key; int idx; } keyidx_t;
static int cmp_keyidx(const void* a, const void* b) {
  const keyidx_t* x = (const keyidx_t*)a; const keyidx_t* y = (const keyidx_t*)b;
  if (x->key < y->key) return -1;
  if (x->key > y->key) return 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}

/* The same order as qsort with cmp_keyidx, for arrays whose idx fields ascend on entry (they are
 * filled 0, 1, 2, ...): stable LSD radix sort on the bit pattern of the keys, 11 bits per pass,
 * passes whose digit is the same for all keys skipped.  qsort with a comparator took most of the
 * time of the large cuts (three sorts of 125 k pairs: 0.25 s of a 0.25 s cut). */
static void sort_keyidx(keyidx_t* a, keyidx_t* tmp, int n) {
  if (n < 256) { qsort(a, (size_t)n, sizeof(keyidx_t), cmp_keyidx); return; }
  unsigned long long* kb = (unsigned long long*)malloc((size_t)n * sizeof(unsigned long long));
  if (!kb) { qsort(a, (size_t)n, sizeof(keyidx_t), cmp_keyidx); return; }
  unsigned long long all_or = 0, all_and = ~0ULL;
  for (int i = 0; i < n; ++i) {
    double d = a[i].key + 0.0;                  /* -0.0 -> +0.0: equal keys, equal bits */
    unsigned long long u;
    memcpy(&u, &d, sizeof(u));
    u = (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
    kb[i] = u; all_or |= u; all_and &= u;
  }
  /* sort (kb, a) pairs: carry the payload index instead of the structs */
  int* ord = (int*)malloc((size_t)n * sizeof(int));
  int* ord2 = (int*)malloc((size_t)n * sizeof(int));
  if (!ord || !ord2) { free(kb); free(ord); free(ord2); qsort(a, (size_t)n, sizeof(keyidx_t), cmp_keyidx); return; }
  for (int i = 0; i < n; ++i) ord[i] = i;
  for (int sh = 0; sh < 64; sh += 11) {
    const unsigned long long mask = 0x7ffULL << sh;
    if (((all_or ^ all_and) & mask) == 0) continue;            /* this digit is the same everywhere */
    int cnt[2049];
    memset(cnt, 0, sizeof(cnt));
    for (int i = 0; i < n; ++i) ++cnt[((kb[ord[i]] >> sh) & 0x7ff) + 1];
    for (int b = 0; b < 2048; ++b) cnt[b + 1] += cnt[b];
    for (int i = 0; i < n; ++i) ord2[cnt[(kb[ord[i]] >> sh) & 0x7ff]++] = ord[i];
    int* t2 = ord; ord = ord2; ord2 = t2;
  }
  for (int i = 0; i < n; ++i) tmp[i] = a[ord[i]];
  memcpy(a, tmp, (size_t)n * sizeof(keyidx_t));
  free(kb); free(ord); free(ord2);
}

#define NLM 8      /* landmarks per sub-graph */

typedef struct {
  const graph_t* g;
  int* part;       /* result: part of every vertex */
  int* tag;        /* sub-graph membership: tag[v] == current id */
  int* loc;        /* position of a member vertex in the current list */
  int* queue;      /* BFS queue / scratch, n ints */
  int* dist;       /* NLM * n hop distances, list-local */
  keyidx_t* ki;    /* (projection, list-local index), sorted along the axis */
  keyidx_t* ki2;   /* scratch of the sort */
  char* side;      /* list-local */
  int* tmp;        /* n ints */
  int tag_store;   /* (the counter next_tag points at, in the root context) */
  int* next_tag;   /* shared by the contexts of one recursion: every cut takes a fresh id (atomic) */
  int cur_tag;
  int smooth;      /* sweeps of neighbour averaging applied to the projection */
  double* z;       /* 3 * n whitened principal coordinates, list-local */
} rb_t;

/* hop distances from `start` inside the current sub-graph; vertices of other components keep -1.
 * Returns the number of vertices reached; *last = the last one. */
static int rb_bfs(rb_t* c, const int* list, int len, int tag, int start, int* dist, int* last) {
  const graph_t* g = c->g;
  for (int i = 0; i < len; ++i) dist[i] = -1;
  int head = 0, tail = 0;
  c->queue[tail++] = start; dist[c->loc[start]] = 0;
  while (head < tail) {
    int u = c->queue[head++], du = dist[c->loc[u]];
    for (int q = g->xadj[u]; q < g->xadj[u + 1]; ++q) {
      int v = g->adj[q];
      if (c->tag[v] == tag && dist[c->loc[v]] < 0) { dist[c->loc[v]] = du + 1; c->queue[tail++] = v; }
    }
  }
  *last = c->queue[tail - 1];
  (void)list;
  return tail;
}

/* Cut the sub-graph `list` in two: side[i] (list-local) = 0 / 1, about `want` of the weight and
 * at least min0 (min1) vertices on side 0 (1).  Returns the number of vertices on side 0 and
 * its weight through *w0.  On return tag / loc still describe this list. */
static int rb_bisect(rb_t* c, const int* list, int len, long long wtot, long long want, int min0, int min1,
                     long long* w0) {
  const graph_t* g = c->g;
  const int tag = __atomic_fetch_add(c->next_tag, 1, __ATOMIC_RELAXED);
  for (int i = 0; i < len; ++i) { c->tag[list[i]] = tag; c->loc[list[i]] = i; }
  c->cur_tag = tag;
  /* landmarks by farthest-point sampling; a vertex another component hides from all landmarks
   * so far is infinitely far and becomes the next landmark */
  int nlm = 0, lm = list[0], last = list[0];
  int* mind = c->tmp;                         /* min distance to the landmarks so far */
  rb_bfs(c, list, len, tag, lm, c->dist, &last);
  lm = last;                                  /* a peripheral vertex of the first component */
  for (int i = 0; i < len; ++i) mind[i] = 1 << 30;
  while (nlm < NLM) {
    int* d = c->dist + (size_t)nlm * len;
    rb_bfs(c, list, len, tag, lm, d, &last);
    ++nlm;
    int far = -1, fard = -1;
    for (int i = 0; i < len; ++i) {
      int di = d[i] < 0 ? (1 << 29) : d[i];   /* unreachable: very far */
      if (di < mind[i]) mind[i] = di;
      if (mind[i] > fard) { fard = mind[i]; far = i; }
    }
    if (fard <= 0) break;
    lm = list[far];
  }
  /* unreachable distances -> (largest finite + 1) of that landmark, so that other components sit
   * beside the landmark's own one instead of dominating the axis */
  for (int j = 0; j < nlm; ++j) {
    int* d = c->dist + (size_t)j * len, mx = 0;
    for (int i = 0; i < len; ++i) if (d[i] > mx) mx = d[i];
    for (int i = 0; i < len; ++i) if (d[i] < 0) d[i] = mx + 1;
  }
  /* principal axis of the nlm-dimensional point cloud (weighted by vertex weight) */
  double mean[NLM], cov[NLM][NLM];
  for (int j = 0; j < nlm; ++j) {
    const int* d = c->dist + (size_t)j * len;
    double s = 0.0;
    for (int i = 0; i < len; ++i) s += (double)g->vw[list[i]] * d[i];
    mean[j] = s / (double)wtot;
  }
  for (int j = 0; j < nlm; ++j)
    for (int l = j; l < nlm; ++l) {
      const int* dj = c->dist + (size_t)j * len; const int* dl = c->dist + (size_t)l * len;
      double s = 0.0;
      for (int i = 0; i < len; ++i) s += (double)g->vw[list[i]] * (dj[i] - mean[j]) * (dl[i] - mean[l]);
      cov[j][l] = cov[l][j] = s;
    }
  /* eigenvectors of the covariance (cyclic Jacobi on the nlm x nlm matrix) */
  double ev[NLM][NLM], lam[NLM];
  for (int j = 0; j < nlm; ++j) for (int l = 0; l < nlm; ++l) ev[j][l] = j == l;
  for (int sweep = 0; sweep < 30; ++sweep) {
    double offd = 0.0;
    for (int p = 0; p < nlm; ++p) for (int q = p + 1; q < nlm; ++q) offd += cov[p][q] * cov[p][q];
    if (offd < 1e-24 * (1.0 + cov[0][0] * cov[0][0])) break;
    for (int p = 0; p < nlm; ++p)
      for (int q = p + 1; q < nlm; ++q) {
        if (fabs(cov[p][q]) < 1e-300) continue;
        double th = (cov[q][q] - cov[p][p]) / (2.0 * cov[p][q]);
        double tt = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
        double cs = 1.0 / sqrt(tt * tt + 1.0), sn = tt * cs;
        for (int k = 0; k < nlm; ++k) { double a = cov[k][p], b = cov[k][q]; cov[k][p] = cs * a - sn * b; cov[k][q] = sn * a + cs * b; }
        for (int k = 0; k < nlm; ++k) { double a = cov[p][k], b = cov[q][k]; cov[p][k] = cs * a - sn * b; cov[q][k] = sn * a + cs * b; }
        for (int k = 0; k < nlm; ++k) { double a = ev[k][p], b = ev[k][q]; ev[k][p] = cs * a - sn * b; ev[k][q] = sn * a + cs * b; }
      }
  }
  for (int j = 0; j < nlm; ++j) lam[j] = cov[j][j];
  int top[3] = {-1, -1, -1}, nd = 0;
  for (int r = 0; r < 3 && r < nlm; ++r) {
    int best = -1;
    for (int j = 0; j < nlm; ++j) {
      if (j == top[0] || j == top[1] || j == top[2]) continue;
      if (best < 0 || lam[j] > lam[best]) best = j;
    }
    if (best < 0 || !(lam[best] > 0.0) || (r > 0 && lam[best] < 0.5 * lam[top[0]])) break;
    top[r] = best; nd = r + 1;
  }
  /* The piece is as long in `nd` directions (a cube: three).  Among them the one to cut across is
   * the one along which the vertices are spread most evenly -- a box projected on one of its
   * axes is uniform, projected on a diagonal it is peaked -- i.e. the direction of least
   * kurtosis in the whitened coordinates: a cut parallel to a face instead of a diagonal one. */
  double wdir[3] = {1.0, 0.0, 0.0};
  double* z = c->z;
  if (nd >= 1) {
    for (int i = 0; i < len; ++i)
      for (int r = 0; r < nd; ++r) {
        double sm = 0.0;
        for (int j = 0; j < nlm; ++j) sm += ev[j][top[r]] * (c->dist[(size_t)j * len + i] - mean[j]);
        z[(size_t)r * len + i] = sm / sqrt(lam[top[r]] / (double)wtot);
      }
  }
  if (nd >= 2) {
    double M4[3][3][3][3];
    memset(M4, 0, sizeof(M4));
    for (int i = 0; i < len; ++i) {
      double zi[3] = {z[i], z[(size_t)len + i], nd > 2 ? z[(size_t)2 * len + i] : 0.0}, w = g->vw[list[i]];
      for (int a = 0; a < nd; ++a) for (int b2 = a; b2 < nd; ++b2) for (int c2 = b2; c2 < nd; ++c2) for (int d2 = c2; d2 < nd; ++d2)
        M4[a][b2][c2][d2] += w * zi[a] * zi[b2] * zi[c2] * zi[d2];
    }
    double best = 1e300;
    const int nsamp = nd == 2 ? 90 : 600;
    for (int q = 0; q < nsamp; ++q) {
      double w3[3];
      if (nd == 2) { double an = 3.14159265358979 * q / nsamp; w3[0] = cos(an); w3[1] = sin(an); w3[2] = 0.0; }
      else {   /* Fibonacci points on the upper hemisphere */
        double zc = (q + 0.5) / nsamp, rr = sqrt(1.0 - zc * zc), an = 2.39996322972865 * q;
        w3[0] = rr * cos(an); w3[1] = rr * sin(an); w3[2] = zc;
      }
      double kq = 0.0;
      for (int a = 0; a < nd; ++a) for (int b2 = a; b2 < nd; ++b2) for (int c2 = b2; c2 < nd; ++c2) for (int d2 = c2; d2 < nd; ++d2) {
        /* multiplicity of the sorted index tuple */
        int cnt[3] = {0, 0, 0}; cnt[a]++; cnt[b2]++; cnt[c2]++; cnt[d2]++;
        double mult = 24.0;
        for (int e = 0; e < 3; ++e) { if (cnt[e] == 2) mult /= 2.0; else if (cnt[e] == 3) mult /= 6.0; else if (cnt[e] == 4) mult /= 24.0; }
        kq += mult * M4[a][b2][c2][d2] * w3[a] * w3[b2] * w3[c2] * w3[d2];
      }
      if (kq < best) { best = kq; wdir[0] = w3[0]; wdir[1] = w3[1]; wdir[2] = w3[2]; }
    }
  }
  /* ... which holds when hop distances are a 1-norm (7-point stencils: measured cut 15 % of the
   * entries instead of 19 %).  With dense stencils they are a max-norm and the rule misleads, so
   * both directions -- least kurtosis and plain first principal axis -- are tried and the one
   * whose median cut crosses fewer edges is kept. */
  double wcand[2][3] = {{wdir[0], wdir[1], wdir[2]}, {1.0, 0.0, 0.0}};
  int ncand = nd >= 2 ? 2 : 1, bestc = 0;
  long long bestcut = -1;
  for (int cd = 0; cd < ncand; ++cd) {
    for (int i = 0; i < len; ++i) {
      double sm = 0.0;
      if (nd == 0) sm = c->dist[i];
      for (int r = 0; r < nd; ++r) sm += wcand[cd][r] * z[(size_t)r * len + i];
      c->ki[i].key = sm;
      c->ki[i].idx = i;
    }
    if (ncand == 1) break;
    sort_keyidx(c->ki, c->ki2, len);
    long long acc2 = 0, cutw = 0;
    int h = 0;
    while (h < len && acc2 < want) acc2 += g->vw[list[c->ki[h++].idx]];
    for (int i = 0; i < len; ++i) c->side[c->ki[i].idx] = i >= h;
    for (int i = 0; i < len; ++i) {
      if (c->side[i]) continue;
      int v = list[i];
      for (int q = g->xadj[v]; q < g->xadj[v + 1]; ++q) {
        int u = g->adj[q];
        if (c->tag[u] == tag && c->side[c->loc[u]]) cutw += g->vw[u];
      }
    }
    if (bestcut < 0 || cutw < bestcut) { bestcut = cutw; bestc = cd; }
  }
  if (ncand > 1)
    for (int i = 0; i < len; ++i) {
      double sm = 0.0;
      for (int r = 0; r < nd; ++r) sm += wcand[bestc][r] * z[(size_t)r * len + i];
      c->ki[i].key = sm;
      c->ki[i].idx = i;
    }
  /* Hop distances are piecewise linear in space (max-norm on a 27-point stencil), so the level
   * sets of their projection have kinks.  A few sweeps of neighbour averaging (damped Jacobi on
   * the graph Laplacian, i.e. steps towards its second eigenvector) smooth them out: the cut
   * gets flatter, the separator thinner. */
  if (c->smooth > 0) {
    double* cur = (double*)c->dist;                       /* the distances are not needed any more */
    double* nxt = cur + len;
    for (int i = 0; i < len; ++i) cur[i] = c->ki[i].key;
    for (int it = 0; it < c->smooth; ++it) {
      for (int i = 0; i < len; ++i) {
        int v = list[i], dg = 0;
        double sm = 0.0;
        for (int q = g->xadj[v]; q < g->xadj[v + 1]; ++q) {
          int u = g->adj[q];
          if (c->tag[u] == tag) { sm += cur[c->loc[u]]; ++dg; }
        }
        nxt[i] = dg ? 0.5 * cur[i] + 0.5 * sm / dg : cur[i];
      }
      double* t2 = cur; cur = nxt; nxt = t2;
    }
    for (int i = 0; i < len; ++i) c->ki[i].key = cur[i];
  }
  sort_keyidx(c->ki, c->ki2, len);
  /* weighted median, at least min0 (min1) vertices per side */
  long long acc = 0, wl = 0;
  int cut = 0;
  for (int i = 0; i < len; ++i) {
    long long w = g->vw[list[c->ki[i].idx]];
    if (i >= min0 && (acc + w - want > want - acc || len - i <= min1)) break;
    acc += w; cut = i + 1;
  }
  if (cut > len - min1) cut = len - min1;
  if (cut < min0) cut = min0;
  for (int i = 0; i < len; ++i) c->side[c->ki[i].idx] = i >= cut;
  for (int i = 0; i < cut; ++i) wl += g->vw[list[c->ki[i].idx]];
  /* straighten the cut: a vertex with more neighbours across than on its own side changes
   * sides while the halves stay within 1 % of their targets (and keep enough vertices) */
  {
    const long long slack = wtot / 100 + 1;
    int nl = cut;
    for (int pass = 0; pass < 3; ++pass) {
      int moved = 0;
      for (int i = 0; i < len; ++i) {
        int v = list[i], sd = c->side[i], own = 0, other = 0;
        for (int q = g->xadj[v]; q < g->xadj[v + 1]; ++q) {
          int u = g->adj[q];
          if (c->tag[u] != tag) continue;
          if (c->side[c->loc[u]] == sd) own += g->vw[u]; else other += g->vw[u];
        }
        if (other <= own) continue;
        long long w = g->vw[v], nwl = sd ? wl + w : wl - w;
        int nnl = sd ? nl + 1 : nl - 1;
        if (nwl > want + slack || nwl < want - slack || nnl < min0 || len - nnl < min1) continue;
        c->side[i] = (char)!sd; wl = nwl; nl = nnl; ++moved;
      }
      if (!moved) break;
    }
    cut = nl;
  }
  *w0 = wl;
  return cut;
}

static void rb_split(rb_t* c, int* list, int len, long long wtot, int k, int base) {
  if (k <= 1) { for (int i = 0; i < len; ++i) c->part[list[i]] = base; return; }
  const int k1 = k / 2, k2 = k - k1;
  long long wl = 0;
  rb_bisect(c, list, len, wtot, (long long)((double)wtot * k1 / k + 0.5), k1, k2, &wl);
  /* stable split of the list */
  int a = 0, b = 0;
  for (int i = 0; i < len; ++i) if (!c->side[i]) list[a++] = list[i]; else c->tmp[b++] = list[i];
  memcpy(list + a, c->tmp, (size_t)b * sizeof(int));
  /* The two halves are independent: the right one works in the part of every list-local buffer that
   * lies behind the left one's (the sub-lists are disjoint ranges of one array, tag / loc are indexed
   * by vertex), so they can run as tasks.  The result does not depend on the schedule.  (A task may
   * read tag[v] of a vertex of ANOTHER sub-list while that one's task relabels it: both the old
   * and the new value differ from the reader's own tag, which is all the reader asks.) */
  rb_t cr = *c;
  cr.queue += a; cr.dist += (size_t)NLM * a; cr.ki += a; cr.ki2 += a; cr.side += a; cr.tmp += a; cr.z += (size_t)3 * a;
  if (len >= 4096) {
#pragma omp task default(shared) firstprivate(cr)
    { rb_t c2 = cr; rb_split(&c2, list + a, b, wtot - wl, k2, base + k1); }
    rb_split(c, list, a, wl, k1, base);
#pragma omp taskwait
  } else {
    rb_split(c, list, a, wl, k1, base);
    rb_split(&cr, list + a, b, wtot - wl, k2, base + k1);
  }
}

static int rb_alloc(rb_t* c, const graph_t* g, int* part) {
  int n = g->n;
  memset(c, 0, sizeof(*c));
  c->g = g; c->part = part; c->tag_store = 1; c->next_tag = &c->tag_store;
  c->tag = (int*)calloc((size_t)n, sizeof(int)); c->loc = (int*)malloc((size_t)n * sizeof(int));
  c->queue = (int*)malloc((size_t)n * sizeof(int)); c->dist = (int*)malloc((size_t)NLM * n * sizeof(int));
  c->ki = (keyidx_t*)malloc((size_t)n * sizeof(keyidx_t)); c->ki2 = (keyidx_t*)malloc((size_t)n * sizeof(keyidx_t));
  c->side = (char*)malloc((size_t)n); c->tmp = (int*)malloc((size_t)n * sizeof(int));
  c->z = (double*)malloc((size_t)3 * n * sizeof(double));
  return !c->tag || !c->loc || !c->queue || !c->dist || !c->ki || !c->ki2 || !c->side || !c->tmp || !c->z;
}
static void rb_free(rb_t* c) {
  free(c->tag); free(c->loc); free(c->queue); free(c->dist); free(c->ki); free(c->ki2); free(c->side); free(c->tmp); free(c->z);
}

static int bisect(const graph_t* g, int k, int* part) {
  int n = g->n;
  rb_t c;
  int* list = (int*)malloc((size_t)n * sizeof(int));
  int rc = rb_alloc(&c, g, part) || !list;
  c.smooth = 0;
  if (!rc) {
    long long wtot = 0;
    for (int v = 0; v < n; ++v) { list[v] = v; wtot += g->vw[v]; }
#pragma omp parallel num_threads(pa_host_threads())
#pragma omp single
    rb_split(&c, list, n, wtot, k, 0);
  }
  rb_free(&c); free(list);
  return rc;
}

/* ---- 3. slivers ------------------------------------------------------------------------------------------- */
/* A cut can separate a sliver from its part.  Every part keeps its largest connected piece; a
 * piece lighter than `limit` joins the neighbouring part it touches most. */
static int join_fragments(const graph_t* g, int k, int* part, long long limit) {
  int n = g->n;
  int* comp = (int*)malloc((size_t)n * sizeof(int));
  int* stack = (int*)malloc((size_t)n * sizeof(int));
  long long* best_w = (long long*)calloc((size_t)k, sizeof(long long));
  int* best_c = (int*)malloc((size_t)k * sizeof(int));
  size_t ccap = 1024;
  int nc = 0;
  long long* cw = (long long*)malloc(ccap * sizeof(long long));
  if (!comp || !stack || !best_w || !best_c || !cw) { free(comp); free(stack); free(best_w); free(best_c); free(cw); return 1; }
  for (int v = 0; v < n; ++v) comp[v] = -1;
  for (int p = 0; p < k; ++p) best_c[p] = -1;
  for (int v0 = 0; v0 < n; ++v0) {
    if (comp[v0] >= 0) continue;
    if ((size_t)nc == ccap) { ccap *= 2; cw = (long long*)realloc(cw, ccap * sizeof(long long)); if (!cw) { free(comp); free(stack); free(best_w); free(best_c); return 1; } }
    int top = 0, p = part[v0];
    long long w = 0;
    stack[top++] = v0; comp[v0] = nc;
    while (top > 0) {
      int u = stack[--top];
      w += g->vw[u];
      for (int q = g->xadj[u]; q < g->xadj[u + 1]; ++q) { int v = g->adj[q]; if (comp[v] < 0 && part[v] == p) { comp[v] = nc; stack[top++] = v; } }
    }
    cw[nc] = w;
    if (w > best_w[p]) { best_w[p] = w; best_c[p] = nc; }
    ++nc;
  }
  /* fragments: collect their vertices again and vote; a part takes fragments only up to 1.25 times the
   * average weight (a star-shaped graph -- one dense row -- is all fragments around its centre) */
  long long* pw = (long long*)calloc((size_t)k, sizeof(long long));
  long long wtot = 0;
  if (!pw) { free(comp); free(stack); free(best_w); free(best_c); free(cw); return 1; }
  for (int v = 0; v < n; ++v) { pw[part[v]] += g->vw[v]; wtot += g->vw[v]; }
  const long long wcap = wtot / k + wtot / (4 * (long long)k) + 1;
  int cp[64]; long long cc[64];
  for (int v0 = 0; v0 < n; ++v0) {
    int c = comp[v0], p = part[v0];
    if (c < 0 || c == best_c[p] || cw[c] >= limit) continue;
    int top = 0, cnt = 0, ncand = 0;
    stack[top++] = v0; comp[v0] = -2 - c;            /* visited marker of this sweep */
    /* the fragment's vertices end up in stack[n - cnt ...] */
    while (top > 0) {
      int u = stack[--top];
      stack[n - 1 - cnt++] = u;
      for (int q = g->xadj[u]; q < g->xadj[u + 1]; ++q) {
        int v = g->adj[q];
        if (part[v] == p) { if (comp[v] == c) { comp[v] = -2 - c; stack[top++] = v; } continue; }
        int i = 0;
        while (i < ncand && cp[i] != part[v]) ++i;
        if (i == ncand) { if (ncand == 64) continue; cp[ncand] = part[v]; cc[ncand] = 0; ++ncand; }
        cc[i] += g->vw[v];
      }
      if (top + cnt >= n) break;                       /* (cannot happen: a fragment is not the whole graph) */
    }
    if (ncand == 0) continue;                          /* an isolated piece of the graph: stays */
    int bi = -1;
    for (int i = 0; i < ncand; ++i) if (pw[cp[i]] + cw[c] <= wcap && (bi < 0 || cc[i] > cc[bi])) bi = i;
    if (bi < 0) continue;                              /* every neighbour is full: the fragment stays */
    for (int i = 0; i < cnt; ++i) part[stack[n - 1 - i]] = cp[bi];
    pw[cp[bi]] += cw[c]; pw[p] -= cw[c];
  }
  free(pw);
  free(comp); free(stack); free(best_w); free(best_c); free(cw);
  return 0;
}

/* ---- nested dissection of one diagonal block (for the sparse block solve, nd.c) --------------------- */
typedef struct {
  rb_t rb;
  long long leaf;      /* a piece of at most this weight is not cut any further */
  int nsn, cap;
  int* parent;         /* per supernode */
  int* vfirst;         /* per supernode: start in vorder */
  int* vorder;         /* vertices in elimination order */
  int nv;
} nd_ctx_t;

static int nd_new(nd_ctx_t* c, const int* verts, int cnt) {
  if (c->nsn == c->cap) {
    c->cap = c->cap ? 2 * c->cap : 256;
    c->parent = (int*)realloc(c->parent, (size_t)c->cap * sizeof(int));
    c->vfirst = (int*)realloc(c->vfirst, ((size_t)c->cap + 1) * sizeof(int));
    if (!c->parent || !c->vfirst) return -1;
  }
  int s = c->nsn++;
  c->parent[s] = -1;
  c->vfirst[s] = c->nv;
  memcpy(c->vorder + c->nv, verts, (size_t)cnt * sizeof(int));
  c->nv += cnt;
  c->vfirst[s + 1] = c->nv;
  return s;
}

/* returns the supernode at the root of the piece, -1 on allocation failure */
static int nd_rec(nd_ctx_t* c, int* list, int len, long long w) {
  const graph_t* g = c->rb.g;
  if (w <= c->leaf || len < 8) return nd_new(c, list, len);
  long long w0 = 0;
  rb_bisect(&c->rb, list, len, w, w / 2, 1, 1, &w0);
  const int tag = c->rb.cur_tag;
  char* side = c->rb.side;
  /* vertex separator: the boundary of one side, whichever is lighter */
  long long b0 = 0, b1 = 0;
  for (int i = 0; i < len; ++i) {
    int v = list[i], sd = side[i], bd = 0;
    for (int q = g->xadj[v]; q < g->xadj[v + 1] && !bd; ++q) {
      int u = g->adj[q];
      bd = c->rb.tag[u] == tag && (side[c->rb.loc[u]] & 1) != sd;
    }
    if (bd) { side[i] = (char)(sd | 2); if (sd) b1 += g->vw[v]; else b0 += g->vw[v]; }
  }
  const int take = b0 <= b1 ? 0 : 1;
  /* A smaller separator than either boundary layer: any vertex cover of the cut edges separates the
   * two sides, and a minimum cover of this bipartite graph (boundary of side 0 | boundary of side
   * 1) comes from a maximum matching (Koenig).  On a jagged cut it is up to a third smaller than
   * the lighter boundary; the fill of the factor goes with the square.  side bit 4 = in the cover. */
  int cover_ok = 0;
  {
    int* mate = c->rb.queue;            /* list-local: matched partner or -1 (boundary vertices only) */
    int* dist = c->rb.tmp;              /* BFS layers of the left side */
    int* stack = (int*)c->rb.ki;        /* DFS stack / BFS queue (len ints fit: keyidx_t is 16 bytes) */
    int* itq = stack + len;             /* per-vertex adjacency cursor */
    int* cand = itq + len;              /* left boundary vertices */
    int nleft = 0;
    for (int i = 0; i < len; ++i) { mate[i] = -1; if (side[i] == 2) cand[nleft++] = i; }
    /* Hopcroft-Karp */
    for (;;) {
      int head = 0, tail = 0, found = 0;
      for (int a = 0; a < nleft; ++a) { int i = cand[a]; if (mate[i] < 0) { dist[i] = 0; stack[tail++] = i; } else dist[i] = -1; }
      while (head < tail) {
        int i = stack[head++], v = list[i];
        for (int q = g->xadj[v]; q < g->xadj[v + 1]; ++q) {
          int u = g->adj[q];
          if (c->rb.tag[u] != tag) continue;
          int j = c->rb.loc[u];
          if (side[j] != 3) continue;
          int k = mate[j];
          if (k < 0) found = 1;
          else if (dist[k] < 0) { dist[k] = dist[i] + 1; stack[tail++] = k; }
        }
      }
      if (!found) break;
      int aug = 0;
      for (int a = 0; a < nleft; ++a) itq[cand[a]] = g->xadj[list[cand[a]]];
      for (int a = 0; a < nleft; ++a) {
        int root = cand[a];
        if (mate[root] >= 0) continue;
        int top = 0;
        stack[top++] = root;
        while (top > 0) {
          int i = stack[top - 1], v = list[i], advanced = 0;
          while (itq[i] < g->xadj[v + 1]) {
            int u = g->adj[itq[i]++];
            if (c->rb.tag[u] != tag) continue;
            int j = c->rb.loc[u];
            if (side[j] != 3) continue;
            int k = mate[j];
            if (k < 0) {                  /* free right vertex: augment along the stack */
              for (int t2 = top - 1; t2 >= 0; --t2) { int li = stack[t2], old = mate[li]; mate[li] = j; mate[j] = li; j = old; }
              top = 0; advanced = 1; ++aug;
              break;
            }
            if (dist[k] == dist[i] + 1) { stack[top++] = k; advanced = 1; break; }
          }
          if (!advanced && top > 0) { dist[i] = -2; --top; }
        }
      }
      if (!aug) break;
    }
    /* Koenig: Z = reachable from free left vertices by alternating paths; cover = (left \ Z) + (right & Z) */
    int head = 0, tail = 0;
    for (int i = 0; i < len; ++i) dist[i] = 0;          /* visited marks */
    for (int a = 0; a < nleft; ++a) { int i = cand[a]; if (mate[i] < 0) { dist[i] = 1; stack[tail++] = i; } }
    while (head < tail) {
      int i = stack[head++], v = list[i];
      for (int q = g->xadj[v]; q < g->xadj[v + 1]; ++q) {
        int u = g->adj[q];
        if (c->rb.tag[u] != tag) continue;
        int j = c->rb.loc[u];
        if (side[j] != 3 || dist[j] || mate[i] == j) continue;
        dist[j] = 1;
        int k = mate[j];
        if (k >= 0 && !dist[k]) { dist[k] = 1; stack[tail++] = k; }
      }
    }
    long long wc = 0;
    for (int i = 0; i < len; ++i) {
      if (side[i] == 2 && !dist[i]) { side[i] |= 4; wc += g->vw[list[i]]; }
      else if (side[i] == 3 && dist[i]) { side[i] |= 4; wc += g->vw[list[i]]; }
    }
    cover_ok = wc <= (b0 <= b1 ? b0 : b1);
    if (!cover_ok) for (int i = 0; i < len; ++i) side[i] &= 3;
  }
  int nl = 0, nr = 0, ns = 0;
  long long wl = 0, wr = 0;
  int* tmp = c->rb.tmp;                 /* [right | separator] while the left part is compacted in place */
#define ND_IS_SEP(i) (cover_ok ? (side[i] & 4) != 0 : ((side[i] & 2) && (side[i] & 1) == take))
  for (int i = 0; i < len; ++i) {
    int v = list[i], sd = side[i] & 1, sep = ND_IS_SEP(i);
    if (sep) ++ns;
    else if (sd) { ++nr; wr += g->vw[v]; }
    else { wl += g->vw[v]; }
  }
  nl = len - nr - ns;
  if (nl == 0 || nr == 0) return nd_new(c, list, len);      /* nothing left on one side: a dense leaf */
  {
    int a = 0, b = 0, d = 0;
    for (int i = 0; i < len; ++i) {
      int v = list[i], sd = side[i] & 1, sep = ND_IS_SEP(i);
      if (sep) tmp[nr + d++] = v; else if (sd) tmp[b++] = v; else list[a++] = v;
    }
#undef ND_IS_SEP
    memcpy(list + nl, tmp, (size_t)(nr + ns) * sizeof(int));
  }
  int l = nd_rec(c, list, nl, wl);
  if (l < 0) return -1;
  int r = nd_rec(c, list + nl, nr, wr);
  if (r < 0) return -1;
  int s = nd_new(c, list + nl + nr, ns);
  if (s < 0) return -1;
  c->parent[l] = s; c->parent[r] = s;
  return s;
}

void pa_nd_tree_free(pa_nd_tree_t* t) {
  free(t->first); free(t->parent); free(t->perm);
  memset(t, 0, sizeof(*t));
}

/* Nested-dissection order of an n x n block with a structurally symmetric pattern (local CSR,
 * diagonal stored): supernodes (leaves and separators) in postorder, each a contiguous range of
 * the new order; rows with identical patterns stay together.  Thread safe. */
int pa_nd_order(int n, const int* rp, const int* ci, int leaf_rows, pa_nd_tree_t* t) {
  memset(t, 0, sizeof(*t));
  int* cid = (int*)malloc((size_t)n * sizeof(int));
  graph_t g;
  memset(&g, 0, sizeof(g));
  if (!cid || build_graph(n, rp, ci, 1, cid, &g)) { free(cid); graph_free(&g); return 1; }
  nd_ctx_t c;
  memset(&c, 0, sizeof(c));
  c.leaf = leaf_rows;
  c.vorder = (int*)malloc((size_t)g.n * sizeof(int));
  int* list = (int*)malloc((size_t)g.n * sizeof(int));
  int rc = rb_alloc(&c.rb, &g, NULL) || !c.vorder || !list;
  c.rb.smooth = 30;
  if (!rc) {
    /* connected components are independent trees: handled by the bisection itself (a cut that
     * separates components has an empty separator) */
    long long w = 0;
    for (int v = 0; v < g.n; ++v) { list[v] = v; w += g.vw[v]; }
    rc = nd_rec(&c, list, g.n, w) < 0;
  }
  if (!rc) {
    /* rows of every vertex, then the row order */
    int* vstart = (int*)calloc((size_t)g.n + 1, sizeof(int));
    int* vrows = (int*)malloc((size_t)n * sizeof(int));
    t->first = (int*)malloc(((size_t)c.nsn + 1) * sizeof(int));
    t->parent = (int*)malloc((size_t)c.nsn * sizeof(int));
    t->perm = (int*)malloc((size_t)n * sizeof(int));
    rc = !vstart || !vrows || !t->first || !t->parent || !t->perm;
    if (!rc) {
      for (int i = 0; i < n; ++i) vstart[cid[i] + 1]++;
      for (int v = 0; v < g.n; ++v) vstart[v + 1] += vstart[v];
      int* fill = c.rb.tmp;
      memcpy(fill, vstart, (size_t)g.n * sizeof(int));
      for (int i = 0; i < n; ++i) vrows[fill[cid[i]]++] = i;
      int pos = 0;
      for (int s = 0; s < c.nsn; ++s) {
        t->first[s] = pos;
        t->parent[s] = c.parent[s];
        for (int q = c.vfirst[s]; q < c.vfirst[s + 1]; ++q) {
          int v = c.vorder[q];
          for (int e = vstart[v]; e < vstart[v + 1]; ++e) t->perm[pos++] = vrows[e];
        }
      }
      t->first[c.nsn] = pos;
      t->nsn = c.nsn;
      if (pos != n) rc = 1;
    }
    free(vstart); free(vrows);
  }
  rb_free(&c.rb); free(c.parent); free(c.vfirst); free(c.vorder); free(list); free(cid); graph_free(&g);
  if (rc) pa_nd_tree_free(t);
  return rc;
}

/* ---- entry point ------------------------------------------------------------------------------------------ */
int preAlps_hip_partition_kway(int N, const int* rowPtr, const int* colInd, int nparts, int* part) {
  if (N < 1 || nparts < 1 || nparts > N || !rowPtr || !colInd || !part)
    return PA_FAIL("invalid arguments (N = %d, nparts = %d)", N, nparts);
  if (nparts == 1) { memset(part, 0, (size_t)N * sizeof(int)); return 0; }
  int* cid = (int*)malloc((size_t)N * sizeof(int));
  if (!cid) return PA_FAIL("out of host memory");
  graph_t g;
  memset(&g, 0, sizeof(g));
  int merge = 1;       /* rows with identical column lists (the dofs of a node) become one weighted vertex */
  const int trace = getenv("PREALPS_SETUP_TRACE") != NULL;
  double t0 = pa_wtime();
  int rc = build_graph(N, rowPtr, colInd, merge, cid, &g);
  if (!rc && merge && g.n < 4 * (long long)nparts && g.n < N) {   /* too few merged vertices per part: plain rows */
    graph_free(&g);
    rc = build_graph(N, rowPtr, colInd, 0, cid, &g);
  }
  if (rc) { graph_free(&g); free(cid); return PA_FAIL("out of host memory for the adjacency graph"); }
  int* cpart = (int*)malloc((size_t)g.n * sizeof(int));
  rc = !cpart;
  double t1 = pa_wtime();
  if (!rc) rc = bisect(&g, nparts, cpart);
  double t2 = pa_wtime();
  if (!rc) rc = join_fragments(&g, nparts, cpart, (long long)N / nparts / 8 + 1);
  if (trace) fprintf(stderr, "[partition] graph of %d vertices %.2f s, bisection %.2f s (%d threads), fragments %.2f s\n",
                     g.n, t1 - t0, t2 - t1, pa_host_threads(), pa_wtime() - t2);
  if (!rc)
    for (int i = 0; i < N; ++i) part[i] = cpart[cid[i]];
  graph_free(&g); free(cid); free(cpart);
  if (rc) return PA_FAIL("graph partitioning failed (out of memory)");
  return 0;
}
