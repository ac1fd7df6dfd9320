/*
 * oracle/ecg_oracle.c -- CPU restatement of the preAlps ECG hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under prealps_amd/ may include, link or
 * call this file.  Only tests/, __graft_entry__.smoke() and the cpu_baseline
 * leg of bench.py use it, and only as the checker / reported CPU baseline.
 *
 * Parity pin: the reference cannot be built in this image without writing
 * stand-in headers (mkl.h, metis.h are absent), so this restatement is pinned
 * against the reference's own recorded outputs in BASELINE.md section 2
 * (per-iteration residual histories of the unmodified reference on Poisson
 * 24^3 np=8 t=4 for Odir and D-Odir, and LFAT5 np=2 t=2), see
 * tests/test_oracle_golden.py.
 *
 * What is restated (reference file:line, all under /root/reference):
 *   orc_spmm            utils/cplm_light/cplm_kernels.c:620-671 (mkl_dcsrmm
 *                       per column block) as driven by
 *                       utils/cplm_v0/cplm_v0_matmult_v2.c:108-343
 *   orc_bj_*            src/preconditioners/block_jacobi.c:26-63 (diag block
 *                       extraction + Cholesky), :93-109 (solve with t rhs)
 *   orc_ecg_*           src/solvers/ecg.c:41-96 (pool), :98-171 (reset),
 *                       :201-221 (split), :223-271 (stopping), :289-400
 *                       (Orthomin + BF-Omin), :402-530 (Orthodir + D-Odir),
 *                       :532-658 (fused Orthodir), :660-677 (wrap-up)
 *   small dense         LAPACK dpotrf/dpstrf(dpstf2)/dlapmt/dgesvd/dgeqrf/
 *                       dormqr as called at ecg.c:318,375,380,455,470-479
 *                       (MKL is closed source; the published LAPACK
 *                       algorithms are restated; SVD by one-sided Jacobi)
 *
 * Formulation: one process plays all P reference ranks ("parts").  Every
 * panel is the vertical concatenation of the P local panels (column major,
 * leading dimension N); per-rank Gram blocks are accumulated part by part
 * and then summed in rank order, which is what MPI_Allreduce(SUM) returns up
 * to rounding.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_ORTHOMIN 0
#define ORC_ORTHODIR 1
#define ORC_ORTHODIR_FUSED 2
#define ORC_ADAPT_BS 0
#define ORC_NO_BS_RED 1

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------ SpMM */
/* Y = A X, X and Y column major with t columns (cplm_kernels.c:620-671). */
void orc_spmm(int n, const int* rowptr, const int* colind, const double* val,
              int t, const double* X, int ldx, double* Y, int ldy) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    double s[128];
    for (int c = 0; c < t; ++c) s[c] = 0.0;
    for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) { /* one pass over the row, all t columns */
      const double v = val[k];
      const double* x = X + colind[k];
      for (int c = 0; c < t; ++c) s[c] += v * x[(size_t)c * ldx];
    }
    for (int c = 0; c < t; ++c) Y[(size_t)c * ldy + i] = s[c];
  }
}

/* -------------------------------------------------- block-Jacobi (exact) */
/* One SPD diagonal block per part, factored as an envelope (skyline)
 * Cholesky L L^T in the block's natural row order: no fill leaves the
 * envelope, so the solve is exact like PARDISO's (block_jacobi.c:48-58). */
typedef struct {
  int n, P;
  int* rowpos;     /* P+1 */
  int* first;      /* n: first local column of the envelope of each row */
  size_t* start;   /* n+1: offset of row i in L (entries first[i]..i) */
  double* L;
  int info;        /* 0 or 1+row of the first non-positive pivot */
} orc_bj_t;

orc_bj_t* orc_bj_create(int n, const int* rowptr, const int* colind,
                        const double* val, int P, const int* rowpos) {
  orc_bj_t* h = (orc_bj_t*)calloc(1, sizeof(orc_bj_t));
  h->n = n; h->P = P;
  h->rowpos = (int*)malloc((P + 1) * sizeof(int));
  memcpy(h->rowpos, rowpos, (P + 1) * sizeof(int));
  h->first = (int*)malloc(n * sizeof(int));
  h->start = (size_t*)malloc((n + 1) * sizeof(size_t));
  /* envelope */
  for (int p = 0; p < P; ++p) {
    int r0 = rowpos[p], r1 = rowpos[p + 1];
    for (int i = r0; i < r1; ++i) {
      int f = i;
      for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
        int j = colind[k];
        if (j >= r0 && j < r1 && j < f) f = j;
      }
      h->first[i] = f - r0;
    }
  }
  h->start[0] = 0;
  for (int p = 0; p < P; ++p) {
    int r0 = rowpos[p], r1 = rowpos[p + 1];
    for (int i = r0; i < r1; ++i)
      h->start[i + 1] = h->start[i] + (size_t)((i - r0) - h->first[i] + 1);
  }
  h->L = (double*)calloc(h->start[n] ? h->start[n] : 1, sizeof(double));
  /* scatter the lower triangle of each diagonal block */
  for (int p = 0; p < P; ++p) {
    int r0 = rowpos[p], r1 = rowpos[p + 1];
    for (int i = r0; i < r1; ++i)
      for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
        int j = colind[k];
        if (j >= r0 && j <= i)
          h->L[h->start[i] + (size_t)((j - r0) - h->first[i])] = val[k];
      }
  }
  int info = 0;
#pragma omp parallel for schedule(dynamic, 1)
  for (int p = 0; p < P; ++p) {
    int r0 = rowpos[p], b = rowpos[p + 1] - r0;
    const int* first = h->first + r0;
    for (int i = 0; i < b; ++i) {
      double* Li = h->L + h->start[r0 + i] - first[i]; /* Li[j] = L(i,j) */
      for (int j = first[i]; j < i; ++j) {
        const double* Lj = h->L + h->start[r0 + j] - first[j];
        int k0 = first[i] > first[j] ? first[i] : first[j];
        double s = Li[j];
        for (int k = k0; k < j; ++k) s -= Li[k] * Lj[k];
        Li[j] = s / Lj[j];
      }
      double d = Li[i];
      for (int k = first[i]; k < i; ++k) d -= Li[k] * Li[k];
      if (!(d > 0.0)) {
#pragma omp critical
        { if (!info) info = r0 + i + 1; }
        d = NAN;
      }
      Li[i] = sqrt(d);
    }
  }
  h->info = info;
  return h;
}

int orc_bj_info(const orc_bj_t* h) { return h->info; }
double orc_bj_factor_bytes(const orc_bj_t* h) { return 8.0 * (double)h->start[h->n]; }

/* Z = blockdiag(A)^-1 X for t columns (block_jacobi.c:93-109). */
int orc_bj_apply(const orc_bj_t* h, int t, const double* X, int ldx, double* Z,
                 int ldz) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int p = 0; p < h->P; ++p) {
    int r0 = h->rowpos[p], b = h->rowpos[p + 1] - r0;
    const int* first = h->first + r0;
    for (int c = 0; c < t; ++c) {
      const double* x = X + (size_t)c * ldx + r0;
      double* z = Z + (size_t)c * ldz + r0;
      for (int i = 0; i < b; ++i) { /* L y = x */
        const double* Li = h->L + h->start[r0 + i] - first[i];
        double s = x[i];
        for (int k = first[i]; k < i; ++k) s -= Li[k] * z[k];
        z[i] = s / Li[i];
      }
      for (int i = b - 1; i >= 0; --i) { /* L^T z = y */
        const double* Li = h->L + h->start[r0 + i] - first[i];
        double zi = z[i] / Li[i];
        z[i] = zi;
        for (int k = first[i]; k < i; ++k) z[k] -= Li[k] * zi;
      }
    }
  }
  return 0;
}

void orc_bj_free(orc_bj_t* h) {
  if (!h) return;
  free(h->rowpos); free(h->first); free(h->start); free(h->L); free(h);
}

/* ------------------------------------------------------ small dense LAPACK */
/* Upper Cholesky W = U^T U in place, column major (LAPACKE_dpotrf 'U'). */
static int potrf_upper(int n, double* W, int ld) {
  for (int j = 0; j < n; ++j) {
    double d = W[j + (size_t)ld * j];
    for (int k = 0; k < j; ++k) d -= W[k + (size_t)ld * j] * W[k + (size_t)ld * j];
    if (!(d > 0.0)) return j + 1;
    d = sqrt(d);
    W[j + (size_t)ld * j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = W[j + (size_t)ld * i];
      for (int k = 0; k < j; ++k) s -= W[k + (size_t)ld * j] * W[k + (size_t)ld * i];
      W[j + (size_t)ld * i] = s / d;
    }
  }
  return 0;
}

/* B <- B U^-1, B is m x n (cblas_dtrsm Right,Upper,NoTrans,NonUnit).  Rows are
 * independent: each thread substitutes along its own rows. */
static void trsm_right_upper(int m, int n, const double* U, int ldu, double* B,
                             int ldb) {
  double rd[128];
  for (int j = 0; j < n; ++j) rd[j] = 1.0 / U[j + (size_t)ldu * j];
#pragma omp parallel for schedule(static) if (m > 4096)
  for (int i = 0; i < m; ++i) {
    for (int j = 0; j < n; ++j) {
      double s = B[i + (size_t)ldb * j];
      for (int k = 0; k < j; ++k) s -= B[i + (size_t)ldb * k] * U[k + (size_t)ldu * j];
      B[i + (size_t)ldb * j] = s * rd[j];
    }
  }
}

/* B <- U^-T B, U is n x n upper, B is n x nrhs (cblas_dtrsm Left,Upper,Trans). */
static void trsm_left_upper_trans(int n, int nrhs, const double* U, int ldu,
                                  double* B, int ldb) {
  for (int c = 0; c < nrhs; ++c) {
    double* b = B + (size_t)ldb * c;
    for (int i = 0; i < n; ++i) {
      double s = b[i];
      for (int k = 0; k < i; ++k) s -= U[k + (size_t)ldu * i] * b[k];
      b[i] = s / U[i + (size_t)ldu * i];
    }
  }
}

/* LAPACK dpstf2 'U' (what dpstrf runs for n <= block size).  tol < 0 means
 * n * eps * max diag.  piv is 1-based.  Returns info (1 = rank deficient). */
static int pstrf_upper(int n, double* A, int ld, int* piv, int* rank, double tol,
                       double* work /* 2n */) {
  int info = 0;
  for (int i = 0; i < n; ++i) piv[i] = i + 1;
  int pvt = 0;
  double ajj = A[0];
  for (int i = 1; i < n; ++i)
    if (A[i + (size_t)ld * i] > ajj) { pvt = i; ajj = A[i + (size_t)ld * i]; }
  if (ajj <= 0.0 || isnan(ajj)) { *rank = 0; return 1; }
  double dstop = tol < 0.0 ? n * (DBL_EPSILON * 0.5) * ajj : tol;
  for (int i = 0; i < n; ++i) work[i] = 0.0;
  int j;
  for (j = 0; j < n; ++j) {
    for (int i = j; i < n; ++i) {
      if (j > 0) work[i] += A[(j - 1) + (size_t)ld * i] * A[(j - 1) + (size_t)ld * i];
      work[n + i] = A[i + (size_t)ld * i] - work[i];
    }
    if (j > 0) {
      pvt = j; ajj = work[n + j];
      for (int i = j + 1; i < n; ++i)
        if (work[n + i] > ajj) { pvt = i; ajj = work[n + i]; }
      if (ajj <= dstop || isnan(ajj)) {
        A[j + (size_t)ld * j] = ajj;
        *rank = j;
        return 1;
      }
    }
    if (j != pvt) {
      A[pvt + (size_t)ld * pvt] = A[j + (size_t)ld * j];
      for (int k = 0; k < j; ++k) { /* swap columns j,pvt above row j */
        double tmp = A[k + (size_t)ld * j];
        A[k + (size_t)ld * j] = A[k + (size_t)ld * pvt];
        A[k + (size_t)ld * pvt] = tmp;
      }
      for (int k = pvt + 1; k < n; ++k) { /* row j <-> row pvt right of pvt */
        double tmp = A[j + (size_t)ld * k];
        A[j + (size_t)ld * k] = A[pvt + (size_t)ld * k];
        A[pvt + (size_t)ld * k] = tmp;
      }
      for (int k = j + 1; k < pvt; ++k) { /* A(j,k) <-> A(k,pvt) */
        double tmp = A[j + (size_t)ld * k];
        A[j + (size_t)ld * k] = A[k + (size_t)ld * pvt];
        A[k + (size_t)ld * pvt] = tmp;
      }
      double dt = work[j]; work[j] = work[pvt]; work[pvt] = dt;
      int it = piv[pvt]; piv[pvt] = piv[j]; piv[j] = it;
    }
    ajj = sqrt(ajj);
    A[j + (size_t)ld * j] = ajj;
    if (j < n - 1) {
      for (int i = j + 1; i < n; ++i) {
        double s = A[j + (size_t)ld * i];
        for (int k = 0; k < j; ++k) s -= A[k + (size_t)ld * j] * A[k + (size_t)ld * i];
        A[j + (size_t)ld * i] = s / ajj;
      }
    }
  }
  *rank = n;
  return info;
}

/* LAPACK dlapmt forward: X(:,j) <- X(:,k[j]) (k 1-based, restored on exit). */
static void lapmt_forward(int m, int n, double* X, int ldx, int* k) {
  if (n <= 1) return;
  for (int i = 0; i < n; ++i) k[i] = -k[i];
  for (int i = 1; i <= n; ++i) {
    if (k[i - 1] > 0) continue;
    int j = i;
    k[j - 1] = -k[j - 1];
    int in = k[j - 1];
    while (k[in - 1] <= 0) {
      for (int ii = 0; ii < m; ++ii) {
        double tmp = X[ii + (size_t)ldx * (j - 1)];
        X[ii + (size_t)ldx * (j - 1)] = X[ii + (size_t)ldx * (in - 1)];
        X[ii + (size_t)ldx * (in - 1)] = tmp;
      }
      k[in - 1] = -k[in - 1];
      j = in;
      in = k[in - 1];
    }
  }
}

/* Left singular vectors and singular values of A (m x n, m <= n), the part of
 * LAPACKE_dgesvd(jobu='O', jobvt='N') that ecg.c:455 uses: on exit A(:,0:m)
 * holds U (m x m) and s the singular values in decreasing order.  One-sided
 * Jacobi on the rows of A. */
static void svd_left(int m, int n, double* A, int lda, double* s,
                     double* work /* m*n + m*m */) {
  double* B = work;          /* n x m, B = A^T */
  double* V = work + (size_t)m * n; /* m x m rotations */
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) B[j + (size_t)n * i] = A[i + (size_t)lda * j];
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) V[i + (size_t)m * j] = (i == j);
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < m - 1; ++p)
      for (int q = p + 1; q < m; ++q) {
        double a = 0, b = 0, c = 0;
        for (int k = 0; k < n; ++k) {
          a += B[k + (size_t)n * p] * B[k + (size_t)n * p];
          b += B[k + (size_t)n * q] * B[k + (size_t)n * q];
          c += B[k + (size_t)n * p] * B[k + (size_t)n * q];
        }
        if (c == 0.0 || fabs(c) <= 1e-300) continue;
        double r = fabs(c) / sqrt(a * b);
        if (r > off) off = r;
        if (r <= DBL_EPSILON * 0.25) continue;
        double zeta = (b - a) / (2.0 * c);
        double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
        for (int k = 0; k < n; ++k) {
          double bp = B[k + (size_t)n * p], bq = B[k + (size_t)n * q];
          B[k + (size_t)n * p] = cs * bp - sn * bq;
          B[k + (size_t)n * q] = sn * bp + cs * bq;
        }
        for (int k = 0; k < m; ++k) {
          double vp = V[k + (size_t)m * p], vq = V[k + (size_t)m * q];
          V[k + (size_t)m * p] = cs * vp - sn * vq;
          V[k + (size_t)m * q] = sn * vp + cs * vq;
        }
      }
    if (off <= DBL_EPSILON * 0.25) break;
  }
  /* A = V * diag(norms) * (normalised B)^T: left vectors are the columns of V */
  for (int p = 0; p < m; ++p) {
    double a = 0;
    for (int k = 0; k < n; ++k) a += B[k + (size_t)n * p] * B[k + (size_t)n * p];
    s[p] = sqrt(a);
  }
  for (int p = 0; p < m; ++p) { /* selection sort, decreasing */
    int best = p;
    for (int q = p + 1; q < m; ++q) if (s[q] > s[best]) best = q;
    if (best != p) {
      double ts = s[p]; s[p] = s[best]; s[best] = ts;
      for (int k = 0; k < m; ++k) {
        double tv = V[k + (size_t)m * p];
        V[k + (size_t)m * p] = V[k + (size_t)m * best];
        V[k + (size_t)m * best] = tv;
      }
    }
  }
  for (int j = 0; j < m; ++j)
    for (int i = 0; i < m; ++i) A[i + (size_t)lda * j] = V[i + (size_t)m * j];
}

/* LAPACK dgeqr2 (dlarfg convention): A (m x n, m>=n... here square). */
static void geqrf(int m, int n, double* A, int lda, double* tau) {
  int kmax = m < n ? m : n;
  for (int k = 0; k < kmax; ++k) {
    double* x = A + k + (size_t)lda * k;
    double alpha = x[0], xnorm = 0.0;
    for (int i = 1; i < m - k; ++i) xnorm += x[i] * x[i];
    xnorm = sqrt(xnorm);
    if (xnorm == 0.0) { tau[k] = 0.0; }
    else {
      double beta = -copysign(sqrt(alpha * alpha + xnorm * xnorm), alpha);
      tau[k] = (beta - alpha) / beta;
      double sc = 1.0 / (alpha - beta);
      for (int i = 1; i < m - k; ++i) x[i] *= sc;
      x[0] = beta;
    }
    /* apply H_k = I - tau v v^T to A(k:m, k+1:n) from the left */
    for (int j = k + 1; j < n; ++j) {
      double* c = A + k + (size_t)lda * j;
      double w = c[0];
      for (int i = 1; i < m - k; ++i) w += x[i] * c[i];
      w *= tau[k];
      c[0] -= w;
      for (int i = 1; i < m - k; ++i) c[i] -= x[i] * w;
    }
  }
}

/* C <- Q^T C (side L, trans T), Q = H_0 ... H_{k-1} from geqrf; C is k x nc. */
static void ormqr_left_trans(int k, int nc, const double* Q, int ldq,
                             const double* tau, double* C, int ldc) {
  for (int r = 0; r < k; ++r) { /* Q^T C = H_{k-1} ... H_0 C: apply H_0 first */
    for (int j = 0; j < nc; ++j) {
      double* c = C + (size_t)ldc * j;
      double w = c[r];
      for (int i = r + 1; i < k; ++i) w += Q[i + (size_t)ldq * r] * c[i];
      w *= tau[r];
      c[r] -= w;
      for (int i = r + 1; i < k; ++i) c[i] -= Q[i + (size_t)ldq * r] * w;
    }
  }
}

/* C <- C Q (side R, trans N); C is m x k. */
static void ormqr_right_notrans(int m, int k, const double* Q, int ldq,
                                const double* tau, double* C, int ldc) {
  for (int r = 0; r < k; ++r) { /* C H_0 H_1 ... */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < m; ++i) {
      double w = C[i + (size_t)ldc * r];
      for (int j = r + 1; j < k; ++j) w += C[i + (size_t)ldc * j] * Q[j + (size_t)ldq * r];
      w *= tau[r];
      C[i + (size_t)ldc * r] -= w;
      for (int j = r + 1; j < k; ++j) C[i + (size_t)ldc * j] -= w * Q[j + (size_t)ldq * r];
    }
  }
}

/* mkl_dimatcopy('C','N', rows, cols, 1.0, A, lda, ldb): in-place repack. */
static void imatcopy_shrink(int rows, int cols, double* A, int lda, int ldb) {
  for (int j = 0; j < cols; ++j)
    for (int i = 0; i < (rows < ldb ? rows : ldb); ++i)
      A[i + (size_t)ldb * j] = A[i + (size_t)lda * j];
}

/* ------------------------------------------------------------ ECG solver */
typedef struct { double* val; int m, n, ld; } pan_t; /* m rows x n cols, col major */

typedef struct {
  int N, P, T;
  int* rowpos;
  int ortho_alg, bs_red, maxIter;
  double tol;
  /* pool (ecg.c:41-96) */
  double* work;
  size_t work_len;
  int iwork[256];
  pan_t X, R, V, AV, Z, alpha, beta, Pd, APd;
  double normb, res;
  int iter, bs, kbs;
  double* gpart; /* P * (2T*2T) partial Gram blocks */
  double* tmp;   /* scratch for small dense */
} orc_ecg_t;

static void setinfo(pan_t* p, int m, int n) { p->m = m; p->n = n; p->ld = m; }

orc_ecg_t* orc_ecg_create(int N, int P, const int* rowpos, int T, int ortho_alg,
                          int bs_red, double tol, int maxIter) {
  if (T > 128) return NULL;
  orc_ecg_t* e = (orc_ecg_t*)calloc(1, sizeof(orc_ecg_t));
  e->N = N; e->P = P; e->T = T;
  e->rowpos = (int*)malloc((P + 1) * sizeof(int));
  memcpy(e->rowpos, rowpos, (P + 1) * sizeof(int));
  e->ortho_alg = ortho_alg; e->bs_red = bs_red; e->tol = tol; e->maxIter = maxIter;
  size_t m = (size_t)N, t = (size_t)T;
  /* ecg.c:55-61 */
  if (ortho_alg == ORC_ORTHOMIN) e->work_len = 5 * m * t + 2 * t * t;
  else if (ortho_alg == ORC_ORTHODIR) e->work_len = 7 * m * t + 3 * t * t;
  else e->work_len = 7 * m * t + 5 * t * t + 2 * t;
  e->work = (double*)calloc(e->work_len, sizeof(double));
  if (ortho_alg == ORC_ORTHOMIN) { /* ecg.c:66-74 */
    e->V.val = e->work; e->AV.val = e->work + m * t; e->Z.val = e->work + 2 * m * t;
    e->R.val = e->work + 3 * m * t; e->X.val = e->work + 4 * m * t;
    e->alpha.val = e->work + 5 * m * t; e->beta.val = e->work + 5 * m * t + t * t;
  } else { /* ecg.c:75-83 */
    e->V.val = e->work; e->AV.val = e->work + 2 * m * t; e->Z.val = e->work + 4 * m * t;
    e->R.val = e->work + 5 * m * t; e->X.val = e->work + 6 * m * t;
    e->alpha.val = e->work + 7 * m * t; e->beta.val = e->work + 7 * m * t + t * t;
  }
  e->Pd.val = e->V.val; e->APd.val = e->AV.val;
  e->gpart = (double*)malloc((size_t)P * 4 * t * t * sizeof(double));
  e->tmp = (double*)malloc((8 * t * t + 8 * t) * sizeof(double));
  return e;
}

void orc_ecg_destroy(orc_ecg_t* e) {
  if (!e) return;
  free(e->rowpos); free(e->work); free(e->gpart); free(e->tmp); free(e);
}

/* C (a x b, ldc) = A^T B summed over ranks; A is N x a, B is N x b
 * (CPLM_MatDenseKernelMatDotProd cplm_kernels.h:62-68 + MPI_Allreduce). */
static void gram(orc_ecg_t* e, const double* A, int lda, int a, const double* B,
                 int ldb, int b, double* C, int ldc) {
  if (a <= 0 || b <= 0) return;
  int ab = a * b;
#pragma omp parallel for schedule(static)
  for (int p = 0; p < e->P; ++p) {
    double* g = e->gpart + (size_t)p * ab;
    int r0 = e->rowpos[p], r1 = e->rowpos[p + 1];
    for (int q = 0; q < ab; ++q) g[q] = 0.0;
    for (int r = r0; r < r1; ++r)       /* one pass over the rows of the rank */
      for (int j = 0; j < b; ++j) {
        const double y = B[r + (size_t)ldb * j];
        for (int i = 0; i < a; ++i) g[i + a * j] += A[r + (size_t)lda * i] * y;
      }
  }
  for (int j = 0; j < b; ++j)
    for (int i = 0; i < a; ++i) {
      double s = 0.0;
      for (int p = 0; p < e->P; ++p) s += e->gpart[(size_t)p * ab + i + a * j];
      C[i + (size_t)ldc * j] = s;
    }
}

/* C (N x c) += sgn * A (N x a) * S (a x c, lds)  (cblas_dgemm N,N, beta=1). */
static void panel_update(int N, const double* A, int lda, int a, const double* S,
                         int lds, int c, double* C, int ldc, double sgn) {
#pragma omp parallel for schedule(static)
  for (int r = 0; r < N; ++r)
    for (int j = 0; j < c; ++j) {
      double s = 0.0;
      for (int k = 0; k < a; ++k) s += A[r + (size_t)lda * k] * S[k + (size_t)lds * j];
      C[r + (size_t)ldc * j] += sgn * s;
    }
}

static void copy_cols(int N, int ncol, const double* src, double* dst) {
  if (ncol <= 0 || src == dst) return;
#pragma omp parallel for schedule(static)
  for (int j = 0; j < ncol; ++j)
    memmove(dst + (size_t)N * j, src + (size_t)N * j, (size_t)N * sizeof(double));
}

/* ecg.c:98-171 + :201-221.  rhs is the concatenation of the per-rank rhs. */
static void ecg_reset(orc_ecg_t* e, const double* rhs, int* rci) {
  int N = e->N, t = e->T;
  setinfo(&e->X, N, t); setinfo(&e->R, N, t); setinfo(&e->Z, N, t);
  if (e->ortho_alg == ORC_ORTHOMIN) {
    setinfo(&e->V, N, t); setinfo(&e->AV, N, t);
    setinfo(&e->alpha, t, t); setinfo(&e->beta, t, t);
  } else {
    setinfo(&e->V, N, 2 * t); setinfo(&e->AV, N, 2 * t);
    setinfo(&e->alpha, t, t); setinfo(&e->beta, 2 * t, t);
  }
  setinfo(&e->Pd, N, t); setinfo(&e->APd, N, t);
  double nb = 0.0;
  for (int p = 0; p < e->P; ++p) { /* per-rank sum, then allreduce */
    double s = 0.0;
    for (int i = e->rowpos[p]; i < e->rowpos[p + 1]; ++i) s += rhs[i] * rhs[i];
    nb += s;
  }
  e->normb = sqrt(nb);
  e->res = 1.0; e->iter = 0; e->bs = t; e->kbs = e->V.n;
  for (int p = 0; p < e->P; ++p) { /* R0(:, rank % t) = rhs */
    int col = p % t;
    for (int i = e->rowpos[p]; i < e->rowpos[p + 1]; ++i)
      e->R.val[i + (size_t)e->R.ld * col] = rhs[i];
  }
  *rci = 0;
}

int orc_ecg_initialize(orc_ecg_t* e, const double* rhs, int* rci) {
  if (e->P < e->T) return 1; /* ecg.c:178-183 aborts */
  memset(e->work, 0, e->work_len * sizeof(double));
  ecg_reset(e, rhs, rci);
  /* the LAPACK warm-up at ecg.c:190-196 acts on a zero pool: a no-op */
  return 0;
}

/* ecg.c:223-271 */
int orc_ecg_stopping(orc_ecg_t* e, int* stop) {
  int N = e->N, t = e->T;
  double* RtR;
  if (e->ortho_alg == ORC_ORTHOMIN) RtR = e->work + 5 * (size_t)N * t + (size_t)t * t;
  else if (e->ortho_alg == ORC_ORTHODIR) RtR = e->work + 7 * (size_t)N * t + 2 * (size_t)t * t;
  else RtR = e->tmp; /* reference leaves the pointer NULL here: never called */
  gram(e, e->R.val, e->R.ld, t, e->R.val, e->R.ld, t, RtR, t);
  double s = 0.0;
  for (int i = 0; i < t; ++i) s += RtR[i + (size_t)t * i];
  e->res = sqrt(s);
  *stop = (e->res > e->normb * e->tol && e->iter < e->maxIter && e->bs > 0) ? 0 : 1;
  return 0;
}

/* ecg.c:289-400 */
static int iterate_omin(orc_ecg_t* e, int* rci) {
  int N = e->N, nrhs = e->T, ierr = 0;
  int t = e->Pd.n;
  double* ws = e->work + 5 * (size_t)N * nrhs + (size_t)nrhs * nrhs;
  if (*rci == 0) {
    gram(e, e->APd.val, e->APd.ld, t, e->Pd.val, e->Pd.ld, t, ws, t);
    ierr = potrf_upper(t, ws, t);
    if (ierr) return 2; /* reference aborts: "P^tAP is not spd" */
    trsm_right_upper(N, t, ws, t, e->Pd.val, e->Pd.ld);
    trsm_right_upper(N, t, ws, t, e->APd.val, e->APd.ld);
    gram(e, e->Pd.val, e->Pd.ld, e->alpha.m, e->R.val, e->R.ld, e->alpha.n,
         e->alpha.val, e->alpha.ld);
    panel_update(N, e->Pd.val, e->Pd.ld, e->Pd.n, e->alpha.val, e->alpha.ld, e->X.n,
                 e->X.val, e->X.ld, 1.0);
    panel_update(N, e->APd.val, e->APd.ld, e->APd.n, e->alpha.val, e->alpha.ld,
                 e->R.n, e->R.val, e->R.ld, -1.0);
    e->iter++;
    *rci = 1;
  } else {
    gram(e, e->APd.val, e->APd.ld, e->beta.m, e->Z.val, e->Z.ld, e->beta.n,
         e->beta.val, e->beta.ld);
    panel_update(N, e->Pd.val, e->Pd.ld, e->Pd.n, e->beta.val, e->beta.ld, e->Z.n,
                 e->Z.val, e->Z.ld, -1.0);
    copy_cols(N, nrhs, e->Z.val, e->Pd.val);
    if (e->bs_red == ORC_ADAPT_BS) {
      gram(e, e->Pd.val, e->Pd.ld, nrhs, e->Pd.val, e->Pd.ld, nrhs, ws, nrhs);
      pstrf_upper(nrhs, ws, nrhs, e->iwork, &t, -1.0, e->tmp);
      lapmt_forward(N, nrhs, e->Pd.val, N, e->iwork);
      trsm_right_upper(N, t, ws, nrhs, e->Pd.val, N);
      setinfo(&e->Pd, N, t); setinfo(&e->APd, N, t);
      setinfo(&e->alpha, t, nrhs); setinfo(&e->beta, t, nrhs);
      e->bs = t;
    }
    *rci = 0;
  }
  return 0;
}

/* Reduction of the search directions shared by Odir (ecg.c:445-497) and the
 * fused variant (ecg.c:593-637).  scratch is t x T (ld t), tau after it. */
static void odir_reduce(orc_ecg_t* e, double* scratch, double* tau, int with_Z) {
  int N = e->N, nrhs = e->T, t = e->Pd.n, t1 = 0;
  double tol = e->tol * e->normb / sqrt((double)nrhs);
  memcpy(scratch, e->alpha.val, sizeof(double) * (size_t)e->alpha.m * e->alpha.n);
  svd_left(t, nrhs, scratch, e->alpha.ld, tau, e->tmp);
  for (int i = 0; i < t; ++i) { if (tau[i] > tol) t1++; else break; }
  if (t1 > 0 && t1 < nrhs && t1 < t) {
    geqrf(t, t, scratch, t, tau);
    ormqr_left_trans(t, nrhs, scratch, t, tau, e->alpha.val, t);
    ormqr_right_notrans(N, t, scratch, t, tau, e->Pd.val, N);
    ormqr_right_notrans(N, t, scratch, t, tau, e->APd.val, N);
    if (with_Z) ormqr_right_notrans(N, t, scratch, t, tau, e->Z.val, N);
    imatcopy_shrink(t, nrhs, e->alpha.val, t, t1);
    setinfo(&e->alpha, t1, nrhs);
    setinfo(&e->Pd, N, t1); setinfo(&e->APd, N, t1); setinfo(&e->Z, N, t1);
    e->bs = t1;
    e->kbs = t + nrhs;
  }
  setinfo(&e->beta, e->kbs, t1);
  setinfo(&e->V, N, e->kbs); setinfo(&e->AV, N, e->kbs);
}

/* ecg.c:402-530 */
static int iterate_odir(orc_ecg_t* e, int* rci) {
  int N = e->N, nrhs = e->T;
  int t = e->Pd.n;
  if (*rci == 0) {
    double* ws = e->work + 7 * (size_t)N * nrhs + (size_t)nrhs * nrhs;
    gram(e, e->APd.val, e->APd.ld, t, e->Pd.val, e->Pd.ld, t, ws, t);
    (void)potrf_upper(t, ws, t); /* failure ignored, ecg.c:431 */
    trsm_right_upper(N, t, ws, t, e->Pd.val, e->Pd.ld);
    trsm_right_upper(N, t, ws, t, e->APd.val, e->APd.ld);
    gram(e, e->Pd.val, e->Pd.ld, e->alpha.m, e->R.val, e->R.ld, e->alpha.n,
         e->alpha.val, e->alpha.ld);
    if (e->bs_red == ORC_ADAPT_BS) odir_reduce(e, ws, ws + (size_t)nrhs * t, 0);
    panel_update(N, e->Pd.val, e->Pd.ld, e->Pd.n, e->alpha.val, e->alpha.ld, e->X.n,
                 e->X.val, e->X.ld, 1.0);
    panel_update(N, e->APd.val, e->APd.ld, e->APd.n, e->alpha.val, e->alpha.ld,
                 e->R.n, e->R.val, e->R.ld, -1.0);
    e->iter++;
    *rci = 1;
  } else {
    gram(e, e->AV.val, e->AV.ld, e->beta.m, e->Z.val, e->Z.ld, e->beta.n,
         e->beta.val, e->beta.ld);
    panel_update(N, e->V.val, e->V.ld, e->V.n, e->beta.val, e->beta.ld, e->Z.n,
                 e->Z.val, e->Z.ld, -1.0);
    copy_cols(N, t, e->V.val, e->V.val + (size_t)N * nrhs);
    copy_cols(N, t, e->AV.val, e->AV.val + (size_t)N * nrhs);
    copy_cols(N, t, e->Z.val, e->V.val);
    *rci = 0;
  }
  return 0;
}

/* ecg.c:532-658 */
static int iterate_odir_fused(orc_ecg_t* e, int* rci) {
  int N = e->N, nrhs = e->T;
  int t = e->Pd.n;
  double* mu = e->alpha.val + 3 * (size_t)nrhs * nrhs;
  double* rtr = e->alpha.val + 4 * (size_t)nrhs * nrhs;
  gram(e, e->Pd.val, e->Pd.ld, e->alpha.m, e->R.val, e->R.ld, e->alpha.n,
       e->alpha.val, e->alpha.ld);
  gram(e, e->AV.val, e->AV.ld, e->beta.m, e->Z.val, e->Z.ld, e->beta.n, e->beta.val,
       e->beta.ld);
  gram(e, e->APd.val, e->APd.ld, t, e->Pd.val, e->Pd.ld, t, mu, t);
  gram(e, e->R.val, e->R.ld, nrhs, e->R.val, e->R.ld, nrhs, rtr, nrhs);
  double s = 0.0;
  for (int i = 0; i < nrhs; ++i) s += rtr[i + (size_t)nrhs * i];
  e->res = sqrt(s);
  *rci = (e->res < e->tol * e->normb || e->iter > e->maxIter) ? 1 : 0;
  (void)potrf_upper(t, mu, t);
  trsm_right_upper(N, e->Pd.n, mu, t, e->Pd.val, e->Pd.ld);
  trsm_right_upper(N, e->APd.n, mu, t, e->APd.val, e->APd.ld);
  trsm_right_upper(e->beta.m, e->beta.n, mu, t, e->beta.val, e->beta.ld);
  trsm_right_upper(N, e->Z.n, mu, t, e->Z.val, e->Z.ld);
  trsm_left_upper_trans(t, nrhs, mu, t, e->alpha.val, t);
  trsm_left_upper_trans(t, t, mu, t, e->beta.val, e->kbs);
  panel_update(N, e->V.val, e->V.ld, e->V.n, e->beta.val, e->beta.ld, e->Z.n,
               e->Z.val, e->Z.ld, -1.0);
  if (e->bs_red == ORC_ADAPT_BS)
    odir_reduce(e, mu, e->alpha.val + 5 * (size_t)nrhs * nrhs, 1);
  panel_update(N, e->Pd.val, e->Pd.ld, e->Pd.n, e->alpha.val, e->alpha.ld, e->X.n,
               e->X.val, e->X.ld, 1.0);
  panel_update(N, e->APd.val, e->APd.ld, e->APd.n, e->alpha.val, e->alpha.ld, e->R.n,
               e->R.val, e->R.ld, -1.0);
  e->iter++;
  copy_cols(N, e->bs, e->V.val, e->V.val + (size_t)N * nrhs);
  copy_cols(N, e->bs, e->AV.val, e->AV.val + (size_t)N * nrhs);
  copy_cols(N, e->bs, e->Z.val, e->V.val);
  return 0;
}

int orc_ecg_iterate(orc_ecg_t* e, int* rci) {
  if (e->ortho_alg == ORC_ORTHOMIN) return iterate_omin(e, rci);
  if (e->ortho_alg == ORC_ORTHODIR) return iterate_odir(e, rci);
  return iterate_odir_fused(e, rci);
}

/* ecg.c:668-677: sol = X * ones */
int orc_ecg_finalize(orc_ecg_t* e, double* sol) {
  for (int i = 0; i < e->N; ++i) {
    double s = 0.0;
    for (int j = 0; j < e->X.n; ++j) s += e->X.val[i + (size_t)e->X.ld * j];
    sol[i] = s;
  }
  return 0;
}

/* accessors for the caller-side half of the RCI protocol */
double* orc_ecg_panel(orc_ecg_t* e, int which) {
  switch (which) {
    case 0: return e->Pd.val;
    case 1: return e->APd.val;
    case 2: return e->R.val;
    case 3: return e->Z.val;
    case 4: return e->X.val;
    case 5: return e->alpha.val;
    case 6: return e->beta.val;
    case 7: return e->V.val;
    case 8: return e->AV.val;
  }
  return NULL;
}
int orc_ecg_bs(const orc_ecg_t* e) { return e->bs; }
int orc_ecg_kbs(const orc_ecg_t* e) { return e->kbs; }
int orc_ecg_iter(const orc_ecg_t* e) { return e->iter; }
int orc_ecg_pn(const orc_ecg_t* e) { return e->Pd.n; }
double orc_ecg_res(const orc_ecg_t* e) { return e->res; }
double orc_ecg_normb(const orc_ecg_t* e) { return e->normb; }

static double now_s(void) {
#ifdef _OPENMP
  return omp_get_wtime();
#else
  return (double)clock() / CLOCKS_PER_SEC;
#endif
}

/* The driver loop of examples/test_ecg_prealps_op.c:203-223 (and of
 * examples/test_ecg_bench_fused.c:243-259 for the fused variant) around the
 * functions above.  res_hist/bs_hist get one entry per stopping test (per
 * call for the fused variant).  timers: [0] total loop, [1] operator,
 * [2] preconditioner. Returns the number of history entries. */
int orc_ecg_solve(orc_ecg_t* e, int n, const int* rowptr, const int* colind,
                  const double* val, const orc_bj_t* bj, const double* rhs,
                  double* sol, double* res_hist, int* bs_hist, int maxhist,
                  double* timers) {
  int rci = 0, stop = 0, nh = 0;
  double t0, top = 0, tpre = 0, tall = now_s();
  if (orc_ecg_initialize(e, rhs, &rci)) return -1;
  int N = e->N;
  t0 = now_s(); orc_bj_apply(bj, e->Pd.n, e->R.val, N, e->Pd.val, N); tpre += now_s() - t0;
  if (e->ortho_alg != ORC_ORTHODIR_FUSED) {
    t0 = now_s(); orc_spmm(n, rowptr, colind, val, e->Pd.n, e->Pd.val, N, e->APd.val, N); top += now_s() - t0;
    while (stop != 1) {
      if (orc_ecg_iterate(e, &rci)) { nh = -2; break; }
      if (rci == 0) {
        t0 = now_s(); orc_spmm(n, rowptr, colind, val, e->Pd.n, e->Pd.val, N, e->APd.val, N); top += now_s() - t0;
      } else {
        orc_ecg_stopping(e, &stop);
        if (nh < maxhist) { res_hist[nh] = e->res; bs_hist[nh] = e->bs; nh++; }
        if (stop == 1) break;
        t0 = now_s();
        if (e->ortho_alg == ORC_ORTHOMIN) orc_bj_apply(bj, e->R.n, e->R.val, N, e->Z.val, N);
        else orc_bj_apply(bj, e->APd.n, e->APd.val, N, e->Z.val, N);
        tpre += now_s() - t0;
      }
    }
  } else {
    while (rci != 1) {
      t0 = now_s(); orc_spmm(n, rowptr, colind, val, e->Pd.n, e->Pd.val, N, e->APd.val, N); top += now_s() - t0;
      t0 = now_s(); orc_bj_apply(bj, e->APd.n, e->APd.val, N, e->Z.val, N); tpre += now_s() - t0;
      orc_ecg_iterate(e, &rci);
      if (nh < maxhist) { res_hist[nh] = e->res; bs_hist[nh] = e->bs; nh++; }
    }
  }
  if (timers) { timers[0] = now_s() - tall; timers[1] = top; timers[2] = tpre; }
  if (sol) orc_ecg_finalize(e, sol);
  return nh;
}
