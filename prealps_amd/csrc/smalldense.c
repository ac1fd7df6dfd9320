/* smalldense.c -- see smalldense.h.  Orders are <= 16, so clarity beats speed. */
#include <float.h>
#include <math.h>
#include <string.h>

#include "smalldense.h"

#define SD_MAX 16

void pa_sd_left_singular(int t, int n, const double* A, int lda, double* U, double* sigma) {
  /* Hestenes one-sided Jacobi on the rows of G = U^T A: rotate row pairs
   * until they are mutually orthogonal; then A = U G with orthogonal rows. */
  double G[SD_MAX][2 * SD_MAX];
  for (int i = 0; i < t; ++i)
    for (int j = 0; j < n; ++j) G[i][j] = A[i + (size_t)lda * j];
  for (int j = 0; j < t; ++j)
    for (int i = 0; i < t; ++i) U[i + (size_t)t * j] = (i == j) ? 1.0 : 0.0;
  const double eps = DBL_EPSILON / 4;
  for (int sweep = 0; sweep < 64; ++sweep) {
    int rotated = 0;
    for (int p = 0; p + 1 < t; ++p)
      for (int q = p + 1; q < t; ++q) {
        double app = 0.0, aqq = 0.0, apq = 0.0;
        for (int j = 0; j < n; ++j) { app += G[p][j] * G[p][j]; aqq += G[q][j] * G[q][j]; apq += G[p][j] * G[q][j]; }
        if (apq == 0.0 || fabs(apq) <= eps * sqrt(app * aqq)) continue;
        rotated = 1;
        double theta = (aqq - app) / (2.0 * apq);
        double tn = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + hypot(1.0, theta));
        double c = 1.0 / sqrt(1.0 + tn * tn), s = c * tn;
        for (int j = 0; j < n; ++j) {
          double gp = G[p][j], gq = G[q][j];
          G[p][j] = c * gp - s * gq;
          G[q][j] = s * gp + c * gq;
        }
        for (int i = 0; i < t; ++i) {
          double up = U[i + (size_t)t * p], uq = U[i + (size_t)t * q];
          U[i + (size_t)t * p] = c * up - s * uq;
          U[i + (size_t)t * q] = s * up + c * uq;
        }
      }
    if (!rotated) break;
  }
  for (int p = 0; p < t; ++p) {
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += G[p][j] * G[p][j];
    sigma[p] = sqrt(s);
  }
  for (int p = 0; p + 1 < t; ++p) { /* decreasing order, columns of U follow */
    int best = p;
    for (int q = p + 1; q < t; ++q) if (sigma[q] > sigma[best]) best = q;
    if (best == p) continue;
    double ts = sigma[p]; sigma[p] = sigma[best]; sigma[best] = ts;
    for (int i = 0; i < t; ++i) {
      double tu = U[i + (size_t)t * p];
      U[i + (size_t)t * p] = U[i + (size_t)t * best];
      U[i + (size_t)t * best] = tu;
    }
  }
}

void pa_sd_qr_q(int t, double* Q) {
  double v[SD_MAX][SD_MAX], tau[SD_MAX];
  /* Householder vectors, LAPACK dlarfg: H = I - tau v v^T, v[0] = 1 */
  for (int k = 0; k < t; ++k) {
    double* x = Q + k + (size_t)t * k;
    int len = t - k;
    double alpha = x[0], xn = 0.0;
    for (int i = 1; i < len; ++i) xn += x[i] * x[i];
    xn = sqrt(xn);
    v[k][0] = 1.0;
    if (xn == 0.0) {
      tau[k] = 0.0;
      for (int i = 1; i < len; ++i) v[k][i] = 0.0;
    } else {
      double beta = -copysign(hypot(alpha, xn), alpha);
      tau[k] = (beta - alpha) / beta;
      for (int i = 1; i < len; ++i) v[k][i] = x[i] / (alpha - beta);
    }
    for (int j = k + 1; j < t; ++j) { /* apply H_k to the trailing columns */
      double* c = Q + k + (size_t)t * j;
      double w = 0.0;
      for (int i = 0; i < len; ++i) w += v[k][i] * c[i];
      w *= tau[k];
      for (int i = 0; i < len; ++i) c[i] -= w * v[k][i];
    }
  }
  /* Q = H_0 H_1 ... H_{t-1} applied to the identity, last reflector first */
  for (int j = 0; j < t; ++j)
    for (int i = 0; i < t; ++i) Q[i + (size_t)t * j] = (i == j) ? 1.0 : 0.0;
  for (int k = t - 1; k >= 0; --k) {
    int len = t - k;
    for (int j = 0; j < t; ++j) {
      double* c = Q + k + (size_t)t * j;
      double w = 0.0;
      for (int i = 0; i < len; ++i) w += v[k][i] * c[i];
      w *= tau[k];
      for (int i = 0; i < len; ++i) c[i] -= w * v[k][i];
    }
  }
}

void pa_sd_qt_times(int t, int n, const double* Q, double* B) {
  double col[SD_MAX];
  for (int j = 0; j < n; ++j) {
    double* b = B + (size_t)t * j;
    for (int i = 0; i < t; ++i) {
      double s = 0.0;
      for (int k = 0; k < t; ++k) s += Q[k + (size_t)t * i] * b[k];
      col[i] = s;
    }
    memcpy(b, col, t * sizeof(double));
  }
}

int pa_sd_pstrf_upper(int n, double* A, int lda, int* piv, int* rank, double tol) {
  /* right-looking form: after step j the trailing block holds the Schur
   * complement, whose diagonal drives the pivot choice */
  double S[SD_MAX][SD_MAX]; /* full symmetric working copy */
  if (n <= 0) { *rank = 0; return 0; }
  S[0][0] = 0.0;
  for (int j = 0; j < n; ++j)
    for (int i = 0; i <= j; ++i) S[i][j] = S[j][i] = A[i + (size_t)lda * j];
  for (int i = 0; i < n; ++i) piv[i] = i + 1;
  double dmax = S[0][0];
  for (int i = 1; i < n; ++i) if (S[i][i] > dmax) dmax = S[i][i];
  if (!(dmax > 0.0)) { *rank = 0; return 1; }
  double stop = tol < 0.0 ? n * (DBL_EPSILON / 2) * dmax : tol;
  int r = n, info = 0;
  for (int j = 0; j < n; ++j) {
    int p = j;
    for (int i = j + 1; i < n; ++i) if (S[i][i] > S[p][p]) p = i;
    if (j > 0 && !(S[p][p] > stop)) { r = j; info = 1; break; }
    if (p != j) { /* symmetric interchange of rows/columns j and p */
      for (int k = 0; k < n; ++k) { double tv = S[j][k]; S[j][k] = S[p][k]; S[p][k] = tv; }
      for (int k = 0; k < n; ++k) { double tv = S[k][j]; S[k][j] = S[k][p]; S[k][p] = tv; }
      int ti = piv[j]; piv[j] = piv[p]; piv[p] = ti;
    }
    double d = sqrt(S[j][j]);
    S[j][j] = d;
    for (int k = j + 1; k < n; ++k) S[j][k] /= d;
    for (int i = j + 1; i < n; ++i)
      for (int k = i; k < n; ++k) { S[i][k] -= S[j][i] * S[j][k]; S[k][i] = S[i][k]; }
    for (int i = j + 1; i < n; ++i) S[i][j] = 0.0;
  }
  for (int j = 0; j < n; ++j)
    for (int i = 0; i <= j; ++i) A[i + (size_t)lda * j] = S[i][j];
  *rank = r;
  return info;
}
