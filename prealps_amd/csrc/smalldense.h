/*
 * smalldense.h -- t x t host kernels used only by the block-size reduction
 * paths (D-Odir ecg.c:445-497, BF-Omin ecg.c:361-393), where the reference
 * calls LAPACKE_dgesvd / dgeqrf / dormqr / dpstrf on matrices of order
 * t <= enlFac.  Column-major throughout.
 */
#ifndef PA_SMALLDENSE_H
#define PA_SMALLDENSE_H
/* Left singular vectors U (t x t, ld t) and singular values (decreasing) of the
 * t x n matrix A (ld lda), t <= n.  A is not modified. */
void pa_sd_left_singular(int t, int n, const double* A, int lda, double* U, double* sigma);
/* Overwrite the t x t matrix Q (ld t) with the orthogonal factor of its
 * Householder QR factorisation (LAPACK dgeqr2 / dorg2r conventions). */
void pa_sd_qr_q(int t, double* Q);
/* B <- Q^T B, Q is t x t (ld t), B is t x n (ld t). */
void pa_sd_qt_times(int t, int n, const double* Q, double* B);
/* Cholesky with complete diagonal pivoting of the upper triangle, P^T A P =
 * U^T U (LAPACK dpstrf 'U').  tol < 0 selects n * eps * max diag.  piv is
 * 1-based; returns 0 for full rank, 1 if the factorisation stopped at *rank. */
int pa_sd_pstrf_upper(int n, double* A, int lda, int* piv, int* rank, double tol);
#endif
