/*
 * dense_ops.c -- the tall-skinny panel kernels of the ECG iteration as stand-alone entry
 * points on device panels, so that each one can be checked against a host computation by
 * itself (a wrong tile of an MFMA kernel would otherwise only show as a drifting residual).
 * They launch exactly what ecg.c launches:
 *   preAlps_hip_panel_gram         C = [A0 | A1]^T B        (cblas_dgemm at ecg.c:311,330,347,425,438,510)
 *   preAlps_hip_panel_update       Z -= [V0 | V1] beta      (ecg.c:354,517)
 *   preAlps_hip_panel_trsm_update  P U^-1, AP U^-1, X += P alpha, R -= AP alpha, column sums of R^2
 *                                  (cblas_dtrsm ecg.c:324-327,434-435 + dgemm :337-338,500-501 + :250)
 */
#include <stdlib.h>
#include <string.h>

#include "pa_host.h"

int preAlps_hip_panel_gram(const CPLM_Mat_Dense_t* A0, const CPLM_Mat_Dense_t* A1, const CPLM_Mat_Dense_t* B,
                           double* host_out, int ld_out) {
  PA_REQUIRE_GPU();
  if (!A0 || !B || !A0->val || !B->val || !host_out) return PA_FAIL(" wrong test 'A0 != NULL && B != NULL'");
  int m = A0->info.m, ts = pa_desc_stride(A0);
  if (B->info.m != m || pa_desc_stride(B) != ts || (A1 && (A1->info.m != m || pa_desc_stride(A1) != ts)))
    return PA_FAIL("panel shapes do not match");
  int a_lo = A0->info.n, a_hi = A1 ? A1->info.n : 0, nb = B->info.n, na = a_lo + a_hi;
  if (ld_out < na) return PA_FAIL("leading dimension %d below %d rows", ld_out, na);
  size_t pbytes = (size_t)pa_gram_max_blocks() * 2 * ts * ts * sizeof(double);
  double* d_part = (double*)pa_rt_malloc(pbytes);
  double* d_out = (double*)pa_rt_malloc((size_t)(na * nb > 0 ? na * nb : 1) * sizeof(double));
  double* h = (double*)malloc((size_t)(na * nb > 0 ? na * nb : 1) * sizeof(double));
  int rc = !d_part || !d_out || !h;
  rc = rc || pa_rt_memset(d_part, 0, pbytes);        /* (the ticket of the wide sum lives in this buffer) */
  rc = rc || pa_k_gram_finish(m, ts, A0->val, A1 ? A1->val : NULL, B->val, d_part, a_lo, a_hi, nb, d_out, na,
                              0, 0, NULL, NULL, NULL);
  rc = rc || pa_rt_d2h(h, d_out, (size_t)na * nb * sizeof(double));
  if (!rc)
    for (int j = 0; j < nb; ++j) memcpy(host_out + (size_t)ld_out * j, h + (size_t)na * j, (size_t)na * sizeof(double));
  pa_rt_free(d_part); pa_rt_free(d_out); free(h);
  if (rc) return PA_FAIL("Gram product failed: %s", pa_rt_error());
  return 0;
}

int preAlps_hip_panel_update(CPLM_Mat_Dense_t* Z, const CPLM_Mat_Dense_t* V0, const CPLM_Mat_Dense_t* V1,
                             const double* host_beta, int ldb) {
  PA_REQUIRE_GPU();
  if (!Z || !V0 || !Z->val || !V0->val || !host_beta) return PA_FAIL(" wrong test 'Z != NULL && V0 != NULL'");
  int m = Z->info.m, ts = pa_desc_stride(Z);
  if (V0->info.m != m || pa_desc_stride(V0) != ts || (V1 && (V1->info.m != m || pa_desc_stride(V1) != ts)))
    return PA_FAIL("panel shapes do not match");
  int a_lo = V0->info.n, a_hi = V1 ? V1->info.n : 0, nc = Z->info.n, na = a_lo + a_hi;
  if (ldb < na) return PA_FAIL("leading dimension %d below %d rows", ldb, na);
  double* d_beta = (double*)pa_rt_malloc((size_t)(ldb * nc > 0 ? ldb * nc : 1) * sizeof(double));
  int rc = !d_beta || pa_rt_h2d(d_beta, host_beta, (size_t)ldb * nc * sizeof(double));
  rc = rc || pa_k_update_z(m, ts, a_lo, a_hi, nc, d_beta, ldb, V0->val, V1 ? V1->val : V0->val, Z->val, NULL, NULL, NULL, NULL, NULL, 0, NULL);
  rc = rc || pa_rt_sync();
  pa_rt_free(d_beta);
  if (rc) return PA_FAIL("panel update failed: %s", pa_rt_error());
  return 0;
}

int preAlps_hip_panel_trsm_update(CPLM_Mat_Dense_t* P, CPLM_Mat_Dense_t* AP, CPLM_Mat_Dense_t* X,
                                  CPLM_Mat_Dense_t* R, const double* host_U, const double* host_alpha,
                                  double* host_res2) {
  PA_REQUIRE_GPU();
  if (!P || !AP || !X || !R || !host_U || !host_alpha) return PA_FAIL(" wrong test 'P, AP, X, R != NULL'");
  int m = P->info.m, ts = pa_desc_stride(P), t = P->info.n, nc = X->info.n;
  if (AP->info.m != m || X->info.m != m || R->info.m != m || pa_desc_stride(AP) != ts || pa_desc_stride(X) != ts ||
      pa_desc_stride(R) != ts || AP->info.n != t || R->info.n != nc)
    return PA_FAIL("panel shapes do not match");
  int nblk = 0;
  double* d_small = (double*)pa_rt_malloc(((size_t)t * t + (size_t)t * nc + 8) * sizeof(double));
  double* d_rtr = (double*)pa_rt_malloc((size_t)pa_gram_max_blocks() * ts * sizeof(double));
  int* d_info = (int*)pa_rt_malloc(8 * sizeof(int));
  int rc = !d_small || !d_rtr || !d_info;
  rc = rc || pa_rt_h2d(d_small, host_U, (size_t)t * t * sizeof(double));
  rc = rc || pa_rt_h2d(d_small + (size_t)t * t, host_alpha, (size_t)t * nc * sizeof(double));
  rc = rc || pa_rt_memset(d_info, 0, 8 * sizeof(int));
  double* d_res2 = d_small + (size_t)t * t + (size_t)t * nc;
  rc = rc || pa_k_trsm_update(m, ts, t, nc, d_small, d_small + (size_t)t * t, P->val, AP->val, X->val, R->val,
                              d_rtr, &nblk, nc, d_res2, d_info, NULL, NULL, NULL);
  double res2[2] = {0.0, 0.0};
  rc = rc || pa_rt_d2h(res2, d_res2, 2 * sizeof(double));
  if (host_res2) *host_res2 = res2[0];
  pa_rt_free(d_small); pa_rt_free(d_rtr); pa_rt_free(d_info);
  if (rc) return PA_FAIL("triangular solve + update failed: %s", pa_rt_error());
  return 0;
}

/* BF-Omin's second half on its own (src/solvers/ecg.c:358-393 of the reference): P(:, c) = Z(:, piv[c]) for the n
 * columns of the panel, then the leading t columns times U^-1 (t x t upper triangular, host, column major).
 * one_pass != 0: the solver's kernel (k_permute_trsm); 0: the three kernels it replaces (copy, permutation,
 * triangular solve), whose result it reproduces bit for bit. */
int preAlps_hip_panel_permute_solve(const CPLM_Mat_Dense_t* Z, CPLM_Mat_Dense_t* P, const int* host_piv, int t,
                                    const double* host_U, int one_pass) {
  PA_REQUIRE_GPU();
  if (!Z || !P || !Z->val || !P->val || !host_piv || !host_U) return PA_FAIL(" wrong test 'Z, P, piv, U != NULL'");
  int m = Z->info.m, ts = pa_desc_stride(Z), n = Z->info.n;
  if (P->info.m != m || pa_desc_stride(P) != ts || P->info.n != n || t < 0 || t > n) return PA_FAIL("panel shapes do not match");
  for (int c = 0; c < n; ++c) if (host_piv[c] < 0 || host_piv[c] >= n) return PA_FAIL("pivot %d outside the panel", host_piv[c]);
  double* d_u = (double*)pa_rt_malloc(((size_t)t * t + 1) * sizeof(double));
  int* d_piv = (int*)pa_rt_malloc((size_t)(n > 0 ? n : 1) * sizeof(int));
  int rc = !d_u || !d_piv;
  rc = rc || pa_rt_h2d(d_u, host_U, (size_t)t * t * sizeof(double)) || pa_rt_h2d(d_piv, host_piv, (size_t)n * sizeof(int));
  if (one_pass) rc = rc || pa_k_permute_trsm(m, ts, n, d_piv, t, d_u, Z->val, P->val);
  else rc = rc || pa_k_copy_cols(m, ts, n, Z->val, P->val) || pa_k_permute_cols(m, ts, n, d_piv, P->val) ||
            pa_k_trsm(m, ts, t, d_u, P->val, NULL);
  rc = rc || pa_rt_sync();
  pa_rt_free(d_u); pa_rt_free(d_piv);
  if (rc) return PA_FAIL("permutation + triangular solve failed: %s", pa_rt_error());
  return 0;
}
