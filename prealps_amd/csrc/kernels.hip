// kernels.hip -- CDNA4 (gfx950) kernels of the ECG block iteration and their
// C launchers (pa_device.h).  All panels are row-interleaved [rows][TS] fp64,
// all small t x t blocks column-major like the reference's work area.
//
// Every kernel here is HBM-bandwidth bound (SpMM: 0.67 flop/B at t = 4; the
// tall-skinny kernels t/8..t/4 flop/B against a ridge of ~10 flop/B), so the
// design rules are coalesced 16-B accesses, LDS staging where rows are reused,
// 64-wide shuffle reductions, and XCD-aware block order for the SpMM gathers.
#include "kernels_common.h"


namespace {
// Window (rows in flight = record length) of a wide block and the register sets per lane it
// needs: the same rule as block_jacobi.c (bj_wide_window).
__host__ __device__ inline int bjw_window(int w) {
  int W = (w + 64 + 63) & ~63;
  if (W <= 1024) return W;
  W = (w + 64 + 127) & ~127;
  if (W <= 2048) return W;
  return (w + 64 + 255) & ~255;
}

// ------------------------------------------------ block-Jacobi setup ----
// Band Cholesky of one diagonal block per workgroup (bands up to PA_BJ_FACTOR_WMAX), right
// looking: the (w+1) x (w+1) window of rows j..j+w lives in LDS (row i in slot i mod (w+1),
// win[slot][d] = A(i, i-d)); step j takes the pivot, scales column j, writes the two sweep
// records the solve kernels read -- forward record j = column j / L(j,j), backward record
// b-1-j = row j / L(j,j) -- and applies the rank-1 update to the rest of the window while
// the next row streams in.  band: the block's rows in factor order, (w+1) doubles each.
__global__ __launch_bounds__(WG) void k_bj_factor(
    const int* __restrict__ list, const int* __restrict__ row0, const int* __restrict__ nrows,
    const int* __restrict__ bw, const long long* __restrict__ off, const long long* __restrict__ boff,
    const double* __restrict__ band, double* __restrict__ Lf, double* __restrict__ Lb,
    double* __restrict__ invd_f, double* __restrict__ invd_b, int* __restrict__ fail) {
  extern __shared__ double win[];
  const int p = list[blockIdx.x];
  const int r0 = row0[p], b = nrows[p], w = bw[p];
  const int ld = w + 1, wr = (w + 2) & ~1;
  const double* __restrict__ A = band + boff[p];
  double* __restrict__ f = Lf + off[p];
  double* __restrict__ g = Lb + off[p];
  const int tid = threadIdx.x;
  const int nfirst = (b < ld ? b : ld) * ld;
  for (int e = tid; e < nfirst; e += WG) win[e] = A[e];
  __syncthreads();
  for (int j = 0; j < b; ++j) {
    const int rj = j % ld, jb = b - 1 - j;
    const double d0 = win[rj * ld];
    if (!(d0 > 0.0) && tid == 0) atomicCAS(fail, 0, r0 + j + 1);
    const double piv = (d0 > 0.0) ? sqrt(d0) : __longlong_as_double(0x7ff8000000000000LL);
    const double invp = 1.0 / piv;
    for (int t = tid; t < w; t += WG) {
      const int dd = t + 1, i = j + dd;
      g[(size_t)jb * wr + t] = (j - dd >= 0) ? win[rj * ld + dd] * invp : 0.0;
      if (i < b) {
        const int ri = i % ld;
        const double l = win[ri * ld + dd] / piv;
        win[ri * ld + dd] = l;
        f[(size_t)j * wr + t] = l * invp;
      }
    }
    if (tid == 0) { invd_f[r0 + j] = invp; invd_b[r0 + jb] = invp; }
    __syncthreads();
    const int nb = (b - 1 - j) < w ? (b - 1 - j) : w;     // rows below the pivot inside the band
    for (int e = tid; e < nb * nb; e += WG) {
      const int a = e / nb + 1, c = e - (a - 1) * nb + 1;
      if (c <= a) {
        const int ri = (j + a) % ld, rk = (j + c) % ld;
        win[ri * ld + (a - c)] -= win[ri * ld + a] * win[rk * ld + c];
      }
    }
    const int in = j + w + 1;                             // the row that takes over slot rj
    if (in < b)
      for (int e = tid; e < ld; e += WG) win[rj * ld + e] = A[(size_t)in * ld + e];
    __syncthreads();
  }
}

// The same factorisation for wider bands (up to 4032), blocked by NB columns, one workgroup of
// 1024 threads per block.  The band is stored diagonal-major (A(i, i-d) at band[d*b + i]) so
// that a wavefront working on one diagonal touches consecutive addresses.  Per block column:
// the NB x NB diagonal block is factored by one thread in LDS; one thread per row solves the
// panel rows against it and leaves them in LDS (w x NB doubles: NB = 16 / 8 / 4 for bands up
// to 1024 / 2048 / 4096); then the trailing window is updated diagonal by diagonal, the
// diagonals dealt out to the wavefronts: A(i, i-d) -= panel[i] . panel[i-d].
template <int NB>
__global__ __launch_bounds__(1024) void k_bj_factor_big(
    const int* __restrict__ list, const int* __restrict__ row0, const int* __restrict__ nrows,
    const int* __restrict__ bw, const long long* __restrict__ boff, double* __restrict__ band,
    int* __restrict__ fail) {
  extern __shared__ double sm[];
  const int p = list[blockIdx.x];
  const int b = nrows[p], w = bw[p];
  double* __restrict__ A = band + boff[p];
  double* panel = sm;                        // [row][NB]
  double* D = sm + (size_t)w * NB;           // [NB][NB], lower triangle of the diagonal block
  const int tid = threadIdx.x, nt = blockDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nw = nt >> 6;
  for (int J = 0; J < b; J += NB) {
    const int nbk = (b - J) < NB ? (b - J) : NB;
    for (int e = tid; e < NB * NB; e += nt) {
      const int r = e / NB, c = e % NB;
      D[e] = (r < nbk && c <= r) ? A[(size_t)(r - c) * b + J + r] : (r == c ? 1.0 : 0.0);
    }
    __syncthreads();
    if (tid == 0) {
      for (int c = 0; c < nbk; ++c) {
        double d = D[c * NB + c];
        for (int k = 0; k < c; ++k) d -= D[c * NB + k] * D[c * NB + k];
        if (!(d > 0.0)) { atomicCAS(fail, 0, row0[p] + J + c + 1); d = __longlong_as_double(0x7ff8000000000000LL); }
        const double piv = sqrt(d);
        D[c * NB + c] = piv;
        for (int r = c + 1; r < nbk; ++r) {
          double v = D[r * NB + c];
          for (int k = 0; k < c; ++k) v -= D[r * NB + k] * D[c * NB + k];
          D[r * NB + c] = v / piv;
        }
      }
    }
    __syncthreads();
    for (int e = tid; e < nbk * NB; e += nt) {
      const int r = e / NB, c = e % NB;
      if (c <= r) A[(size_t)(r - c) * b + J + r] = D[e];
    }
    // panel: rows below the diagonal block that reach into these columns
    const int i0 = J + nbk;
    const int last = (J + nbk + w) < b ? (J + nbk + w) : b;
    const int np_ = last - i0;
    for (int ip = tid; ip < np_; ip += nt) {
      const int i = i0 + ip;
      double x[NB];
#pragma unroll
      for (int c = 0; c < NB; ++c) {
        const int dd = i - (J + c);
        x[c] = (c < nbk && dd <= w) ? A[(size_t)dd * b + i] : 0.0;
      }
#pragma unroll
      for (int c = 0; c < NB; ++c) {
        if (c < nbk) {
          double v = x[c];
#pragma unroll
          for (int k = 0; k < c; ++k) v -= x[k] * D[c * NB + k];
          x[c] = v / D[c * NB + c];
        }
      }
#pragma unroll
      for (int c = 0; c < NB; ++c) {
        const int dd = i - (J + c);
        panel[(size_t)ip * NB + c] = x[c];
        if (c < nbk && dd <= w) A[(size_t)dd * b + i] = x[c];
      }
    }
    __syncthreads();
    // trailing update of the window, diagonal by diagonal
    for (int ipb = 0; ipb < np_; ipb += 64) {
      const int ip = ipb + lane;
      const bool valid = ip < np_;
      double pr[NB];
#pragma unroll
      for (int c = 0; c < NB; ++c) pr[c] = valid ? panel[(size_t)ip * NB + c] : 0.0;
      const int dtop = (ipb + 63) < w ? (ipb + 63) : w;
      for (int d = wave; d <= dtop; d += nw) {
        if (valid && d <= ip) {
          const double* q = panel + (size_t)(ip - d) * NB;
          double sum = 0.0;
#pragma unroll
          for (int c = 0; c < NB; ++c) sum = fma(pr[c], q[c], sum);
          A[(size_t)d * b + i0 + ip] -= sum;
        }
      }
    }
    __syncthreads();
  }
}

// band[off[e]] = val[e]: assembly of the wide bands from the block's lower-triangle entries
// (shipping the mostly empty band itself would cost gigabytes over PCIe)
__global__ __launch_bounds__(WG) void k_scatter(size_t n, const long long* __restrict__ off,
                                               const double* __restrict__ val, double* __restrict__ dst) {
  const size_t stride = (size_t)gridDim.x * WG;
  for (size_t e = (size_t)blockIdx.x * WG + threadIdx.x; e < n; e += stride) dst[off[e]] = val[e];
}

// Sweep records and 1/L(j,j) from a factored diagonal-major band: forward record j = column j
// of L over L(j,j), backward record j = row b-1-j over its diagonal; `wide` records are in
// window-slot order (k_bj_wide), the others [d = 1..w | 0] (k_bj_apply).
__global__ __launch_bounds__(WG) void k_bj_layout_big(
    const int* __restrict__ list, const int* __restrict__ row0, const int* __restrict__ nrows,
    const int* __restrict__ bw, const long long* __restrict__ off, const long long* __restrict__ boff,
    const double* __restrict__ band, int wide_from, double* __restrict__ Lf, double* __restrict__ Lb,
    double* __restrict__ invd_f, double* __restrict__ invd_b) {
  const int p = list[blockIdx.y];
  const int b = nrows[p], w = bw[p], r0 = row0[p];
  const double* __restrict__ A = band + boff[p];
  const bool wide = w > wide_from;
  const size_t reclen = wide ? (size_t)bjw_window(w) : (size_t)((w + 2) & ~1);
  double* __restrict__ f = Lf + off[p];
  double* __restrict__ g = Lb + off[p];
  for (int j = blockIdx.x; j < b; j += gridDim.x) {
    const int jr = b - 1 - j;
    const double idf = 1.0 / A[j], idb = 1.0 / A[jr];
    if (threadIdx.x == 0) { invd_f[r0 + j] = idf; invd_b[r0 + j] = idb; }
    for (int dd = 1 + threadIdx.x; dd <= w; dd += WG) {
      const size_t slot = wide ? (size_t)(j + dd) % reclen : (size_t)(dd - 1);
      if (j + dd < b) f[(size_t)j * reclen + slot] = A[(size_t)dd * b + j + dd] * idf;
      if (jr - dd >= 0) g[(size_t)j * reclen + slot] = A[(size_t)dd * b + jr] * idb;
    }
  }
}

// ------------------------------------------------ HBM calibration ----
// What this device sustains on the plainest streaming kernels, measured in the same process
// as the solver kernels (bench.py quotes the SpMM against the 8 TB/s spec and against this).
__global__ __launch_bounds__(WG) void k_probe_copy(size_t n2, const double2* __restrict__ src,
                                                   double2* __restrict__ dst) {
  const size_t stride = (size_t)gridDim.x * WG;
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n2; i += stride) dst[i] = src[i];
}
__global__ __launch_bounds__(WG) void k_probe_read(size_t n2, const double2* __restrict__ src,
                                                   double* __restrict__ out) {
  const size_t stride = (size_t)gridDim.x * WG;
  double s = 0.0;
#pragma unroll 4
  for (size_t i = (size_t)blockIdx.x * WG + threadIdx.x; i < n2; i += stride) { const double2 v = src[i]; s += v.x + v.y; }
  if (s == 12345.678) out[0] = s;   // keeps the loads alive, never true for the zero-filled buffer
}

// 16-column panels: the same Gram product on the f64 matrix cores.  One v_mfma_f64_16x16x4
// takes four panel rows: lane l supplies A[row l>>4][col l&15] and B[row l>>4][col l&15] --
// a wave-wide load of either operand is 512 contiguous bytes -- and accumulates
// C[i][j] = sum_rows A[row][i] B[row][j] (C/D map: col = lane&15, row = (lane>>4) + 4*reg).
// The register-tiled kernel above re-reads every panel segment from L1 four to eight times at
// this width (139 us for 395 MB); this one reads each byte once.  Same partial-block layout.
template <int NPAN>
__global__ __launch_bounds__(WG) void k_gram_mfma16(int m, const double* __restrict__ A0,
                                                    const double* __restrict__ A1,
                                                    const double* __restrict__ B,
                                                    double* __restrict__ partials) {
  constexpr int TS = 16, LDP = NPAN * TS;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int col = lane & 15, rsub = lane >> 4;
  mfma_d4 acc[NPAN];
#pragma unroll
  for (int p = 0; p < NPAN; ++p) acc[p] = mfma_d4{0.0, 0.0, 0.0, 0.0};
  const size_t nquad = ((size_t)m + 3) >> 2;
  const size_t qstride = (size_t)gridDim.x * (WG / 64);
  constexpr int U = 4;                                 // quads in flight per wavefront
  for (size_t q0 = (size_t)blockIdx.x * (WG / 64) + wave; q0 < nquad; q0 += U * qstride) {
    double a[U][NPAN], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t row = (q0 + u * qstride) * 4 + rsub;
      const bool ok = row < (size_t)m;
      b[u] = ok ? B[row * TS + col] : 0.0;
      a[u][0] = ok ? A0[row * TS + col] : 0.0;
      if (NPAN > 1) a[u][NPAN - 1] = ok ? A1[row * TS + col] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int p = 0; p < NPAN; ++p) acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][p], b[u], acc[p], 0, 0, 0);
  }
  __shared__ double red[WG / 64][LDP * TS];
#pragma unroll
  for (int p = 0; p < NPAN; ++p)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(p * TS + rsub + 4 * r) + LDP * col] = acc[p][r];
  __syncthreads();
  for (int e = tid; e < LDP * TS; e += WG) {
    double sum = red[0][e];
#pragma unroll
    for (int w2 = 1; w2 < WG / 64; ++w2) sum += red[w2][e];
    partials[(size_t)blockIdx.x * (LDP * TS) + e] = sum;
  }
}

// 8-column panels: [A0 | A1] fills the 16 rows of one tile, B its first 8 columns.
template <int NPAN>
__global__ __launch_bounds__(WG) void k_gram_mfma8(int m, const double* __restrict__ A0,
                                                   const double* __restrict__ A1,
                                                   const double* __restrict__ B,
                                                   double* __restrict__ partials) {
  constexpr int TS = 8, LDP = NPAN * TS;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int col = lane & 15, rsub = lane >> 4;
  const double* __restrict__ Ap = (col < TS) ? A0 : A1;      // panel this lane's A column lives in
  const bool a_on = col < LDP, b_on = col < TS;
  const int ac = col & (TS - 1);
  mfma_d4 acc = mfma_d4{0.0, 0.0, 0.0, 0.0};
  const size_t nquad = ((size_t)m + 3) >> 2;
  const size_t qstride = (size_t)gridDim.x * (WG / 64);
  constexpr int U = 4;
  for (size_t q0 = (size_t)blockIdx.x * (WG / 64) + wave; q0 < nquad; q0 += U * qstride) {
    double a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t row = (q0 + u * qstride) * 4 + rsub;
      const bool ok = row < (size_t)m;
      a[u] = (ok && a_on) ? Ap[row * TS + ac] : 0.0;
      b[u] = (ok && b_on) ? B[row * TS + col] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
  }
  __shared__ double red[WG / 64][LDP * TS];
  if (b_on)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (rsub + 4 * r < LDP) red[wave][(rsub + 4 * r) + LDP * col] = acc[r];
  __syncthreads();
  for (int e = tid; e < LDP * TS; e += WG) {
    double sum = red[0][e];
#pragma unroll
    for (int w2 = 1; w2 < WG / 64; ++w2) sum += red[w2][e];
    partials[(size_t)blockIdx.x * (LDP * TS) + e] = sum;
  }
}

// ------------------------------------------------ small finishing steps ----
// Measured and rejected: letting the last workgroup of the producing kernel (ticket counter)
// do these sums.  The device-scope release every workgroup needs before taking its ticket
// writes the whole dirty L2 back on gfx950 (one L2 per XCD): +130 us per iteration.
// Sum the per-workgroup partial blocks (fixed order) and scatter the active
// sub-block into the reference's t x t layout.  BS threads, red = BS doubles.
template <int BS>
__device__ __forceinline__ void finish_sum(const double* partials, int nblk, int npan, int ts,
                                           int a_lo, int a_hi, int nb, double* out, int ld_out,
                                           double* red, int ner_cap = BS, int g0 = 0, int gstride = 1) {
  const int ldp = npan * ts;
  const int na = a_lo + a_hi;
  const int ne = na * nb;
  int ner = 1;
  while (ner < ne && ner < ner_cap) ner <<= 1;        // elements summed side by side
  const int nsl = BS / ner;                           // slices of the partial blocks per element
  const int tid = threadIdx.x;
  const int e0 = tid % ner, s = tid / ner;
  // groups of `ner` elements, dealt out to the workgroups of the launch
  for (int base = g0 * ner; base < ne; base += gstride * ner) {
    const int e = base + e0;
    double sum = 0.0;
    int i = 0, j = 0;
    if (e < ne) {
      i = e % na; j = e / na;
      const int src = (i < a_lo ? i : ts + (i - a_lo)) + ldp * j;
      const double* q = partials + src;
      const size_t bstride = (size_t)ldp * ts;
      int b = s;
      for (; b + 7 * nsl < nblk; b += 8 * nsl) {       // eight loads in flight, added in order
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = q[(size_t)(b + u * nsl) * bstride];
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += v[u];
      }
      for (; b < nblk; b += nsl) sum += q[(size_t)b * bstride];
    }
    red[tid] = sum;
    __syncthreads();
    if (s == 0 && e < ne) {
      double tot = 0.0;
      for (int q2 = 0; q2 < nsl; ++q2) tot += red[q2 * ner + e0];
      out[i + ld_out * j] = tot;
    }
    __syncthreads();
  }
}

// The lanes of ONE wavefront see each other's LDS stores once both of these have been passed (DS operations of
// a wavefront execute in order; this keeps the compiler from moving them and drains the counter).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// In-place upper Cholesky of the t x t column-major W in LDS (LAPACK dpotf2 'U': on failure
// the failing pivot is stored and the rest of W is left untouched), called by a whole
// workgroup; t <= 16.  The first wavefront does the work, right looking: pivot j, row j of U in
// parallel over the lanes, then the rank-1 update of the trailing triangle, t^2 / 64 entries per lane --
// three LDS round trips per pivot instead of the t^3 / 3 dependent LDS reads of one thread walking
// dpotf2's loops (21 us per iteration at 8 columns, round 3).  Every entry sees the same operations in
// the same order as in dpotf2 (the terms u_kj u_ki are taken off one k after the other, then one
// division), so the factor is bitwise the same; a pivot that is not positive -- rare, and by then the
// trailing entries are no longer dpotf2's -- sends one thread through dpotf2's own loops on a copy.
__device__ __forceinline__ void potrf_upper_wg(double* W, int t, int* info) {
  __shared__ double keep[256];
  __shared__ int s_fail;
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    for (int e = lane; e < t * t; e += 64) keep[e] = W[e];
    int fail = 0;
    // the lane's (up to four) entries of the upper triangle: row / column worked out once (a division by the
    // run-time t costs more than a pivot step)
    int er[4], ec[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = lane + 64 * q;
      ec[q] = e / t; er[q] = e - ec[q] * t;
      if (e >= t * t || er[q] > ec[q]) er[q] = -1;      // (below the diagonal or beyond the block: never touched)
    }
    wave_lds_sync();
    for (int j = 0; j < t; ++j) {
      double d = W[j + t * j];
      if (!(d > 0.0)) { fail = j + 1; break; }       // (the same value in every lane)
      d = sqrt(d);
      double u = 0.0;
      if (lane > j && lane < t) u = W[j + t * lane] / d;
      wave_lds_sync();
      if (lane == j) W[j + t * j] = d;
      if (lane > j && lane < t) W[j + t * lane] = u;
      wave_lds_sync();
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (er[q] > j) W[lane + 64 * q] -= W[j + t * er[q]] * W[j + t * ec[q]];
      wave_lds_sync();
    }
    if (fail) {
      for (int e = lane; e < t * t; e += 64) W[e] = keep[e];
      wave_lds_sync();
      if (lane == 0) {
        fail = 0;
        for (int j = 0; j < t; ++j) {
          double d = W[j + t * j];
          for (int k = 0; k < j; ++k) d -= W[k + t * j] * W[k + t * j];
          if (!(d > 0.0)) { W[j + t * j] = d; fail = j + 1; break; }
          d = sqrt(d);
          W[j + t * j] = d;
          for (int i = j + 1; i < t; ++i) {
            double sv = W[j + t * i];
            for (int k = 0; k < j; ++k) sv -= W[k + t * j] * W[k + t * i];
            W[j + t * i] = sv / d;
          }
        }
        s_fail = fail;
      }
    } else if (lane == 0) s_fail = 0;
    if (lane == 0 && info) *info = s_fail;
  }
  __syncthreads();
}

// [W ; G^T] ((t+T) x t, ld t+T) -> mu = chol(W) (t x t, ld t), alpha = U^-T G (t x T, ld t);
// called by a whole workgroup, W / G = 256 doubles of LDS each.  (ecg.c:431 + :438 with the
// Gram of the un-normalised P: (P U^-1)^T R = U^-T (P^T R).)
__device__ __forceinline__ void potrf_alpha_wg(const double* buf, int t, int T, double* mu,
                                               double* alpha, int* info, double* W, double* G) {
  const int ld = t + T, nt = blockDim.x;
  for (int e = threadIdx.x; e < t * t; e += nt) W[e] = buf[(e % t) + ld * (e / t)];
  for (int e = threadIdx.x; e < t * T; e += nt) { const int i = e % t, c = e / t; G[e] = buf[(t + c) + ld * i]; }
  __syncthreads();
  potrf_upper_wg(W, t, info);
  // forward substitution with U^T on the T columns of G at once, by the first wavefront: row i is divided by
  // its pivot, then taken off the rows below (t T / 64 entries per lane) -- per entry the operations and the
  // order of one lane walking its column (g_i - u_0i a_0 - u_1i a_1 ... , then the division)
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    int gk[4], gc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = lane + 64 * q;
      gc[q] = e / t; gk[q] = e - gc[q] * t;
      if (e >= t * T) gk[q] = -1;
    }
    for (int i = 0; i < t; ++i) {
      const double d = W[i + t * i];
      if (lane < T) G[i + t * lane] = G[i + t * lane] / d;
      wave_lds_sync();
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (gk[q] > i) G[lane + 64 * q] -= W[i + t * gk[q]] * G[i + t * gc[q]];
      wave_lds_sync();
    }
  }
  __syncthreads();
  // (mu / alpha / info may be null: the factor and alpha stay in W and G for the caller)
  if (mu) for (int e = threadIdx.x; e < t * t; e += nt) mu[e] = W[e];
  if (alpha) for (int e = threadIdx.x; e < t * T; e += nt) alpha[e] = G[e];
}

// Residual norm from the per-workgroup column sums: fixed-order tree over WG threads.
// res2[1] carries the Cholesky status so the host fetches both with one copy; `host`
// (pinned, device-visible) receives the same two values when given.
__device__ __forceinline__ void trace_finish_wg(const double* rtr, int nblk, int ts, int nc,
                                                double* res2, const int* info, double* host,
                                                double* red, double seq = 0.0) {
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += WG)
    for (int c = 0; c < nc; ++c) s += rtr[(size_t)b * ts + c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = WG / 2; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double r2 = red[0], st = info ? (double)info[0] : 0.0;
    res2[0] = r2; res2[1] = st;
    // host[2] = seq (when given) tells a polling host that the two words are there (pa_k_note_seq)
    if (host) {
      __hip_atomic_store(host, r2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(host + 1, st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __threadfence_system();
      if (seq != 0.0) __hip_atomic_store(host + 2, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---------------------------------------------------------------- Gram ----
// C = [A0 | A1]^T B over the local rows.  Each lane owns a TI x TI tile of C
// for a strided set of rows; NPAN*(TS/TI)^2 lanes cover one row.  Lanes are
// then folded with wavefront shuffles (64 wide), waves through LDS, and each
// workgroup writes one partial block (summed by k_finish in a fixed order, so
// results are bitwise reproducible).
template <int TS, int NPAN>
__global__ __launch_bounds__(WG) void k_gram(int m, const double* __restrict__ A0,
                                             const double* __restrict__ A1,
                                             const double* __restrict__ B,
                                             double* __restrict__ partials) {
  constexpr int TI = TS < 4 ? TS : 4;
  constexpr int TD = TS / TI;
  constexpr int LPR = NPAN * TD * TD;
  constexpr int LDP = NPAN * TS;
  const int tid = threadIdx.x;
  const int li = tid % LPR;
  const int pan = li / (TD * TD), ti = (li / TD) % TD, tj = li % TD;
  const double* __restrict__ A = (pan == 0) ? A0 : A1;
  double acc[TI][TI];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TI; ++j) acc[i][j] = 0.0;
  const size_t rstride = (size_t)gridDim.x * WG / LPR;
  for (size_t row = ((size_t)blockIdx.x * WG + tid) / LPR; row < (size_t)m; row += rstride) {
    double a[TI], b[TI];
    const double2* ap = reinterpret_cast<const double2*>(A + row * TS + ti * TI);
    const double2* bp = reinterpret_cast<const double2*>(B + row * TS + tj * TI);
#pragma unroll
    for (int i = 0; i < TI / 2; ++i) {
      double2 va = ap[i], vb = bp[i];
      a[2 * i] = va.x; a[2 * i + 1] = va.y;
      b[2 * i] = vb.x; b[2 * i + 1] = vb.y;
    }
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TI; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
  }
  // fold the 64/LPR row groups of the wave
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TI; ++j) acc[i][j] += __shfl_xor(acc[i][j], off);
  __shared__ double red[WG / 64][LDP * TS];
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  if (lane < LPR) {
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TI; ++j)
        red[wave][(pan * TS + ti * TI + i) + LDP * (tj * TI + j)] = acc[i][j];
  }
  __syncthreads();
  for (int e = tid; e < LDP * TS; e += WG) {
    double s = red[0][e];
#pragma unroll
    for (int w = 1; w < WG / 64; ++w) s += red[w][e];
    partials[(size_t)blockIdx.x * (LDP * TS) + e] = s;
  }
}

__global__ __launch_bounds__(1024) void k_finish(const double* __restrict__ partials, int nblk,
                                               int npan, int ts, int a_lo, int a_hi, int nb,
                                               double* __restrict__ out, int ld_out) {
  __shared__ double red[1024];
  // several workgroups (large blocks, 16-column panels): 64 elements at a time each
  if (gridDim.x > 1) finish_sum<1024>(partials, nblk, npan, ts, a_lo, a_hi, nb, out, ld_out, red, 64, blockIdx.x, gridDim.x);
  else finish_sum<1024>(partials, nblk, npan, ts, a_lo, a_hi, nb, out, ld_out, red);
}

// k_finish and k_trace_finish in one launch: the Gram block that is about to be all-reduced and,
// right behind it (res2), the squared residual norm from the column sums the update kernel left
// (same order of additions as k_trace_finish), so that one collective carries both.
__global__ __launch_bounds__(1024) void k_finish_trace(const double* __restrict__ partials, int nblk,
                                                     int npan, int ts, int a_lo, int a_hi, int nb,
                                                     double* __restrict__ out, int ld_out,
                                                     const double* __restrict__ rtr, int rtr_nblk, int nc,
                                                     double* __restrict__ res2, const int* __restrict__ info) {
  __shared__ double red[1024];
  finish_sum<1024>(partials, nblk, npan, ts, a_lo, a_hi, nb, out, ld_out, red);
  __syncthreads();
  const int tid = threadIdx.x;
  if (tid < WG) {
    double s = 0.0;
    for (int b = tid; b < rtr_nblk; b += WG)
      for (int c = 0; c < nc; ++c) s += rtr[(size_t)b * ts + c];
    red[tid] = s;
  }
  __syncthreads();
  for (int off = WG / 2; off > 0; off >>= 1) {
    if (tid < off) red[tid] += red[tid + off];
    __syncthreads();
  }
  if (tid == 0) { res2[0] = red[0]; res2[1] = info ? (double)info[0] : 0.0; }
}

// k_finish followed by k_potrf_alpha on its output, one launch (single-process runs, where no
// all-reduce sits between the two).
__global__ __launch_bounds__(1024) void k_finish_potrf_alpha(const double* __restrict__ partials,
                                                           int nblk, int npan, int ts, int t, int T,
                                                           double* out, double* __restrict__ mu,
                                                           double* __restrict__ alpha,
                                                           int* __restrict__ info) {
  __shared__ double red[1024];
  finish_sum<1024>(partials, nblk, npan, ts, t, T, t, out, t + T, red);
  __threadfence_block();
  __syncthreads();
  potrf_alpha_wg(out, t, T, mu, alpha, info, red, red + 256);
}

// Wide Gram blocks (panels of 8 / 16 columns: 128 / 512 doubles per partial block, 512 of them = 0.5 / 2 MB,
// which ONE workgroup needs 9 / 36 us to read -- k_finish_potrf_alpha took 17-21 us per iteration at 8 columns,
// k_finish x 2 + k_potrf_alpha + k_trace_finish 43 us at 16): FINW_WG workgroups sum a contiguous share of the
// blocks each, element by element, coalesced; the share goes out with device-scope stores, a ticket elects the
// last workgroup (k_finish32's protocol: no fence), which adds the shares in their fixed order, scatters the
// active sub-block into `out` as finish_sum does and, as asked, factors it (t > 0: k_finish_potrf_alpha) and /
// or sums the residual norm next to it (rtr: k_finish_trace).  The ticket lives behind the shares in `scratch`
// (per buffer, zero when the buffer is made, set back to zero by the last workgroup).
constexpr int FINW_WG = 32;
constexpr int GRAM_SCRATCH_BLOCKS = FINW_WG + 1;
__global__ __launch_bounds__(WG) void k_finish_wide(const double* __restrict__ partials, int nblk, int npan, int ts,
                                                    int a_lo, int a_hi, int nb, double* out, int ld_out,
                                                    double* scratch, int t, int T, double* __restrict__ mu,
                                                    double* __restrict__ alpha, int* __restrict__ info,
                                                    const double* __restrict__ rtr, int rtr_nblk, int rtr_nc,
                                                    double* __restrict__ res2) {
  __shared__ double red[512];
  __shared__ int s_last;
  const int NB = npan * ts * ts;                 // doubles per partial block (128, 256 or 512)
  const int tid = threadIdx.x;
  const int lpb = NB < WG ? NB : WG;             // lanes that cover one block side by side
  const int nsl = WG / lpb, ept = NB / lpb;      // blocks side by side; elements per lane (1 or 2)
  const int l = tid % lpb, sl = tid / lpb;
  const int per = (nblk + FINW_WG - 1) / FINW_WG;
  const int b0 = blockIdx.x * per, b1 = min(nblk, b0 + per);
  double acc[2] = {0.0, 0.0};
  {
    // eight blocks' loads in flight at a time (one after the other, each paid a trip to memory: 16 of them were
    // most of this kernel's 21 us at 16 columns); the additions keep their order
    const int second = ept > 1 ? lpb : 0;
    int b = b0 + sl;
    for (; b + 7 * nsl < b1; b += 8 * nsl) {
      double v[8][2];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double* __restrict__ q = partials + (size_t)(b + u * nsl) * NB + l;
        v[u][0] = q[0];
        v[u][1] = q[second];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc[0] += v[u][0]; acc[1] += v[u][1]; }
    }
    for (; b < b1; b += nsl) {
      const double* __restrict__ q = partials + (size_t)b * NB + l;
      acc[0] += q[0];
      acc[1] += q[second];
    }
  }
  // (behind the LARGEST shares this buffer can see, 2 ts^2 doubles each: a call with one panel must not look for
  // its ticket where a call with two panels leaves share data)
  unsigned* ticket = reinterpret_cast<unsigned*>(scratch + (size_t)FINW_WG * 2 * ts * ts);
  // the side-by-side slices of this workgroup (up to four), then the share
  if (nsl > 1) {
    if (sl > 0) red[(sl - 1) * lpb + l] = acc[0];
    __syncthreads();
    if (sl == 0) for (int k = 1; k < nsl; ++k) acc[0] += red[(k - 1) * lpb + l];
  }
  if (sl == 0) {
    __hip_atomic_store(scratch + (size_t)blockIdx.x * NB + l, acc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ept > 1) __hip_atomic_store(scratch + (size_t)blockIdx.x * NB + lpb + l, acc[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) s_last = (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1);
  __syncthreads();
  if (!s_last) return;
  {
    // every share, in order: the two halves of the workgroup take 32 shares each when a block is 128 doubles
    double tot[2] = {0.0, 0.0};
    const int g0 = sl * (FINW_WG / nsl), g1 = g0 + FINW_WG / nsl;
    // (sixteen / eight shares at a time, all loads of a round in flight: a device-scope load is a trip to memory)
    int g = g0;
    for (; g + 16 <= g1; g += 16) {
      double v[16][2];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        v[u][0] = __hip_atomic_load(scratch + (size_t)(g + u) * NB + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v[u][1] = ept > 1 ? __hip_atomic_load(scratch + (size_t)(g + u) * NB + lpb + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) { tot[0] += v[u][0]; tot[1] += v[u][1]; }
    }
    for (; g < g1; g += 8) {
      double v[8][2];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v[u][0] = __hip_atomic_load(scratch + (size_t)(g + u) * NB + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v[u][1] = ept > 1 ? __hip_atomic_load(scratch + (size_t)(g + u) * NB + lpb + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { tot[0] += v[u][0]; tot[1] += v[u][1]; }
    }
    __syncthreads();
    if (nsl > 1 && sl > 0) red[(sl - 1) * lpb + l] = tot[0];
    __syncthreads();
    if (sl == 0) {
      for (int k = 1; k < nsl; ++k) tot[0] += red[(k - 1) * lpb + l];
      // element e of the partial block = (row, column) of [panel 0 | panel 1]^T B: scatter the active part
      const int ldp = npan * ts, na = a_lo + a_hi;
#pragma unroll
      for (int qx = 0; qx < 2; ++qx) {
        if (qx < ept) {
          const int e = l + qx * lpb, r = e % ldp, j = e / ldp;
          const int i = r < ts ? (r < a_lo ? r : -1) : (r - ts < a_hi ? a_lo + (r - ts) : -1);
          if (i >= 0 && j < nb && i < na) out[i + (size_t)ld_out * j] = tot[qx];
        }
      }
    }
  }
  if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (t > 0) {
    __threadfence_block();
    __syncthreads();
    potrf_alpha_wg(out, t, T, mu, alpha, info, red, red + 256);
  }
  if (rtr) {
    __syncthreads();
    trace_finish_wg(rtr, rtr_nblk, ts, rtr_nc, res2, info, nullptr, red);
  }
}

// The [W ; G^T] block of 4-column panels (8 x 4) from many partial blocks (one per workgroup of
// the SpMM that formed them; 1 MB on the headline problem, which one workgroup needs 24 us to
// read): FIN32_WG workgroups sum a contiguous share each (32-byte loads, 32 blocks side by
// side, fixed order) into scratch; the last one to finish (ticket counter) adds
// the shares in order and, t > 0, factors and forms alpha as k_finish_potrf_alpha does.
constexpr int FIN32_WG = 64;
__global__ __launch_bounds__(WG) void k_finish32(const double* __restrict__ partials, int nblk,
                                                 double* scratch, int t, int T, double* out,
                                                 double* __restrict__ mu, double* __restrict__ alpha,
                                                 int* __restrict__ info, const double* __restrict__ rtr,
                                                 int rtr_nblk, int rtr_ts, int rtr_nc, double* __restrict__ res2) {
  __shared__ double red[32 * 32];
  __shared__ int s_last;
  typedef double d4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, e4 = tid & 7, sl = tid >> 3;
  const int per = (nblk + FIN32_WG - 1) / FIN32_WG;
  const int b0 = blockIdx.x * per, b1 = min(nblk, b0 + per);
  const d4* __restrict__ q = reinterpret_cast<const d4*>(partials) + e4;
  d4 sum = {0.0, 0.0, 0.0, 0.0};
  // four blocks per thread in flight, the last round too (blocks beyond the share: the thread's first block
  // again, added as zero) -- with 4174 / 5670 partial blocks a share is 66 / 89 blocks, i.e. two or three per
  // thread: the remainder loop this replaces took them one memory round trip after the other
  for (int b = b0 + sl; b < b1; b += 4 * 32) {
    d4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = q[(size_t)(b + u * 32 < b1 ? b + u * 32 : b) * 8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (b + u * 32 < b1) sum += v[u];
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) red[sl * 32 + e4 * 4 + u] = sum[u];
  __syncthreads();
  for (int half = 16; half >= 1; half >>= 1) {       // tree over the 32 slices
    for (int e = tid; e < half * 32; e += WG) red[e] += red[half * 32 + e];
    __syncthreads();
  }
  // the share goes out with device-scope (write-through) stores, the ticket is taken once they have
  // completed, and the last workgroup reads the shares with device-scope loads: no fence, which
  // costs 10 us on gfx950 even when the L2 holds nothing dirty
  if (tid < 32) __hip_atomic_store(scratch + blockIdx.x * 32 + tid, red[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // (the ticket lives behind the shares of THIS call's scratch -- zero when the buffer is made, set back by the
  // last workgroup -- so that two solver objects, or two streams, never elect across each other's launches)
  unsigned* ticket = reinterpret_cast<unsigned*>(scratch + FIN32_WG * 32);
  if (tid == 0) s_last = (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1);
  __syncthreads();
  if (!s_last) return;
  {
    // eight shares per thread, all loads in flight at once (one after the other they cost 0.3 us each)
    const int e = tid & 31, g8 = tid >> 5;
    double v[FIN32_WG / 8];
#pragma unroll
    for (int u = 0; u < FIN32_WG / 8; ++u)
      v[u] = __hip_atomic_load(scratch + (g8 * (FIN32_WG / 8) + u) * 32 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double tot = 0.0;
#pragma unroll
    for (int u = 0; u < FIN32_WG / 8; ++u) tot += v[u];
    red[g8 * 32 + e] = tot;
  }
  __syncthreads();
  if (tid < 32) {
    double tot = red[tid];
#pragma unroll
    for (int g = 1; g < 8; ++g) tot += red[g * 32 + tid];
    out[tid] = tot;
  }
  if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (t > 0) {
    __threadfence_block();
    __syncthreads();
    potrf_alpha_wg(out, t, T, mu, alpha, info, red, red + 256);
  }
  if (rtr) {      // the residual norm next to the block, as k_finish_trace (same order of additions as k_trace_finish)
    __syncthreads();
    trace_finish_wg(rtr, rtr_nblk, rtr_ts, rtr_nc, res2, info, nullptr, red);
  }
}

// t x t upper Cholesky, one lane (t <= 16).  LAPACK dpotf2 'U': on failure
// the failing pivot is stored and the rest of W is left untouched.
__global__ void k_potrf(double* __restrict__ Wg, int t, int* __restrict__ info) {
  __shared__ double W[16 * 16];
  for (int e = threadIdx.x; e < t * t; e += 64) W[e] = Wg[e];
  __syncthreads();
  potrf_upper_wg(W, t, info);
  for (int e = threadIdx.x; e < t * t; e += 64) Wg[e] = W[e];
}

// Small t x t work of the fused Orthodir step (ecg.c:577-587), one lane:
// mu = U^T U ; beta <- beta U^-1 (bm x bn) ; alpha <- U^-T alpha (t x nrhs) ;
// beta(0:t, 0:t) <- U^-T beta(0:t, 0:t).
__global__ void k_fused_small(double* __restrict__ mu, int t, int nrhs, int bm, int bn, int ldb,
                              double* __restrict__ alpha, double* __restrict__ beta,
                              int* __restrict__ info) {
  if (threadIdx.x != 0) return;
  int fail = 0;
  for (int j = 0; j < t; ++j) {
    double d = mu[j + t * j];
    for (int k = 0; k < j; ++k) d -= mu[k + t * j] * mu[k + t * j];
    if (!(d > 0.0)) { mu[j + t * j] = d; fail = j + 1; break; }
    d = sqrt(d);
    mu[j + t * j] = d;
    for (int i = j + 1; i < t; ++i) {
      double s = mu[j + t * i];
      for (int k = 0; k < j; ++k) s -= mu[k + t * j] * mu[k + t * i];
      mu[j + t * i] = s / d;
    }
  }
  *info = fail;
  for (int j = 0; j < bn && j < t; ++j) {      // beta <- beta U^-1
    for (int k = 0; k < j; ++k) {
      const double u = mu[k + t * j];
      for (int i = 0; i < bm; ++i) beta[i + ldb * j] -= beta[i + ldb * k] * u;
    }
    const double d = 1.0 / mu[j + t * j];
    for (int i = 0; i < bm; ++i) beta[i + ldb * j] *= d;
  }
  for (int c = 0; c < nrhs; ++c)                 // alpha <- U^-T alpha
    for (int i = 0; i < t; ++i) {
      double s = alpha[i + t * c];
      for (int k = 0; k < i; ++k) s -= mu[k + t * i] * alpha[k + t * c];
      alpha[i + t * c] = s / mu[i + t * i];
    }
  for (int c = 0; c < t; ++c)                    // beta(0:t,0:t) <- U^-T beta(0:t,0:t)
    for (int i = 0; i < t; ++i) {
      double s = beta[i + ldb * c];
      for (int k = 0; k < i; ++k) s -= mu[k + t * i] * beta[k + ldb * c];
      beta[i + ldb * c] = s / mu[i + t * i];
    }
}

// -------------------------------------------------------------- updates ----
// P <- P U^-1 (and AP) by forward substitution along each row, one row/lane.
template <int TS>
__global__ __launch_bounds__(WG) void k_trsm(int m, int t, const double* __restrict__ U,
                                             double* __restrict__ P, double* __restrict__ AP) {
  __shared__ double su[TS * TS];
  __shared__ double sd[TS];
  for (int e = threadIdx.x; e < t * t; e += WG) su[e] = U[e];
  __syncthreads();
  if (threadIdx.x < t) sd[threadIdx.x] = 1.0 / su[threadIdx.x + t * threadIdx.x];
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * WG;
  for (size_t row = (size_t)blockIdx.x * WG + threadIdx.x; row < (size_t)m; row += stride) {
    double p[TS];
    load_row<TS>(P, row, p);
#pragma unroll
    for (int j = 0; j < TS; ++j) {
      if (j < t) {
        double s = p[j];
#pragma unroll
        for (int k = 0; k < j; ++k) s = fma(-p[k], su[k + t * j], s);
        p[j] = s * sd[j];
      }
    }
    store_row<TS>(P, row, p);
    if (AP) {
      load_row<TS>(AP, row, p);
#pragma unroll
      for (int j = 0; j < TS; ++j) {
        if (j < t) {
          double s = p[j];
#pragma unroll
          for (int k = 0; k < j; ++k) s = fma(-p[k], su[k + t * j], s);
          p[j] = s * sd[j];
        }
      }
      store_row<TS>(AP, row, p);
    }
  }
}

template <int TS>
__device__ __forceinline__ void block_sum_cols(double (&v)[TS], double* __restrict__ out) {
  __shared__ double red[WG / 64][TS];
#pragma unroll
  for (int off = 1; off < 64; off <<= 1)
#pragma unroll
    for (int c = 0; c < TS; ++c) v[c] += __shfl_xor(v[c], off);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (lane == 0)
#pragma unroll
    for (int c = 0; c < TS; ++c) red[wave][c] = v[c];
  __syncthreads();
  if (threadIdx.x < TS) {
    double s = red[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < WG / 64; ++w) s += red[w][threadIdx.x];
    out[threadIdx.x] = s;
  }
}

// X += P alpha, R -= AP alpha, plus per-workgroup sums of R(:,c)^2.
template <int TS>
__global__ __launch_bounds__(WG) void k_update_xr(int m, int t, int nc,
                                                  const double* __restrict__ alpha,
                                                  const double* __restrict__ P,
                                                  const double* __restrict__ AP,
                                                  double* __restrict__ X, double* __restrict__ R,
                                                  double* __restrict__ rtr) {
  __shared__ double sa[TS * TS];
  for (int e = threadIdx.x; e < t * nc; e += WG) sa[e] = alpha[e];
  __syncthreads();
  double rr[TS];
#pragma unroll
  for (int c = 0; c < TS; ++c) rr[c] = 0.0;
  const size_t stride = (size_t)gridDim.x * WG;
  for (size_t row = (size_t)blockIdx.x * WG + threadIdx.x; row < (size_t)m; row += stride) {
    double p[TS], ap[TS], x[TS], r[TS];
    load_row<TS>(P, row, p);
    load_row<TS>(AP, row, ap);
    load_row<TS>(X, row, x);
    load_row<TS>(R, row, r);
#pragma unroll
    for (int c = 0; c < TS; ++c) {
      if (c < nc) {
        double sx = 0.0, sr = 0.0;
#pragma unroll
        for (int k = 0; k < TS; ++k) {
          if (k < t) {
            const double a = sa[k + t * c];
            sx = fma(p[k], a, sx);
            sr = fma(ap[k], a, sr);
          }
        }
        x[c] += sx;
        r[c] -= sr;
        rr[c] = fma(r[c], r[c], rr[c]);
      }
    }
    store_row<TS>(X, row, x);
    store_row<TS>(R, row, r);
  }
  block_sum_cols<TS>(rr, rtr + (size_t)blockIdx.x * TS);
}

__global__ void k_potrf_alpha(const double* __restrict__ buf, int t, int T, double* __restrict__ mu,
                              double* __restrict__ alpha, int* __restrict__ info) {
  __shared__ double W[16 * 16];
  __shared__ double G[16 * 16];
  potrf_alpha_wg(buf, t, T, mu, alpha, info, W, G);
}

// P <- P U^-1, AP <- AP U^-1, X += P alpha, R -= AP alpha and the column sums of
// R^2 in one pass over the four panels (ecg.c:434-435 + :500-501 + :250).
// gram != null (runs of several processes, where an all-reduce of [W ; G^T] sits between the Gram
// kernel and this one): every workgroup factors W and forms alpha itself (k_potrf_alpha's
// arithmetic, a microsecond), workgroup 0 stores them and the status -- one launch less.
template <int TS>
__global__ __launch_bounds__(WG) void k_trsm_update(int m, int t, int nc, double* U, double* alpha,
                                                    double* __restrict__ P, double* __restrict__ AP,
                                                    double* __restrict__ X, double* __restrict__ R,
                                                    double* __restrict__ rtr, const double* gram, int* info,
                                                    double* __restrict__ ukeep, int xnt) {
  __shared__ double su[TS * TS];
  __shared__ double sd[TS];
  __shared__ double sa[TS * TS];
  if (gram) {
    const bool first = blockIdx.x == 0;
    potrf_alpha_wg(gram, t, nc, first ? U : nullptr, first ? alpha : nullptr, first ? info : nullptr, su, sa);
  } else {
    for (int e = threadIdx.x; e < t * t; e += WG) su[e] = U[e];
    for (int e = threadIdx.x; e < t * nc; e += WG) sa[e] = alpha[e];
  }
  __syncthreads();
  if (threadIdx.x < t) sd[threadIdx.x] = 1.0 / su[threadIdx.x + t * threadIdx.x];
  // ukeep (lazy normalisation, ecg.c): P and AP stay as they are -- the rows below are normalised in
  // registers for X and R only -- and the factor is kept for the kernels that meet the raw panels later
  if (ukeep && blockIdx.x == 0) for (int e = threadIdx.x; e < t * t; e += WG) ukeep[e] = su[e];
  __syncthreads();
  double rr[TS];
#pragma unroll
  for (int c = 0; c < TS; ++c) rr[c] = 0.0;
  const size_t stride = (size_t)gridDim.x * WG;
  for (size_t row = (size_t)blockIdx.x * WG + threadIdx.x; row < (size_t)m; row += stride) {
    double p[TS], ap[TS], x[TS], r[TS];
    load_row<TS>(P, row, p);
    load_row<TS>(AP, row, ap);
    // X is the one panel nobody reads again before the caches have turned over (its next reader is this kernel, an
    // iteration later): read and written with the nontemporal hint its 2 x 33 MB do not push R -- which the block
    // solve reads next -- out of the caches, nor wait there as dirty lines: the block solve behind this kernel
    // 125.5 -> 121.0 us, this kernel +0.5 us (six alternations in one box, profiles/r04_nontemporal_x_ab.txt).  The
    // hint on the store alone or on the load alone does nothing; on the loads of P / AP, or of P / P_prev in
    // k_update_z, it costs 2-5 us; on the X accesses of k_trsm_update_mfma<8 / 16> it changes nothing.
    // xnt = 0 (the launcher: a panel below 16 MiB, e.g. one GPU's share of a small problem): X stays in the
    // caches from one iteration to the next and the hint would send it to memory
    if (xnt) load_row_nt<TS>(X, row, x); else load_row<TS>(X, row, x);
    load_row<TS>(R, row, r);
#pragma unroll
    for (int j = 0; j < TS; ++j) {
      if (j < t) {
        double s1 = p[j], s2 = ap[j];
#pragma unroll
        for (int k = 0; k < j; ++k) { const double u = su[k + t * j]; s1 = fma(-p[k], u, s1); s2 = fma(-ap[k], u, s2); }
        p[j] = s1 * sd[j];
        ap[j] = s2 * sd[j];
      }
    }
#pragma unroll
    for (int c = 0; c < TS; ++c) {
      if (c < nc) {
        double sx = 0.0, sr = 0.0;
#pragma unroll
        for (int k = 0; k < TS; ++k) {
          if (k < t) {
            const double a = sa[k + t * c];
            sx = fma(p[k], a, sx);
            sr = fma(ap[k], a, sr);
          }
        }
        x[c] += sx;
        r[c] -= sr;
        rr[c] = fma(r[c], r[c], rr[c]);
      }
    }
    if (!ukeep) {
      store_row<TS>(P, row, p);
      store_row<TS>(AP, row, ap);
    }
    if (xnt) store_row_nt<TS>(X, row, x); else store_row<TS>(X, row, x);
    store_row<TS>(R, row, r);
  }
  block_sum_cols<TS>(rr, rtr + (size_t)blockIdx.x * TS);
}

// The same pass for panels of 8 and 16 columns on the f64 matrix cores.  With Ui = U^-1 (formed
// once per workgroup, t <= 16) and B = Ui alpha, all four results are products of the OLD tiles:
//   P <- P Ui,  AP <- AP Ui,  X += P B,  R -= AP B,
// so a tile of 16 rows of P / AP is loaded once in the A layout of v_mfma_f64_16x16x4 (lane l:
// row l&15, k = 4s + (l>>4)) and multiplied by per-lane constants (B layout: k = 4s + (l>>4),
// column l&15); X and R pass through the accumulator (C/D layout: four coalesced 512-byte rows).
// Columns beyond t ride along unchanged (Ui is padded with the identity, B with zeros).
// TS = 16: one tile per panel.  TS = 8: [P | AP] is one 16-wide A operand, blockdiag(Ui, Ui) and
// [B 0; 0 -B] the B operands, [X | R] the accumulator.
template <int TS>
__global__ __launch_bounds__(WG) void k_trsm_update_mfma(int m, int t, int nc, double* Ug, double* alphag,
                                                         double* __restrict__ P, double* __restrict__ AP,
                                                         double* __restrict__ X, double* __restrict__ R,
                                                         double* __restrict__ rtr, const double* gram, int* info,
                                                         double* __restrict__ ukeep) {
  static_assert(TS == 8 || TS == 16, "matrix-core variant: panels of 8 or 16 columns");
  __shared__ double su[16 * 16];    // U (column major, leading dimension 16, identity beyond t)
  __shared__ double si[16 * 16];    // Ui = U^-1
  __shared__ double sb[16 * 16];    // B = Ui alpha (16 x 16, zero beyond t x nc)
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lo = lane & 15, hi = lane >> 4;
  __shared__ double sw[16 * 16];    // gram != null: U and alpha as this workgroup computes them (k_trsm_update)
  __shared__ double sg[16 * 16];
  const double* U = Ug;
  const double* alpha = alphag;
  if (gram) {
    const bool first = blockIdx.x == 0;
    potrf_alpha_wg(gram, t, nc, first ? Ug : nullptr, first ? alphag : nullptr, first ? info : nullptr, sw, sg);
    __syncthreads();
    U = sw; alpha = sg;
  }
  for (int e = tid; e < 256; e += WG) {
    const int i = e & 15, j = e >> 4;
    su[e] = (i < t && j < t) ? U[i + t * j] : (i == j ? 1.0 : 0.0);
    sb[e] = 0.0;
  }
  // ukeep (lazy normalisation): P and AP are left as they are, the factor is kept for pa_k_update_z
  if (ukeep && blockIdx.x == 0) for (int e = tid; e < t * t; e += WG) ukeep[e] = U[e];
  __syncthreads();
  if (tid < 16) {
    // column tid of Ui by back substitution: U x = e_tid (upper triangular)
    double x[16];
#pragma unroll
    for (int i = 15; i >= 0; --i) {
      double sv = (i == tid) ? 1.0 : 0.0;
#pragma unroll
      for (int k = i + 1; k < 16; ++k) sv -= su[i + 16 * k] * x[k];
      x[i] = sv / su[i + 16 * i];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) si[i + 16 * tid] = x[i];
  }
  __syncthreads();
  for (int e = tid; e < 256; e += WG) {
    const int i = e & 15, c = e >> 4;
    double sv = 0.0;
    if (i < t && c < nc) for (int k = i; k < t; ++k) sv += si[i + 16 * k] * alpha[k + t * c];
    sb[e] = sv;
  }
  __syncthreads();
  // per-lane B operands: k = 4 s + hi
  double bu[4], bb[4];
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2) {
    const int k = 4 * s2 + hi;
    if (TS == 16) { bu[s2] = si[k + 16 * lo]; bb[s2] = sb[k + 16 * lo]; }
    else {
      // blockdiag(Ui, Ui): rows / columns 0..7 act on P, 8..15 on AP;  [B 0; 0 -B]
      const int kk = k & 7, cc = lo & 7;
      const bool same = (k < 8) == (lo < 8);
      bu[s2] = same ? si[kk + 16 * cc] : 0.0;
      bb[s2] = same ? (k < 8 ? sb[kk + 16 * cc] : -sb[kk + 16 * cc]) : 0.0;
    }
  }
  double rr = 0.0;   // sum of squares of the new R in column lo (TS = 8: lo - 8)
  const size_t ntile = ((size_t)m + 15) >> 4;
  const size_t tstride = (size_t)gridDim.x * (WG / 64);
  for (size_t tl = (size_t)blockIdx.x * (WG / 64) + wave; tl < ntile; tl += tstride) {
    const size_t r0 = tl << 4;
    const size_t arow = r0 + lo;
    const bool aok = arow < (size_t)m;
    if (TS == 16) {
      double ap_[4], aap[4];
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        ap_[s2] = aok ? P[arow * 16 + 4 * s2 + hi] : 0.0;
        aap[s2] = aok ? AP[arow * 16 + 4 * s2 + hi] : 0.0;
      }
      mfma_d4 x, r, pn = mfma_d4{0.0, 0.0, 0.0, 0.0}, apn = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t row = r0 + hi + 4 * q;
        const bool ok = row < (size_t)m;
        x[q] = ok ? X[row * 16 + lo] : 0.0;
        r[q] = ok ? R[row * 16 + lo] : 0.0;
      }
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        if (!ukeep) {
          pn = __builtin_amdgcn_mfma_f64_16x16x4f64(ap_[s2], bu[s2], pn, 0, 0, 0);
          apn = __builtin_amdgcn_mfma_f64_16x16x4f64(aap[s2], bu[s2], apn, 0, 0, 0);
        }
        x = __builtin_amdgcn_mfma_f64_16x16x4f64(ap_[s2], bb[s2], x, 0, 0, 0);
        r = __builtin_amdgcn_mfma_f64_16x16x4f64(aap[s2], -bb[s2], r, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t row = r0 + hi + 4 * q;
        if (row < (size_t)m) {
          if (!ukeep) { P[row * 16 + lo] = pn[q]; AP[row * 16 + lo] = apn[q]; }
          X[row * 16 + lo] = x[q]; R[row * 16 + lo] = r[q];
          if (lo < nc) rr = fma(r[q], r[q], rr);
        }
      }
    } else {
      double a[4];
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        const int k = 4 * s2 + hi;
        a[s2] = aok ? (k < 8 ? P[arow * 8 + k] : AP[arow * 8 + k - 8]) : 0.0;
      }
      mfma_d4 xr, pn = mfma_d4{0.0, 0.0, 0.0, 0.0};
      double* __restrict__ XR = lo < 8 ? X : R;
      double* __restrict__ PA = lo < 8 ? P : AP;
      const int cc = lo & 7;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t row = r0 + hi + 4 * q;
        xr[q] = row < (size_t)m ? XR[row * 8 + cc] : 0.0;
      }
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        if (!ukeep) pn = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s2], bu[s2], pn, 0, 0, 0);
        xr = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s2], bb[s2], xr, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t row = r0 + hi + 4 * q;
        if (row < (size_t)m) {
          if (!ukeep) PA[row * 8 + cc] = pn[q];
          XR[row * 8 + cc] = xr[q];
          if (lo >= 8 && cc < nc) rr = fma(xr[q], xr[q], rr);
        }
      }
    }
  }
  // column sums of R^2: lanes with the same column (4 per wave), then the waves
  __shared__ double red[WG / 64][TS];
  rr += __shfl_xor(rr, 16);
  rr += __shfl_xor(rr, 32);
  if (TS == 16) { if (hi == 0) red[wave][lo] = rr; }
  else if (hi == 0 && lo >= 8) red[wave][lo - 8] = rr;
  __syncthreads();
  if (tid < TS) {
    double sv = red[0][tid];
#pragma unroll
    for (int w2 = 1; w2 < WG / 64; ++w2) sv += red[w2][tid];
    rtr[(size_t)blockIdx.x * TS + tid] = sv;
  }
}

template <int TS>
__global__ __launch_bounds__(WG) void k_colnorm2(int m, const double* __restrict__ R,
                                                 double* __restrict__ rtr) {
  double rr[TS];
#pragma unroll
  for (int c = 0; c < TS; ++c) rr[c] = 0.0;
  const size_t stride = (size_t)gridDim.x * WG;
  for (size_t row = (size_t)blockIdx.x * WG + threadIdx.x; row < (size_t)m; row += stride) {
    double r[TS];
    load_row<TS>(R, row, r);
#pragma unroll
    for (int c = 0; c < TS; ++c) rr[c] = fma(r[c], r[c], rr[c]);
  }
  block_sum_cols<TS>(rr, rtr + (size_t)blockIdx.x * TS);
}

__global__ __launch_bounds__(WG) void k_trace_finish(const double* __restrict__ rtr, int nblk,
                                                     int ts, int nc, double* __restrict__ res2,
                                                     const int* __restrict__ info, double* host, double seq) {
  __shared__ double red[WG];
  trace_finish_wg(rtr, nblk, ts, nc, res2, info, host, red, seq);
}

// Z(:, :nc) -= [V0(:, :a_lo) | V1(:, :a_hi)] beta
template <int TS>
__global__ __launch_bounds__(WG) void k_update_z(int m, int a_lo, int a_hi, int nc,
                                                 const double* __restrict__ beta, int ldb,
                                                 const double* __restrict__ V0,
                                                 const double* __restrict__ V1,
                                                 double* __restrict__ Z,
    const double* note_src, double* note_host, double note_seq,
    const double* __restrict__ ucur, const double* __restrict__ uprev,
    const int* __restrict__ pk_off, const int* __restrict__ pk_slot, double* __restrict__ sendbuf) {
  // pk_off != null (several processes, pa_k_update_z_pack): the new rows of Z are the next product's X -- the rows
  // the neighbours need go into the send buffer from here (row r into the slots pk_slot[pk_off[r] .. pk_off[r + 1])),
  // k_pack_rows is not launched
  __shared__ double sb[2 * TS * TS];
  __shared__ double sc[7 * TS * TS];
  // (note_host: two words the host is waiting for -- the all-reduced residual norm and the
  // factorisation status next to beta -- go out to pinned memory from here: no copy, no extra launch)
  if (note_host && blockIdx.x == 0 && threadIdx.x == 0) {
    __hip_atomic_store(note_host, note_src[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(note_host + 1, note_src[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    if (note_seq != 0.0) __hip_atomic_store(note_host + 2, note_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const int na = a_lo + a_hi;
  for (int e = threadIdx.x; e < na * nc; e += WG) sb[e] = beta[(e % na) + ldb * (e / na)];
  // (lazy normalisation: the two factors come in with the same round trip as beta -- read from memory inside
  // the back substitution below, behind run-time conditions, they cost several trips in a row at the head of
  // every workgroup)
  __shared__ double sfac[2 * TS * TS];
  if (ucur) {
    const int tt0 = a_lo * a_lo;
    for (int e = threadIdx.x; e < 2 * tt0; e += WG) sfac[e] = e < tt0 ? ucur[e] : (uprev ? uprev[e - tt0] : 0.0);
  }
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * WG;
  if (ucur) {
    // Lazy normalisation (ecg.c: Orthodir without block-size reduction, panels of up to 4 columns): the panels
    // were never multiplied by U^-1 -- V0 = P_raw and Z = M^-1 AP_raw belong to the factor U = ucur of this
    // iteration, V1 = P_prev_raw to uprev -- and `beta` holds the RAW Gram blocks G1 = AP_raw^T Z_raw, G2 =
    // AP_prev_raw^T Z_raw.  With Ui = U^-1, Up = uprev^-1 the reference's update Z_n - P_n beta1 - P_prev_n beta2
    // (ecg.c:510-517 on the normalised panels) is  Z_raw C0 - P_raw C1 - P_prev_raw C2,  C0 = Ui,
    // C1 = Ui (Ui^T G1 Ui), C2 = Up (Up^T G2 Ui): sixteen threads form the three t x t blocks in LDS.
    const int t = a_lo, tt = t * t, tid = threadIdx.x;
    double* Ui = sc; double* Up = sc + TS * TS; double* T1 = sc + 2 * TS * TS; double* T2 = sc + 3 * TS * TS;
    double* C0 = sc + 4 * TS * TS; double* C1 = sc + 5 * TS * TS; double* C2 = sc + 6 * TS * TS;
    if (tid < 64) {
      if (tid < 2 * t) {          // column c of the inverse of an upper-triangular factor: back substitution on e_c
        const double* Uf = tid < t ? sfac : sfac + tt;
        double* inv = tid < t ? Ui : Up;
        const int c = tid < t ? tid : tid - t;
        double x[TS];
#pragma unroll
        for (int i = TS - 1; i >= 0; --i) {
          x[i] = 0.0;
          if (i < t && i <= c) {
            double sv = i == c ? 1.0 : 0.0;
#pragma unroll
            for (int k = i + 1; k < TS; ++k) if (k < t && k <= c) sv = fma(-Uf[i + t * k], x[k], sv);
            x[i] = sv / Uf[i + t * i];
          }
        }
#pragma unroll
        for (int i = 0; i < TS; ++i) if (i < t) inv[i + t * c] = x[i];
      }
      wave_lds_sync();
      const int r = tid % (t > 0 ? t : 1), c = tid / (t > 0 ? t : 1);
      const bool on = tid < tt;
      if (on) {                     // T1 = G1 Ui, T2 = G2 Ui
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < t; ++k) { s1 = fma(sb[r + na * k], Ui[k + t * c], s1); if (a_hi > 0) s2 = fma(sb[a_lo + r + na * k], Ui[k + t * c], s2); }
        T1[tid] = s1; T2[tid] = s2;
      }
      wave_lds_sync();
      double b1 = 0.0, b2 = 0.0;
      if (on) {                     // beta1 = Ui^T T1, beta2 = Up^T T2
        for (int k = 0; k < t; ++k) { b1 = fma(Ui[k + t * r], T1[k + t * c], b1); b2 = fma(Up[k + t * r], T2[k + t * c], b2); }
      }
      wave_lds_sync();
      if (on) { T1[tid] = b1; T2[tid] = b2; }
      wave_lds_sync();
      if (on) {                     // C1 = Ui beta1, C2 = Up beta2, C0 = Ui
        double c1 = 0.0, c2 = 0.0;
        for (int k = 0; k < t; ++k) { c1 = fma(Ui[r + t * k], T1[k + t * c], c1); c2 = fma(Up[r + t * k], T2[k + t * c], c2); }
        C0[tid] = Ui[tid]; C1[tid] = c1; C2[tid] = c2;
      }
    }
    __syncthreads();
    for (size_t row = (size_t)blockIdx.x * WG + threadIdx.x; row < (size_t)m; row += stride) {
      double z[TS], v0[TS], v1[TS], o[TS];
      load_row<TS>(Z, row, z);
      load_row<TS>(V0, row, v0);
      if (a_hi > 0) load_row<TS>(V1, row, v1);
#pragma unroll
      for (int c = 0; c < TS; ++c) {
        o[c] = z[c];
        if (c < nc) {
          double s = 0.0;
#pragma unroll
          for (int k = 0; k < TS; ++k)
            if (k < t) { s = fma(z[k], C0[k + t * c], s); s = fma(-v0[k], C1[k + t * c], s); if (a_hi > 0) s = fma(-v1[k], C2[k + t * c], s); }
          o[c] = s;
        }
      }
      store_row<TS>(Z, row, o);
      if (pk_off) for (int k = pk_off[row], k1 = pk_off[row + 1]; k < k1; ++k) store_row<TS>(sendbuf, (size_t)pk_slot[k], o);
    }
    return;
  }
  for (size_t row = (size_t)blockIdx.x * WG + threadIdx.x; row < (size_t)m; row += stride) {
    double z[TS], v0[TS], v1[TS];
    load_row<TS>(Z, row, z);
    load_row<TS>(V0, row, v0);
    if (a_hi > 0) load_row<TS>(V1, row, v1);
#pragma unroll
    for (int c = 0; c < TS; ++c) {
      if (c < nc) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < TS; ++k)
          if (k < a_lo) s = fma(v0[k], sb[k + na * c], s);
        if (a_hi > 0) {
#pragma unroll
          for (int k = 0; k < TS; ++k)
            if (k < a_hi) s = fma(v1[k], sb[a_lo + k + na * c], s);
        }
        z[c] -= s;
      }
    }
    store_row<TS>(Z, row, z);
    if (pk_off) for (int k = pk_off[row], k1 = pk_off[row + 1]; k < k1; ++k) store_row<TS>(sendbuf, (size_t)pk_slot[k], z);
  }
}

// 16-column panels on the f64 matrix cores: a tile of 16 rows of Z is the C/D operand (lane l
// holds Z[row (l>>4) + 4r][col l&15]: four fully coalesced 512-byte accesses), the rows of
// [V0 | V1] are the A operand (A[row l&15][k = 4s + (l>>4)], 32-byte pieces of 16 rows per
// load, every cache line used up over four loads) and -beta the B operand, a per-lane constant.
// Lazy normalisation, panels of 8 / 16 columns (see k_update_z<TS>): C0 = Ui, C1 = Ui (Ui^T G1 Ui), C2 = Up (Up^T G2 Ui)
// from the raw Gram blocks in `beta` and the two kept factors, as 16 x 16 blocks (leading dimension 16, zero
// beyond t) in LDS; called by a whole workgroup of WG threads.  sc = 7 * 256 doubles.
__device__ __forceinline__ void lazy_coeffs16(const double* __restrict__ beta, int ldb, int t, int a_hi,
                                              const double* __restrict__ ucur, const double* __restrict__ uprev,
                                              double* sc) {
  double* Ui = sc; double* Up = sc + 256; double* T1 = sc + 512; double* T2 = sc + 768;
  double* C0 = sc + 1024; double* C1 = sc + 1280; double* C2 = sc + 1536;
  const int tid = threadIdx.x;
  for (int e = tid; e < 256; e += WG) {       // the factors, identity beyond t (T1 / T2 serve as staging)
    const int i = e & 15, j = e >> 4;
    T1[e] = (i < t && j < t) ? ucur[i + t * j] : (i == j ? 1.0 : 0.0);
    T2[e] = (i < t && j < t) ? uprev[i + t * j] : (i == j ? 1.0 : 0.0);
    // the raw Gram blocks with the same round trip (C1 / C2 serve as staging until they are written at the end):
    // read from memory inside the loop over k below they cost t trips in a row at the head of every workgroup
    C1[e] = (i < t && j < t) ? beta[i + ldb * j] : 0.0;
    C2[e] = (a_hi > 0 && i < t && j < t) ? beta[t + i + ldb * j] : 0.0;
  }
  __syncthreads();
  if (tid < 32) {                             // column c of an inverse by back substitution on e_c
    const double* Uf = tid < 16 ? T1 : T2;
    double* inv = tid < 16 ? Ui : Up;
    const int c = tid & 15;
    double x[16];
#pragma unroll
    for (int i = 15; i >= 0; --i) {
      double sv = (i == c) ? 1.0 : 0.0;
#pragma unroll
      for (int k = i + 1; k < 16; ++k) sv -= Uf[i + 16 * k] * x[k];
      x[i] = sv / Uf[i + 16 * i];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) inv[i + 16 * c] = x[i];
  }
  __syncthreads();
  const int r = tid & 15, c = tid >> 4;       // WG = 256: one entry each
  const bool on = r < t && c < t;
  {
    double s1 = 0.0, s2 = 0.0;                // T = G Ui
    if (on) for (int k = 0; k < t; ++k) {
      s1 = fma(C1[r + 16 * k], Ui[k + 16 * c], s1);
      if (a_hi > 0) s2 = fma(C2[r + 16 * k], Ui[k + 16 * c], s2);
    }
    __syncthreads();
    T1[tid] = s1; T2[tid] = s2;
  }
  __syncthreads();
  double b1 = 0.0, b2 = 0.0;                  // beta1 = Ui^T T1, beta2 = Up^T T2
  if (on) for (int k = 0; k < t; ++k) { b1 = fma(Ui[k + 16 * r], T1[k + 16 * c], b1); b2 = fma(Up[k + 16 * r], T2[k + 16 * c], b2); }
  __syncthreads();
  T1[tid] = b1; T2[tid] = b2;
  __syncthreads();
  double c1 = 0.0, c2 = 0.0;                  // C1 = Ui beta1, C2 = Up beta2
  if (on) for (int k = 0; k < t; ++k) { c1 = fma(Ui[r + 16 * k], T1[k + 16 * c], c1); c2 = fma(Up[r + 16 * k], T2[k + 16 * c], c2); }
  C0[tid] = on ? Ui[tid] : 0.0; C1[tid] = c1; C2[tid] = c2;
  __syncthreads();
}

// Z^T Z next to the update (BF-Omin forms it right behind, ecg.c:361 of the reference: G = P^T P with P = Z): a
// tile of the new Z in the accumulator layout -- lane (lo, hi): row hi + 4 r, column lo -- is the A operand AND the
// B operand of step r as it stands (A[i = lo][k = hi], B[k = hi][j = lo], k = the row), so the product costs four
// matrix instructions per tile and no data movement.  zzc = columns that count (the rest of a 16-wide tile is 0).
__device__ __forceinline__ void update_z_gram_step(const mfma_d4& z, mfma_d4& zz, size_t r0, int hi, int lo, int m, int zzc) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double zm = (r0 + hi + 4 * r < (size_t)m && lo < zzc) ? z[r] : 0.0;
    zz = __builtin_amdgcn_mfma_f64_16x16x4f64(zm, zm, zz, 0, 0, 0);
  }
}
// the four wavefronts' sums -> one TS x TS block per workgroup (column major, the layout of k_gram's partial blocks)
template <int TS>
__device__ __forceinline__ void update_z_gram_out(const mfma_d4& zz, double* sc, double* __restrict__ zzp, int wave, int lane) {
  __syncthreads();                       // (sc may still hold the coefficient blocks of the prologue)
#pragma unroll
  for (int r = 0; r < 4; ++r) sc[wave * 256 + r * 64 + lane] = zz[r];
  __syncthreads();
  const int tid = threadIdx.x;           // WG = 256: entry (r, lane) of the tile each
  const int r = tid >> 6, l = tid & 63, i = (l >> 4) + 4 * r, j = l & 15;
  double sm = sc[tid];
#pragma unroll
  for (int w = 1; w < WG / 64; ++w) sm += sc[w * 256 + tid];
  if (i < TS && j < TS) zzp[(size_t)blockIdx.x * TS * TS + i + TS * j] = sm;
}

__global__ __launch_bounds__(WG, 4) void k_update_z_mfma16(int m, int a_lo, int a_hi, int nc,
                                                        const double* __restrict__ beta, int ldb,
                                                        const double* __restrict__ V0,
                                                        const double* __restrict__ V1,
                                                        double* __restrict__ Z,
    const double* note_src, double* note_host, double note_seq,
    const double* __restrict__ ucur, const double* __restrict__ uprev, double* __restrict__ zzp, int zzc) {
  constexpr int TS = 16;
  __shared__ double sc[7 * 256];
  // (note_host: two words the host is waiting for -- the all-reduced residual norm and the
  // factorisation status next to beta -- go out to pinned memory from here: no copy, no extra launch)
  if (note_host && blockIdx.x == 0 && threadIdx.x == 0) {
    __hip_atomic_store(note_host, note_src[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(note_host + 1, note_src[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    if (note_seq != 0.0) __hip_atomic_store(note_host + 2, note_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lo = lane & 15, hi = lane >> 4;
  // B[k][j] = -beta(k, j), k = 4s + hi (s < 4: rows of beta that meet V0, s >= 4: V1), j = lo
  double bneg[8];
#pragma unroll
  for (int s2 = 0; s2 < 8; ++s2) {
    const int k = 4 * (s2 & 3) + hi;
    const bool first = s2 < 4;
    const bool ok = lo < nc && (first ? k < a_lo : k < a_hi);
    bneg[s2] = ok ? -beta[(first ? k : a_lo + k) + ldb * lo] : 0.0;
  }
  const size_t ntile = ((size_t)m + 15) >> 4;
  const size_t tstride = (size_t)gridDim.x * (WG / 64);
  if (ucur) {
    // lazy normalisation: Z <- Z C0 - V0 C1 - V1 C2 (k_update_z<TS>); Z is an A operand like the other two.
    // The k index of a matrix-core step is free as long as both operands agree on it: step s2 of lane (lo, hi)
    // takes k = 8 (s2 >> 1) + 2 hi + (s2 & 1), so that the lane's four entries of a row are two 16-byte loads and
    // the four lanes of a row read 64 contiguous bytes per load (k = 4 s2 + hi, one double per load and 32 B
    // between the lanes of a row, touched 16 lines a quarter each per load: 129 us against 94 us for the in-place
    // form at 16 columns; whole rows turned through a padded LDS tile per wavefront: 193 us).  The first tile is
    // requested in front of the coefficient prologue (a 16 x 16 inverse and three products per workgroup), every
    // further one in front of the stores of the tile before it.
    const double* __restrict__ V1p = a_hi > 0 ? V1 : V0;       // (no second panel: its coefficients are zero)
    lazy_coeffs16(beta, ldb, a_lo, a_hi, ucur, uprev, sc);
    double b0[4], b1[4], b2[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      const int k = 8 * (s2 >> 1) + 2 * hi + (s2 & 1);
      b0[s2] = sc[1024 + k + 16 * lo]; b1[s2] = -sc[1280 + k + 16 * lo]; b2[s2] = a_hi > 0 ? -sc[1536 + k + 16 * lo] : 0.0;
    }
    double2 rg[6];
    size_t t = (size_t)blockIdx.x * (WG / 64) + wave;
    {
      const size_t tl = t < ntile ? t : ntile - 1, arow = (tl << 4) + lo;
      const size_t base = (arow < (size_t)m ? arow : 0) * TS + 2 * hi;
      rg[0] = *reinterpret_cast<const double2*>(Z + base); rg[1] = *reinterpret_cast<const double2*>(Z + base + 8);
      rg[2] = *reinterpret_cast<const double2*>(V0 + base); rg[3] = *reinterpret_cast<const double2*>(V0 + base + 8);
      rg[4] = *reinterpret_cast<const double2*>(V1p + base); rg[5] = *reinterpret_cast<const double2*>(V1p + base + 8);
    }
    while (t < ntile) {
      const size_t r0 = t << 4;
      const bool aok = r0 + lo < (size_t)m;
      const double az[4] = {aok ? rg[0].x : 0.0, aok ? rg[0].y : 0.0, aok ? rg[1].x : 0.0, aok ? rg[1].y : 0.0};
      const double a0[4] = {aok ? rg[2].x : 0.0, aok ? rg[2].y : 0.0, aok ? rg[3].x : 0.0, aok ? rg[3].y : 0.0};
      const double a1[4] = {aok ? rg[4].x : 0.0, aok ? rg[4].y : 0.0, aok ? rg[5].x : 0.0, aok ? rg[5].y : 0.0};
      mfma_d4 z = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        z = __builtin_amdgcn_mfma_f64_16x16x4f64(az[s2], b0[s2], z, 0, 0, 0);
        z = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s2], b1[s2], z, 0, 0, 0);
        z = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s2], b2[s2], z, 0, 0, 0);
      }
      const size_t tn = t + tstride;
      {
        // the next tile's rows (the last round asks for its own tile once more: no branch around a load)
        const size_t tl = tn < ntile ? tn : t, arow = (tl << 4) + lo;
        const size_t base = (arow < (size_t)m ? arow : 0) * TS + 2 * hi;
        rg[0] = *reinterpret_cast<const double2*>(Z + base); rg[1] = *reinterpret_cast<const double2*>(Z + base + 8);
        rg[2] = *reinterpret_cast<const double2*>(V0 + base); rg[3] = *reinterpret_cast<const double2*>(V0 + base + 8);
        rg[4] = *reinterpret_cast<const double2*>(V1p + base); rg[5] = *reinterpret_cast<const double2*>(V1p + base + 8);
      }
      asm volatile("" ::: "memory");     // (every lane of the tile has read its rows of Z before any is overwritten)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const size_t row = r0 + hi + 4 * r;
        if (row < (size_t)m && lo < nc) Z[row * TS + lo] = z[r];
      }
      t = tn;
    }
    return;
  }
  mfma_d4 zz = mfma_d4{0.0, 0.0, 0.0, 0.0};
  for (size_t t = (size_t)blockIdx.x * (WG / 64) + wave; t < ntile; t += tstride) {
    const size_t r0 = t << 4;
    mfma_d4 z;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t row = r0 + hi + 4 * r;
      z[r] = row < (size_t)m ? Z[row * TS + lo] : 0.0;
    }
    const size_t arow = r0 + lo;
    const bool aok = arow < (size_t)m;
    double a[8];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) a[s2] = aok ? V0[arow * TS + 4 * s2 + hi] : 0.0;
    if (a_hi > 0) {
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) a[4 + s2] = aok ? V1[arow * TS + 4 * s2 + hi] : 0.0;
    }
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) z = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s2], bneg[s2], z, 0, 0, 0);
    if (a_hi > 0) {
#pragma unroll
      for (int s2 = 4; s2 < 8; ++s2) z = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s2], bneg[s2], z, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t row = r0 + hi + 4 * r;
      if (row < (size_t)m && lo < nc) Z[row * TS + lo] = z[r];
    }
    if (zzp) update_z_gram_step(z, zz, r0, hi, lo, m, zzc);
  }
  if (zzp) update_z_gram_out<TS>(zz, sc, zzp, wave, lane);
}

// 8-column panels: [V0 | V1] is one 16-wide A operand (four k-steps), Z uses half the tile.
__global__ __launch_bounds__(WG) void k_update_z_mfma8(int m, int a_lo, int a_hi, int nc,
                                                       const double* __restrict__ beta, int ldb,
                                                       const double* __restrict__ V0,
                                                       const double* __restrict__ V1,
                                                       double* __restrict__ Z,
    const double* note_src, double* note_host, double note_seq,
    const double* __restrict__ ucur, const double* __restrict__ uprev, double* __restrict__ zzp, int zzc) {
  constexpr int TS = 8;
  __shared__ double sc[7 * 256];
  // (note_host: two words the host is waiting for -- the all-reduced residual norm and the
  // factorisation status next to beta -- go out to pinned memory from here: no copy, no extra launch)
  if (note_host && blockIdx.x == 0 && threadIdx.x == 0) {
    __hip_atomic_store(note_host, note_src[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(note_host + 1, note_src[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    if (note_seq != 0.0) __hip_atomic_store(note_host + 2, note_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lo = lane & 15, hi = lane >> 4;
  double bneg[4];
  bool von[4];
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2) {
    const int k = 4 * s2 + hi;                 // column of [V0 | V1]
    const bool first = k < TS;
    const int kk = first ? k : k - TS;
    von[s2] = first ? kk < a_lo : kk < a_hi;
    bneg[s2] = (von[s2] && lo < nc) ? -beta[(first ? kk : a_lo + kk) + ldb * lo] : 0.0;
  }
  const size_t ntile = ((size_t)m + 15) >> 4;
  const size_t tstride = (size_t)gridDim.x * (WG / 64);
  if (ucur) {
    // lazy normalisation: Z <- [V0 | V1 | Z] [-C1 ; -C2 ; C0], 24 columns in six steps of four.  As in
    // k_update_z_mfma16: k = 2 hi + (s2 & 1) within the panel s2 >> 1, so a lane's two entries of a row are one
    // 16-byte load and the four lanes of a row read its 64 bytes; every tile but the first is requested in front
    // of the stores of the tile before it.
    const double* __restrict__ V1p = a_hi > 0 ? V1 : V0;       // (no second panel: its coefficients are zero)
    lazy_coeffs16(beta, ldb, a_lo, a_hi, ucur, uprev, sc);
    double bb[6];
#pragma unroll
    for (int s2 = 0; s2 < 6; ++s2) {
      const int kk = 2 * hi + (s2 & 1);
      const double v = s2 < 2 ? -sc[1280 + kk + 16 * (lo & 7)] : s2 < 4 ? (a_hi > 0 ? -sc[1536 + kk + 16 * (lo & 7)] : 0.0) : sc[1024 + kk + 16 * (lo & 7)];
      bb[s2] = lo < TS ? v : 0.0;
    }
    double2 rg[3];
    size_t t = (size_t)blockIdx.x * (WG / 64) + wave;
    {
      const size_t tl = t < ntile ? t : ntile - 1, arow = (tl << 4) + lo;
      const size_t base = (arow < (size_t)m ? arow : 0) * TS + 2 * hi;
      rg[0] = *reinterpret_cast<const double2*>(V0 + base);
      rg[1] = *reinterpret_cast<const double2*>(V1p + base);
      rg[2] = *reinterpret_cast<const double2*>(Z + base);
    }
    while (t < ntile) {
      const size_t r0 = t << 4;
      const bool aok = r0 + lo < (size_t)m;
      const double a[6] = {aok ? rg[0].x : 0.0, aok ? rg[0].y : 0.0, aok ? rg[1].x : 0.0, aok ? rg[1].y : 0.0,
                           aok ? rg[2].x : 0.0, aok ? rg[2].y : 0.0};
      mfma_d4 z = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s2 = 0; s2 < 6; ++s2) z = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s2], bb[s2], z, 0, 0, 0);
      const size_t tn = t + tstride;
      {
        const size_t tl = tn < ntile ? tn : t, arow = (tl << 4) + lo;
        const size_t base = (arow < (size_t)m ? arow : 0) * TS + 2 * hi;
        rg[0] = *reinterpret_cast<const double2*>(V0 + base);
        rg[1] = *reinterpret_cast<const double2*>(V1p + base);
        rg[2] = *reinterpret_cast<const double2*>(Z + base);
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const size_t row = r0 + hi + 4 * r;
        if (row < (size_t)m && lo < nc) Z[row * TS + lo] = z[r];
      }
      t = tn;
    }
    return;
  }
  mfma_d4 zz = mfma_d4{0.0, 0.0, 0.0, 0.0};
  for (size_t t = (size_t)blockIdx.x * (WG / 64) + wave; t < ntile; t += tstride) {
    const size_t r0 = t << 4;
    mfma_d4 z;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t row = r0 + hi + 4 * r;
      z[r] = (row < (size_t)m && lo < TS) ? Z[row * TS + lo] : 0.0;
    }
    const size_t arow = r0 + lo;
    const bool aok = arow < (size_t)m;
    double a[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      const int k = 4 * s2 + hi;
      a[s2] = (aok && von[s2]) ? (k < TS ? V0[arow * TS + k] : V1[arow * TS + k - TS]) : 0.0;
    }
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) z = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s2], bneg[s2], z, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t row = r0 + hi + 4 * r;
      if (row < (size_t)m && lo < nc) Z[row * TS + lo] = z[r];
    }
    if (zzp) update_z_gram_step(z, zz, r0, hi, lo, m, zzc);
  }
  if (zzp) update_z_gram_out<TS>(zz, sc, zzp, wave, lane);
}

template <int TS>
__global__ __launch_bounds__(WG) void k_copy_cols(int m, int nc, const double* __restrict__ src,
                                                  double* __restrict__ dst) {
  const size_t stride = (size_t)gridDim.x * WG;
  for (size_t row = (size_t)blockIdx.x * WG + threadIdx.x; row < (size_t)m; row += stride) {
    double s[TS], d[TS];
    load_row<TS>(src, row, s);
    load_row<TS>(dst, row, d);
#pragma unroll
    for (int c = 0; c < TS; ++c) if (c < nc) d[c] = s[c];
    store_row<TS>(dst, row, d);
  }
}

template <int TS>
__global__ __launch_bounds__(WG) void k_right_mult(int m, int t, const double* __restrict__ Q,
                                                   double* __restrict__ A) {
  __shared__ double sq[TS * TS];
  for (int e = threadIdx.x; e < t * t; e += WG) sq[e] = Q[e];
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * WG;
  for (size_t row = (size_t)blockIdx.x * WG + threadIdx.x; row < (size_t)m; row += stride) {
    double a[TS], o[TS];
    load_row<TS>(A, row, a);
#pragma unroll
    for (int c = 0; c < TS; ++c) {
      o[c] = a[c];
      if (c < t) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < TS; ++k) if (k < t) s = fma(a[k], sq[k + t * c], s);
        o[c] = s;
      }
    }
    store_row<TS>(A, row, o);
  }
}

template <int TS>
__global__ __launch_bounds__(WG) void k_permute_cols(int m, int n, const int* __restrict__ piv,
                                                     double* __restrict__ A) {
  __shared__ int sp[TS];
  if (threadIdx.x < n) sp[threadIdx.x] = piv[threadIdx.x];
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * WG;
  for (size_t row = (size_t)blockIdx.x * WG + threadIdx.x; row < (size_t)m; row += stride) {
    double a[TS], o[TS];
    load_row<TS>(A, row, a);
#pragma unroll
    for (int c = 0; c < TS; ++c) {
      o[c] = a[c];
      if (c < n) {
        const int s = sp[c];
        double v = a[0];
#pragma unroll
        for (int k = 1; k < TS; ++k) v = (s == k) ? a[k] : v;
        o[c] = v;
      }
    }
    store_row<TS>(A, row, o);
  }
}

// BF-Omin (ecg.c:358-393 of the reference: copy Z -> P, dlapmt, dtrsm on the leading `t` columns) in one pass:
// dst(:, c) = src(:, piv[c]) for c < n, then the first t columns times U^-1 -- the substitution of k_trsm, same
// order of operations, so the result equals the three kernels' bit for bit.
template <int TS>
__global__ __launch_bounds__(WG) void k_permute_trsm(int m, int n, const int* __restrict__ piv, int t,
                                                     const double* __restrict__ U, const double* __restrict__ src,
                                                     double* __restrict__ dst) {
  __shared__ double su[TS * TS];
  __shared__ double sd[TS];
  __shared__ int sp[TS];
  for (int e = threadIdx.x; e < t * t; e += WG) su[e] = U[e];
  if (threadIdx.x < n) sp[threadIdx.x] = piv[threadIdx.x];
  __syncthreads();
  if (threadIdx.x < t) sd[threadIdx.x] = 1.0 / su[threadIdx.x + t * threadIdx.x];
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * WG;
  for (size_t row = (size_t)blockIdx.x * WG + threadIdx.x; row < (size_t)m; row += stride) {
    double a[TS], p[TS];
    load_row<TS>(src, row, a);
    if (n < TS) load_row<TS>(dst, row, p);      // (columns beyond the panel's width stay what they are)
#pragma unroll
    for (int c = 0; c < TS; ++c) {
      if (c < n) {
        const int s_ = sp[c];
        double v = a[0];
#pragma unroll
        for (int k = 1; k < TS; ++k) v = (s_ == k) ? a[k] : v;
        p[c] = v;
      }
    }
#pragma unroll
    for (int j = 0; j < TS; ++j) {
      if (j < t) {
        double s_ = p[j];
#pragma unroll
        for (int k = 0; k < j; ++k) s_ = fma(-p[k], su[k + t * j], s_);
        p[j] = s_ * sd[j];
      }
    }
    store_row<TS>(dst, row, p);
  }
}

template <int TS>
__global__ __launch_bounds__(WG) void k_rowsum(int m, int nc, const double* __restrict__ X,
                                               double* __restrict__ sol) {
  const size_t stride = (size_t)gridDim.x * WG;
  for (size_t row = (size_t)blockIdx.x * WG + threadIdx.x; row < (size_t)m; row += stride) {
    double x[TS];
    load_row<TS>(X, row, x);
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < TS; ++c) if (c < nc) s += x[c];
    sol[row] = s;
  }
}

// -------------------------------------------------------- block-Jacobi ----
// Exact solve with one SPD diagonal block per wavefront: banded Cholesky
// factor (RCM order, factored at setup) applied as two systolic sweeps.  The
// W = 64*R rows in flight live in registers, row (j mod W) in lane (j mod 64)
// of register set (j/64 mod R); at step j the pivot y_j is broadcast with
// v_readlane and every lane updates the rows j+1..j+w it holds.
//
// The band is the only large operand and it is read exactly once per sweep:
// step j needs the record [L(j+1..j+w, j) / L(j,j) | 0] (wr doubles).
// Records are streamed HBM -> LDS in chunks of CH steps with LDS-DMA
// (global_load_lds_dwordx4: 1 KiB per wave instruction, no VGPRs), double
// buffered per wave so chunk c+1 is in flight while chunk c is consumed.
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

template <int CH>
__device__ __forceinline__ void bj_issue_chunk(const double* __restrict__ rec, int wr, int chunk,
                                               double* lbuf, int lane) {
  const int nbytes = CH * wr * 8;
  const char* g = reinterpret_cast<const char*>(rec) + (size_t)chunk * nbytes + lane * 16;
  char* l = reinterpret_cast<char*>(lbuf);
  for (int o = 0; o < nbytes; o += 1024)
    __builtin_amdgcn_global_load_lds((glb_void_ptr)(g + o), (lds_void_ptr)(l + o), 16, 0, 0);
}

// Record of step j (wr = w + 1 rounded up to even, doubles):
//   r[d-1] = L(j+d, j) / L(j, j), d = 1..w      r[w] = 0
// The band is pre-divided by its pivot, so a step is a_i -= (L_ij / L_jj) a_j
// with a_j read straight from its lane (no multiply on the critical path);
// y_j = a_j / L_jj is formed once per row when its block of 64 is stored.
// A lane that holds row j+d reads r[min(d-1, w)] (d-1 taken as unsigned): rows
// outside the band, the pivot row itself (d = 0) and rows already solved
// (d < 0) all land on the zero, so the update needs no branch.  A solved row is
// never touched again, so the block's 64 results stay in their lanes and are
// scaled and stored together when the block is done.
template <int R>
struct bj_vals {
  double lv[R];
};

template <int R, int K>
__device__ __forceinline__ void bj_load_step(const double* __restrict__ r, int l, int w, int lane,
                                             bj_vals<R>& v) {
#pragma unroll
  for (int k2 = 0; k2 < R; ++k2) {
    const int rel = (k2 - K + R) % R;
    v.lv[k2] = 0.0;
    if (rel == 0 || l >= rel * 64 - w) {       // wave-uniform: does set k2 touch the band at all?
      const unsigned d1 = (unsigned)(rel * 64 + lane - l - 1);
      const unsigned idx = min(d1, (unsigned)w);
      v.lv[k2] = r[idx];
    }
  }
}

// A full chunk of CH steps with the first NA register sets (counted from the pivots' own set)
// inside the band: straight-line code, the band values of four steps at a time read from LDS
// up front, no branch and no scalar bookkeeping per step.  (Sets beyond NA would only meet the zero slot.)
template <int TS, int R, int CH, int K, int NA>
__device__ __forceinline__ void bj_chunk_fast(double (&acc)[R][TS], const double* cur, int lc, int w,
                                              int wr, int lane) {
  constexpr int G = 4;                              // steps whose band values are in registers at once
#pragma unroll
  for (int s0 = 0; s0 < CH; s0 += G) {
    double cf[G][NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const int a0 = a * 64 + lane - lc - s0 - 1;   // d - 1 of step s0 for this set
#pragma unroll
      for (int s = 0; s < G; ++s) {
        const unsigned idx = min((unsigned)(a0 - s), (unsigned)w);
        cf[s][a] = cur[(s0 + s) * wr + idx];
      }
    }
#pragma unroll
    for (int s = 0; s < G; ++s) {
      double y[TS];
#pragma unroll
      for (int c = 0; c < TS; ++c) y[c] = readlane_f64(acc[K][c], lc + s0 + s);
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        const int k2 = (K + a) % R;
#pragma unroll
        for (int c = 0; c < TS; ++c) acc[k2][c] = fma(-cf[s][a], y[c], acc[k2][c]);
      }
    }
  }
}

// The same chunk from PAIRED records (k_bj_pairs): a chunk is two sub-blocks of four steps, a
// sub-block holds, for each of its two pivot pairs, one double2 per target row rho = row - first
// pivot of the sub-block (0 .. w + 3; row 0 is all zero and doubles as the slot of every row outside
// the band): the band values a lane needs for four steps are two ds_read_b128 at one index instead
// of four ds_read_b64 at four clamped indices.
template <int TS, int R, int CH, int K, int NA>
__device__ __forceinline__ void bj_chunk_pairs(double (&acc)[R][TS], const double* cur, int lc, int w, int lane) {
  const int nr = w + 4;
#pragma unroll
  for (int sb = 0; sb < CH / 4; ++sb) {
    const double2* blk = reinterpret_cast<const double2*>(cur + (size_t)sb * 4 * nr);
    double2 cf[NA][2];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const unsigned rho = (unsigned)(a * 64 + lane - lc - 4 * sb);
      const unsigned idx = rho < (unsigned)nr ? rho : 0u;
      cf[a][0] = blk[idx];
      cf[a][1] = blk[nr + idx];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      double y[TS];
#pragma unroll
      for (int c = 0; c < TS; ++c) y[c] = readlane_f64(acc[K][c], lc + 4 * sb + s);
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        const int k2 = (K + a) % R;
        const double v = (s & 1) ? cf[a][s >> 1].y : cf[a][s >> 1].x;
#pragma unroll
        for (int c = 0; c < TS; ++c) acc[k2][c] = fma(-v, y[c], acc[k2][c]);
      }
    }
  }
}

// The chunks of a sweep go through a ring of `nbuf` LDS buffers (`lstride` doubles apart).  Two
// buffers: chunk c+1 is in flight while chunk c is consumed, and c has landed once every
// outstanding VMEM operation of the wave is done.  Three buffers (narrow bands): c+1 AND c+2 are in
// flight -- at 5 TB/s the loaded memory latency is longer than the 8 steps a chunk lasts, so one
// chunk of look-ahead left the wave waiting -- and c has landed once at most the `nld` load
// instructions of chunk c+1 are outstanding (loads return in order; younger ones only make the
// wait stricter).
__device__ __forceinline__ void bj_wait_chunk(int nld_allowed) {
  switch (nld_allowed) {
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

template <int TS, int R, int CH, int K, int LAY = 0>
__device__ __forceinline__ void bj_block(double (&acc)[R][TS], int lim, int& chunk, int b, int w, int wr,
                                         const double* __restrict__ rec, double* lds0, int lstride, int nbuf,
                                         int nld, int lane) {
  for (int lc = 0; lc < lim; lc += CH, ++chunk) {
    bj_wait_chunk((nbuf == 3 && (chunk + 1) * CH < b) ? nld : 0);
    const double* cur = lds0 + (size_t)(chunk % nbuf) * lstride;
    {
      const int ahead = chunk + nbuf - 1;
      if (ahead * CH < b) bj_issue_chunk<CH>(rec, wr, ahead, lds0 + (size_t)(ahead % nbuf) * lstride, lane);
    }
    if constexpr (LAY == 1) {            // paired records: every chunk is whole (zero padded)
      const int lmax = lc + CH - 1;
      if (R >= 3 && lmax >= 128 - w) bj_chunk_pairs<TS, R, CH, K, (R >= 3 ? 3 : 1)>(acc, cur, lc, w, lane);
      else if (R >= 2 && lmax >= 64 - w) bj_chunk_pairs<TS, R, CH, K, (R >= 2 ? 2 : 1)>(acc, cur, lc, w, lane);
      else bj_chunk_pairs<TS, R, CH, K, 1>(acc, cur, lc, w, lane);
      continue;
    }
    const int send = (lim - lc) < CH ? (lim - lc) : CH;
    if constexpr (R <= 3 && TS <= 4) {   // (no gain measured at 8 columns; 16 would spill)
      if (send == CH) {
        // sets the last step of the chunk reaches (wave-uniform): rel is inside the band from
        // step l >= rel*64 - w on
        const int lmax = lc + CH - 1;
        if (R >= 3 && lmax >= 128 - w) bj_chunk_fast<TS, R, CH, K, (R >= 3 ? 3 : 1)>(acc, cur, lc, w, wr, lane);
        else if (R >= 2 && lmax >= 64 - w) bj_chunk_fast<TS, R, CH, K, (R >= 2 ? 2 : 1)>(acc, cur, lc, w, wr, lane);
        else bj_chunk_fast<TS, R, CH, K, 1>(acc, cur, lc, w, wr, lane);
        continue;
      }
    }
    bj_vals<R> nv;
    bj_load_step<R, K>(cur, lc, w, lane, nv);
    for (int s = 0; s < send; ++s) {
      const int l = lc + s;
      const bj_vals<R> cv = nv;
      if (s + 1 < send) bj_load_step<R, K>(cur + (s + 1) * wr, l + 1, w, lane, nv);
      double y[TS];
#pragma unroll
      for (int c = 0; c < TS; ++c) y[c] = readlane_f64(acc[K][c], l);
#pragma unroll
      for (int k2 = 0; k2 < R; ++k2) {
        const int rel = (k2 - K + R) % R;
        if (rel == 0 || l >= rel * 64 - w) {
#pragma unroll
          for (int c = 0; c < TS; ++c) acc[k2][c] = fma(-cv.lv[k2], y[c], acc[k2][c]);
        }
      }
    }
  }
}

template <int TS, int R, int CH, int K, int XS, int LAY = 0>
__device__ __forceinline__ void bj_blocks(double (&acc)[R][TS], int (&rowid)[R], int jb, int& chunk, int b,
                                          int w, int wr, const double* __restrict__ rec,
                                          const double* __restrict__ invd,
                                          const int* __restrict__ iomap, size_t rowbase,
                                          const double* __restrict__ src, double* __restrict__ dst,
                                          double* lds0, int lstride, int nbuf, int nld, int lane) {
  if constexpr (K < R) {
    constexpr int W = 64 * R;
    const int j0 = jb + K * 64;
    if (j0 < b) {
      double nxt[TS];
      double idl = 0.0;
      int nrow = 0;
      const int jn = j0 + W + lane;
      if (j0 + lane < b) idl = invd[j0 + lane];
      if (jn < b) { nrow = iomap[jn]; load_row_s<TS, XS>(src, rowbase + nrow, nxt); }
      else
#pragma unroll
        for (int c = 0; c < TS; ++c) nxt[c] = 0.0;
      const int lim = (b - j0) < 64 ? (b - j0) : 64;
      bj_block<TS, R, CH, K, LAY>(acc, lim, chunk, b, w, wr, rec, lds0, lstride, nbuf, nld, lane);
      if (lane < lim) {
        double y[TS];
#pragma unroll
        for (int c = 0; c < TS; ++c) y[c] = acc[K][c] * idl;
        store_row_s<TS, XS>(dst, rowbase + rowid[K], y);
      }
#pragma unroll
      for (int c = 0; c < TS; ++c) acc[K][c] = nxt[c];
      rowid[K] = nrow;
    }
    bj_blocks<TS, R, CH, K + 1, XS, LAY>(acc, rowid, jb, chunk, b, w, wr, rec, invd, iomap, rowbase, src, dst, lds0,
                                lstride, nbuf, nld, lane);
  }
}

template <int TS, int R, int CH, int XS, int LAY = 0>
__device__ __forceinline__ void bj_sweep(int b, int w, int wr, const double* __restrict__ rec,
                                         const double* __restrict__ invd,
                                         const int* __restrict__ iomap, size_t rowbase,
                                         const double* __restrict__ src, double* __restrict__ dst,
                                         double* lds0, int lstride, int nbuf, int nld, int lane) {
  constexpr int W = 64 * R;
  double acc[R][TS];
  int rowid[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int j = k * 64 + lane;
    rowid[k] = 0;
    if (j < b) { rowid[k] = iomap[j]; load_row_s<TS, XS>(src, rowbase + rowid[k], acc[k]); }
    else
#pragma unroll
      for (int c = 0; c < TS; ++c) acc[k][c] = 0.0;
  }
  bj_issue_chunk<CH>(rec, wr, 0, lds0, lane);
  if (nbuf == 3 && CH < b) bj_issue_chunk<CH>(rec, wr, 1, lds0 + lstride, lane);
  int chunk = 0;
  for (int jb = 0; jb < b; jb += W)
    bj_blocks<TS, R, CH, 0, XS, LAY>(acc, rowid, jb, chunk, b, w, wr, rec, invd, iomap, rowbase, src, dst, lds0, lstride,
                            nbuf, nld, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// XS = panel stride: XS = TS, or a multiple of it when the panel is split by columns among XS/TS
// wavefronts (opt-in, see bj_launch).
// OCC = wavefronts per SIMD the register allocation must leave room for (1: no constraint).  All
// blocks of a class take equally long, so what counts is whether they fit the chip in ONE round:
// 5670 blocks on 1024 SIMDs need 6 resident wavefronts per SIMD (4 meant a second, nearly empty
// round behind the first).
template <int TS, int R, int CH, int XS, int OCC>
__global__ __launch_bounds__(256, OCC) void k_bj_apply(
    const int* __restrict__ list, int count, const int* __restrict__ row0,
    const int* __restrict__ nrows, const int* __restrict__ bw, const long long* __restrict__ off,
    const int* __restrict__ map_f, const int* __restrict__ map_b, const double* __restrict__ Lf,
    const double* __restrict__ Lb, const double* __restrict__ invd_f,
    const double* __restrict__ invd_b, int lds_per_wave, int nbuf, const double* __restrict__ in,
    double* __restrict__ out) {
  extern __shared__ double smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  constexpr int NS = XS / TS;
  const int unit = blockIdx.x * (blockDim.x >> 6) + wave;
  const int pi = unit / NS;
  if (pi >= count) return;
  const int coff = (unit % NS) * TS;
  // everything that steers the sweep is wave-uniform: keep it in SGPRs
  const int p = __builtin_amdgcn_readfirstlane(list[pi]);
  const int r0 = __builtin_amdgcn_readfirstlane(row0[p]);
  const int b = __builtin_amdgcn_readfirstlane(nrows[p]);
  const int w = __builtin_amdgcn_readfirstlane(bw[p]);
  const int wr = (w + 2) & ~1;
  const long long o64 = off[p];
  const size_t o = ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(o64 >> 32)) << 32) |
                   (unsigned)__builtin_amdgcn_readfirstlane((int)o64);
  double* lds0 = smem + (size_t)wave * lds_per_wave;
  const int lstride = lds_per_wave / nbuf;
  const int nld = (CH * wr * 8 + 1023) >> 10;      // load instructions per chunk of THIS block
  const int nb = nld <= 4 ? nbuf : 2;               // (the counted wait knows 1..4)
  // forward: L y = x (y goes to `out`), backward: L^T z = y in place
  bj_sweep<TS, R, CH, XS>(b, w, wr, Lf + o, invd_f + r0, map_f + r0, (size_t)r0, in + coff, out + coff, lds0, lstride, nb, nld, lane);
  __threadfence_block();
  bj_sweep<TS, R, CH, XS>(b, w, wr, Lb + o, invd_b + r0, map_b + r0, (size_t)r0, out + coff, out + coff, lds0, lstride, nb, nld, lane);
}

// k_bj_apply on paired records (bj_chunk_pairs): Lf2 / Lb2 at off2[p], 8 (w + 4) doubles per chunk.
template <int TS, int R, int CH, int XS>
__global__ __launch_bounds__(256) void k_bj_apply_pairs(
    const int* __restrict__ list, int count, const int* __restrict__ row0,
    const int* __restrict__ nrows, const int* __restrict__ bw, const long long* __restrict__ off2,
    const int* __restrict__ map_f, const int* __restrict__ map_b, const double* __restrict__ Lf2,
    const double* __restrict__ Lb2, const double* __restrict__ invd_f,
    const double* __restrict__ invd_b, int lds_per_wave, int nbuf, const double* __restrict__ in,
    double* __restrict__ out) {
  extern __shared__ double smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  constexpr int NS = XS / TS;
  const int unit = blockIdx.x * (blockDim.x >> 6) + wave;
  const int pi = unit / NS;
  if (pi >= count) return;
  const int coff = (unit % NS) * TS;
  const int p = __builtin_amdgcn_readfirstlane(list[pi]);
  const int r0 = __builtin_amdgcn_readfirstlane(row0[p]);
  const int b = __builtin_amdgcn_readfirstlane(nrows[p]);
  const int w = __builtin_amdgcn_readfirstlane(bw[p]);
  const int wr2 = w + 4;
  const int nld = (CH * wr2 * 8 + 1023) >> 10;
  const int nb = nld <= 4 ? nbuf : 2;
  const long long o64 = off2[p];
  const size_t o = ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(o64 >> 32)) << 32) |
                   (unsigned)__builtin_amdgcn_readfirstlane((int)o64);
  double* lds0 = smem + (size_t)wave * lds_per_wave;
  const int lstride = lds_per_wave / nbuf;
  bj_sweep<TS, R, CH, XS, 1>(b, w, wr2, Lf2 + o, invd_f + r0, map_f + r0, (size_t)r0, in + coff, out + coff, lds0, lstride, nb, nld, lane);
  __threadfence_block();
  bj_sweep<TS, R, CH, XS, 1>(b, w, wr2, Lb2 + o, invd_b + r0, map_b + r0, (size_t)r0, out + coff, out + coff, lds0, lstride, nb, nld, lane);
}

// Paired records from the plain ones: sub-block q = steps 4q .. 4q + 3; entry (pair, rho) =
// the two pivots' coefficients for row 4q + rho (zero outside the band and past the last step).
__global__ __launch_bounds__(WG) void k_bj_pairs(const int* __restrict__ list, const int* __restrict__ nrows,
                                                 const int* __restrict__ bw, const long long* __restrict__ off,
                                                 const long long* __restrict__ off2, const double* __restrict__ L,
                                                 double* __restrict__ L2) {
  const int p = list[blockIdx.x];
  const int b = nrows[p], w = bw[p], wr = (w + 2) & ~1, nr = w + 4;
  const double* __restrict__ rec = L + off[p];
  double2* __restrict__ dst = reinterpret_cast<double2*>(L2 + off2[p]);
  const int nsub = 2 * ((b + 7) / 8);
  const int total = nsub * 2 * nr;
  for (int e = threadIdx.x; e < total; e += WG) {
    const int q = e / (2 * nr), r = e - q * 2 * nr, pr = r / nr, rho = r - pr * nr;
    double v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int l = 4 * q + 2 * pr + h, d = rho - (2 * pr + h);      // pivot, distance row - pivot
      v[h] = (l < b && d >= 1 && d <= w) ? rec[(size_t)l * wr + d - 1] : 0.0;
    }
    dst[e] = make_double2(v[0], v[1]);
  }
}

// Wide bands (RCM bandwidth > 448: few, large subdomains).  One workgroup of up to 16
// wavefronts per subdomain; the W = 64*R*NW rows in flight are spread over the waves'
// registers exactly as in the one-wave kernel.  Per step the owning wave broadcasts the
// pivot through LDS (double buffered, one raw s_barrier per step), then every wave updates
// its rows.  Records are stored in window-slot order -- the value for target row i sits at
// column i mod W -- so a lane reads the same column of every record: no index arithmetic,
// coalesced, and prefetched D steps ahead in registers.
// ---- the same sweeps on the f64 matrix cores (panels of 8 / 16 columns) ----
// The window is kept as NT tiles of 16 rows in the C/D layout of v_mfma_f64_16x16x4 (lane l,
// register r <-> tile row (l>>4) + 4r, panel column l&15).  Four pivots at a time: they are
// the four rows of one register of the pivot tile, so once they are settled among themselves
// (three ds_bpermute steps) that register *is* the B operand
// Y[k = l>>4][j = l&15] of the rank-4 update  tile -= L[rows of tile][4 pivots] * Y  -- no
// data movement.  The A operand is the band value of (row l&15 of the tile, pivot l>>4), one
// LDS read per lane and tile.  Per step this costs a quarter of the v_readlane / v_fma_f64
// recurrence above at 16 columns.  Same records, same LDS-DMA chunks of 8 steps.
template <int TS, int NT, int TP>
__device__ __forceinline__ void bjm_tile(mfma_d4 (&acc)[NT], int (&rid)[NT][4], int g, int& chunk, int b,
                                         int w, int wr, const double* __restrict__ rec,
                                         const double* __restrict__ invd, const int* __restrict__ iomap,
                                         size_t rowbase, const double* __restrict__ src,
                                         double* __restrict__ dst, double* lds0, double* lds1, int lane) {
  const int lo = lane & 15, hi = lane >> 4;
  const double* cur = lds0;
  // what the hand-over at the end needs -- 1/L(j,j) of the tile's rows, the ids of the rows
  // that take over the slot, then their values -- is fetched right *behind* the two chunk
  // waits, so that every s_waitcnt vmcnt(0) only meets loads issued eight steps earlier
  double idl[4], nxt[4];
  int nid[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { idl[r] = 0.0; nid[r] = 0; nxt[r] = 0.0; }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if ((r & 1) == 0) {   // a new chunk of 8 steps starts with this group of four pivots
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      cur = (chunk & 1) ? lds1 : lds0;
      if ((chunk + 1) * 8 < b) bj_issue_chunk<8>(rec, wr, chunk + 1, (chunk & 1) ? lds0 : lds1, lane);
      ++chunk;
      if (r == 0) {
#pragma unroll
        for (int r2 = 0; r2 < 4; ++r2) {   // branch-free: past the end, entry 0 (masked when used)
          const int j = 16 * g + hi + 4 * r2, jn = j + 16 * NT;
          idl[r2] = invd[j < b ? j : 0];
          nid[r2] = iomap[jn < b ? jn : 0];
        }
      } else {              // the ids have landed: the four row loads go out back to back
        const double* sp[4];
#pragma unroll
        for (int r2 = 0; r2 < 4; ++r2) sp[r2] = src + (rowbase + nid[r2]) * TS + (lo < TS ? lo : 0);
#pragma unroll
        for (int r2 = 0; r2 < 4; ++r2) nxt[r2] = *sp[r2];
      }
    }
    const double* grec = cur + (size_t)(4 * (r & 1)) * wr;     // records of this group's pivots
    const double* prec = grec + (size_t)hi * wr;               // record of "my" pivot (k = hi)
    // every band value this group needs, read with explicit ds_read_b64 + one wait: a
    // compiler-visible LDS read of an LDS-DMA target drains all outstanding VMEM first (the
    // chunk in flight), once per read
    double ct[NT], cg[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {   // pivot a of the group against row hi of the group
      const unsigned ad = (unsigned)(uintptr_t)(lds_void_ptr)(grec + (size_t)a * wr + min(max(hi - a - 1, 0), w));
      asm volatile("ds_read_b64 %0, %1" : "=v"(cg[a]) : "v"(ad));
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const unsigned d1 = (unsigned)(16 * t + lo - 4 * r - hi - 1);
      const unsigned ad = (unsigned)(uintptr_t)(lds_void_ptr)(prec + min(d1, (unsigned)w));
      asm volatile("ds_read_b64 %0, %1" : "=v"(ct[t]) : "v"(ad));
    }
    if constexpr (NT == 4)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cg[0]), "+v"(cg[1]), "+v"(cg[2]), "+v"(ct[0]), "+v"(ct[1]), "+v"(ct[2]), "+v"(ct[3]));
    else if constexpr (NT == 6)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cg[0]), "+v"(cg[1]), "+v"(cg[2]), "+v"(ct[0]), "+v"(ct[1]), "+v"(ct[2]), "+v"(ct[3]), "+v"(ct[4]), "+v"(ct[5]));
    else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cg[0]), "+v"(cg[1]), "+v"(cg[2]), "+v"(ct[0]), "+v"(ct[1]), "+v"(ct[2]), "+v"(ct[3]), "+v"(ct[4]), "+v"(ct[5]), "+v"(ct[6]), "+v"(ct[7]));
    // the four pivots among themselves: the value of pivot a (lanes hi == a) goes down to the
    // rows below it with ds_bpermute -- through the builtin, NOT by hand: `y` is the result of a
    // double-precision matrix instruction (and then of an FMA), which needs wait states before a DS
    // instruction may read it, and the hazard recogniser does not look into inline assembly (bj_g4.hip
    // read stale registers that way; here the nearest producer was 13 instructions ahead: safe by
    // distance only).  The permute touches no LDS memory, so the compiler does not drain the LDS-DMA
    // in flight in front of it (checked in the ISA).  The matrix pipe is the bottleneck of this
    // kernel, so these three steps are not worth four masked MFMAs.
    double y = acc[TP][r];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int from = (a * 16 + lo) * 4;
      const int plo = __builtin_amdgcn_ds_bpermute(from, __double2loint(y));
      const int phi = __builtin_amdgcn_ds_bpermute(from, __double2hiint(y));
      const double ya = __hiloint2double(phi, plo);
      y = fma((hi > a) ? -cg[a] : 0.0, ya, y);
    }
    acc[TP][r] = y;
    // rank-4 update of every tile the band reaches (rows of this group in the pivot tile: done)
    // (tiles past the band meet the record's zero slot: no branch -- a uniform skip makes the
    // compiler merge register states with thousands of moves)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      double cf = ct[t];
      if (t == 0 && (lo >> 2) == r) cf = 0.0;
      acc[(TP + t) % NT] = __builtin_amdgcn_mfma_f64_16x16x4f64(-cf, y, acc[(TP + t) % NT], 0, 0, 0);
    }
  }
  asm volatile("" ::: "memory");   // the next LDS-DMA into these buffers stays behind the reads
  // the tile is solved: scale, store, and take the tile NT further down
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = 16 * g + hi + 4 * r;
    if (j < b && lo < TS) dst[(rowbase + rid[TP][r]) * TS + lo] = acc[TP][r] * idl[r];
    acc[TP][r] = (j + 16 * NT < b && lo < TS) ? nxt[r] : 0.0;
    rid[TP][r] = nid[r];
  }
  __builtin_amdgcn_sched_barrier(0);   // keep the next tile's prefetches from piling up here
}

template <int TS, int NT, int TP>
__device__ __forceinline__ void bjm_tiles(mfma_d4 (&acc)[NT], int (&rid)[NT][4], int g0, int& chunk, int b,
                                          int w, int wr, const double* __restrict__ rec,
                                          const double* __restrict__ invd, const int* __restrict__ iomap,
                                          size_t rowbase, const double* __restrict__ src,
                                          double* __restrict__ dst, double* lds0, double* lds1, int lane) {
  if constexpr (TP < NT) {
    if (16 * (g0 + TP) < b) {
      bjm_tile<TS, NT, TP>(acc, rid, g0 + TP, chunk, b, w, wr, rec, invd, iomap, rowbase, src, dst, lds0, lds1, lane);
      bjm_tiles<TS, NT, TP + 1>(acc, rid, g0, chunk, b, w, wr, rec, invd, iomap, rowbase, src, dst, lds0, lds1, lane);
    }
  }
}

template <int TS, int NT>
__device__ __forceinline__ void bjm_sweep(int b, int w, int wr, const double* __restrict__ rec,
                                          const double* __restrict__ invd, const int* __restrict__ iomap,
                                          size_t rowbase, const double* __restrict__ src,
                                          double* __restrict__ dst, double* lds0, double* lds1, int lane) {
  const int lo = lane & 15, hi = lane >> 4;
  mfma_d4 acc[NT];
  int rid[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 16 * t + hi + 4 * r;
      double v = 0.0;
      int id = 0;
      if (j < b) { id = iomap[j]; if (lo < TS) v = src[(rowbase + id) * TS + lo]; }
      acc[t][r] = v;
      rid[t][r] = id;
    }
  bj_issue_chunk<8>(rec, wr, 0, lds0, lane);
  int chunk = 0;
  for (int g0 = 0; 16 * g0 < b; g0 += NT)
    bjm_tiles<TS, NT, 0>(acc, rid, g0, chunk, b, w, wr, rec, invd, iomap, rowbase, src, dst, lds0, lds1, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int TS, int NT>
__global__ __launch_bounds__(256) void k_bj_mfma(
    const int* __restrict__ list, int count, const int* __restrict__ row0,
    const int* __restrict__ nrows, const int* __restrict__ bw, const long long* __restrict__ off,
    const int* __restrict__ map_f, const int* __restrict__ map_b, const double* __restrict__ Lf,
    const double* __restrict__ Lb, const double* __restrict__ invd_f,
    const double* __restrict__ invd_b, int lds_per_wave, const double* __restrict__ in,
    double* __restrict__ out) {
  extern __shared__ double smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int pi = blockIdx.x * (blockDim.x >> 6) + wave;
  if (pi >= count) return;
  const int p = __builtin_amdgcn_readfirstlane(list[pi]);
  const int r0 = __builtin_amdgcn_readfirstlane(row0[p]);
  const int b = __builtin_amdgcn_readfirstlane(nrows[p]);
  const int w = __builtin_amdgcn_readfirstlane(bw[p]);
  const int wr = (w + 2) & ~1;
  const long long o64 = off[p];
  const size_t o = ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(o64 >> 32)) << 32) |
                   (unsigned)__builtin_amdgcn_readfirstlane((int)o64);
  double* lds0 = smem + (size_t)wave * lds_per_wave;
  double* lds1 = lds0 + (lds_per_wave >> 1);
  bjm_sweep<TS, NT>(b, w, wr, Lf + o, invd_f + r0, map_f + r0, (size_t)r0, in, out, lds0, lds1, lane);
  __threadfence_block();
  bjm_sweep<TS, NT>(b, w, wr, Lb + o, invd_b + r0, map_b + r0, (size_t)r0, out, out, lds0, lds1, lane);
}

// The sweep is blocked by 64 pivots.  Phase A: the wave that holds the block's rows
// eliminates them among themselves (in-wave, v_readlane; its 64 x 64 coefficients were
// brought into LDS by LDS-DMA during the previous block) and publishes the 64 solved rows in
// LDS.  Barrier.  Phase B: every wave applies the 64 pivots to its other rows, reading them
// back as LDS broadcasts; the band values come through a register ring that keeps D steps
// (D*R loads per lane, 16-32 KiB per wave) in flight across block boundaries.  Barrier.
// Two barriers per 64 steps instead of one per step; every record entry is read once.
// Always 64 steps: in a short last block the missing pivots are zero rows and their (clamped)
// coefficients multiply zeros, so no step is conditional.  The coefficients arrive in LDS by
// LDS-DMA; they are read with explicit ds_read_b64 + s_waitcnt (8 steps per batch) because a
// compiler-visible LDS read of a DMA target makes the compiler drain every outstanding VMEM
// load of the wave (the whole prefetch ring) first.  `la` = LDS byte address of dl[lane].
template <int TS, int R, int K>
__device__ __forceinline__ void bjb_diag(double (&acc)[R][TS], unsigned la) {
  constexpr int DA = 8;
#pragma unroll
  for (int l0 = 0; l0 < 64; l0 += DA) {
    double q[DA];
#pragma unroll
    for (int u = 0; u < DA; ++u)
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(q[u]) : "v"(la), "n"((l0 + u) * 512));
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]));
#pragma unroll
    for (int u = 0; u < DA; ++u) {
      const int l = l0 + u;
#pragma unroll
      for (int c = 0; c < TS; ++c) {
        const double y = readlane_f64(acc[K][c], l);
        acc[K][c] = fma(-q[u], y, acc[K][c]);
      }
    }
  }
  asm volatile("" ::: "memory");   // the next LDS-DMA into this buffer stays behind these reads
}

// 64 records x 64 slots of the diagonal block starting at record `rec0` -> LDS, 1 KiB (two
// records) per instruction; records past the end of the block are clamped (never used).
__device__ __forceinline__ void bjb_issue_diag(const double* __restrict__ rec, int W, int jb, int b,
                                               double* dl, int lane) {
  const int sb = jb % W;
  const int half = lane >> 5, l16 = lane & 31;
  for (int i = 0; i < 32; ++i) {
    int r = jb + 2 * i + half;
    r = r < b ? r : b - 1;
    const char* g = reinterpret_cast<const char*>(rec + (size_t)r * W + sb) + l16 * 16;
    __builtin_amdgcn_global_load_lds((glb_void_ptr)g, (lds_void_ptr)(reinterpret_cast<char*>(dl) + i * 1024), 16, 0, 0);
  }
}

// Phase B of one block; q is the ring (step jb + l sits in q[l % D]), refilled D steps ahead
// (into the next block's records, clamped at the end of the subdomain).  recb = first record
// of the block (wave-uniform, so the loads take a scalar base + the lane's slot offset);
// kskip = the owner's register set, whose slots of these records belong to phase A (-1: none).
template <int TS, int R, int D>
__device__ __forceinline__ void bjb_update(double (&acc)[R][TS], double (&q)[D][R],
                                           const double* __restrict__ recb, int s0, int W, int nleft,
                                           const double (*yb)[TS], int kskip) {
  // the pivots are read P steps ahead of their use (LDS broadcasts, ~100 cycles each)
  constexpr int P = TS * R <= 8 ? 2 : 1;
  double2 yq[P][TS / 2];
#pragma unroll
  for (int a = 0; a < P; ++a)
#pragma unroll
    for (int c = 0; c < TS / 2; ++c) yq[a][c] = reinterpret_cast<const double2*>(yb[a])[c];
#pragma unroll
  for (int l = 0; l < 64; ++l) {
    const int u = l % D;
    double y[TS];
#pragma unroll
    for (int c = 0; c < TS / 2; ++c) { y[2 * c] = yq[l % P][c].x; y[2 * c + 1] = yq[l % P][c].y; }
    if (l + P < 64) {
#pragma unroll
      for (int c = 0; c < TS / 2; ++c) yq[l % P][c] = reinterpret_cast<const double2*>(yb[l + P])[c];
    }
    const int ln = (l + D) < nleft ? (l + D) : nleft - 1;
    const double* __restrict__ pr = recb + (size_t)ln * W;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const double lv = (k == kskip) ? 0.0 : q[u][k];
      q[u][k] = pr[s0 + k * 64];
#pragma unroll
      for (int c = 0; c < TS; ++c) acc[k][c] = fma(-lv, y[c], acc[k][c]);
    }
    // keep the issue order of the source: the scheduler otherwise sinks the LDS reads back
    // next to their uses to save registers and every step eats the full LDS latency
    __builtin_amdgcn_sched_barrier(0);
  }
}

// phase A + hand-over of the owner wave for register set K.  The id of the row that takes
// over the slot was fetched one ownership earlier (rownext), so its load does not hang on a
// fresh index load behind the whole prefetch ring.
template <int TS, int R, int K>
__device__ __forceinline__ void bjb_own(double (&acc)[R][TS], int (&rowid)[R], int (&rownext)[R],
                                        const double* dl, int W, int lim, int jb, int b,
                                        const double* __restrict__ invd, const int* __restrict__ iomap,
                                        size_t rowbase, const double* __restrict__ src,
                                        double* __restrict__ dst, double (*yb)[TS], int lane) {
  double nxt[TS];
  double idl = 0.0;
  const int nrow = rownext[K];
  const int jn = jb + W + lane;
  if (jb + lane < b) idl = invd[jb + lane];
  if (jn < b) load_row<TS>(src, rowbase + nrow, nxt);
  else
#pragma unroll
    for (int c = 0; c < TS; ++c) nxt[c] = 0.0;
  int rn = (jn + W < b) ? iomap[jn + W] : 0;
  bjb_diag<TS, R, K>(acc, (unsigned)(uintptr_t)(lds_void_ptr)(dl + lane));
  // lane l now holds the solved (unscaled) row jb + l: publish, scale, store, take the next row
  double2* yq = reinterpret_cast<double2*>(yb[lane]);
#pragma unroll
  for (int c = 0; c < TS / 2; ++c) yq[c] = make_double2(acc[K][2 * c], acc[K][2 * c + 1]);
  double v[TS];
#pragma unroll
  for (int c = 0; c < TS; ++c) v[c] = acc[K][c] * idl;
  if (lane < lim) store_row<TS>(dst, rowbase + rowid[K], v);
  // the incoming row and the id prefetched for the next ownership are touched here, where
  // their loads have long landed: otherwise the compiler sinks the copy into the loop latch
  // (draining every wave's prefetch ring there) and waits for the id at the start of the next
  // ownership, behind the whole ring
#pragma unroll
  for (int c = 0; c < TS; ++c) { asm volatile("" : "+v"(nxt[c])); acc[K][c] = nxt[c]; }
  rowid[K] = nrow;
  asm volatile("" : "+v"(rn));
  rownext[K] = rn;
}

template <int TS, int R, int D>
__device__ __forceinline__ void bjw_sweep(int b, int W, const double* __restrict__ rec,
                                          const double* __restrict__ invd,
                                          const int* __restrict__ iomap, size_t rowbase,
                                          const double* __restrict__ src, double* __restrict__ dst,
                                          double (*ybuf)[64][TS], double (*dlbuf)[64 * 64], int wave,
                                          int lane, bool active) {
  const int s0 = (active ? wave : 0) * 64 * R + lane;   // slot of register set 0 of this lane
  double acc[R][TS];
  int rowid[R], rownext[R];
  // diagonal coefficients of the first two blocks; later ones are fetched two blocks ahead by
  // the owner that has just finished with the buffer
  if (wave == 0) {
    bjb_issue_diag(rec, W, 0, b, dlbuf[0], lane);
    if (64 < b) bjb_issue_diag(rec, W, 64, b, dlbuf[1], lane);
  }
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int j = s0 + k * 64;
    rowid[k] = 0;
    rownext[k] = (active && j + W < b) ? iomap[j + W] : 0;
    if (active && j < b) { rowid[k] = iomap[j]; load_row<TS>(src, rowbase + rowid[k], acc[k]); }
    else
#pragma unroll
      for (int c = 0; c < TS; ++c) acc[k][c] = 0.0;
  }
  double q[D][R];
#pragma unroll
  for (int u = 0; u < D; ++u)
#pragma unroll
    for (int k = 0; k < R; ++k) q[u][k] = rec[(size_t)(u < b ? u : b - 1) * W + s0 + k * 64];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int jb = 0; jb < b; jb += 64) {
    const int sb = jb % W;
    const int ow = sb / (64 * R), ok = (sb >> 6) % R;   // wave / register set that owns this block
    const bool mine = active && wave == ow;
    const int lim = (b - jb) < 64 ? (b - jb) : 64;
    const int par = (jb >> 6) & 1;
    double (*yb)[TS] = ybuf[par];
    if (mine) {
      double* dl = dlbuf[par];
      if (ok == 0) bjb_own<TS, R, 0>(acc, rowid, rownext, dl, W, lim, jb, b, invd, iomap, rowbase, src, dst, yb, lane);
      if constexpr (R > 1) if (ok == 1) bjb_own<TS, R, 1>(acc, rowid, rownext, dl, W, lim, jb, b, invd, iomap, rowbase, src, dst, yb, lane);
      if constexpr (R > 2) {
        if (ok == 2) bjb_own<TS, R, 2>(acc, rowid, rownext, dl, W, lim, jb, b, invd, iomap, rowbase, src, dst, yb, lane);
        if (ok == 3) bjb_own<TS, R, 3>(acc, rowid, rownext, dl, W, lim, jb, b, invd, iomap, rowbase, src, dst, yb, lane);
      }
    }
    // everything this wave has in flight lands before the barrier (the compiler drains all
    // counters in front of an s_barrier on gfx9 anyway): the published pivots, and the owner's
    // coefficient fetch of the previous block
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // the owner's coefficient buffer is free again: fetch the block after next into it.  The
    // fetch is complete at the next barrier, one block before it is read.
    if (mine && jb + 128 < b) bjb_issue_diag(rec, W, jb + 128, b, dlbuf[par], lane);
    if (active) bjb_update<TS, R, D>(acc, q, rec + (size_t)jb * W, s0, W, b - jb, yb, mine ? ok : -1);
  }
}

// NT = threads the launch may use: with at most 12 wavefronts (3 per SIMD) a lane has 168
// VGPRs and the ring can hold twice as many steps.
template <int TS, int R, int NT>
__global__ __launch_bounds__(NT) void k_bj_wide(
    const int* __restrict__ list, int count, const int* __restrict__ row0,
    const int* __restrict__ nrows, const int* __restrict__ bw, const long long* __restrict__ off,
    const int* __restrict__ map_f, const int* __restrict__ map_b, const double* __restrict__ Lf,
    const double* __restrict__ Lb, const double* __restrict__ invd_f,
    const double* __restrict__ invd_b, const double* __restrict__ in, double* __restrict__ out) {
  __shared__ double ybuf[2][64][TS];
  __shared__ double dl[2][64 * 64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int p = __builtin_amdgcn_readfirstlane(list[blockIdx.x]);
  const int r0 = __builtin_amdgcn_readfirstlane(row0[p]);
  const int b = __builtin_amdgcn_readfirstlane(nrows[p]);
  const int w = __builtin_amdgcn_readfirstlane(bw[p]);
  const int W = bjw_window(w);
  const bool active = wave < W / (64 * R);
  const size_t o = (size_t)off[p];
  constexpr int D = (TS == 16 ? 8 : 16) * (NT <= 768 ? 2 : 1) / R;
  bjw_sweep<TS, R, D>(b, W, Lf + o, invd_f + r0, map_f + r0, (size_t)r0, in, out, ybuf, dl, wave, lane, active);
  __threadfence_block();
  __syncthreads();
  bjw_sweep<TS, R, D>(b, W, Lb + o, invd_b + r0, map_b + r0, (size_t)r0, out, out, ybuf, dl, wave, lane, active);
}

// ------------------------------------------------ sparse block solve (large blocks) ----
// Sparse block solve with the nested-dissection factor of nd.c.  Every front (n pivot columns,
// m rows below) is stored as the panel P = [T ; -G], T = strictly lower part of (L_11 D^-1)^-1,
// G = (L_21 D^-1) (L_11 D^-1)^-1, D = diag(L_11) -- the "selective inversion" form of a supernodal
// factor: both sweeps become products of dense panels with short vectors, there is no recurrence
// inside a front and one launch takes a whole level of the tree.
//   forward   w = x(pivot rows) + children's contributions;  a = w + T w;  y = D^-1 a  -> Y
//             contribution to the parent = children's contributions(rows below) + (-G) w
//   backward  v = [D^-1 y ; z(rows below, final)];  z_k = v_k + sum_{i > k} P(i, k) v_i
// Two copies of P: column major (forward: a thread owns a front row, lanes = consecutive rows) and
// row major (backward: a thread owns a pivot column, lanes = consecutive columns).
//
// A thread adds coefficient(j) * ys[j] over 16-wide groups of an index range (j = 16 g + u, kept
// to lo <= j < hi); consecutive j are `stride` doubles apart at p.  Q threads (wavefronts) share
// one output, groups dealt round robin, and meet in LDS afterwards.  The vector is staged in LDS
// in rounds; the coefficients of a group are requested one group ahead, across the rounds'
// barriers, so a workgroup that is alone on its CU still keeps 16 loads per thread in flight.
__device__ __forceinline__ void nd_dot_load(double (&cf)[16], const double* __restrict__ p, size_t stride, int g, int gtot,
                                            int lo, int hi, bool on) {
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int j = 16 * g + u;
    cf[u] = (on && g < gtot && j >= lo && j < hi) ? p[(size_t)j * stride] : 0.0;
  }
}

template <int TS, int Q, int YN>
__device__ __forceinline__ void nd_dot(double (&acc)[TS], double (&cf)[16], const double* __restrict__ p, size_t stride,
                                       int& g, int gend, int gtot, int lo, int hi, bool on, const double (*ys)[TS],
                                       int y0) {
  double nx[16];
#pragma unroll 1
  for (; g < gend; g += Q) {
    nd_dot_load(nx, p, stride, g + Q, gtot, lo, hi, on);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const double2* yq = reinterpret_cast<const double2*>(ys[min(16 * g + u - y0, YN - 1)]);
#pragma unroll
      for (int c = 0; c < TS / 2; ++c) {
        const double2 yv = yq[c];
        acc[2 * c] = fma(cf[u], yv.x, acc[2 * c]);
        acc[2 * c + 1] = fma(cf[u], yv.y, acc[2 * c + 1]);
      }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) cf[u] = nx[u];
  }
}

struct nd_args {
  const int* n; const int* m; const int* ld; const long long* offF; const long long* offB; const int* rows_off;
  const int* coff; const int* ccoff; const int* rows; const int* src; const double* dinv; const double* F;
  const double* B; double* contrib; double* Y;
};
constexpr int ND_CHUNK = 256;                 /* front rows per forward workgroup */
constexpr int ND_COLS = 64;                   /* pivot columns per backward workgroup */
constexpr int ND_STAGE = 512;                 /* entries of w staged per round (forward) */

template <int TS, int XS>
__device__ __forceinline__ void nd_gather_w(const nd_args& a, const int* __restrict__ rows, const int* __restrict__ src,
                                            int cc0, int cc1, int j, int n, const double* __restrict__ in, int coff,
                                            double (&w)[TS]) {
  const int s0 = src[2 * j], s1 = src[2 * j + 1];
  if (j < n) load_row_s<TS, XS>(in + coff, (size_t)rows[j], w);
  double t[TS];
  if (s0 >= 0) {
    load_row_s<TS, XS>(a.contrib + coff, (size_t)(cc0 + s0), t);
#pragma unroll
    for (int c = 0; c < TS; ++c) w[c] += t[c];
  }
  if (s1 >= 0) {
    load_row_s<TS, XS>(a.contrib + coff, (size_t)(cc1 + s1), t);
#pragma unroll
    for (int c = 0; c < TS; ++c) w[c] += t[c];
  }
}

// Forward, one workgroup per (front, chunk of ND_CHUNK front rows); Q threads per row.
template <int TS, int XS, int Q>
__global__ __launch_bounds__(ND_CHUNK * Q) void k_nd_forward(nd_args a, const int* __restrict__ cfront,
                                                             const int* __restrict__ crow0,
                                                             const double* __restrict__ in) {
  constexpr int STG = TS >= 16 ? ND_STAGE / 2 : ND_STAGE;   // entries of w staged per round (32 KiB at 8 and 16 columns)
  __shared__ double ws[STG][TS];
  __shared__ double red[Q > 1 ? Q - 1 : 1][ND_CHUNK][TS];
  const int s = cfront[blockIdx.x], r0 = crow0[blockIdx.x], coff = blockIdx.y * TS;
  const int n = a.n[s], f = n + a.m[s], ld = a.ld[s];
  const double* __restrict__ P = a.F + a.offF[s];
  const int* __restrict__ rows = a.rows + a.rows_off[s];
  const int* __restrict__ src = a.src + 2 * (size_t)a.rows_off[s];
  const int cc0 = a.ccoff[2 * s], cc1 = a.ccoff[2 * s + 1];
  const int tid = threadIdx.x, rl = tid % ND_CHUNK, q = __builtin_amdgcn_readfirstlane(tid / ND_CHUNK);
  const int r = r0 + rl;
  const bool on = r < f;
  const int hi = min(r, n);                   // columns 0 .. hi - 1 of front row r
  const int gtot = (hi + 15) >> 4;
  const int jwg = min(n, r0 + ND_CHUNK);      // columns this workgroup meets
  const double* __restrict__ p = P + (on ? r : 0);
  double cf[16];
  int g = q;
  nd_dot_load(cf, p, (size_t)ld, g, gtot, 0, hi, on);
  double acc[TS], own[TS];
#pragma unroll
  for (int c = 0; c < TS; ++c) { acc[c] = 0.0; own[c] = 0.0; }
  if (q == 0 && on && r >= n) nd_gather_w<TS, XS>(a, rows, src, cc0, cc1, r, n, in, coff, own);
  for (int c0 = 0; c0 < jwg; c0 += STG) {
    if (c0 > 0) __syncthreads();
    for (int jj = tid; jj < STG; jj += ND_CHUNK * Q) {
      double w[TS];
#pragma unroll
      for (int c = 0; c < TS; ++c) w[c] = 0.0;
      if (c0 + jj < jwg) nd_gather_w<TS, XS>(a, rows, src, cc0, cc1, c0 + jj, n, in, coff, w);
      double2* wq = reinterpret_cast<double2*>(ws[jj]);
#pragma unroll
      for (int c = 0; c < TS / 2; ++c) wq[c] = make_double2(w[2 * c], w[2 * c + 1]);
    }
    __syncthreads();
    if (q == 0 && on && r < n && r >= c0 && r < c0 + STG) {
#pragma unroll
      for (int c = 0; c < TS; ++c) own[c] = ws[r - c0][c];
    }
    nd_dot<TS, Q, STG>(acc, cf, p, (size_t)ld, g, min(gtot, (c0 + STG) >> 4), gtot, 0, hi, on, ws, c0);
  }
  if constexpr (Q > 1) {
    if (q > 0) {
#pragma unroll
      for (int c = 0; c < TS; ++c) red[q - 1][rl][c] = acc[c];
    }
    __syncthreads();
  }
  if (q == 0 && on) {
#pragma unroll
    for (int c = 0; c < TS; ++c) {
      double sm = acc[c];
      if constexpr (Q > 1) {
#pragma unroll
        for (int k = 0; k < Q - 1; ++k) sm += red[k][rl][c];
      }
      own[c] += sm;
    }
    if (r < n) {
      const int gr = rows[r];
      const double id = a.dinv[gr];
#pragma unroll
      for (int c = 0; c < TS; ++c) own[c] *= id;
      store_row_s<TS, XS>(a.Y + coff, (size_t)gr, own);
    } else {
      store_row_s<TS, XS>(a.contrib + coff, (size_t)(a.coff[s] + r - n), own);
    }
  }
}

// Backward, one workgroup per (front, block of ND_COLS pivot columns); W wavefronts share the rows.
template <int TS, int XS, int W>
__global__ __launch_bounds__(64 * W) void k_nd_backward(nd_args a, const int* __restrict__ cfront,
                                                        const int* __restrict__ ccol0, double* __restrict__ out) {
  constexpr int NSTG = TS >= 16 ? 256 : TS >= 8 ? 512 : 1024;  // rows of v staged per round
  __shared__ double vs[NSTG][TS];
  __shared__ double red[W - 1][ND_COLS][TS];
  const int s = cfront[blockIdx.x], k0 = ccol0[blockIdx.x], coff = blockIdx.y * TS;
  const int n = a.n[s], f = n + a.m[s], ldb = PA_ND_LD(n);
  const double* __restrict__ U = a.B + a.offB[s];
  const int* __restrict__ rows = a.rows + a.rows_off[s];
  const int tid = threadIdx.x, kk = tid & 63, q = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = k0 + kk;
  const bool on = k < n;
  const int gtot = (f + 15) >> 4;
  const double* __restrict__ p = U + (on ? k : 0);
  double cf[16];
  int g = (k0 >> 4) + q;
  nd_dot_load(cf, p, (size_t)ldb, g, gtot, k + 1, f, on);
  double acc[TS], own[TS];
#pragma unroll
  for (int c = 0; c < TS; ++c) { acc[c] = 0.0; own[c] = 0.0; }
  for (int c0 = k0; c0 < f; c0 += NSTG) {
    if (c0 > k0) __syncthreads();
    for (int ii = tid; ii < NSTG; ii += 64 * W) {
      const int i = c0 + ii;
      double v[TS];
#pragma unroll
      for (int c = 0; c < TS; ++c) v[c] = 0.0;
      if (i < f) {
        const int gr = rows[i];
        if (i < n) {
          load_row_s<TS, XS>(a.Y + coff, (size_t)gr, v);
          const double id = a.dinv[gr];
#pragma unroll
          for (int c = 0; c < TS; ++c) v[c] *= id;
        } else {
          load_row_s<TS, XS>(out + coff, (size_t)gr, v);
        }
      }
      double2* vq = reinterpret_cast<double2*>(vs[ii]);
#pragma unroll
      for (int c = 0; c < TS / 2; ++c) vq[c] = make_double2(v[2 * c], v[2 * c + 1]);
    }
    __syncthreads();
    if (q == 0 && on && k >= c0 && k < c0 + NSTG) {
#pragma unroll
      for (int c = 0; c < TS; ++c) own[c] = vs[k - c0][c];
    }
    nd_dot<TS, W, NSTG>(acc, cf, p, (size_t)ldb, g, min(gtot, (c0 + NSTG) >> 4), gtot, k + 1, f, on, vs, c0);
  }
  if (q > 0) {
#pragma unroll
    for (int c = 0; c < TS; ++c) red[q - 1][kk][c] = acc[c];
  }
  __syncthreads();
  if (q == 0 && on) {
#pragma unroll
    for (int c = 0; c < TS; ++c) {
      double sm = acc[c];
#pragma unroll
      for (int j = 0; j < W - 1; ++j) sm += red[j][kk][c];
      own[c] += sm;
    }
    store_row_s<TS, XS>(out + coff, (size_t)rows[k], own);
  }
}

inline int grid_rows(int m, int per_thread_rows = 1) {
  long long blocks = ((long long)m + (long long)WG * per_thread_rows - 1) / ((long long)WG * per_thread_rows);
  if (blocks < 1) blocks = 1;
  const long long cap = 2048;
  return (int)(blocks < cap ? blocks : cap);
}

}  // namespace

// Sequence number for the next launch that writes its two words to pinned host memory (host[2] =
// seq behind them): the host then polls that word instead of waiting for an event, whose record
// costs the stream 5-6 us of idle time.  One-shot: taken by that launch, 0 = no number.
static double g_note_seq = 0.0;
static inline double take_note_seq(const double* host) {
  if (!host) return 0.0;
  const double v = g_note_seq;
  g_note_seq = 0.0;
  return v;
}


template <int TS, int CH, int XS>
static int bj_launch_ch(const pa_bj_plan_t* pl, int R, int wmax, const int* list, int count,
                        const double* in, double* out) {
  // LDS per wave: two chunk buffers of CH records of the widest band in this class
  const int wr = (wmax + 2) & ~1;
  // Two chunk buffers.  (A ring of three with counted waits measured neutral in round 2 -- 183.6 vs 187.1 us
  // on elasticity, 173.6 vs 172.1 on Poisson, same box: the sweep does not wait for its band -- and its
  // switch is gone; the kernels still take the ring depth as an argument.)
  const int cbuf = (CH * wr + 127) & ~127;      // doubles, each buffer a multiple of 1 KiB
  const int nbuf = 2;
  int per_wave = nbuf * cbuf;
  int waves = (160 * 1024) / (per_wave * 8);
  if (waves > 4) waves = 4;
  if (waves < 1) { snprintf(g_kerr, sizeof(g_kerr), "block-Jacobi band too wide for LDS (R=%d)", R); return 1; }
  const size_t lds = (size_t)waves * per_wave * 8;
  const int units = count * (XS / TS);   // one wavefront per (subdomain, column group)
  const int blocks = (units + waves - 1) / waves;
#define BJ_CASE(RR)                                                                               \
  case RR: {                                                                                      \
    static size_t configured = 0;                                                                 \
    if (lds > 64 * 1024 && lds > configured) {                                                    \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bj_apply<TS, RR, CH, XS, 1>),          \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
        return kfail("hipFuncSetAttribute(k_bj_apply)");                                          \
      configured = lds;                                                                           \
    }                                                                                             \
    PA_LAUNCH((k_bj_apply<TS, RR, CH, XS, 1>), dim3(blocks), dim3(64 * waves), lds,          \
                       cur_stream(), list, count, pl->row0, pl->nrows, pl->bw, pl->off,           \
                       pl->map_f, pl->map_b, pl->Lf, pl->Lb, pl->invd_f, pl->invd_b, per_wave, nbuf, in, out); \
  } break;
  // paired records (pa_k_bj_pairs ran at setup): classes R = 2, 3 at up to 4 columns
  if constexpr (TS <= 4) {
    if (pl->Lf2 && (R == 2 || R == 3)) {
      const int wr2 = wmax + 4;
      const int cb2 = (CH * wr2 + 127) & ~127;
      const int nbuf2 = 2;
      const int pw2 = nbuf2 * cb2;
      int wv2 = (160 * 1024) / (pw2 * 8);
      if (wv2 > 4) wv2 = 4;
      if (wv2 >= 1) {
        const size_t lds2 = (size_t)wv2 * pw2 * 8;
        const int blocks2 = (units + wv2 - 1) / wv2;
        static size_t conf2[2] = {0, 0};
        const void* fn = R == 2 ? reinterpret_cast<const void*>(&k_bj_apply_pairs<TS, 2, CH, XS>)
                                : reinterpret_cast<const void*>(&k_bj_apply_pairs<TS, 3, CH, XS>);
        if (lds2 > 64 * 1024 && lds2 > conf2[R - 2]) {
          if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)
            return kfail("hipFuncSetAttribute(k_bj_apply_pairs)");
          conf2[R - 2] = lds2;
        }
        if (R == 2)
          PA_LAUNCH((k_bj_apply_pairs<TS, 2, CH, XS>), dim3(blocks2), dim3(64 * wv2), lds2, cur_stream(), list, count,
                             pl->row0, pl->nrows, pl->bw, pl->off2, pl->map_f, pl->map_b, pl->Lf2, pl->Lb2, pl->invd_f,
                             pl->invd_b, pw2, nbuf2, in, out);
        else
          PA_LAUNCH((k_bj_apply_pairs<TS, 3, CH, XS>), dim3(blocks2), dim3(64 * wv2), lds2, cur_stream(), list, count,
                             pl->row0, pl->nrows, pl->bw, pl->off2, pl->map_f, pl->map_b, pl->Lf2, pl->Lb2, pl->invd_f,
                             pl->invd_b, pw2, nbuf2, in, out);
        return kfail("k_bj_apply_pairs");
      }
    }
  }
  switch (R) {
    BJ_CASE(1) BJ_CASE(2) BJ_CASE(3) BJ_CASE(4) BJ_CASE(5) BJ_CASE(6) BJ_CASE(7) BJ_CASE(8)
    default:
      snprintf(g_kerr, sizeof(g_kerr), "block-Jacobi bandwidth class R=%d unsupported", R);
      return 1;
  }
#undef BJ_CASE
  return kfail("k_bj_apply");
}

template <int TS>
static int bj_launch(const pa_bj_plan_t* pl, int R, int wmax, const int* list, int count,
                     const double* in, double* out) {
  // chunks of 8 steps: measured equal or better than 16 and 32 (smaller LDS footprint,
  // more workgroups per CU)
  // The matrix-core sweep (k_bj_mfma, bands up to 112) costs 210-250 us per apply whatever the
  // panel width (latency: three wavefronts per SIMD, the chain of three lane moves and the
  // pivot tile's MFMA per group of pivots), the register recurrence 172 / 246 / 443 us at
  // 4 / 8 / 16 columns: so it takes the panels of 8 and 16 columns.
  // PREALPS_BJ_MFMA=0: never; 2: always.
  static int use_mfma = -1;
  if (use_mfma < 0) { const char* e = getenv("PREALPS_BJ_MFMA"); use_mfma = e ? atoi(e) : 1; }
  if (wmax <= 112 && ((TS >= 8 && use_mfma == 1) || use_mfma == 2)) {
    const int wr = (wmax + 2) & ~1;
    int per_wave = 2 * ((8 * wr + 127) & ~127);
    int waves = 4;
    const size_t lds = (size_t)waves * per_wave * 8;
    const int blocks = (count + waves - 1) / waves;
#define BJM_LAUNCH(NTT)                                                                              \
    PA_LAUNCH((k_bj_mfma<TS, NTT>), dim3(blocks), dim3(64 * waves), lds, cur_stream(), list, count, \
                       pl->row0, pl->nrows, pl->bw, pl->off, pl->map_f, pl->map_b, pl->Lf, pl->Lb,   \
                       pl->invd_f, pl->invd_b, per_wave, in, out)
    if (wmax <= 48) BJM_LAUNCH(4); else if (wmax <= 80) BJM_LAUNCH(6); else BJM_LAUNCH(8);
#undef BJM_LAUNCH
    return kfail("k_bj_mfma");
  }
  // (Panels of 8 / 16 columns as 2 / 4 wavefronts of 4 columns each per subdomain measured slower in round 1 --
  // each wavefront streams the factor again through L2: 257 vs 243 us at 8 columns, 485 vs 434 us at 16 -- and
  // the variant is gone.)
  // Few blocks (a GPU of a multi-GPU run holds 1/8 of them): below one wavefront per SIMD the
  // sweep is latency bound, so a 4-column panel is shared by two wavefronts of 2 columns each
  // (half the FMAs and pivot broadcasts per wavefront; the band is read twice, from L2).
  if constexpr (TS == 4) {
    const int simds = 4 * (pa_rt_num_cus() > 0 ? pa_rt_num_cus() : 256);
    if (count < simds) return bj_launch_ch<2, 8, 4>(pl, R, wmax, list, count, in, out);
  }
  return bj_launch_ch<TS, 8, TS>(pl, R, wmax, list, count, in, out);
}

// R = register sets per lane (1 / 2 / 4 for windows up to 1024 / 2048 / 4096 rows)
template <int TS, int R>
static int bj_launch_wide(const pa_bj_plan_t* pl, int wmax, const int* list, int count, const double* in,
                          double* out) {
  const int W = bjw_window(wmax);
  const int nw = (W + 64 * R - 1) / (64 * R);
  if (nw > 16 || TS * R > 16) {
    snprintf(g_kerr, sizeof(g_kerr),
             "block-Jacobi: bandwidth %d is too wide for panel stride %d (window %d rows); use more subdomains",
             wmax, TS, W);
    fprintf(stderr, "[prealps_hip] %s\n", g_kerr);
    return 1;
  }
  if constexpr (TS * R <= 16) {
    if (nw <= 12)
      PA_LAUNCH((k_bj_wide<TS, R, 768>), dim3(count), dim3(64 * nw), 0, cur_stream(), list, count,
                         pl->row0, pl->nrows, pl->bw, pl->off, pl->map_f, pl->map_b, pl->Lf, pl->Lb,
                         pl->invd_f, pl->invd_b, in, out);
    else
      PA_LAUNCH((k_bj_wide<TS, R, 1024>), dim3(count), dim3(64 * nw), 0, cur_stream(), list, count,
                         pl->row0, pl->nrows, pl->bw, pl->off, pl->map_f, pl->map_b, pl->Lf, pl->Lb,
                         pl->invd_f, pl->invd_b, in, out);
  }
  return kfail("k_bj_wide");
}

template <int R>
static int bj_wide_dispatch(const pa_bj_plan_t* pl, int ts, int wmax, const int* list, int count,
                            const double* in, double* out) {
  switch (ts) {
    case 2: return bj_launch_wide<2, R>(pl, wmax, list, count, in, out);
    case 4: return bj_launch_wide<4, R>(pl, wmax, list, count, in, out);
    case 8: return bj_launch_wide<8, R>(pl, wmax, list, count, in, out);
    case 16: return bj_launch_wide<16, R>(pl, wmax, list, count, in, out);
    default: return 1;
  }
}

template <int NB>
static int bj_factor_big_launch(const int* list, int count, int wmax, const int* row0, const int* nrows,
                                const int* bw, const long long* boff, double* band, int* fail) {
  const size_t lds = ((size_t)wmax * NB + NB * NB) * 8;
  static size_t configured = 0;
  if (lds > 64 * 1024 && lds > configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bj_factor_big<NB>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return kfail("hipFuncSetAttribute(k_bj_factor_big)");
    configured = lds;
  }
  PA_LAUNCH((k_bj_factor_big<NB>), dim3(count), dim3(1024), lds, cur_stream(), list, row0, nrows, bw,
                     boff, band, fail);
  return kfail("k_bj_factor_big");
}

// 16-column panels: all columns in one pass over the factor (172 VGPRs, two wavefronts per SIMD: 2.6 TB/s
// of factor bytes) instead of two 8-column passes at 4.5 TB/s each: 4.37 against 5.09 ms per apply on the
// 64-block elasticity problem (round 3, same-call A/B).
// one level of the tree; launches of few workgroups put more threads on each output
template <int XS>
static int nd_launch_fwd(const nd_args& a, const int* cfront, const int* crow0, int nwg, const double* in) {
  if (nwg <= 0) return 0;
  constexpr int few = 1024;
  if constexpr (XS == 16) {                   // all 16 columns in one pass over the factor
    if (nwg < few) PA_LAUNCH((k_nd_forward<16, 16, 2>), dim3(nwg), dim3(ND_CHUNK * 2), 0, cur_stream(), a, cfront, crow0, in);
    else PA_LAUNCH((k_nd_forward<16, 16, 1>), dim3(nwg), dim3(ND_CHUNK), 0, cur_stream(), a, cfront, crow0, in);
    return kfail("k_nd_forward");
  }
  constexpr int TS = XS <= 8 ? XS : 8;
  constexpr int QB = TS >= 8 ? 2 : 4;
  const dim3 grid(nwg, XS / TS);
  if (nwg < few) PA_LAUNCH((k_nd_forward<TS, XS, QB>), grid, dim3(ND_CHUNK * QB), 0, cur_stream(), a, cfront, crow0, in);
  else PA_LAUNCH((k_nd_forward<TS, XS, 1>), grid, dim3(ND_CHUNK), 0, cur_stream(), a, cfront, crow0, in);
  return kfail("k_nd_forward");
}

template <int XS>
static int nd_launch_bwd(const nd_args& a, const int* cfront, const int* ccol0, int nwg, double* out) {
  if (nwg <= 0) return 0;
  constexpr int few = 2048;
  if constexpr (XS == 16) {
    PA_LAUNCH((k_nd_backward<16, 16, 4>), dim3(nwg), dim3(256), 0, cur_stream(), a, cfront, ccol0, out);
    return kfail("k_nd_backward");
  }
  constexpr int TS = XS <= 8 ? XS : 8;
  constexpr int WB = TS >= 8 ? 8 : 16;
  const dim3 grid(nwg, XS / TS);
  if (nwg < few) PA_LAUNCH((k_nd_backward<TS, XS, WB>), grid, dim3(64 * WB), 0, cur_stream(), a, cfront, ccol0, out);
  else PA_LAUNCH((k_nd_backward<TS, XS, 4>), grid, dim3(256), 0, cur_stream(), a, cfront, ccol0, out);
  return kfail("k_nd_backward");
}

extern "C" {

int pa_bj_max_R(void) { return 8; }
/* partial blocks a Gram buffer must hold: the kernels' grid cap + the shares and the ticket of k_finish_wide */
int pa_gram_max_blocks(void) { return GRAM_WIDE_BLOCKS + GRAM_SCRATCH_BLOCKS; }

void pa_k_note_seq(double seq) { g_note_seq = seq; }

int pa_finish32_scratch_blocks(void) { return FIN32_WG + 1; }     /* the shares + the block that holds the ticket */

int pa_k_finish32(const double* partials, int nblk, double* scratch, int t, int T, double* out, double* mu,
                  double* alpha, int* info) {
  PA_LAUNCH(k_finish32, dim3(FIN32_WG), dim3(WG), 0, cur_stream(), partials, nblk, scratch, t, T, out, mu,
            alpha, info, (const double*)nullptr, 0, 0, 0, (double*)nullptr);
  return kfail("k_finish32");
}

int pa_k_finish32_trace(const double* partials, int nblk, double* scratch, double* out, const double* rtr_partials,
                        int rtr_nblk, int ts, int nc, double* res2, int* info) {
  PA_LAUNCH(k_finish32, dim3(FIN32_WG), dim3(WG), 0, cur_stream(), partials, nblk, scratch, 0, 0, out,
            (double*)nullptr, (double*)nullptr, info, rtr_partials, rtr_nblk, ts, nc, res2);
  return kfail("k_finish32");
}

/* One wavefront that does nothing for `us` microseconds (phase timers, context.c: the launch of a timed region is
 * already queued when the spacer ends, so the event pair around it measures the kernels and not the dispatch
 * latency of a launch onto an idle stream). */
__global__ void k_spacer(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
int pa_k_spacer(int us) {
  PA_LAUNCH(k_spacer, dim3(1), dim3(64), 0, cur_stream(), (long long)us * 100);     /* (100 MHz counter) */
  return kfail("k_spacer");
}

int pa_k_probe(int which, size_t bytes, const double* src, double* dst) {
  const size_t n2 = bytes / 16;
  if (which == 0) PA_LAUNCH(k_probe_copy, dim3(8192), dim3(WG), 0, cur_stream(), n2, (const double2*)src, (double2*)dst);
  else PA_LAUNCH(k_probe_read, dim3(8192), dim3(WG), 0, cur_stream(), n2, (const double2*)src, dst);
  return kfail("k_probe");
}

int pa_k_gram(int m, int ts, const double* A0, const double* A1, const double* B, double* partials,
              int* nblk) {
  int blocks = grid_rows(m, 4);
  // (8 columns: up to 1024 workgroups -- four wavefronts per SIMD keep more of the three panel streams in flight:
  // 42.2 -> 38.2 us, the sum of the 1024 partial blocks +1.9 us.  k_gram<4, 2> measured best at 512 in round 2;
  // 16 columns lose with 1024: 69.6 -> 72.5 us and +5 us in the sum.)
  const int cap = ts == 8 ? GRAM_WIDE_BLOCKS : GRAM_MAX_BLOCKS;
  if (blocks > cap) blocks = cap;
  *nblk = blocks;
  if (ts == 16) {   // matrix cores (k_gram_mfma16)
    if (A1) PA_LAUNCH((k_gram_mfma16<2>), dim3(blocks), dim3(WG), 0, cur_stream(), m, A0, A1, B, partials);
    else PA_LAUNCH((k_gram_mfma16<1>), dim3(blocks), dim3(WG), 0, cur_stream(), m, A0, A1, B, partials);
    return kfail("k_gram_mfma16");
  }
  if (ts == 8) {
    if (A1) PA_LAUNCH((k_gram_mfma8<2>), dim3(blocks), dim3(WG), 0, cur_stream(), m, A0, A1, B, partials);
    else PA_LAUNCH((k_gram_mfma8<1>), dim3(blocks), dim3(WG), 0, cur_stream(), m, A0, A1, B, partials);
    return kfail("k_gram_mfma8");
  }
  if (A1) {
    TS_DISPATCH(ts, PA_LAUNCH((k_gram<TS_, 2>), dim3(blocks), dim3(WG), 0, cur_stream(), m,
                                       A0, A1, B, partials));
  } else {
    TS_DISPATCH(ts, PA_LAUNCH((k_gram<TS_, 1>), dim3(blocks), dim3(WG), 0, cur_stream(), m,
                                       A0, A1, B, partials));
  }
  return kfail("k_gram");
}

/* The sum of wide partial blocks by FINW_WG workgroups (k_finish_wide); the shares and the ticket lie behind the
 * GRAM_WIDE_BLOCKS partial blocks of the buffer (pa_gram_max_blocks() counts them in). */
static int finish_wide(const double* partials, int nblk, int npan, int ts, int a_lo, int a_hi, int nb, double* out,
                       int ld_out, int t, int T, double* mu, double* alpha, int* info, const double* rtr, int rtr_nblk,
                       int rtr_nc, double* res2) {
  double* scratch = const_cast<double*>(partials) + (size_t)GRAM_WIDE_BLOCKS * 2 * ts * ts;
  PA_LAUNCH(k_finish_wide, dim3(FINW_WG), dim3(WG), 0, cur_stream(), partials, nblk, npan, ts, a_lo, a_hi, nb, out, ld_out,
            scratch, t, T, mu, alpha, info, rtr, rtr_nblk, rtr_nc, res2);
  return kfail("k_finish_wide");
}

int pa_k_gram_finish(int m, int ts, const double* A0, const double* A1, const double* B,
                     double* partials, int a_lo, int a_hi, int nb, double* out, int ld_out, int t,
                     int T, double* mu, double* alpha, int* info) {
  int nblk = 0;
  if ((a_lo + a_hi) * nb <= 0) return 0;
  if (pa_k_gram(m, ts, A0, A1, B, partials, &nblk)) return 1;
  if (ts >= 8) {
    if (t > 0 && (a_lo != t || a_hi != T || nb != t || ld_out != t + T || !A1)) {
      snprintf(g_kerr, sizeof(g_kerr), "pa_k_gram_finish: [W ; G^T] layout expected");
      return 1;
    }
    return finish_wide(partials, nblk, A1 ? 2 : 1, ts, a_lo, a_hi, nb, out, ld_out, t, T, mu, alpha, info, nullptr, 0, 0, nullptr);
  }
  if (t > 0) {
    if (a_lo != t || a_hi != T || nb != t || ld_out != t + T || !A1) {
      snprintf(g_kerr, sizeof(g_kerr), "pa_k_gram_finish: [W ; G^T] layout expected");
      return 1;
    }
    if ((t + T) * t > 128) {   // large block: spread the sum, then factor
      if (pa_k_finish(partials, nblk, 2, ts, t, T, t, out, t + T)) return 1;
      return pa_k_potrf_alpha(out, t, T, mu, alpha, info);
    }
    PA_LAUNCH(k_finish_potrf_alpha, dim3(1), dim3(1024), 0, cur_stream(), partials, nblk, 2, ts,
                       t, T, out, mu, alpha, info);
    return kfail("k_finish_potrf_alpha");
  }
  return pa_k_finish(partials, nblk, A1 ? 2 : 1, ts, a_lo, a_hi, nb, out, ld_out);
}

int pa_k_gram_finish_trace(int m, int ts, const double* A0, const double* A1, const double* B, double* partials,
                           int a_lo, int a_hi, int nb, double* out, int ld_out, const double* rtr_partials,
                           int rtr_nblk, int nc, double* res2, const int* info) {
  int nblk = 0;
  const int ne = (a_lo + a_hi) * nb;
  if (ne > 0 && ts >= 8) {     /* wide blocks: several workgroups sum, the last one adds the norm */
    if (pa_k_gram(m, ts, A0, A1, B, partials, &nblk)) return 1;
    return finish_wide(partials, nblk, A1 ? 2 : 1, ts, a_lo, a_hi, nb, out, ld_out, 0, 0, nullptr, nullptr, const_cast<int*>(info),
                       rtr_partials, rtr_nblk, nc, res2);
  }
  if (ne <= 0 || ne > 128)     /* no Gram block, or one that several workgroups sum: the two launches */
    return pa_k_trace_finish(rtr_partials, rtr_nblk, ts, nc, res2, info, NULL) ||
           pa_k_gram_finish(m, ts, A0, A1, B, partials, a_lo, a_hi, nb, out, ld_out, 0, 0, NULL, NULL, NULL);
  if (pa_k_gram(m, ts, A0, A1, B, partials, &nblk)) return 1;
  PA_LAUNCH(k_finish_trace, dim3(1), dim3(1024), 0, cur_stream(), partials, nblk, A1 ? 2 : 1, ts, a_lo, a_hi, nb,
            out, ld_out, rtr_partials, rtr_nblk, nc, res2, info);
  return kfail("k_finish_trace");
}

int pa_k_finish(const double* partials, int nblk, int npan, int ts, int a_lo, int a_hi, int nb,
                double* out, int ld_out) {
  const int ne = (a_lo + a_hi) * nb;
  if (ne <= 0) return 0;
  // wide blocks in the Gram buffer of a solver (pa_gram_max_blocks() blocks: the shares and the ticket of
  // k_finish_wide lie behind them) -- BF-Omin's Z^T Z, up to 2048 blocks from the update kernel
  if (ts >= 8 && (long long)nblk * npan * ts * ts <= (long long)GRAM_WIDE_BLOCKS * 2 * ts * ts)
    return finish_wide(partials, nblk, npan, ts, a_lo, a_hi, nb, out, ld_out, 0, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr);
  const int groups = ne > 128 ? (ne + 63) / 64 : 1;    // one workgroup unless the block is large
  PA_LAUNCH(k_finish, dim3(groups), dim3(1024), 0, cur_stream(), partials, nblk, npan, ts, a_lo,
                     a_hi, nb, out, ld_out);
  return kfail("k_finish");
}

int pa_k_potrf(double* W, int t, int* info) {
  PA_LAUNCH(k_potrf, dim3(1), dim3(64), 0, cur_stream(), W, t, info);
  return kfail("k_potrf");
}

int pa_k_fused_small(double* mu, int t, int nrhs, int bm, int bn, int ldb, double* alpha,
                     double* beta, int* info) {
  PA_LAUNCH(k_fused_small, dim3(1), dim3(64), 0, cur_stream(), mu, t, nrhs, bm, bn, ldb,
                     alpha, beta, info);
  return kfail("k_fused_small");
}

int pa_k_trsm(int m, int ts, int t, const double* U, double* P, double* AP) {
  if (t <= 0) return 0;
  TS_DISPATCH(ts, PA_LAUNCH((k_trsm<TS_>), dim3(grid_rows(m)), dim3(WG), 0, cur_stream(),
                                     m, t, U, P, AP));
  return kfail("k_trsm");
}

int pa_k_update_xr(int m, int ts, int t, int nc, const double* alpha, const double* P,
                   const double* AP, double* X, double* R, double* rtr_partials, int* nblk,
                   int trace_nc, double* res2, const int* info, double* host) {
  int blocks = grid_rows(m, 2);
  if (blocks > GRAM_MAX_BLOCKS) blocks = GRAM_MAX_BLOCKS;
  *nblk = blocks;
  TS_DISPATCH(ts, PA_LAUNCH((k_update_xr<TS_>), dim3(blocks), dim3(WG), 0, cur_stream(), m,
                                     t, nc, alpha, P, AP, X, R, rtr_partials));
  if (kfail("k_update_xr")) return 1;
  if (trace_nc <= 0) return 0;
  PA_LAUNCH(k_trace_finish, dim3(1), dim3(WG), 0, cur_stream(), rtr_partials, blocks, ts, trace_nc,
                     res2, info, host, take_note_seq(host));
  return kfail("k_trace_finish");
}

int pa_k_potrf_alpha(const double* buf, int t, int T, double* mu, double* alpha, int* info) {
  PA_LAUNCH(k_potrf_alpha, dim3(1), dim3(64), 0, cur_stream(), buf, t, T, mu, alpha, info);
  return kfail("k_potrf_alpha");
}

int pa_k_trsm_update(int m, int ts, int t, int nc, double* U, double* alpha, double* P,
                     double* AP, double* X, double* R, double* rtr_partials, int* nblk, int trace_nc,
                     double* res2, int* info, double* host, const double* gram, double* ukeep) {
  int blocks = grid_rows(m, 2);
  if (blocks > GRAM_MAX_BLOCKS) blocks = GRAM_MAX_BLOCKS;
  *nblk = blocks;
  // PREALPS_TRSM_MFMA=0: lane-per-row substitution at every width (the matrix-core variant forms U^-1)
  static int use_mfma = -1;
  if (use_mfma < 0) { const char* e = getenv("PREALPS_TRSM_MFMA"); use_mfma = e ? atoi(e) : 1; }
  if (ts == 16 && use_mfma)
    PA_LAUNCH((k_trsm_update_mfma<16>), dim3(blocks), dim3(WG), 0, cur_stream(), m, t, nc, U, alpha, P, AP, X, R, rtr_partials, gram, info, ukeep);
  else if (ts == 8 && use_mfma)
    PA_LAUNCH((k_trsm_update_mfma<8>), dim3(blocks), dim3(WG), 0, cur_stream(), m, t, nc, U, alpha, P, AP, X, R, rtr_partials, gram, info, ukeep);
  else {
    TS_DISPATCH(ts, PA_LAUNCH((k_trsm_update<TS_>), dim3(blocks), dim3(WG), 0, cur_stream(), m,
                                       t, nc, U, alpha, P, AP, X, R, rtr_partials, gram, info, ukeep,
                                       (size_t)m * ts * sizeof(double) >= ((size_t)16 << 20) ? 1 : 0));
  }
  if (kfail("k_trsm_update")) return 1;
  if (trace_nc <= 0) return 0;
  PA_LAUNCH(k_trace_finish, dim3(1), dim3(WG), 0, cur_stream(), rtr_partials, blocks, ts, trace_nc,
                     res2, (const int*)info, host, take_note_seq(host));
  return kfail("k_trace_finish");
}

int pa_k_colnorm2(int m, int ts, const double* R, double* rtr_partials, int* nblk) {
  int blocks = grid_rows(m, 4);
  if (blocks > GRAM_MAX_BLOCKS) blocks = GRAM_MAX_BLOCKS;
  *nblk = blocks;
  TS_DISPATCH(ts, PA_LAUNCH((k_colnorm2<TS_>), dim3(blocks), dim3(WG), 0, cur_stream(), m,
                                     R, rtr_partials));
  return kfail("k_colnorm2");
}

int pa_k_trace_finish(const double* rtr_partials, int nblk, int ts, int nc, double* res2,
                      const int* info, double* host) {
  PA_LAUNCH(k_trace_finish, dim3(1), dim3(WG), 0, cur_stream(), rtr_partials, nblk, ts, nc,
                     res2, info, host, take_note_seq(host));
  return kfail("k_trace_finish");
}

/* One-shot: the next pa_k_update_z on a panel of up to 4 columns also packs the send rows of the Z it writes
 * (pa_operator_pack_hint).  Returns 0 when that launch would not take it (wider panels). */
static struct { const int* off; const int* slot; double* buf; } g_zpack;
int pa_k_update_z_pack(int ts, const int* pk_off, const int* pk_slot, double* sendbuf) {
  g_zpack.off = g_zpack.slot = nullptr; g_zpack.buf = nullptr;
  if (ts > 4 || !pk_off || !pk_slot || !sendbuf) return 0;
  g_zpack.off = pk_off; g_zpack.slot = pk_slot; g_zpack.buf = sendbuf;
  return 1;
}

int pa_k_update_z(int m, int ts, int a_lo, int a_hi, int nc, const double* beta, int ldb,
                  const double* V0, const double* V1, double* Z, const double* note_src, double* note_host,
                  const double* ucur, const double* uprev, double* zz_part, int zz_cols, int* zz_nblk) {
  const auto pk = g_zpack;
  g_zpack.off = g_zpack.slot = nullptr; g_zpack.buf = nullptr;
  if (zz_nblk) *zz_nblk = 0;
  if (nc <= 0) return 0;
  if (ucur && (!uprev || nc != a_lo || (a_hi != 0 && a_hi != a_lo))) {
    snprintf(g_kerr, sizeof(g_kerr), "pa_k_update_z: lazy normalisation needs square blocks");
    return 1;
  }
  const double seq_ = take_note_seq(note_host);
  // zz_part (8 / 16 columns, panels normalised in place): the launch also leaves Z^T Z of the new Z behind, one
  // ts x ts block per workgroup (*zz_nblk of them, the partial blocks pa_k_finish sums); elsewhere *zz_nblk stays 0
  double* zzp = (zz_part && zz_nblk && !ucur && (ts == 8 || ts == 16)) ? zz_part : nullptr;
  if (ts == 16) {   // matrix cores (k_update_z_mfma16), one 16-row tile per wavefront and step
    const int grid = grid_rows(m, 4);
    PA_LAUNCH(k_update_z_mfma16, dim3(grid), dim3(WG), 0, cur_stream(), m, a_lo, a_hi, nc,
                       beta, ldb, V0, V1, Z, note_src, note_host, seq_, ucur, uprev, zzp, zz_cols);
    if (zzp) *zz_nblk = grid;
    return kfail("k_update_z_mfma16");
  }
  if (ts == 8) {
    const int grid = grid_rows(m, 4);
    PA_LAUNCH(k_update_z_mfma8, dim3(grid), dim3(WG), 0, cur_stream(), m, a_lo, a_hi, nc,
                       beta, ldb, V0, V1, Z, note_src, note_host, seq_, ucur, uprev, zzp, zz_cols);
    if (zzp) *zz_nblk = grid;
    return kfail("k_update_z_mfma8");
  }
  TS_DISPATCH(ts, PA_LAUNCH((k_update_z<TS_>), dim3(grid_rows(m, 2)), dim3(WG), 0,
                                     cur_stream(), m, a_lo, a_hi, nc, beta, ldb, V0, V1, Z, note_src, note_host, seq_, ucur, uprev,
                                     pk.off, pk.slot, pk.buf));
  return kfail("k_update_z");
}

int pa_k_copy_cols(int m, int ts, int nc, const double* src, double* dst) {
  if (nc <= 0) return 0;
  TS_DISPATCH(ts, PA_LAUNCH((k_copy_cols<TS_>), dim3(grid_rows(m, 2)), dim3(WG), 0,
                                     cur_stream(), m, nc, src, dst));
  return kfail("k_copy_cols");
}

int pa_k_right_mult(int m, int ts, int t, const double* Q, double* A) {
  if (t <= 0) return 0;
  TS_DISPATCH(ts, PA_LAUNCH((k_right_mult<TS_>), dim3(grid_rows(m, 2)), dim3(WG), 0,
                                     cur_stream(), m, t, Q, A));
  return kfail("k_right_mult");
}

int pa_k_permute_trsm(int m, int ts, int n, const int* piv, int t, const double* U, const double* src, double* dst) {
  if (m <= 0) return 0;
  TS_DISPATCH(ts, PA_LAUNCH((k_permute_trsm<TS_>), dim3(grid_rows(m, 2)), dim3(WG), 0, cur_stream(), m, n, piv, t, U,
                            src, dst));
  return kfail("k_permute_trsm");
}

int pa_k_permute_cols(int m, int ts, int n, const int* piv, double* A) {
  if (n <= 0) return 0;
  TS_DISPATCH(ts, PA_LAUNCH((k_permute_cols<TS_>), dim3(grid_rows(m, 2)), dim3(WG), 0,
                                     cur_stream(), m, n, piv, A));
  return kfail("k_permute_cols");
}

int pa_k_rowsum(int m, int ts, int nc, const double* X, double* sol) {
  TS_DISPATCH(ts, PA_LAUNCH((k_rowsum<TS_>), dim3(grid_rows(m, 2)), dim3(WG), 0,
                                     cur_stream(), m, nc, X, sol));
  return kfail("k_rowsum");
}

int pa_bj_factor_wmax(void) { return 96; }

int pa_k_bj_factor(const int* list, int count, int wmax, const int* row0, const int* nrows, const int* bw,
                   const long long* off, const long long* boff, const double* band, double* Lf, double* Lb,
                   double* invd_f, double* invd_b, int* fail) {
  if (count <= 0) return 0;
  const size_t lds = (size_t)(wmax + 1) * (wmax + 1) * 8;
  static size_t configured = 0;
  if (lds > 64 * 1024 && lds > configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bj_factor),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return kfail("hipFuncSetAttribute(k_bj_factor)");
    configured = lds;
  }
  PA_LAUNCH(k_bj_factor, dim3(count), dim3(WG), lds, cur_stream(), list, row0, nrows, bw, off, boff,
                     band, Lf, Lb, invd_f, invd_b, fail);
  return kfail("k_bj_factor");
}

int pa_k_scatter(size_t n, const long long* off, const double* val, double* dst) {
  if (n == 0) return 0;
  size_t blocks = (n + WG - 1) / WG;
  if (blocks > 65535) blocks = 65535;
  PA_LAUNCH(k_scatter, dim3((unsigned)blocks), dim3(WG), 0, cur_stream(), n, off, val, dst);
  return kfail("k_scatter");
}

int pa_k_bj_factor_big(const int* list, int count, int wmax, int wide_from, const int* row0, const int* nrows,
                       const int* bw, const long long* off, const long long* boff, double* band, double* Lf, double* Lb,
                       double* invd_f, double* invd_b, int* fail) {
  if (count <= 0) return 0;
  int rc;
  if (wmax <= 1024) rc = bj_factor_big_launch<16>(list, count, wmax, row0, nrows, bw, boff, band, fail);
  else if (wmax <= 2048) rc = bj_factor_big_launch<8>(list, count, wmax, row0, nrows, bw, boff, band, fail);
  else rc = bj_factor_big_launch<4>(list, count, wmax, row0, nrows, bw, boff, band, fail);
  if (rc) return rc;
  PA_LAUNCH(k_bj_layout_big, dim3(512, count), dim3(WG), 0, cur_stream(), list, row0, nrows, bw, off,
                     boff, band, wide_from, Lf, Lb, invd_f, invd_b);
  return kfail("k_bj_layout_big");
}

int pa_nd_chunk_rows(void) { return ND_CHUNK; }
int pa_nd_block_cols(void) { return ND_COLS; }

// levels are listed bottom-up: forward in that order, backward reversed
int pa_k_nd_apply(const pa_nd_plan_t* pl, int ts, const double* in, double* out) {
  nd_args a{pl->n, pl->m, pl->ld, pl->offF, pl->offB, pl->rows_off, pl->coff, pl->ccoff, pl->rows, pl->src,
            pl->dinv, pl->F, pl->B, pl->contrib, pl->Y};
  for (int i = 0; i < pl->nlevel; ++i) {
    int rc = 0;
    TS_DISPATCH(ts, rc = nd_launch_fwd<TS_>(a, pl->f_front[i], pl->f_row0[i], pl->f_count[i], in));
    if (rc) return rc;
  }
  for (int i = pl->nlevel - 1; i >= 0; --i) {
    int rc = 0;
    TS_DISPATCH(ts, rc = nd_launch_bwd<TS_>(a, pl->b_front[i], pl->b_col0[i], pl->b_count[i], out));
    if (rc) return rc;
  }
  return 0;
}

int pa_k_bj_pairs(const int* list, int count, const int* nrows, const int* bw, const long long* off,
                  const long long* off2, const double* L, double* L2) {
  if (count <= 0) return 0;
  PA_LAUNCH(k_bj_pairs, dim3(count), dim3(WG), 0, cur_stream(), list, nrows, bw, off, off2, L, L2);
  return kfail("k_bj_pairs");
}

/* A Gram block requested from the next block solve in -> out (pa_k_bj_gram_arm), as g_sg for the SpMM */
static struct { const double* in; const double* out; const double* prev; double* partials; int cap, count, armed; } g_bg;

void pa_k_bj_gram_arm(const double* in, const double* out, const double* prev, double* partials, int cap) {
  g_bg.in = in; g_bg.out = out; g_bg.prev = prev; g_bg.partials = partials; g_bg.cap = cap; g_bg.count = 0;
  g_bg.armed = (in && out && prev && partials && cap > 0);
}
static long long g_bg_applies = 0;
long long pa_k_bj_gram_applies(void) { return g_bg_applies; }
void pa_k_bj_gram_disarm(const double* owner) { if (!owner || owner == g_bg.partials) { g_bg.armed = 0; g_bg.count = 0; } }
int pa_k_bj_gram_take(const double* in, const double* out) {
  if (!g_bg.armed || in != g_bg.in || out != g_bg.out) return 0;
  const int n = g_bg.count;
  g_bg.armed = 0; g_bg.count = 0;
  return n;
}

int pa_k_bj_apply(const pa_bj_plan_t* pl, int ts, const double* in, double* out) {
  for (int c = 0; c < pl->nclass; ++c) {
    if (pl->class_count[c] <= 0) continue;
    int rc = 1;
    static int g4_wide = -1;      /* PREALPS_BJ_G4_WIDE=0: 8-column panels stay with k_bj_mfma */
    if (g4_wide < 0) { const char* e = getenv("PREALPS_BJ_G4_WIDE"); g4_wide = e ? atoi(e) : 1; }
    if (pl->Lg4 && pl->class_g4[c] && (ts <= 4 || (ts == 8 && g4_wide && pl->class_wmax[c] <= pa_bj_g4_max_band8()))) {   /* one copy of the factor, matrix cores (bj_g4.hip) */
      if (ts == 4 && g_bg.armed && pl->nclass == 1 && in == g_bg.in && out == g_bg.out && pl->class_count[c] <= g_bg.cap) {
        pa_k_bj_g4_gram(g_bg.prev, g_bg.partials);      // this apply also leaves [in | prev]^T out (pa_k_bj_gram_arm)
        g_bg.count = pl->class_count[c];
        ++g_bg_applies;
      } else if (g_bg.armed && in == g_bg.in && out == g_bg.out) {
        g_bg.count = 0;
      }
      rc = pa_k_bj_g4(pl, pl->class_list[c], pl->class_count[c], pl->class_wmax[c], pl->class_bmax[c], ts, ts, in, out);
      if (rc) return rc;
      continue;
    }
    if (pl->class_R[c] < 0) {   /* wide bands: one workgroup per subdomain, -class_R register sets */
      const int Rw = -pl->class_R[c];
      if (Rw == 1) rc = bj_wide_dispatch<1>(pl, ts, pl->class_wmax[c], pl->class_list[c], pl->class_count[c], in, out);
      else if (Rw == 2) rc = bj_wide_dispatch<2>(pl, ts, pl->class_wmax[c], pl->class_list[c], pl->class_count[c], in, out);
      else rc = bj_wide_dispatch<4>(pl, ts, pl->class_wmax[c], pl->class_list[c], pl->class_count[c], in, out);
      if (rc) return rc;
      continue;
    }
    TS_DISPATCH(ts, rc = bj_launch<TS_>(pl, pl->class_R[c], pl->class_wmax[c], pl->class_list[c],
                                        pl->class_count[c], in, out));
    if (rc) return rc;
  }
  return 0;
}

}  // extern "C"
