// spmm.hip -- the block operator AP = A P (utils/operator.c:334-351 -> cplm_v0_matmult_v2.c:108-343 ->
// mkl_dcsrmm of the reference): SELL-64 CSR x panel kernels, the row pack of the halo exchange, launchers.
#include "kernels_common.h"

namespace {

// The Gram epilogue of one slice (see k_spmm_runs_gram): acc = the lane's row of Y, xrow(r) = where row r of the
// slice's X lies (LDS or memory), R rows from memory; gw / gg = the wavefront's two accumulators.
// the lane's four entries of R for spmm_gram_slice: step st = row g4 + st of the slice, column c
__device__ __forceinline__ void spmm_gram_load_r(double (&rv)[4], int nr, int row_s, int lane, const double* __restrict__ Rg) {
  const int g4 = lane & ~3, c = lane & 3;
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const int rr = min(g4 + st, nr - 1);          // (rows beyond the slice: clamped here, masked in the slice step)
    rv[st] = Rg[(size_t)(row_s + rr) * 4 + c];
  }
}
template <typename XROW>
__device__ __forceinline__ void spmm_gram_slice(const double (&acc)[4], int nr, int row_s, int lane,
                                                const double (&rin)[4], XROW xrow, double& gw, double& gg) {
  const int g4 = lane & ~3, c = lane & 3;
  double rv[4], xv[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) {         // step st: the quad stands for row g4 + st of the slice
    const int rr = g4 + st;
    const bool on = rr < nr;
    rv[st] = on ? rin[st] : 0.0;
    xv[st] = on ? xrow(rr)[c] : 0.0;
  }
  double ty[4] = {acc[0], acc[1], acc[2], acc[3]};
  quad_transpose4(ty, c);
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    gw = __builtin_amdgcn_mfma_f64_4x4x4f64(ty[st], xv[st], gw, 0, 0, 0);
    gg = __builtin_amdgcn_mfma_f64_4x4x4f64(rv[st], xv[st], gg, 0, 0, 0);
  }
}

// ... and of the workgroup: the four blocks of a lane's row of 16 lanes (row_ror 4 / 8), then the four
// wavefronts through the first 128 doubles of the staging area (no wavefront reads it any more after the
// barrier), one 8 x 4 partial block out.
__device__ __forceinline__ void spmm_gram_tail(double gw, double gg, double* sx, int wave, int lane, int tid,
                                               double* __restrict__ gblock) {
  gw += dpp_mov_f64<0x124>(gw); gw += dpp_mov_f64<0x128>(gw);
  gg += dpp_mov_f64<0x124>(gg); gg += dpp_mov_f64<0x128>(gg);
  __syncthreads();
  if ((lane & 12) == 0) {
    const int i = lane >> 4, j = lane & 3;
    sx[wave * 32 + i + 8 * j] = gw;
    sx[wave * 32 + 4 + i + 8 * j] = gg;
  }
  __syncthreads();
  if (tid < 32) gblock[tid] = ((sx[tid] + sx[32 + tid]) + sx[64 + tid]) + sx[96 + tid];
}

// ---------------------------------------------------------------- SpMM ----
// SELL-64 SpMM.  One workgroup per block of slices of one subdomain; the
// subdomain's own X rows (where ~90 % of the nonzeros of a box partition
// point) are staged once into LDS with coalesced 16-B loads, every wavefront
// then walks whole slices: lane r owns row r of the slice, keeps its TS sums
// in registers and reads val/col of entry k at [off + k*64 + r] -- one fully
// coalesced 512-B / 256-B load per wave instruction for the 12 B/nonzero
// stream.  Columns outside the window (neighbour subdomains, halo rows) are
// gathered as whole 8*TS-byte rows from L2.
template <int TS>
__device__ __forceinline__ void spmm_fma_row(double (&acc)[TS], double v, const double* __restrict__ xr) {
  const double2* q = reinterpret_cast<const double2*>(xr);
#pragma unroll
  for (int i = 0; i < TS / 2; ++i) {
    const double2 x = q[i];
    acc[2 * i] = fma(v, x.x, acc[2 * i]);
    acc[2 * i + 1] = fma(v, x.y, acc[2 * i + 1]);
  }
}

template <int TS, bool GRAM>
__device__ __forceinline__ void spmm_body(
    int m, const long long* __restrict__ sl_off, const int* __restrict__ sl_len,
    const int* __restrict__ sl_row0, const int* __restrict__ sl_nrows,
    const int* __restrict__ col, const double* __restrict__ val,
    const int* __restrict__ blk_slice, const int* __restrict__ blk_win,
    const int* __restrict__ order, int nlist, int win_cap, const double* __restrict__ X,
    const double* __restrict__ Xh, double* __restrict__ Y,
    const double* __restrict__ Rg, double* __restrict__ gpart, int gbase) {
  static_assert(!GRAM || TS == 4, "the fused Gram block is built for 4-column panels");
  extern __shared__ double sx[];
  // XCD-aware order: consecutive logical blocks (which share X rows) run on one XCD.
  const int cpx = (nlist + 7) >> 3;
  const int logical = (blockIdx.x & 7) * cpx + (blockIdx.x >> 3);
  if (logical >= nlist) return;
  const int b = order[logical];
  const int s0 = blk_slice[b], s1 = blk_slice[b + 1];
  const int w0 = blk_win[2 * b];
  const int wlen = min(blk_win[2 * b + 1] - w0, win_cap);
  const int tid = threadIdx.x;
  {
    const double2* xsrc = reinterpret_cast<const double2*>(X + (size_t)w0 * TS);
    const int nx2 = (wlen * TS) >> 1;
    for (int i = tid; i < nx2; i += WG) reinterpret_cast<double2*>(sx)[i] = xsrc[i];
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  double gw = 0.0, gg = 0.0;      // GRAM: as in k_spmm_runs_gram
  for (int s = s0 + wave; s < s1; s += WG / 64) {
    const long long off = sl_off[s];
    const int len = sl_len[s];
    const int* __restrict__ cp = col + off + lane;
    const double* __restrict__ vp = val + off + lane;
    double acc[TS];
#pragma unroll
    for (int c = 0; c < TS; ++c) acc[c] = 0.0;
#pragma unroll 4
    for (int k = 0; k < len; ++k) {
      // (nontemporal loads of the matrix stream measured 30 % slower in round 1: the stream is partly served
      // by the Infinity Cache)
      const double v = vp[(size_t)k * 64];
      const int cidx = cp[(size_t)k * 64];
      const unsigned wi = (unsigned)(cidx - w0);
      if (wi < (unsigned)wlen) spmm_fma_row<TS>(acc, v, sx + (size_t)wi * TS);
      else if (cidx < m) spmm_fma_row<TS>(acc, v, X + (size_t)cidx * TS);
      else spmm_fma_row<TS>(acc, v, Xh + (size_t)(cidx - m) * TS);
    }
    const int nr = sl_nrows[s], row_s = sl_row0[s];
    if (lane < nr) store_row<TS>(Y, (size_t)(row_s + lane), acc);
    if constexpr (GRAM) {
      // a row of the slice's own X: in the window, or (a window cut short by win_cap) in memory
      double rin[4];
      spmm_gram_load_r(rin, nr, row_s, lane, Rg);
      spmm_gram_slice(acc, nr, row_s, lane, rin, [&](int rr) {
        const unsigned wi = (unsigned)(row_s + rr - w0);
        return wi < (unsigned)wlen ? (const double*)(sx + (size_t)wi * 4) : X + (size_t)(row_s + rr) * 4;
      }, gw, gg);
    }
  }
  if constexpr (GRAM) spmm_gram_tail(gw, gg, sx, wave, lane, tid, gpart + (size_t)(gbase + logical) * 32);
}

template <int TS>
__global__ __launch_bounds__(WG) void k_spmm(
    int m, const long long* __restrict__ sl_off, const int* __restrict__ sl_len,
    const int* __restrict__ sl_row0, const int* __restrict__ sl_nrows,
    const int* __restrict__ col, const double* __restrict__ val,
    const int* __restrict__ blk_slice, const int* __restrict__ blk_win,
    const int* __restrict__ order, int nlist, int win_cap, const double* __restrict__ X,
    const double* __restrict__ Xh, double* __restrict__ Y) {
  spmm_body<TS, false>(m, sl_off, sl_len, sl_row0, sl_nrows, col, val, blk_slice, blk_win, order, nlist, win_cap,
                           X, Xh, Y, nullptr, nullptr, 0);
}

// 4 columns with the Gram block [Y | R]^T X (see k_spmm_runs_gram)
__global__ __launch_bounds__(WG) void k_spmm_gram(
    int m, const long long* __restrict__ sl_off, const int* __restrict__ sl_len,
    const int* __restrict__ sl_row0, const int* __restrict__ sl_nrows,
    const int* __restrict__ col, const double* __restrict__ val,
    const int* __restrict__ blk_slice, const int* __restrict__ blk_win,
    const int* __restrict__ order, int nlist, int win_cap, const double* __restrict__ X,
    const double* __restrict__ Xh, double* __restrict__ Y,
    const double* __restrict__ Rg, double* __restrict__ gpart, int gbase) {
  spmm_body<4, true>(m, sl_off, sl_len, sl_row0, sl_nrows, col, val, blk_slice, blk_win, order, nlist, win_cap,
                            X, Xh, Y, Rg, gpart, gbase);
}

// Staged SpMM over runs of three consecutive LDS slots (rows whose nonzeros sit in groups
// of neighbouring columns, e.g. the 3 dofs of a node): per run one coalesced 2-B slot, three
// coalesced 8-B values and 3*TS/2 ds_read_b128 off one address -- 8.67 B of matrix stream
// per nonzero instead of 10 and a third of the index arithmetic.  The staging area is
// [external rows below | own rows | external rows above | two zero rows].
// XS = panel stride in doubles.  XS = 2 TS splits a wide panel by columns: two workgroups per
// block (neighbours in the dispatch order of one XCD, so the second one finds the matrix slice
// in that XCD's L2), each staging and computing TS of the XS columns.
//
// GRAM (4-column panels): the workgroup also leaves the block's share of [Y | R]^T X -- the
// Gram block the ECG iteration forms right after A P (ecg.c:425-436: W = AP^T P, G^T = R^T P)
// -- in gpart[gbase + logical] (8 x 4, column major, the layout of k_gram<4, 2>), so that the
// panels are not read a second time.  Y rows sit one per lane in registers and X rows in the
// staging area: a 4 x 4 transpose inside each quad of lanes puts 16 rows x 4 columns into the
// operand layout of v_mfma_f64_4x4x4 (lane 4g + c = column c of row g), four of which cover the
// 64 rows of a slice; R is read in that layout directly.
// One group of a slice's entries: RU(RL) slots and RL values per slot for every lane, all coalesced (RL = 3: runs of
// three consecutive staged rows; RL = 1: one staged row per slot -- the staged plan of matrices without such
// runs).  `g` is clamped by the caller, so that the loads are unconditional (a conditional load makes the
// compiler's wait-count bookkeeping join two histories at the next use, and the join waits for everything in
// flight); an entry beyond the slice's last one takes that one's slot and its values from a block of zeros -- the
// choice is between two scalar addresses, the vector code is the same for every group.
template <int RL> struct spmm_ru { static constexpr int n = RL == 3 ? 3 : 6; };     // 12 loads per group either way
// (1, 2 and 4 runs per group measured the same as 3: profiles/r04_spmm_variants_ab.txt)
__device__ const double g_zero_run[192] = {0.0};
template <int RL>
__device__ __forceinline__ void spmm_runs_load(const unsigned short* __restrict__ cp, const double* __restrict__ vp,
                                               unsigned lane, int g, int len, int (&sl)[spmm_ru<RL>::n],
                                               double (&vv)[RL * spmm_ru<RL>::n]) {
  constexpr int RU = spmm_ru<RL>::n;
  // cp / vp: the slice's slots and values, wavefront-uniform; the entry index is uniform too, so every address
  // is a scalar base plus the lane
#pragma unroll
  for (int j = 0; j < RU; ++j) {
    const int k = g * RU + j, kc = min(k, len - 1);
    const unsigned short* __restrict__ ck = cp + (size_t)kc * 64;
    typedef const __attribute__((address_space(1))) double* gdp;     // (a select of two pointers would go flat)
    const gdp vk = k < len ? (gdp)(vp + (size_t)(RL * k) * 64) : (gdp)g_zero_run;
    sl[j] = ck[lane];
#pragma unroll
    for (int i = 0; i < RL; ++i) vv[RL * j + i] = vk[64 * i + lane];
  }
}
// Row stride of the staging area in doubles.  8 columns: 64-byte rows lie on only four different bank quads of
// the 64-bank LDS, so the 16 lanes of a ds_read_b128 group, which gather up to 16 different rows, conflict up to
// four ways: SQ_LDS_BANK_CONFLICT counted 53.4 M of the kernel's 77.6 M LDS cycles (126 us of LDS time per CU in
// a 180 us kernel; the two 8-column halves of a 16-column product: 252 of 372 us), while the HBM counters showed
// 3.0-4.5 TB/s -- at 8 and 16 columns the product was bound by the LDS, not by memory.  Rows 80 bytes apart
// (r * 20 mod 64 visits all sixteen quads) read without conflicts.  4 columns keep 32-byte rows: the padded
// area would cost the fifth workgroup per CU, and there the kernel is bound by HBM.
template <int TS> struct spmm_row { static constexpr int stride = TS == 8 ? TS + 2 : TS; };

template <int TS, int RL>
__device__ __forceinline__ void spmm_runs_fma(const double* sx, const int (&sl)[spmm_ru<RL>::n],
                                              const double (&vv)[RL * spmm_ru<RL>::n], double (&acc)[TS]) {
  constexpr int TSP = spmm_row<TS>::stride;
#pragma unroll
  for (int j = 0; j < spmm_ru<RL>::n; ++j) {
    const double* __restrict__ xr = sx + (size_t)sl[j] * TSP;
#pragma unroll
    for (int i = 0; i < RL; ++i) spmm_fma_row<TS>(acc, vv[RL * j + i], xr + i * TSP);
  }
}

template <int TS, int XS, bool GRAM, int RL>
__device__ __forceinline__ void spmm_runs_block(
    int m, const long long* __restrict__ sl_off, const int* __restrict__ sl_len,
    const int* __restrict__ sl_row0, const int* __restrict__ sl_nrows,
    const unsigned short* __restrict__ slot16, const double* __restrict__ val,
    const int* __restrict__ blk_slice, const int* __restrict__ blk_ext_off,
    const int* __restrict__ blk_nlow, const int* __restrict__ ext_rows,
    const int* __restrict__ order, const double* __restrict__ X,
    const double* __restrict__ Xh, double* __restrict__ Y,
    const double* __restrict__ Rg, double* __restrict__ gpart, int gbase, int logical, int coff);

template <int TS, int XS, bool GRAM, int RL = 3>
__device__ __forceinline__ void spmm_runs_body(
    int m, const long long* __restrict__ sl_off, const int* __restrict__ sl_len,
    const int* __restrict__ sl_row0, const int* __restrict__ sl_nrows,
    const unsigned short* __restrict__ slot16, const double* __restrict__ val,
    const int* __restrict__ blk_slice, const int* __restrict__ blk_ext_off,
    const int* __restrict__ blk_nlow, const int* __restrict__ ext_rows,
    const int* __restrict__ order, int nlist, const double* __restrict__ X,
    const double* __restrict__ Xh, double* __restrict__ Y,
    const double* __restrict__ Rg, double* __restrict__ gpart, int gbase) {
  constexpr int NS = XS / TS;
  const int cpx = (nlist + 7) >> 3;
  const int idx = blockIdx.x >> 3;
  const int logical = (blockIdx.x & 7) * cpx + idx / NS;
  if (logical >= nlist) return;
  spmm_runs_block<TS, XS, GRAM, RL>(m, sl_off, sl_len, sl_row0, sl_nrows, slot16, val, blk_slice, blk_ext_off, blk_nlow,
                                         ext_rows, order, X, Xh, Y, Rg, gpart, gbase, logical, (idx % NS) * TS);
}

// One block of slices: `logical` = its place in the launch's list, `coff` = first of the TS columns.
template <int TS, int XS, bool GRAM, int RL>
__device__ __forceinline__ void spmm_runs_block(
    int m, const long long* __restrict__ sl_off, const int* __restrict__ sl_len,
    const int* __restrict__ sl_row0, const int* __restrict__ sl_nrows,
    const unsigned short* __restrict__ slot16, const double* __restrict__ val,
    const int* __restrict__ blk_slice, const int* __restrict__ blk_ext_off,
    const int* __restrict__ blk_nlow, const int* __restrict__ ext_rows,
    const int* __restrict__ order, const double* __restrict__ X,
    const double* __restrict__ Xh, double* __restrict__ Y,
    const double* __restrict__ Rg, double* __restrict__ gpart, int gbase, int logical, int coff) {
  static_assert(!GRAM || (TS == 4 && XS == 4), "the fused Gram block is built for 4-column panels");
  extern __shared__ double sx[];
  const int b = order[logical];
  const int s0 = blk_slice[b], s1 = blk_slice[b + 1];
  const int r0 = sl_row0[s0];
  const int nown = sl_row0[s1 - 1] + sl_nrows[s1 - 1] - r0;
  const int e0 = blk_ext_off[b], next = blk_ext_off[b + 1] - e0, nlow = blk_nlow ? blk_nlow[b] : 0;
  const int tid = threadIdx.x;
  // (the wavefront's number as a scalar: the slice loop and everything indexed by it stay in SGPRs)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  constexpr int H = TS / 2;  // double2 per staged row

  // The matrix stream of the wavefront's first slice is requested BEFORE the staging of X: its first group
  // is on its way while the rows are gathered (two dependent memory latencies) and the workgroup meets at
  // the barrier.
  constexpr int RU = spmm_ru<RL>::n;
  int slA[RU], slB[RU];
  double vA[RL * RU], vB[RL * RU];
  int s = s0 + wave;
  int len = 0;
  const unsigned short* __restrict__ cp = slot16;
  const double* __restrict__ vp = val;
  if (s < s1) {
    const long long off = sl_off[s];
    len = sl_len[s];
    cp = slot16 + off;
    vp = val + RL * off;
    if (len > 0) spmm_runs_load<RL>(cp, vp, lane, 0, len, slA, vA);
  }

  // Staging: LDS row L of [external rows below | own rows | external rows above] comes from global row
  // id(L); H lanes per row, WG / H rows per pass, SB passes in flight at a time: all ids, then all rows,
  // then the LDS stores -- two memory latencies per batch instead of two per pass.
  {
    constexpr int RPP = WG / H, SB = 8, HP = spmm_row<TS>::stride / 2;      // HP: double2 per LDS row (padded)
    double2* dst = reinterpret_cast<double2*>(sx);
    const int j = tid % H, l0 = tid / H, nst = nown + next;
    // (no conditional anywhere: lanes beyond the last row repeat it -- same bytes to the same place --
    // and every lane reads an id from ext_rows, which has one spare entry at its end)
    const int emax = max(next - 1, 0);
    for (int base = 0; base < nst; base += SB * RPP) {
      int L[SB], id[SB];
#pragma unroll
      for (int it = 0; it < SB; ++it) {
        L[it] = min(base + it * RPP + l0, nst - 1);
        id[it] = ext_rows[e0 + min(max(L[it] < nlow ? L[it] : L[it] - nown, 0), emax)];
      }
      double2 v[SB];
#pragma unroll
      for (int it = 0; it < SB; ++it) {
        const bool own = L[it] >= nlow && L[it] < nlow + nown;
        const int r = own ? r0 + (L[it] - nlow) : id[it];
        const double* src = (r < m ? X + (size_t)r * XS : Xh + (size_t)(r - m) * XS) + coff;
        v[it] = reinterpret_cast<const double2*>(src)[j];
      }
#pragma unroll
      for (int it = 0; it < SB; ++it) dst[(size_t)L[it] * HP + j] = v[it];
    }
    if constexpr (RL == 3) { if (tid < 2 * H) dst[(size_t)(nst + tid / H) * HP + tid % H] = make_double2(0.0, 0.0); }    // (a run may reach two rows past the last one)
  }
  __syncthreads();
  double gw = 0.0, gg = 0.0;      // GRAM: D[i][j] of the lane's 4 x 4 block (lane = 16 i + 4 q + j)
  for (; s < s1; s += WG / 64) {
    double acc[TS];
#pragma unroll
    for (int c = 0; c < TS; ++c) acc[c] = 0.0;
    // GRAM: the slice's rows of R are requested here, in the operand layout, and consumed behind the matrix
    // loop (round 3 touched them here and read them again in the epilogue: the PMC counters showed the
    // 33 MB fetched twice -- a line does not survive the slice's 45 KB of matrix stream in the L2)
    double rin[4] = {0.0, 0.0, 0.0, 0.0};
    if constexpr (GRAM) spmm_gram_load_r(rin, sl_nrows[s], sl_row0[s], lane, Rg);
    if (len > 0) {
      // two register sets: while one group is summed up the next one is in flight
      const int ngt = (len + RU - 1) / RU;
      // (one exit at the bottom and the odd group behind the loop: with an exit in the middle the register
      // allocator copies the set that is in flight at the end of every round, behind a full wait)
      for (int i = 0; i < (ngt >> 1); ++i) {
        spmm_runs_load<RL>(cp, vp, lane, 2 * i + 1, len, slB, vB);
        __builtin_amdgcn_sched_barrier(0);      // (keeps the requests in front of the sums of the group before)
        spmm_runs_fma<TS, RL>(sx, slA, vA, acc);
        __builtin_amdgcn_sched_barrier(0);
        spmm_runs_load<RL>(cp, vp, lane, min(2 * i + 2, ngt - 1), len, slA, vA);
        __builtin_amdgcn_sched_barrier(0);
        spmm_runs_fma<TS, RL>(sx, slB, vB, acc);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ngt & 1) spmm_runs_fma<TS, RL>(sx, slA, vA, acc);
    }
    const int nr = sl_nrows[s], row_s = sl_row0[s];
    if (lane < nr) {
      double2* q = reinterpret_cast<double2*>(Y + (size_t)(row_s + lane) * XS + coff);
#pragma unroll
      for (int i = 0; i < TS / 2; ++i) q[i] = make_double2(acc[2 * i], acc[2 * i + 1]);
    }
    if constexpr (GRAM) {
      const double* own = sx + (size_t)(nlow + row_s - r0) * 4;      // (GRAM: 4 columns, rows not padded)
      spmm_gram_slice(acc, nr, row_s, lane, rin, [&](int rr) { return own + (size_t)rr * 4; }, gw, gg);
    }
    if (s + WG / 64 < s1) {       // (a block of more than four slices: the wavefront's next one)
      const int sn = s + WG / 64;
      const long long off = sl_off[sn];
      len = sl_len[sn];
      cp = slot16 + off;
      vp = val + RL * off;
      if (len > 0) spmm_runs_load<RL>(cp, vp, lane, 0, len, slA, vA);
    }
  }
  if constexpr (GRAM) spmm_gram_tail(gw, gg, sx, wave, lane, tid, gpart + (size_t)(gbase + logical) * 32);
}

// (registers: five wavefronts per SIMD is what 32 KiB of staging per workgroup allow at 4 columns -- 96
// VGPRs; the wider panels stage 48 KiB, three workgroups per CU)
template <int TS, int XS, int RL = 3>
__global__ __launch_bounds__(WG, (TS <= 4 ? 5 : 3)) void k_spmm_runs(
    int m, const long long* __restrict__ sl_off, const int* __restrict__ sl_len,
    const int* __restrict__ sl_row0, const int* __restrict__ sl_nrows,
    const unsigned short* __restrict__ slot16, const double* __restrict__ val,
    const int* __restrict__ blk_slice, const int* __restrict__ blk_ext_off,
    const int* __restrict__ blk_nlow, const int* __restrict__ ext_rows,
    const int* __restrict__ order, int nlist, const double* __restrict__ X,
    const double* __restrict__ Xh, double* __restrict__ Y) {
  spmm_runs_body<TS, XS, false, RL>(m, sl_off, sl_len, sl_row0, sl_nrows, slot16, val, blk_slice, blk_ext_off, blk_nlow,
                                    ext_rows, order, nlist, X, Xh, Y, nullptr, nullptr, 0);
}

// 4 columns with the Gram block; five wavefronts per SIMD as k_spmm_runs<4, 4> (32 KiB of staging each)
template <int RL>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(5))) void k_spmm_runs_gram(
    int m, const long long* __restrict__ sl_off, const int* __restrict__ sl_len,
    const int* __restrict__ sl_row0, const int* __restrict__ sl_nrows,
    const unsigned short* __restrict__ slot16, const double* __restrict__ val,
    const int* __restrict__ blk_slice, const int* __restrict__ blk_ext_off,
    const int* __restrict__ blk_nlow, const int* __restrict__ ext_rows,
    const int* __restrict__ order, int nlist, const double* __restrict__ X,
    const double* __restrict__ Xh, double* __restrict__ Y,
    const double* __restrict__ Rg, double* __restrict__ gpart, int gbase) {
  spmm_runs_body<4, 4, true, RL>(m, sl_off, sl_len, sl_row0, sl_nrows, slot16, val, blk_slice, blk_ext_off, blk_nlow,
                             ext_rows, order, nlist, X, Xh, Y, Rg, gpart, gbase);
}

template <int TS>
__global__ __launch_bounds__(WG) void k_pack_rows(int n, const int* __restrict__ idx,
                                                  const double* __restrict__ X,
                                                  double* __restrict__ out) {
  const int i = (blockIdx.x * WG + threadIdx.x) / TS, c = threadIdx.x % TS;
  if (i < n) out[(size_t)i * TS + c] = X[(size_t)idx[i] * TS + c];
}

}  // namespace

// A Gram block requested from the next SpMM that reads X and writes Y (pa_k_spmm_gram_arm):
// k_spmm_runs_gram leaves one partial block per workgroup, `count` of them so far.
static struct {
  const double* X; const double* Y; const double* R;
  double* partials; int cap, count, armed;
} g_sg;

static long long g_sg_launches = 0;

template <int TS>
static int launch_spmm(const pa_spmm_plan_t* pl, const int* order, int nlist, const double* X,
                       const double* Xh, double* Y) {
  if (nlist <= 0) return 0;
  if constexpr (TS == 4) {
    const size_t lds = (size_t)pl->stage_cap * TS * 8;
    if (g_sg.armed && (pl->runs || pl->staged) && lds <= 64 * 1024 && X == g_sg.X && Y == g_sg.Y && g_sg.count >= 0 &&
        g_sg.count + nlist <= g_sg.cap) {
      const int cpx = (nlist + 7) / 8;
      if (pl->runs)
        PA_LAUNCH(k_spmm_runs_gram<3>, dim3(cpx * 8), dim3(WG), lds, cur_stream(), pl->m, pl->sl_off,
                  pl->sl_len, pl->sl_row0, pl->sl_nrows, pl->col16, pl->val, pl->blk_slice, pl->blk_ext_off,
                  pl->blk_nlow, pl->ext_rows, order, nlist, X, Xh, Y, g_sg.R, g_sg.partials, g_sg.count);
      else         // one staged row per slot (matrices without runs of three): no low / high split, no zero rows
        PA_LAUNCH(k_spmm_runs_gram<1>, dim3(cpx * 8), dim3(WG), lds, cur_stream(), pl->m, pl->sl_off,
                  pl->sl_len, pl->sl_row0, pl->sl_nrows, pl->col16, pl->val, pl->blk_slice, pl->blk_ext_off,
                  (const int*)nullptr, pl->ext_rows, order, nlist, X, Xh, Y, g_sg.R, g_sg.partials, g_sg.count);
      g_sg.count += nlist;
      ++g_sg_launches;
      return kfail("k_spmm_runs");
    }
    if (g_sg.armed && !pl->runs && !pl->staged && X == g_sg.X && Y == g_sg.Y && g_sg.count >= 0 &&
        g_sg.count + nlist <= g_sg.cap) {                             // the window kernel (e.g. 7-point Poisson)
      int win_cap = pl->win_cap;
      if ((size_t)win_cap * TS * 8 > 32 * 1024) win_cap = (32 * 1024) / (TS * 8);
      const size_t ldsw = (size_t)win_cap * TS * 8;
      if (ldsw >= 1024) {
        const int cpx = (nlist + 7) / 8;
        PA_LAUNCH(k_spmm_gram, dim3(cpx * 8), dim3(WG), ldsw, cur_stream(), pl->m, pl->sl_off, pl->sl_len,
                  pl->sl_row0, pl->sl_nrows, pl->col, pl->val, pl->blk_slice, pl->blk_win, order, nlist, win_cap,
                  X, Xh, Y, g_sg.R, g_sg.partials, g_sg.count);
        g_sg.count += nlist;
        ++g_sg_launches;
        return kfail("k_spmm_gram");
      }
    }
    if (g_sg.armed && X == g_sg.X && Y == g_sg.Y) g_sg.count = -1;    // this product leaves no Gram block
  }
  if (pl->runs) {
    // a plan cut for half the panel stride (16 columns): two workgroups per block, 8 of the 16 columns each.
    // (Round 4 measured the alternative the round-3 review proposed -- ONE workgroup that stages and computes
    // the two halves one after the other, so that the second pass over the block's matrix slice comes from
    // the caches: 410-414 us against 372-375 us for the two workgroups, same process, profiles/r04_t16_spmm_seq_ab.txt.)
    constexpr int TC = TS >= 16 ? TS / 2 : TS;
    const int ns = TS / TC;
    const size_t lds = (size_t)pl->stage_cap * spmm_row<TC>::stride * 8;
    static size_t configured = 0;
    if (lds > 64 * 1024 && lds > configured) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spmm_runs<TC, TS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return kfail("hipFuncSetAttribute(k_spmm_runs)");
      configured = lds;
    }
    const int cpx = (nlist + 7) / 8;
    PA_LAUNCH((k_spmm_runs<TC, TS>), dim3(cpx * 8 * ns), dim3(WG), lds, cur_stream(), pl->m, pl->sl_off,
                       pl->sl_len, pl->sl_row0, pl->sl_nrows, pl->col16, pl->val, pl->blk_slice,
                       pl->blk_ext_off, pl->blk_nlow, pl->ext_rows, order, nlist, X, Xh, Y);
    return kfail("k_spmm_runs");
  }
  if (pl->staged) {
    // one staged row per slot: the same kernel with run length 1 (batched staging, double-buffered groups)
    const size_t lds = (size_t)pl->stage_cap * spmm_row<TS>::stride * 8;
    static size_t configured = 0;
    if (lds > 64 * 1024 && lds > configured) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spmm_runs<TS, TS, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return kfail("hipFuncSetAttribute(k_spmm_runs)");
      configured = lds;
    }
    const int cpx = (nlist + 7) / 8;
    PA_LAUNCH((k_spmm_runs<TS, TS, 1>), dim3(cpx * 8), dim3(WG), lds, cur_stream(), pl->m, pl->sl_off,
                       pl->sl_len, pl->sl_row0, pl->sl_nrows, pl->col16, pl->val, pl->blk_slice,
                       pl->blk_ext_off, (const int*)nullptr, pl->ext_rows, order, nlist, X, Xh, Y);
    return kfail("k_spmm_runs");
  }
  // the X window shares the 160 KiB LDS of a CU with other workgroups: at most 32 KiB of rows
  int win_cap = pl->win_cap;
  if ((size_t)win_cap * TS * 8 > 32 * 1024) win_cap = (32 * 1024) / (TS * 8);
  const size_t lds = (size_t)win_cap * TS * 8;
  const int cpx = (nlist + 7) / 8;
  PA_LAUNCH((k_spmm<TS>), dim3(cpx * 8), dim3(WG), lds, cur_stream(), pl->m, pl->sl_off,
                     pl->sl_len, pl->sl_row0, pl->sl_nrows, pl->col, pl->val, pl->blk_slice,
                     pl->blk_win, order, nlist, win_cap, X, Xh, Y);
  return kfail("k_spmm");
}

extern "C" {

void pa_k_spmm_gram_arm(const double* X, const double* Y, const double* R, double* partials, int cap) {
  g_sg.X = X; g_sg.Y = Y; g_sg.R = R; g_sg.partials = partials; g_sg.cap = cap; g_sg.count = 0;
  g_sg.armed = (X && Y && R && partials && cap > 0);
}

long long pa_k_spmm_gram_launches(void) { return g_sg_launches; }

/* owner = the partial-block buffer of the request (every solver object has its own): only that object's request
 * is ended; NULL ends whatever is armed */
void pa_k_spmm_gram_disarm(const double* owner) { if (!owner || owner == g_sg.partials) { g_sg.armed = 0; g_sg.count = 0; } }

/* The number of partial blocks the armed products X -> Y have left since the request (0: none,
 * or the operator was applied with another kernel); the request stays armed for the same pointers. */
int pa_k_spmm_gram_take(const double* X, const double* Y) {
  if (!g_sg.armed || X != g_sg.X || Y != g_sg.Y) return 0;
  const int n = g_sg.count > 0 ? g_sg.count : 0;
  g_sg.armed = 0; g_sg.count = 0;
  return n;
}

int pa_k_spmm(const pa_spmm_plan_t* pl, int ts, const double* X, const double* Xhalo, double* Y,
              int phase) {
  const int* order = pl->order;
  int n = pl->nblk;
  if (phase != 1 && g_sg.armed && X == g_sg.X && Y == g_sg.Y) g_sg.count = 0;   /* a new product starts */
  if (phase == 0) n = pl->n_interior;
  else if (phase == 1) { order += pl->n_interior; n = pl->nblk - pl->n_interior; }
  const double* Xh = Xhalo ? Xhalo : X;
  TS_DISPATCH(ts, return launch_spmm<TS_>(pl, order, n, X, Xh, Y));
  return 0;
}

int pa_k_pack_rows(int n, int ts, const int* idx, const double* X, double* sendbuf) {
  if (n <= 0) return 0;
  const int blocks = (int)(((long long)n * ts + WG - 1) / WG);
  TS_DISPATCH(ts, PA_LAUNCH((k_pack_rows<TS_>), dim3(blocks), dim3(WG), 0, cur_stream(), n,
                                     idx, X, sendbuf));
  return kfail("k_pack_rows");
}

}  // extern "C"
