// bj_g4.hip -- the band block solve for panels of up to 4 columns on the f64 matrix cores, from ONE
// copy of the factor (block_jacobi.c:93-109 of the reference: PARDISO phase 33 on every diagonal
// block; here Z = blockdiag(A)^-1 X with the band Cholesky factors of block_jacobi.c / k_bj_factor).
//
// The register recurrence (k_bj_apply_pairs, kernels.hip) streams two copies of the band -- columns of
// L for the forward sweep, rows for the backward one -- and spends ~20 wave instructions per pivot.
// Here both sweeps read the same records, blocked by groups of FOUR pivots:
//
//   * With Lt = L D^-1 (unit lower, D = diag L) the solve is z = Lt^-T D^-2 Lt^-1 x.  The record of
//     group g = pivots 4g .. 4g+3 holds the strictly lower 4 x 4 corner Lt(g, g) and the w rows
//     below it, Lt(4g+4 .. 4g+3+w, g): the plain band entries, (w + 4) rows x 4 pivots.
//     Forward:  y_g = Lt(g, g)^-1 x_g by substitution (three steps), then x(below) -= Lt(below, g) y_g;
//     backward: r = y_g - Lt(below, g)^T z(below), then z_g = Lt(g, g)^-T r by substitution:
//     the exact transpose, from the same record -- 8 N (w + 4) bytes stored for BOTH sweeps instead of
//     16 N (w + 1).  Same arithmetic as the classic sweeps, other summation order.
//     (A first version stored the group in selective-inversion form, [Lt(g,g)^-1 - I ; -Lt(below,g)
//     Lt(g,g)^-1], which needs no substitution at all.  It was 2.5e-8 off on 2 of 5670 blocks of the
//     elasticity problem (1.3e-15 elsewhere) and was dropped -- at a time when the kernel still issued its DS
//     lane moves by hand behind matrix instructions, a hazard that was found later and that produced exactly
//     such isolated, run-to-run different errors.  So the inversion was not shown to be
//     at fault; it was not taken up again because an explicit inverse of a corner whose entries reach 1e5 has
//     no error bound of the substitution's kind, and parity with an exact solve is what this kernel is for.)
//   * v_mfma_f64_4x4x4 computes four independent 4 x 4 x 4 products; on gfx950 the operands sit as
//         A[i][k] of block q: lane 16 k + 4 q + i     B[k][j]: lane 16 k + 4 q + j
//         D[i][j] of block q: lane 16 i + 4 q + j     (tools/probe/mfma_f64_4x4x4.hip)
//     so a register pair that holds 16 rows x 4 columns of the panel as D (row 4 q + i of the tile in
//     block q, column j) IS the B operand of the backward product (contraction over the tile's rows)
//     and the accumulator of the forward one.  The whole block (up to 16 tiles = 256 rows) lives in
//     registers between the sweeps: the forward result is never written out.
//   * The four rows of a group sit in the four 16-lane rows of their tile register (one quad of each).
//     The 4 x 4 substitution runs on the matrix cores too, in place: three dependent 4 x 4 x 4 products
//     whose A operand holds one column of the negated corner and whose B operand and accumulator are the
//     tile itself (g4_corner_solve).  One DPP sequence then copies the group's quad into every quad (the
//     B operand of the update of the rows below); no LDS round trip and no lane-by-lane selection on the
//     chain from one group to the next.
//   * Records are streamed HBM -> LDS by LDS-DMA in chunks of two groups, double buffered per wave,
//     forward in ascending and backward in descending order; a lane fetches its A-operand entry with
//     one ds_read_b64 (rows outside the record are clamped onto its all-zero row).
//
// Stored: 8 N (w + 4) bytes.  HBM bytes per apply: each sweep streams them once, 16 N (w + 4) + 16 N t
// (PMC: 702-713 MB per launch on the headline problem = 1.08-1.09 x the algorithmic bytes, 6.1-6.3 TB/s,
// 0.97-0.99 of what a plain read kernel streams on the same device; profiles/r0*_pmc_hbm_traffic_elasticity.json).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "pa_device.h"

#ifndef G4_NT
#define G4_NT 12      /* this translation unit: 12 register tiles; bj_g4_nt14.hip / bj_g4_nt16.hip set 14 / 16 */
#endif

// a launch that a replayed graph segment makes in its place is skipped (runtime.hip: pa_rt_skip)
#define PA_LAUNCH(...) do { if (!pa_rt_skipping()) hipLaunchKernelGGL(__VA_ARGS__); } while (0)

// One-shot request for the Gram block of the next launch (pa_k_bj_g4_gram; defined in the 12-tile unit)
extern "C" {
extern const double* pa_g4_gram_prev;
extern double* pa_g4_gram_part;
}

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

inline hipStream_t cur_stream() { return (hipStream_t)pa_rt_stream(); }

// ---------------------------------------------------------------------------------- set-up ----
// Records from the plain forward records of kernels.hip (rec[j * wr + d - 1] = L(j+d, j) / L(j, j)):
// group g of block p at Lg4[off2[p] + g * 4 (w + 4)], four doubles [position][pivot]:
//   positions 0 .. w-1     rows rho = w+3 .. 4 of the group (row 4g + rho of the block), NEGATED: the
//                          product adds them;
//   position  w            zeros (where every row outside the record is sent);
//   positions w+1 .. w+3   rows 1 .. 3 of the strictly lower corner Lt(g, g) (zero on and above the
//                          diagonal), negated as well.
// Chunks of two groups; the groups a block does not have are zero.
__global__ __launch_bounds__(256) void k_bj_g4_setup(const int* __restrict__ list, const int* __restrict__ nrows,
                                                      const int* __restrict__ bw, const long long* __restrict__ off,
                                                      const long long* __restrict__ off2,
                                                      const double* __restrict__ L, double* __restrict__ Lg4) {
  const int p = list[blockIdx.x];
  const int b = nrows[p], w = bw[p], wr = (w + 2) & ~1, nr = w + 4;
  const double* __restrict__ rec = L + off[p];
  double* __restrict__ dst = Lg4 + off2[p];
  const int ngrp = 2 * ((b + 7) / 8);
  for (int e = threadIdx.x; e < ngrp * nr * 4; e += blockDim.x) {
    const int g = e / (4 * nr), r = e - g * 4 * nr, pos = r >> 2, pv = r & 3;
    const int piv = 4 * g + pv;
    double v = 0.0;
    if (pos != w) {
      const int rho = pos < w ? w + 3 - pos : pos - w;     // row of the group
      const int row = 4 * g + rho, d = row - piv;
      if (row < b && piv < b && d >= 1 && d <= w) v = -rec[(size_t)piv * wr + d - 1];
    }
    dst[e] = v;
  }
}

// ----------------------------------------------------------------------------------- apply ----
__device__ __forceinline__ void g4_issue_chunk(const double* __restrict__ rec, int chunk_doubles, int chunk,
                                               double* lbuf, int lane) {
  const int nbytes = chunk_doubles * 8;
  const char* g = reinterpret_cast<const char*>(rec) + (size_t)chunk * nbytes + lane * 16;
  char* l = reinterpret_cast<char*>(lbuf);
  for (int o = 0; o < nbytes; o += 1024)
    __builtin_amdgcn_global_load_lds((glb_ptr)(g + o), (lds_ptr)(l + o), 16, 0, 0);
}

// wait until at most `n` of the wave's VMEM operations are outstanding (they complete in order: the
// youngest n are the LDS-DMA instructions of the chunk that was requested last)
__device__ __forceinline__ void g4_wait_vm(int n) {
  switch (n < 15 ? n : 15) {      // (a smaller count than allowed only waits longer)
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// Few blocks (the pipelined chain below): ring depth, pieces per chunk and the buffer stride are compile-time
// constants -- G4F_RING buffers of NLD = DQ KiB (a chunk of a band w <= 16 DQ - 16 is 8 (w + 4) doubles <= DQ KiB)
// plus one more that receives the requests of chunks a block does not have: EVERY chunk step requests exactly NLD
// pieces, so the wait counts are immediates (the run-time count went through a compare-and-branch tree of ~20
// scalar instructions, twice per chunk), the buffer addresses are constants off the wavefront's base and the
// request loop is unrolled.  A request reads NLD KiB from the chunk's start: up to 1 KiB - 8 bytes beyond the
// chunk's end, i.e. into the next chunk / block or the slack behind the last one (block_jacobi.c allocates it).
constexpr int G4F_RING = 4;
template <int N>
__device__ __forceinline__ void g4_wait_vm_c() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
template <int NLD>
__device__ __forceinline__ void g4_issue_chunk_c(const double* __restrict__ src, double* lbuf, int lane) {
  const char* g = reinterpret_cast<const char*>(src) + lane * 16;
  char* l = reinterpret_cast<char*>(lbuf);
#pragma unroll
  for (int o = 0; o < NLD; ++o)
    __builtin_amdgcn_global_load_lds((glb_ptr)(g + o * 1024), (lds_ptr)(l + o * 1024), 16, 0, 0);
}
// chunk cn of the block when the block has it (valid), chunk 0 into the spare buffer otherwise
template <int NLD>
__device__ __forceinline__ void g4_request_c(const double* __restrict__ rec, int chunk_doubles, int cn, bool valid,
                                             double* lds0, int lane) {
  const double* src = rec + (valid ? (size_t)cn * chunk_doubles : (size_t)0);
  double* dst = lds0 + (valid ? (cn & (G4F_RING - 1)) : G4F_RING) * (NLD * 128);
  g4_issue_chunk_c<NLD>(src, dst, lane);
}

// a value of a 16-lane row rotated by 4 * n lanes inside the row (DPP row_ror)
template <int N>
__device__ __forceinline__ double row_ror(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x120 | N, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x120 | N, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

// s_waitcnt lgkmcnt(0) that the compiler sees as the producer of everything read before it
template <int DQ>
__device__ __forceinline__ void g4_wait_cf(double (&cf)[DQ], double& c0, double& c1, double& c2) {
  if constexpr (DQ == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(cf[0]), "+v"(cf[1]), "+v"(cf[2]));
  else if constexpr (DQ == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(cf[0]), "+v"(cf[1]), "+v"(cf[2]), "+v"(cf[3]));
  else if constexpr (DQ == 5) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(cf[0]), "+v"(cf[1]), "+v"(cf[2]), "+v"(cf[3]), "+v"(cf[4]));
  else if constexpr (DQ == 6) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(cf[0]), "+v"(cf[1]), "+v"(cf[2]), "+v"(cf[3]), "+v"(cf[4]), "+v"(cf[5]));
  else if constexpr (DQ == 7) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(cf[0]), "+v"(cf[1]), "+v"(cf[2]), "+v"(cf[3]), "+v"(cf[4]), "+v"(cf[5]), "+v"(cf[6]));
  else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(cf[0]), "+v"(cf[1]), "+v"(cf[2]), "+v"(cf[3]), "+v"(cf[4]), "+v"(cf[5]), "+v"(cf[6]), "+v"(cf[7]));
}

// Lane moves in the vector ALU, compiler-visible (an earlier version issued ds_bpermute by hand and read
// stale registers: the hazard recogniser does not look into inline assembly, and a double-precision
// result needs wait states before a DS instruction may read it).  Lane = 16 hi + 4 blk + lo
// (tools/probe/permlane_swap.hip): DPP row_ror:4n gives quad q of every 16-lane row the quad (q - n) mod 4;
// bank_mask picks the quads that are written.
template <int GQ>
__device__ __forceinline__ double g4_quad_bcast(double x) {      // quad GQ of every row into all four quads
  int lo = __double2loint(x), hi = __double2hiint(x);
  const int slo = lo, shi = hi;
  lo = __builtin_amdgcn_update_dpp(lo, slo, 0x124, 0xF, 1 << ((GQ + 1) & 3), false);
  hi = __builtin_amdgcn_update_dpp(hi, shi, 0x124, 0xF, 1 << ((GQ + 1) & 3), false);
  lo = __builtin_amdgcn_update_dpp(lo, slo, 0x128, 0xF, 1 << ((GQ + 2) & 3), false);
  hi = __builtin_amdgcn_update_dpp(hi, shi, 0x128, 0xF, 1 << ((GQ + 2) & 3), false);
  lo = __builtin_amdgcn_update_dpp(lo, slo, 0x12C, 0xF, 1 << ((GQ + 3) & 3), false);
  hi = __builtin_amdgcn_update_dpp(hi, shi, 0x12C, 0xF, 1 << ((GQ + 3) & 3), false);
  return __hiloint2double(hi, lo);
}

// Per-lane constants of a sweep (lane = 16 hi + 4 blk + lo):
//   cX     w + 3 - (row of this lane inside a tile as the A operand of this sweep sees it)
//   aX     byte offset of this lane's pivot column inside a record row
//   cg     byte offset, from position w of a record, of this lane's entry of the 4 x 4 corner as the A
//          operand of the substitution: forward Lt(lo, hi) (row lo, column hi), backward Lt(hi, lo)
struct g4_lane { int cX; unsigned aX; unsigned cg; int hi, blk; };

// The 4 x 4 substitution on the matrix cores, in place in the pivot tile: step k adds (column k of the
// negated strictly lower corner) x (row k of the group) to the group's rows -- a 4 x 4 x 4 product whose A
// operand is zero but for that column, and zero altogether in the three blocks that do not hold the group,
// whose B operand and accumulator are both the tile itself.  Three dependent matrix instructions, no lane
// moves; the same operations in the same order as the scalar substitution (the other three terms of each
// product are exact zeros).  A lane gets its A entries by reading either its corner entry or the record's
// zero row: `ck` = LDS addresses chosen per step.
template <int NC, int NT, int Q>
__device__ __forceinline__ void g4_corner_solve(double (&T)[NC * NT], double a0, double a1, double a2) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    T[c * NT + Q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, T[c * NT + Q], T[c * NT + Q], 0, 0, 0);
    T[c * NT + Q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, T[c * NT + Q], T[c * NT + Q], 0, 0, 0);
    T[c * NT + Q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a2, T[c * NT + Q], T[c * NT + Q], 0, 0, 0);
  }
}

// One group of four pivots, forward: Q = tile of the pivots, GQ = their block inside it, `cur` = LDS
// byte address of the group's record.  NC = sets of four panel columns the wavefront carries (1, or 2
// for 8-column panels: the record is read once for both), set c in T[c * NT ..].
template <int NC, int NT, int DQ, int Q, int GQ>
__device__ __forceinline__ void g4_fwd_group(double (&T)[NC * NT], unsigned cur, int w, g4_lane ln) {
  // (opaque copy: without it the compiler keeps the clamped index of every (tile offset, group) pair
  // of the unrolled sweeps alive in registers and spills)
  asm volatile("" : "+v"(ln.cX));
  const unsigned zero_ad = cur + (unsigned)w * 32u, mine = zero_ad + ln.cg;
  const bool here = ln.blk == GQ;
  double a0, a1, a2;          // steps 0, 1, 2: column k of the corner lives in the lanes hi == k
  {
    const unsigned ad0 = (here && ln.hi == 0) ? mine : zero_ad;
    const unsigned ad1 = (here && ln.hi == 1) ? mine : zero_ad;
    const unsigned ad2 = (here && ln.hi == 2) ? mine : zero_ad;
    asm volatile("ds_read_b64 %0, %1" : "=v"(a0) : "v"(ad0));
    asm volatile("ds_read_b64 %0, %1" : "=v"(a1) : "v"(ad1));
    asm volatile("ds_read_b64 %0, %1" : "=v"(a2) : "v"(ad2));
  }
  double cf[DQ];
#pragma unroll
  for (int dq = 0; dq < DQ; ++dq) {
    cf[dq] = 0.0;
    if (Q + dq < NT) {
      const unsigned s = min((unsigned)(ln.cX - (16 * dq - 4 * GQ)), (unsigned)w);
      const unsigned ad = cur + s * 32u + ln.aX;
      asm volatile("ds_read_b64 %0, %1" : "=v"(cf[dq]) : "v"(ad));
    }
  }
  g4_wait_cf<DQ>(cf, a0, a1, a2);
  g4_corner_solve<NC, NT, Q>(T, a0, a1, a2);            // y_g = Lt(g, g)^-1 x_g, in place
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const double yg = g4_quad_bcast<GQ>(T[c * NT + Q]);  // B operand: the group's rows in every block
#pragma unroll
    for (int dq = 0; dq < DQ; ++dq)
      if (Q + dq < NT) T[c * NT + Q + dq] = __builtin_amdgcn_mfma_f64_4x4x4f64(cf[dq], yg, T[c * NT + Q + dq], 0, 0, 0);
  }
}

// The same group, backward: r = y_g - Lt(below, g)^T z(below), then z_g = Lt(g, g)^-T r.
template <int NC, int NT, int DQ, int Q, int GQ>
__device__ __forceinline__ void g4_bwd_group(double (&T)[NC * NT], unsigned cur, int w, g4_lane ln) {
  asm volatile("" : "+v"(ln.cX));
  const unsigned zero_ad = cur + (unsigned)w * 32u, mine = zero_ad + ln.cg;
  const bool here = ln.blk == GQ;
  double a3, a2, a1;          // steps 3, 2, 1: row k of the corner, transposed, lives in the lanes hi == k
  {
    const unsigned ad3 = (here && ln.hi == 3) ? mine : zero_ad;
    const unsigned ad2 = (here && ln.hi == 2) ? mine : zero_ad;
    const unsigned ad1 = (here && ln.hi == 1) ? mine : zero_ad;
    asm volatile("ds_read_b64 %0, %1" : "=v"(a3) : "v"(ad3));
    asm volatile("ds_read_b64 %0, %1" : "=v"(a2) : "v"(ad2));
    asm volatile("ds_read_b64 %0, %1" : "=v"(a1) : "v"(ad1));
  }
  double cf[DQ];
#pragma unroll
  for (int dq = 0; dq < DQ; ++dq) {
    cf[dq] = 0.0;
    if (Q + dq < NT) {
      const unsigned s = min((unsigned)(ln.cX - (16 * dq - 4 * GQ)), (unsigned)w);
      const unsigned ad = cur + s * 32u + ln.aX;
      asm volatile("ds_read_b64 %0, %1" : "=v"(cf[dq]) : "v"(ad));
    }
  }
  g4_wait_cf<DQ>(cf, a3, a2, a1);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    double acc = 0.0;
#pragma unroll
    for (int dq = 0; dq < DQ; ++dq)
      if (Q + dq < NT) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(cf[dq], T[c * NT + Q + dq], acc, 0, 0, 0);
    // every block of the accumulator holds the sum over ITS four rows of each tile: add the four
    // blocks (rotations by 4 and 8 lanes inside the 16-lane rows); the group's lanes take the total
    acc += row_ror<4>(acc);
    acc += row_ror<8>(acc);
    T[c * NT + Q] += here ? acc : 0.0;
  }
  g4_corner_solve<NC, NT, Q>(T, a3, a2, a1);
}

// The chunks of a sweep (two groups each) go through a ring of `ring` LDS buffers (a power of two, lstride
// doubles apart), chunk c in buffer c mod ring.  Entering chunk c, the buffer of chunk c - 1 is free: the
// chunk ring - 1 further on is requested FIRST, then the wave waits for everything but the requests made
// after chunk c's own -- ring - 1 chunks are in flight while it computes.  What a lone block pays per
// chunk is then its arithmetic, not the memory latency: it decides how long the blocks of the last,
// partly filled round take, and everything when a GPU holds fewer blocks than it has SIMDs.
template <int NC, int NT, int DQ, int Q, int H>
__device__ __forceinline__ void g4_fwd_chunk(double (&T)[NC * NT], int b, int w, const double* __restrict__ rec,
                                             int chunk_doubles, double* lds0, int lstride, int lane, g4_lane ln,
                                             int ring) {
  constexpr int C = 2 * Q + H;
  const int nch = (b + 7) >> 3, nld = (chunk_doubles + 127) >> 7;
  const int cn = C + ring - 1;
  if (cn < nch) g4_issue_chunk(rec, chunk_doubles, cn, lds0 + (cn & (ring - 1)) * lstride, lane);
  g4_wait_vm(min(ring - 1, nch - 1 - C) * nld);
  const unsigned cur = (unsigned)(uintptr_t)(lds_ptr)(lds0 + (C & (ring - 1)) * lstride);
  g4_fwd_group<NC, NT, DQ, Q, 2 * H>(T, cur, w, ln);
  g4_fwd_group<NC, NT, DQ, Q, 2 * H + 1>(T, cur + (unsigned)(w + 4) * 32u, w, ln);
  asm volatile("" ::: "memory");
}
template <int NC, int NT, int DQ, int Q>
__device__ __forceinline__ void g4_fwd_tiles(double (&T)[NC * NT], int b, int w, const double* __restrict__ rec,
                                             int chunk_doubles, double* lds0, int lstride, int lane, g4_lane ln,
                                             int ring) {
  if constexpr (Q < NT) {
    if (16 * Q < b) {
      g4_fwd_chunk<NC, NT, DQ, Q, 0>(T, b, w, rec, chunk_doubles, lds0, lstride, lane, ln, ring);
      if (16 * Q + 8 < b) g4_fwd_chunk<NC, NT, DQ, Q, 1>(T, b, w, rec, chunk_doubles, lds0, lstride, lane, ln, ring);
      g4_fwd_tiles<NC, NT, DQ, Q + 1>(T, b, w, rec, chunk_doubles, lds0, lstride, lane, ln, ring);
    }
  }
}

// Backward: the last `ring` chunks are still where the forward sweep left them; chunk c requests chunk
// c - (ring - 1) unless that one is among them.
//
// GP (the Gram block [in | prev]^T out of pa_k_bj_g4_gram): tile Q is final when its chunk H = 0 is done, and
// its share prev(tile)^T z(tile) needs the tile's rows of `prev`.  Loaded behind the sweep by every wavefront at
// once they arrive as one burst after the factor stream has ended (round 3: 137 us against 112 us for the
// plain launch).  Here chunk (Q, 0) requests the rows of tile Q right behind its chunk request -- one more
// VMEM operation among the LDS-DMAs, hand-issued so that the wait counts stay ours (the compiler's own count
// would have to cover the chunk request and wait for it) -- and chunk (Q - 1, 1), whose wait covers everything
// older than ITS chunk request, takes the product.  One register pair in flight; tile 0 is taken behind the sweep.
struct g4_gram { const double* prev; unsigned base, xs; double ap, gp; };     // prev + (base + map * xs): the lane's entry of a row

template <int NC, int NT, int DQ, int Q, int H, bool GP>
__device__ __forceinline__ void g4_bwd_chunk(double (&T)[NC * NT], int b, int w, const double* __restrict__ rec,
                                             int chunk_doubles, double* lds0, int lstride, int lane, g4_lane ln,
                                             int ring, g4_gram& gr, int trow, const unsigned (&mpk)[(NT + 3) / 4]) {
  constexpr int C = 2 * Q + H;
  const int nch = (b + 7) >> 3, nld = (chunk_doubles + 127) >> 7;
  const int cn = C - (ring - 1);
  if (cn >= 0 && C < nch - 1) g4_issue_chunk(rec, chunk_doubles, cn, lds0 + (cn & (ring - 1)) * lstride, lane);
  int extra = 0;
  if constexpr (GP && H == 0) {
    const double* src = gr.prev + (gr.base + ((mpk[Q >> 2] >> (8 * (Q & 3))) & 255u) * gr.xs);
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(gr.ap) : "v"(src) : "memory");
    extra = ring == 2 ? 1 : 0;          // (deeper rings: not counted, the wait then covers one piece more)
  }
  {
    // chunks requested by this sweep that lie below c: indices max(0, c - ring + 1) .. min(c - 1, nch - ring - 1)
    const int lo = max(0, C - ring + 1), hi = min(C - 1, nch - ring - 1);
    g4_wait_vm(max(0, hi - lo + 1) * nld + extra);
  }
  if constexpr (GP && H == 1 && Q + 1 < NT) {
    if (16 * (Q + 1) < b) {             // the rows of tile Q + 1 are here (requested before this chunk's own request)
      asm volatile("" : "+v"(gr.ap));
      const double z = 16 * (Q + 1) + trow < b ? T[Q + 1] : 0.0;
      gr.gp = __builtin_amdgcn_mfma_f64_4x4x4f64(gr.ap, z, gr.gp, 0, 0, 0);
    }
  }
  const unsigned cur = (unsigned)(uintptr_t)(lds_ptr)(lds0 + (C & (ring - 1)) * lstride);
  g4_bwd_group<NC, NT, DQ, Q, 2 * H + 1>(T, cur + (unsigned)(w + 4) * 32u, w, ln);
  g4_bwd_group<NC, NT, DQ, Q, 2 * H>(T, cur, w, ln);
  asm volatile("" ::: "memory");
}
template <int NC, int NT, int DQ, int Q, bool GP>
__device__ __forceinline__ void g4_bwd_tiles(double (&T)[NC * NT], int b, int w, const double* __restrict__ rec,
                                             int chunk_doubles, double* lds0, int lstride, int lane, g4_lane ln,
                                             int ring, g4_gram& gr, int trow, const unsigned (&mpk)[(NT + 3) / 4]) {
  if constexpr (Q >= 0) {
    if (16 * Q < b) {
      if (16 * Q + 8 < b) g4_bwd_chunk<NC, NT, DQ, Q, 1, GP>(T, b, w, rec, chunk_doubles, lds0, lstride, lane, ln, ring, gr, trow, mpk);
      g4_bwd_chunk<NC, NT, DQ, Q, 0, GP>(T, b, w, rec, chunk_doubles, lds0, lstride, lane, ln, ring, gr, trow, mpk);
    }
    g4_bwd_tiles<NC, NT, DQ, Q - 1, GP>(T, b, w, rec, chunk_doubles, lds0, lstride, lane, ln, ring, gr, trow, mpk);
  }
}

// ---- few blocks: the group chain, software-pipelined -------------------------------------------------------
// A GPU that holds fewer blocks than it has SIMDs (one of eight GPUs on the headline problem: 709) runs one
// wavefront per SIMD, and the solve is a chain of 2 x b / 4 dependent group steps: operands from LDS (8 reads,
// ~130 cycles), three dependent matrix instructions for the corner, a quad broadcast, the updates.  Here the
// operands of the NEXT group are requested in front of the matrix instructions of the current one and waited for
// behind them: inside a chunk that is free; across chunks the wait for the next chunk moves half a chunk forward
// (it was requested ring - 1 >= 3 chunks ago: with the deep ring of this configuration it has long arrived),
// while the request for the chunk ring - 1 ahead stays where it was, at the chunk's entry, when both groups of
// the buffer it overwrites have their operands in registers.  Only instantiated for NC = 1 and launched when a GPU
// holds fewer than two blocks per SIMD (or PREALPS_BJ_G4_RING = 4 / 8), with the compile-time ring of G4F_RING
// buffers described above; the memory-bound launch (ring 2, six wavefronts per SIMD) keeps the plain chain:
// there the early wait would delay the next request.
template <int DQ> struct g4_ops { double a0, a1, a2; double cf[DQ]; };

template <int DQ>
__device__ __forceinline__ void g4_ops_wait(g4_ops<DQ>& o) { g4_wait_cf<DQ>(o.cf, o.a0, o.a1, o.a2); }

template <int NT, int DQ, int Q, int GQ, bool FWD>
__device__ __forceinline__ void g4_read_ops(unsigned cur, int w, g4_lane ln, g4_ops<DQ>& o) {
  asm volatile("" : "+v"(ln.cX));
  const unsigned zero_ad = cur + (unsigned)w * 32u, mine = zero_ad + ln.cg;
  const bool here = ln.blk == GQ;
  // forward: steps 0, 1, 2 = columns of the corner (lanes hi == k); backward: steps 3, 2, 1 = its rows, transposed
  const unsigned adA = (here && ln.hi == (FWD ? 0 : 3)) ? mine : zero_ad;
  const unsigned adB = (here && ln.hi == (FWD ? 1 : 2)) ? mine : zero_ad;
  const unsigned adC = (here && ln.hi == (FWD ? 2 : 1)) ? mine : zero_ad;
  asm volatile("ds_read_b64 %0, %1" : "=v"(o.a0) : "v"(adA));
  asm volatile("ds_read_b64 %0, %1" : "=v"(o.a1) : "v"(adB));
  asm volatile("ds_read_b64 %0, %1" : "=v"(o.a2) : "v"(adC));
#pragma unroll
  for (int dq = 0; dq < DQ; ++dq) {
    o.cf[dq] = 0.0;
    if (Q + dq < NT) {
      const unsigned s = min((unsigned)(ln.cX - (16 * dq - 4 * GQ)), (unsigned)w);
      const unsigned ad = cur + s * 32u + ln.aX;
      asm volatile("ds_read_b64 %0, %1" : "=v"(o.cf[dq]) : "v"(ad));
    }
  }
}
// Few blocks: where a lane reads its operands depends on (GQ, dq) and the band only, not on the chunk -- with the
// compile-time ring the chunk's buffer is an immediate offset of the read.  The 12 + 4 DQ LDS addresses of a
// sweep are formed once (they were 25 vector instructions per group: compare, select, clamp, shift-add) and the
// eight reads of a group take no address arithmetic at all.  An odd group is the second record of its chunk.
template <int DQ> struct g4_pre { unsigned ca[4], cb[4], cc[4], cf[DQ][4]; };
template <int DQ, bool FWD>
__device__ __forceinline__ void g4_make_pre(unsigned base, int w, g4_lane ln, g4_pre<DQ>& p) {
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const unsigned cur = base + (unsigned)(gq & 1) * (unsigned)(w + 4) * 32u;
    const unsigned zero_ad = cur + (unsigned)w * 32u, mine = zero_ad + ln.cg;
    const bool here = ln.blk == gq;
    p.ca[gq] = (here && ln.hi == (FWD ? 0 : 3)) ? mine : zero_ad;
    p.cb[gq] = (here && ln.hi == (FWD ? 1 : 2)) ? mine : zero_ad;
    p.cc[gq] = (here && ln.hi == (FWD ? 2 : 1)) ? mine : zero_ad;
    asm volatile("" : "+v"(p.ca[gq]), "+v"(p.cb[gq]), "+v"(p.cc[gq]));
#pragma unroll
    for (int dq = 0; dq < DQ; ++dq) {
      const unsigned sidx = min((unsigned)(ln.cX - (16 * dq - 4 * gq)), (unsigned)w);
      p.cf[dq][gq] = cur + sidx * 32u + ln.aX;
      asm volatile("" : "+v"(p.cf[dq][gq]));
    }
  }
}
// the operands of group GQ of tile Q out of ring buffer BUF (LSB = bytes per buffer)
template <int NT, int DQ, int Q, int GQ, int BUF>
__device__ __forceinline__ void g4_read_ops_c(const g4_pre<DQ>& p, g4_ops<DQ>& o) {
  constexpr int OFF = BUF * DQ * 1024;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(o.a0) : "v"(p.ca[GQ]), "n"(OFF));
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(o.a1) : "v"(p.cb[GQ]), "n"(OFF));
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(o.a2) : "v"(p.cc[GQ]), "n"(OFF));
#pragma unroll
  for (int dq = 0; dq < DQ; ++dq) {
    o.cf[dq] = 0.0;
    if (Q + dq < NT) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(o.cf[dq]) : "v"(p.cf[dq][GQ]), "n"(OFF));
  }
}
template <int NT, int DQ, int Q, int GQ>
__device__ __forceinline__ void g4_fwd_math(double (&T)[NT], const g4_ops<DQ>& o) {
  g4_corner_solve<1, NT, Q>(T, o.a0, o.a1, o.a2);
  const double yg = g4_quad_bcast<GQ>(T[Q]);
#pragma unroll
  for (int dq = 0; dq < DQ; ++dq)
    if (Q + dq < NT) T[Q + dq] = __builtin_amdgcn_mfma_f64_4x4x4f64(o.cf[dq], yg, T[Q + dq], 0, 0, 0);
}
template <int NT, int DQ, int Q, int GQ>
__device__ __forceinline__ void g4_bwd_math(double (&T)[NT], const g4_ops<DQ>& o, g4_lane ln) {
  double acc = 0.0;
#pragma unroll
  for (int dq = 0; dq < DQ; ++dq)
    if (Q + dq < NT) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(o.cf[dq], T[Q + dq], acc, 0, 0, 0);
  acc += row_ror<4>(acc);
  acc += row_ror<8>(acc);
  T[Q] += (ln.blk == GQ) ? acc : 0.0;
  g4_corner_solve<1, NT, Q>(T, o.a0, o.a1, o.a2);
}

// forward chunk C = 2 Q + H; on entry A holds the operands of its first group (waited for)
template <int NT, int DQ, int Q, int H>
__device__ __forceinline__ void g4_fwd_chunk_p(double (&T)[NT], int b, const double* __restrict__ rec,
                                               int chunk_doubles, double* lds0, int lane, const g4_pre<DQ>& pre,
                                               g4_ops<DQ>& A, g4_ops<DQ>& B) {
  constexpr int C = 2 * Q + H, NLD = DQ, LS = NLD * 128;
  const int nch = (b + 7) >> 3;
  constexpr int cn = C + G4F_RING - 1;
  g4_request_c<NLD>(rec, chunk_doubles, cn, cn < nch, lds0, lane);
  g4_read_ops_c<NT, DQ, Q, 2 * H + 1, (C & (G4F_RING - 1))>(pre, B);
  g4_fwd_math<NT, DQ, Q, 2 * H>(T, A);
  g4_ops_wait<DQ>(B);
  if (C + 1 < nch) {
    // the next chunk was requested G4F_RING - 2 chunk steps before this one: all but the requests since
    g4_wait_vm_c<(G4F_RING - 2) * NLD>();
    constexpr int Qn = H ? Q + 1 : Q, Gn = H ? 0 : 2;
    if constexpr (Qn < NT) g4_read_ops_c<NT, DQ, Qn, Gn, ((C + 1) & (G4F_RING - 1))>(pre, A);
  }
  g4_fwd_math<NT, DQ, Q, 2 * H + 1>(T, B);
  if (C + 1 < nch) g4_ops_wait<DQ>(A);
  asm volatile("" ::: "memory");
}
template <int NT, int DQ, int Q>
__device__ __forceinline__ void g4_fwd_tiles_p(double (&T)[NT], int b, const double* __restrict__ rec,
                                               int chunk_doubles, double* lds0, int lane, const g4_pre<DQ>& pre,
                                               g4_ops<DQ>& A, g4_ops<DQ>& B) {
  if constexpr (Q < NT) {
    if (16 * Q < b) {
      g4_fwd_chunk_p<NT, DQ, Q, 0>(T, b, rec, chunk_doubles, lds0, lane, pre, A, B);
      if (16 * Q + 8 < b) g4_fwd_chunk_p<NT, DQ, Q, 1>(T, b, rec, chunk_doubles, lds0, lane, pre, A, B);
      g4_fwd_tiles_p<NT, DQ, Q + 1>(T, b, rec, chunk_doubles, lds0, lane, pre, A, B);
    }
  }
}

// backward chunk C: groups 2 H + 1, then 2 H; on entry A holds the operands of group 2 H + 1 -- loaded here when
// C is the sweep's first chunk (the last G4F_RING chunks are still where the forward sweep left them: the
// requests of chunks the block does not have went to the spare buffer)
template <int NT, int DQ, int Q, int H>
__device__ __forceinline__ void g4_bwd_chunk_p(double (&T)[NT], int b, const double* __restrict__ rec,
                                               int chunk_doubles, double* lds0, int lane, g4_lane ln, const g4_pre<DQ>& pre,
                                               g4_ops<DQ>& A, g4_ops<DQ>& B) {
  constexpr int C = 2 * Q + H, NLD = DQ, LS = NLD * 128;
  const int nch = (b + 7) >> 3;
  constexpr int cn = C - (G4F_RING - 1);
  g4_request_c<NLD>(rec, chunk_doubles, cn >= 0 ? cn : 0, cn >= 0 && C < nch - 1, lds0, lane);
  if (C == nch - 1) {
    g4_read_ops_c<NT, DQ, Q, 2 * H + 1, (C & (G4F_RING - 1))>(pre, A);
    g4_ops_wait<DQ>(A);
  }
  g4_read_ops_c<NT, DQ, Q, 2 * H, (C & (G4F_RING - 1))>(pre, B);
  g4_bwd_math<NT, DQ, Q, 2 * H + 1>(T, A, ln);
  g4_ops_wait<DQ>(B);
  if (C > 0) {
    // chunk C - 1: requested G4F_RING - 2 chunk steps before this one, or still there from the forward sweep
    g4_wait_vm_c<(G4F_RING - 2) * NLD>();
    constexpr int Qn = H ? Q : Q - 1, Gn = H ? 1 : 3;
    if constexpr (Qn >= 0) g4_read_ops_c<NT, DQ, Qn, Gn, ((C - 1) & (G4F_RING - 1))>(pre, A);
  }
  g4_bwd_math<NT, DQ, Q, 2 * H>(T, B, ln);
  if (C > 0) g4_ops_wait<DQ>(A);
  asm volatile("" ::: "memory");
}
template <int NT, int DQ, int Q>
__device__ __forceinline__ void g4_bwd_tiles_p(double (&T)[NT], int b, const double* __restrict__ rec,
                                               int chunk_doubles, double* lds0, int lane, g4_lane ln, const g4_pre<DQ>& pre,
                                               g4_ops<DQ>& A, g4_ops<DQ>& B) {
  if constexpr (Q >= 0) {
    if (16 * Q < b) {
      if (16 * Q + 8 < b) g4_bwd_chunk_p<NT, DQ, Q, 1>(T, b, rec, chunk_doubles, lds0, lane, ln, pre, A, B);
      g4_bwd_chunk_p<NT, DQ, Q, 0>(T, b, rec, chunk_doubles, lds0, lane, ln, pre, A, B);
    }
    g4_bwd_tiles_p<NT, DQ, Q - 1>(T, b, rec, chunk_doubles, lds0, lane, ln, pre, A, B);
  }
}

// One wavefront per block.  NT tiles of 16 rows (b <= 16 NT), DQ = tiles a group's record reaches
// (w + 15 < 16 DQ).  `xs` = row stride of the panels in doubles (2, 4; 8 / 16 when the kernel is
// launched on a 4-column slice of a wider panel), `ncol` <= 4 columns starting at `in` / `out`.
template <int NC, int NT, int DQ, int OCC, bool GP, bool PIPE = false>
__global__ __launch_bounds__(256, OCC) void k_bj_g4(
    const int* __restrict__ list, int count, const int* __restrict__ row0, const int* __restrict__ nrows,
    const int* __restrict__ bw, const long long* __restrict__ off2, const int* __restrict__ map_f,
    const double* __restrict__ Lg4, const double* __restrict__ invd_f, int lds_per_wave, int ring, int xs, int ncol,
    const double* __restrict__ in, double* __restrict__ out, const double* __restrict__ gprev,
    double* __restrict__ gpart) {
  extern __shared__ double smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int pi = blockIdx.x * (blockDim.x >> 6) + wave;
  if (pi >= count) return;
  const int p = __builtin_amdgcn_readfirstlane(list[pi]);
  const int r0 = __builtin_amdgcn_readfirstlane(row0[p]);
  const int b = __builtin_amdgcn_readfirstlane(nrows[p]);
  const int w = __builtin_amdgcn_readfirstlane(bw[p]);
  const long long o64 = off2[p];
  const size_t o = ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(o64 >> 32)) << 32) |
                   (unsigned)__builtin_amdgcn_readfirstlane((int)o64);
  const double* __restrict__ rec = Lg4 + o;
  double* lds0 = smem + (size_t)wave * lds_per_wave;
  const int lstride = ((lds_per_wave / ring) >> 7) << 7;      // (whole KiB: the launcher may pad lds_per_wave)
  const int chunk_doubles = 8 * (w + 4);
  const int hi = lane >> 4, blk = (lane >> 2) & 3, lo = lane & 3;
  // tile layout (D / B operand): row 4 blk + hi of the tile, column lo
  const int trow = 4 * blk + hi;
  // A operand, forward: row 4 blk + lo of the tile against pivot hi; backward: row 4 blk + hi against pivot lo
  g4_lane lf;
  lf.cX = w + 3 - (4 * blk + lo);
  lf.aX = (unsigned)hi * 8u;
  lf.cg = (unsigned)lo * 32u + (unsigned)hi * 8u;      // Lt(lo, hi): row lo of the corner, column hi
  lf.hi = hi;
  lf.blk = blk;

  if constexpr (PIPE) g4_request_c<DQ>(rec, chunk_doubles, 0, true, lds0, lane);
  else g4_issue_chunk(rec, chunk_doubles, 0, lds0, lane);
  double T[NC * NT];
  const int* __restrict__ mp = map_f + r0;
  // Where the block's rows lie in the panel: row r0 + map[j], map < b <= 256 -- four of them to a register,
  // kept across both sweeps (the addresses themselves would be NT registers; gathered again behind the
  // backward sweep they put a dependent memory latency at the end of every block).
  unsigned mpk[(NT + 3) / 4];        // (the host checks that m * xs fits 31 bits)
#pragma unroll
  for (int q = 0; q < (NT + 3) / 4; ++q) mpk[q] = 0u;
  {
    unsigned mpv[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int j = 16 * q + trow;
      mpv[q] = (unsigned)mp[j < b ? j : 0];
    }
    // (every load unconditional, rows and columns beyond the block's clamped onto valid ones and masked
    // afterwards: a load inside a branch makes the compiler wait for everything in flight at the join, and
    // the NT loads of a block would come one memory latency after the other)
    const unsigned loc = (unsigned)min(lo, ncol - 1);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const unsigned cc = (unsigned)min(4 * c, max(ncol - 1 - (int)loc, 0));
#pragma unroll
      for (int q = 0; q < NT; ++q) T[c * NT + q] = in[((unsigned)r0 + mpv[q]) * (unsigned)xs + loc + cc];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int q = 0; q < NT; ++q)
        T[c * NT + q] = (16 * q + trow < b && 4 * c + lo < ncol) ? T[c * NT + q] : 0.0;
#pragma unroll
    for (int q = 0; q < NT; ++q) mpk[q >> 2] |= mpv[q] << (8 * (q & 3));
  }
  // the rest of the ring behind the panel loads, so that the newest requests are all chunks
  if constexpr (PIPE) {
#pragma unroll
    for (int c = 1; c < G4F_RING - 1; ++c) g4_request_c<DQ>(rec, chunk_doubles, c, 8 * c < b, lds0, lane);
  } else {
    for (int c = 1; c < ring - 1 && 8 * c < b; ++c) g4_issue_chunk(rec, chunk_doubles, c, lds0 + c * lstride, lane);
  }

  g4_ops<DQ> opA, opB;       // PIPE: the operands of the group at hand and of the next one
  if constexpr (PIPE) {
    static_assert(NC == 1, "the pipelined chain is built for panels of up to 4 columns");
    g4_pre<DQ> pf;
    g4_make_pre<DQ, true>((unsigned)(uintptr_t)(lds_ptr)lds0, w, lf, pf);
    g4_wait_vm_c<(G4F_RING - 2) * DQ>();           // chunk 0 (requested first, the rest of the ring behind it)
    g4_read_ops_c<NT, DQ, 0, 0, 0>(pf, opA);
    g4_ops_wait<DQ>(opA);
    g4_fwd_tiles_p<NT, DQ, 0>(T, b, rec, chunk_doubles, lds0, lane, pf, opA, opB);
  } else {
    g4_fwd_tiles<NC, NT, DQ, 0>(T, b, w, rec, chunk_doubles, lds0, lstride, lane, lf, ring);
  }

  // y = D^-2 a
  double ga_mid = 0.0;
  {
    const double* __restrict__ dv = invd_f + r0;
    double d[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) { const int j = 16 * q + trow; d[q] = dv[j < b ? j : 0]; }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        const double a = T[c * NT + q];
        T[c * NT + q] *= d[q] * d[q];
        // in^T out = in^T L^-T D^-2 L^-1 in = a^T (D^-2 a): the first Gram block of pa_k_bj_g4_gram costs no
        // load at all here, both operands are this tile (rows beyond the block masked out)
        if constexpr (GP) {
          const bool on = 16 * q + trow < b;
          ga_mid = __builtin_amdgcn_mfma_f64_4x4x4f64(on ? a : 0.0, on ? T[q] : 0.0, ga_mid, 0, 0, 0);
        }
      }
  }
  // (the last chunk of the forward sweep is the first of the backward one: it is still in its buffer,
  // and so is the one before it -- g4_bwd_chunk does not fetch that one again)
  g4_gram gr;
  gr.prev = gprev; gr.base = 0u; gr.xs = (unsigned)xs; gr.ap = 0.0; gr.gp = 0.0;
  {
    int l2 = lane;
    asm volatile("" : "+v"(l2));                   // (recomputed here rather than kept across the forward sweep)
    const int hi2 = l2 >> 4, blk2 = (l2 >> 2) & 3, lo2 = l2 & 3;
    g4_lane lb;
    lb.cX = w + 3 - (4 * blk2 + hi2);
    lb.aX = (unsigned)lo2 * 8u;
    lb.cg = (unsigned)hi2 * 32u + (unsigned)lo2 * 8u;    // Lt(hi, lo): the transposed corner
    lb.hi = hi2;
    lb.blk = blk2;
    gr.base = (unsigned)r0 * (unsigned)xs + (unsigned)lo2;
    if constexpr (PIPE) {
      // (few blocks: nothing to gain from spreading the rows of gprev over the sweep -- one wavefront per SIMD,
      // no burst -- so they are all requested here, in front of it, and multiplied in behind it)
      double apv[NT];
      if constexpr (GP) {
#pragma unroll
        for (int q = 0; q < NT; ++q) apv[q] = gprev[gr.base + ((mpk[q >> 2] >> (8 * (q & 3))) & 255u) * gr.xs];
      }
      g4_pre<DQ> pb;
      g4_make_pre<DQ, false>((unsigned)(uintptr_t)(lds_ptr)lds0, w, lb, pb);
      g4_bwd_tiles_p<NT, DQ, NT - 1>(T, b, rec, chunk_doubles, lds0, lane, lb, pb, opA, opB);
      // (the last steps' requests went to the spare buffer and nobody waits for them: no LDS-DMA in flight when the
      // wavefront's LDS is given to the next workgroup)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if constexpr (GP) {
#pragma unroll
        for (int q = 1; q < NT; ++q) {
          const double z = 16 * q + 4 * blk2 + hi2 < b ? T[q] : 0.0;
          gr.gp = __builtin_amdgcn_mfma_f64_4x4x4f64(apv[q], z, gr.gp, 0, 0, 0);
        }
        gr.ap = apv[0];          // (tile 0 is taken below, like the spread-out variant's last request)
      }
    } else {
      g4_bwd_tiles<NC, NT, DQ, NT - 1, GP>(T, b, w, rec, chunk_doubles, lds0, lstride, lane, lb, ring, gr, 4 * blk2 + hi2, mpk);
    }
  }

  {
    int l3 = lane;
    asm volatile("" : "+v"(l3));                   // (addresses recomputed, not carried through both sweeps)
    const int trow3 = 4 * ((l3 >> 2) & 3) + (l3 >> 4), lo3 = l3 & 3;
    unsigned rowoff[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q)
      rowoff[q] = ((unsigned)r0 + ((mpk[q >> 2] >> (8 * (q & 3))) & 255u)) * (unsigned)xs + lo3;
    if constexpr (GP) {          // tile 0's rows of gprev: the one operation still in flight (taken before the stores)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(gr.ap) :: "memory");
      gr.gp = __builtin_amdgcn_mfma_f64_4x4x4f64(gr.ap, trow3 < b ? T[0] : 0.0, gr.gp, 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int q = 0; q < NT; ++q)
        if (16 * q + trow3 < b && 4 * c + lo3 < ncol) out[rowoff[q] + 4 * c] = T[c * NT + q];
    // GP (4-column panels, pa_k_bj_g4_gram): the block's share of [in | gprev]^T out -- the Gram block the ECG
    // iteration forms right after the apply (beta = [AP | AP_prev]^T Z, ecg.c:510) -- while the result is still
    // in registers.  A tile (lane = 16 hi + 4 blk + lo: row 4 blk + hi, column lo) is the B operand of
    // v_mfma_f64_4x4x4 as it stands (k = hi); the A operand, row 4 blk + k of the other panel in column i = lo,
    // sits at the tile's own address.  8 x 4 per block, the layout of k_gram<4, 2>.  in^T out was formed between
    // the sweeps (ga_mid); the rows of gprev came in during the backward sweep (g4_bwd_chunk), tile 0 last.
    if constexpr (GP) {
      double ga = ga_mid, gp = gr.gp;
      ga += row_ror<4>(ga); ga += row_ror<8>(ga);
      gp += row_ror<4>(gp); gp += row_ror<8>(gp);
      if (((l3 >> 2) & 3) == 0) {
        const int i = l3 >> 4;
        double* gq = gpart + (size_t)pi * 32;
        gq[i + 8 * lo3] = ga;
        gq[4 + i + 8 * lo3] = gp;
      }
    }
  }
}

char g_err[256];
int kfail(const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 0;
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  fprintf(stderr, "[prealps_hip] kernel launch failed: %s\n", g_err);
  return 1;
}

template <int NC, int NT, int DQ, int OCC>
int launch_occ(const int* list, int count, const pa_bj_plan_t* pl, int wmax, int xs, int ncol, const double* in, double* out) {
  const int cbuf = (8 * (wmax + 4) + 127) & ~127;        // doubles per ring buffer, a multiple of 1 KiB
  const int nld = cbuf >> 7;
  // Ring depth.  Many blocks (the chip is filled several wavefronts deep): two buffers, the LDS then
  // allows six wavefronts per SIMD.  Few blocks (fewer than two per SIMD: a shard of a multi-GPU run):
  // a lone wavefront cannot hide the memory latency behind other wavefronts, so up to eight chunks in
  // flight.  PREALPS_BJ_G4_RING overrides (2, 4, 8).
  static int ring_env = -1;
  if (ring_env < 0) { const char* e = getenv("PREALPS_BJ_G4_RING"); ring_env = e ? atoi(e) : 0; }
  const int simds = 4 * (pa_rt_num_cus() > 0 ? pa_rt_num_cus() : 256);
  int ring = ring_env == 2 || ring_env == 4 || ring_env == 8 ? ring_env : (count < 2 * simds ? 8 : 2);
  while (ring > 2 && ((ring - 1) * nld > 15 || (size_t)ring * cbuf * 8 > 40 * 1024)) ring >>= 1;
  // (Fewer resident blocks, so that a block's records survive in the Infinity Cache between its two sweeps, was
  // tried with padded LDS in round 3: 130.4 / 144.9 us against 122.1 us.  The switch is gone.)
  const int per_wave = ring * cbuf;
  int waves = (160 * 1024) / (per_wave * 8);
  if (waves > 4) waves = 4;
  if (waves < 1) return 1;
  const size_t lds = (size_t)waves * per_wave * 8;
  static size_t configured = 0;
  if (lds > 64 * 1024 && lds > configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bj_g4<NC, NT, DQ, OCC, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        (NC == 1 && hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bj_g4<NC, NT, DQ, OCC, NC == 1>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess))
      return kfail("hipFuncSetAttribute(k_bj_g4)");
    configured = lds;
  }
  const int blocks = (count + waves - 1) / waves;
  if constexpr (NC == 1 && NT == 12 && DQ <= 5) {
    // few blocks (fewer than two per SIMD): the software-pipelined group chain with its compile-time ring
    // (G4F_RING buffers of DQ KiB and the spare one per wavefront)
    if (ring >= 4) {
      const int per_wave_f = (G4F_RING + 1) * DQ * 128;
      // single-wavefront workgroups: they fill the 160 KiB of a CU to the last buffer set (six blocks per CU at
      // DQ = 5; four-wavefront workgroups of 100 KiB leave room for one).  Measured 1 / 2 / 4 wavefronts per
      // workgroup: 20.7 / 24.8 / 20.7 us on 709 blocks (1/8 of the headline problem), 28.5 / 28.3 / 28.9 us on 1418
      const int waves_f = 1;
      const size_t lds_f = (size_t)waves_f * per_wave_f * 8;
      static size_t configured_p = 0;
      if (lds_f > 64 * 1024 && lds_f > configured_p) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bj_g4<NC, NT, DQ, 1, true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bj_g4<NC, NT, DQ, 1, false, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f) != hipSuccess)
          return kfail("hipFuncSetAttribute(k_bj_g4, pipelined)");
        configured_p = lds_f;
      }
      const int blocks_f = (count + waves_f - 1) / waves_f;
      if (pa_g4_gram_part && xs == 4 && ncol == 4)
        PA_LAUNCH((k_bj_g4<NC, NT, DQ, 1, true, true>), dim3(blocks_f), dim3(64 * waves_f), lds_f, cur_stream(), list, count, pl->row0,
                  pl->nrows, pl->bw, pl->off2, pl->map_f, pl->Lg4, pl->invd_f, per_wave_f, G4F_RING, xs, ncol, in, out,
                  pa_g4_gram_prev, pa_g4_gram_part);
      else
        PA_LAUNCH((k_bj_g4<NC, NT, DQ, 1, false, true>), dim3(blocks_f), dim3(64 * waves_f), lds_f, cur_stream(), list, count, pl->row0,
                  pl->nrows, pl->bw, pl->off2, pl->map_f, pl->Lg4, pl->invd_f, per_wave_f, G4F_RING, xs, ncol, in, out,
                  (const double*)nullptr, (double*)nullptr);
      return kfail("k_bj_g4");
    }
  }
  if constexpr (NC == 1) {
    if (pa_g4_gram_part && xs == 4 && ncol == 4) {
      PA_LAUNCH((k_bj_g4<NC, NT, DQ, OCC, true>), dim3(blocks), dim3(64 * waves), lds, cur_stream(), list, count, pl->row0,
                pl->nrows, pl->bw, pl->off2, pl->map_f, pl->Lg4, pl->invd_f, per_wave, ring, xs, ncol, in, out,
                pa_g4_gram_prev, pa_g4_gram_part);
      return kfail("k_bj_g4");
    }
  }
  PA_LAUNCH((k_bj_g4<NC, NT, DQ, OCC, false>), dim3(blocks), dim3(64 * waves), lds, cur_stream(), list, count, pl->row0,
            pl->nrows, pl->bw, pl->off2, pl->map_f, pl->Lg4, pl->invd_f, per_wave, ring, xs, ncol, in, out,
            (const double*)nullptr, (double*)nullptr);
  return kfail("k_bj_g4");
}

template <int NC, int NT, int DQ>
int launch(const int* list, int count, const pa_bj_plan_t* pl, int wmax, int xs, int ncol, const double* in, double* out) {
  return launch_occ<NC, NT, DQ, (NC == 1 && NT <= 12 && DQ <= 5) ? 5 : (NC == 1 ? 4 : 3)>(list, count, pl, wmax, xs, ncol, in, out);
}

template <int NC, int NT>
int launch_dq(const int* list, int count, const pa_bj_plan_t* pl, int wmax, int xs, int ncol, const double* in, double* out) {
  const int dq = ((wmax + 15) >> 4) + 1;
  switch (dq) {
    case 1: case 2: case 3: return launch<NC, NT, 3>(list, count, pl, wmax, xs, ncol, in, out);
    case 4: return launch<NC, NT, 4>(list, count, pl, wmax, xs, ncol, in, out);
    case 5: return launch<NC, NT, 5>(list, count, pl, wmax, xs, ncol, in, out);
    case 6: if constexpr (NC <= 2) return launch<NC, NT, 6>(list, count, pl, wmax, xs, ncol, in, out); else return 1;
    case 7: if constexpr (NC == 1) return launch<NC, NT, 7>(list, count, pl, wmax, xs, ncol, in, out); else return 1;
    case 8: if constexpr (NC == 1) return launch<NC, NT, 8>(list, count, pl, wmax, xs, ncol, in, out); else return 1;
    default: return 1;
  }
}

}  // namespace

extern "C" {

#if G4_NT == 12
const double* pa_g4_gram_prev = nullptr;
double* pa_g4_gram_part = nullptr;
/* The next pa_k_bj_g4 on a 4-column panel also leaves, per block (in the order of `list`), the 8 x 4 block
 * [in | prev]^T out in part (32 doubles each).  Cleared by that call. */
void pa_k_bj_g4_gram(const double* prev, double* part) { pa_g4_gram_prev = prev; pa_g4_gram_part = part; }

int pa_bj_g4_max_rows(void) { return 256; }
int pa_bj_g4_max_band(void) { return 112; }
int pa_bj_g4_max_band8(void) { return 80; }     /* panels of 5 .. 8 columns (two column sets per wavefront) */

int pa_k_bj_g4_setup(const int* list, int count, const int* nrows, const int* bw, const long long* off,
                      const long long* off2, const double* L, double* Lg4) {
  if (count <= 0) return 0;
  PA_LAUNCH(k_bj_g4_setup, dim3(count), dim3(256), 0, cur_stream(), list, nrows, bw, off, off2, L, Lg4);
  return kfail("k_bj_g4_setup");
}

int pa_k_bj_g4_nt14(const pa_bj_plan_t* pl, const int* list, int count, int wmax, int xs, int ncol, const double* in, double* out);
int pa_k_bj_g4_nt16(const pa_bj_plan_t* pl, const int* list, int count, int wmax, int xs, int ncol, const double* in, double* out);

/* One class of blocks (all with at most bmax rows and bands up to wmax) on a panel of row stride xs:
 * the ncol <= 8 columns starting at `in` / `out` (more than 4: bands up to pa_bj_g4_max_band8()).  The
 * kernels for 12 / 14 / 16 register tiles are compiled in three translation units (bj_g4.hip,
 * bj_g4_nt14.hip, bj_g4_nt16.hip: the same source, G4_NT set) so that they build in parallel. */
int pa_k_bj_g4(const pa_bj_plan_t* pl, const int* list, int count, int wmax, int bmax, int xs, int ncol,
               const double* in, double* out) {
  if (count <= 0) return 0;
  int rc;
  if (bmax > 224) rc = pa_k_bj_g4_nt16(pl, list, count, wmax, xs, ncol, in, out);
  else if (bmax > 192) rc = pa_k_bj_g4_nt14(pl, list, count, wmax, xs, ncol, in, out);
  else rc = ncol > 4 ? launch_dq<2, 12>(list, count, pl, wmax, xs, ncol, in, out)
                     : launch_dq<1, 12>(list, count, pl, wmax, xs, ncol, in, out);
  pa_g4_gram_prev = nullptr; pa_g4_gram_part = nullptr;
  return rc;
}
#elif G4_NT == 14
int pa_k_bj_g4_nt14(const pa_bj_plan_t* pl, const int* list, int count, int wmax, int xs, int ncol, const double* in, double* out) {
  return ncol > 4 ? launch_dq<2, 14>(list, count, pl, wmax, xs, ncol, in, out)
                  : launch_dq<1, 14>(list, count, pl, wmax, xs, ncol, in, out);
}
#else
int pa_k_bj_g4_nt16(const pa_bj_plan_t* pl, const int* list, int count, int wmax, int xs, int ncol, const double* in, double* out) {
  return ncol > 4 ? launch_dq<2, 16>(list, count, pl, wmax, xs, ncol, in, out)
                  : launch_dq<1, 16>(list, count, pl, wmax, xs, ncol, in, out);
}
#endif

}  // extern "C"
