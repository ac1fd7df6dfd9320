// kernels_common.h -- what the kernel translation units share: launch macro, the library stream,
// launch-error report, panel row loads / stores and lane moves.  (Included inside each unit: every
// unit has its own copy of the anonymous-namespace helpers.)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "pa_device.h"

// a launch that a replayed graph segment makes in its place is skipped (runtime.hip: pa_rt_skip)
#define PA_LAUNCH(...) do { if (!pa_rt_skipping()) hipLaunchKernelGGL(__VA_ARGS__); } while (0)

namespace {

constexpr int WG = 256;           // 4 wavefronts of 64
constexpr int GRAM_MAX_BLOCKS = 512;
constexpr int GRAM_WIDE_BLOCKS = 1024;    // Gram kernels of 8 / 16-column panels (their partial blocks are summed by k_finish_wide)

inline hipStream_t cur_stream() { return (hipStream_t)pa_rt_stream(); }

char g_kerr[256];
int kfail(const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 0;
  snprintf(g_kerr, sizeof(g_kerr), "%s: %s", what, hipGetErrorString(e));
  fprintf(stderr, "[prealps_hip] kernel launch failed: %s\n", g_kerr);
  return 1;
}

typedef double mfma_d4 __attribute__((ext_vector_type(4)));   // C/D operand of v_mfma_f64_16x16x4

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}

// Load / store one panel row of TS doubles with 16-byte accesses.
template <int TS>
__device__ __forceinline__ void load_row(const double* __restrict__ p, size_t row, double (&r)[TS]) {
  const double2* q = reinterpret_cast<const double2*>(p + row * TS);
#pragma unroll
  for (int i = 0; i < TS / 2; ++i) {
    double2 v = q[i];
    r[2 * i] = v.x;
    r[2 * i + 1] = v.y;
  }
}
// The same with the nontemporal hint: a row that nobody reads again before the caches have turned over (see
// k_trsm_update) does not take a line away from the panels that are read again right away.
template <int TS>
__device__ __forceinline__ void load_row_nt(const double* __restrict__ p, size_t row, double (&r)[TS]) {
  typedef double d2v __attribute__((ext_vector_type(2)));
  if constexpr (TS >= 2) {
#pragma unroll
    for (int i = 0; i < TS / 2; ++i) {
      const d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(p + row * TS) + i);
      r[2 * i] = v.x; r[2 * i + 1] = v.y;
    }
  } else r[0] = __builtin_nontemporal_load(p + row);
}
template <int TS>
__device__ __forceinline__ void store_row_nt(double* __restrict__ p, size_t row, const double (&r)[TS]) {
  typedef double d2v __attribute__((ext_vector_type(2)));
  if constexpr (TS >= 2) {
#pragma unroll
    for (int i = 0; i < TS / 2; ++i) {
      d2v v; v.x = r[2 * i]; v.y = r[2 * i + 1];
      __builtin_nontemporal_store(v, reinterpret_cast<d2v*>(p + row * TS) + i);
    }
  } else __builtin_nontemporal_store(r[0], p + row);
}
template <int TS>
__device__ __forceinline__ void store_row(double* __restrict__ p, size_t row, const double (&r)[TS]) {
  double2* q = reinterpret_cast<double2*>(p + row * TS);
#pragma unroll
  for (int i = 0; i < TS / 2; ++i) q[i] = make_double2(r[2 * i], r[2 * i + 1]);
}

// The same for TS columns of a panel whose rows are XS doubles apart (p already points at the
// first of those columns).
template <int TS, int XS>
__device__ __forceinline__ void load_row_s(const double* __restrict__ p, size_t row, double (&r)[TS]) {
  const double2* q = reinterpret_cast<const double2*>(p + row * XS);
#pragma unroll
  for (int i = 0; i < TS / 2; ++i) {
    double2 v = q[i];
    r[2 * i] = v.x;
    r[2 * i + 1] = v.y;
  }
}
template <int TS, int XS>
__device__ __forceinline__ void store_row_s(double* __restrict__ p, size_t row, const double (&r)[TS]) {
  double2* q = reinterpret_cast<double2*>(p + row * XS);
#pragma unroll
  for (int i = 0; i < TS / 2; ++i) q[i] = make_double2(r[2 * i], r[2 * i + 1]);
}

template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// a[s] of lane c  <-  a[c] of lane s, within each quad of lanes (c = lane & 3)
__device__ __forceinline__ void quad_transpose4(double (&a)[4], int c) {
  const bool b0 = c & 1, b1 = c & 2;
#pragma unroll
  for (int p = 0; p < 4; p += 2) {          // partner lane ^ 1 (quad_perm [1,0,3,2]), registers (p, p+1)
    const double recv = dpp_mov_f64<0xB1>(b0 ? a[p] : a[p + 1]);
    a[p] = b0 ? recv : a[p];
    a[p + 1] = b0 ? a[p + 1] : recv;
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {             // partner lane ^ 2 (quad_perm [2,3,0,1]), registers (p, p+2)
    const double recv = dpp_mov_f64<0x4E>(b1 ? a[p] : a[p + 2]);
    a[p] = b1 ? recv : a[p];
    a[p + 2] = b1 ? a[p + 2] : recv;
  }
}

}  // namespace

#define TS_DISPATCH(ts, CALL)                      \
  switch (ts) {                                    \
    case 2: { constexpr int TS_ = 2; CALL; } break;   \
    case 4: { constexpr int TS_ = 4; CALL; } break;   \
    case 8: { constexpr int TS_ = 8; CALL; } break;   \
    case 16: { constexpr int TS_ = 16; CALL; } break; \
    default: snprintf(g_kerr, sizeof(g_kerr), "unsupported panel stride %d", ts); return 1; \
  }
