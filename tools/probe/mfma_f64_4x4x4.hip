// Probe of v_mfma_f64_4x4x4f64 on gfx950: operand lane layouts (A, B, C/D), the A-block
// broadcast modifiers (cbsz / abid) and the issue rate.  Standalone: hipcc --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int CBSZ, int ABID>
__global__ void k_one(const double* a, const double* b, double* d) {
  int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, CBSZ, ABID, 0);
}

__global__ void k_rate(double* out, int iters) {
  int l = threadIdx.x & 63;
  double a = 1.0 + l * 1e-9, b = 1.0 - l * 1e-9;
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
    c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
    c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}
__global__ void k_rate_dep(double* out, int iters) {   // one dependent chain: latency
  int l = threadIdx.x & 63;
  double a = 1e-3 + l * 1e-9, b = 1e-3 - l * 1e-9, c0 = 0;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0;
}
__global__ void k_rate16(double* out, int iters) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  int l = threadIdx.x & 63;
  double a = 1.0 + l * 1e-9, b = 1.0 - l * 1e-9;
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ void k_rate_fma(double* out, int iters) {
  int l = threadIdx.x & 63;
  double a = 1.0 + l * 1e-9, b = 1e-9;
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
  for (int i = 0; i < iters; ++i) {
    c0 = fma(a, b, c0); c1 = fma(a, b, c1); c2 = fma(a, b, c2); c3 = fma(a, b, c3);
    c4 = fma(a, b, c4); c5 = fma(a, b, c5); c6 = fma(a, b, c6); c7 = fma(a, b, c7);
    asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}

template <int CBSZ, int ABID>
static int run(const std::vector<double>& a, const std::vector<double>& b, std::vector<double>& d) {
  double *da, *db, *dd;
  CK(hipMalloc(&da, 512)); CK(hipMalloc(&db, 512)); CK(hipMalloc(&dd, 512));
  CK(hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice));
  k_one<CBSZ, ABID><<<1, 64>>>(da, db, dd);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost));
  hipFree(da); hipFree(db); hipFree(dd);
  return 0;
}

int main() {
  std::vector<double> a(64), b(64), d(64);
  // 1. which (A lane, B lane) pairs feed which D lane: unit impulses
  printf("# D lane <- list of (A lane, B lane) contributing, cbsz=0\n");
  std::vector<std::vector<std::pair<int, int>>> contrib(64);
  for (int la = 0; la < 64; ++la) {
    // all B = distinct primes-like weights so that a single run per A lane identifies the B lane
    for (int i = 0; i < 64; ++i) { a[i] = 0.0; b[i] = (double)(i + 1); }
    a[la] = 1.0;
    if (run<0, 0>(a, b, d)) return 1;
    for (int ld = 0; ld < 64; ++ld)
      if (d[ld] != 0.0) contrib[ld].push_back({la, (int)d[ld] - 1});
  }
  for (int ld = 0; ld < 64; ++ld) {
    printf("D%02d:", ld);
    for (auto& p : contrib[ld]) printf(" (A%02d,B%02d)", p.first, p.second);
    printf("\n");
  }
  // 2. broadcast of A block: cbsz=2, abid=1: expect every block to use A from block 1
  for (int i = 0; i < 64; ++i) { a[i] = 100.0 + i; b[i] = (i % 16 == 0) ? 1.0 : 0.0; }
  if (run<2, 1>(a, b, d)) return 1;
  printf("# cbsz=2 abid=1, A[i]=100+i, B=1 at lanes 0,16,32,48:\n");
  for (int ld = 0; ld < 64; ++ld) printf("%s%6.0f", ld % 16 == 0 ? "\n" : " ", d[ld]);
  printf("\n");
  if (run<0, 0>(a, b, d)) return 1;
  printf("# cbsz=0 abid=0 same inputs:\n");
  for (int ld = 0; ld < 64; ++ld) printf("%s%6.0f", ld % 16 == 0 ? "\n" : " ", d[ld]);
  printf("\n");
  // 3. rates
  double* out; CK(hipMalloc(&out, 256 * 1024 * 8 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int dev = 0; hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, dev));
  double clk = pr.clockRate * 1e3;   // Hz
  printf("# %s, %d CUs, clock %.0f MHz\n", pr.name, pr.multiProcessorCount, clk / 1e6);
  const int iters = 20000;
  for (int which = 0; which < 4; ++which) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      // one wave per SIMD: 256 CUs * 4 waves
      if (which == 0) k_rate<<<pr.multiProcessorCount, 256>>>(out, iters);
      if (which == 1) k_rate_dep<<<pr.multiProcessorCount, 256>>>(out, iters);
      if (which == 2) k_rate16<<<pr.multiProcessorCount, 256>>>(out, iters);
      if (which == 3) k_rate_fma<<<pr.multiProcessorCount, 256>>>(out, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      int per_iter = which == 0 ? 8 : which == 1 ? 4 : which == 2 ? 4 : 8;
      double cyc = ms * 1e-3 * clk / ((double)iters * per_iter);
      if (rep) printf("%s: %.3f ms, %.2f cycles per instruction per wave (nominal clock)\n",
                      which == 0 ? "mfma_f64_4x4x4 x8 independent" : which == 1 ? "mfma_f64_4x4x4 dependent chain" :
                      which == 2 ? "mfma_f64_16x16x4 x4 independent" : "v_fma_f64 x8 independent", ms, cyc);
    }
  }
  return 0;
}
