// Probe of v_permlane16_swap / v_permlane32_swap and DPP row_ror on gfx950: which lanes end up where.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned l = threadIdx.x;
  unsigned a = l, b = 100 + l;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[l] = r[0]; out[64 + l] = r[1];
  auto s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[128 + l] = s[0]; out[192 + l] = s[1];
  out[256 + l] = __builtin_amdgcn_update_dpp(0, (int)l, 0x124, 0xF, 0xF, false);   // row_ror:4
  out[320 + l] = __builtin_amdgcn_update_dpp(999, (int)l, 0x124, 0xF, 0x2, false); // row_ror:4, bank_mask 0b0010
}
int main() {
  unsigned* d; hipMalloc(&d, 384 * 4);
  k<<<1, 64>>>(d);
  unsigned h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[6] = {"permlane16_swap vdst (a=lane, b=100+lane)", "permlane16_swap src", "permlane32_swap vdst", "permlane32_swap src", "row_ror:4 of lane id", "row_ror:4 bank_mask 2 (old 999)"};
  for (int t = 0; t < 6; ++t) { printf("%s:\n", names[t]); for (int i = 0; i < 64; ++i) printf("%4u%s", h[64 * t + i], i % 16 == 15 ? "\n" : ""); }
  return 0;
}
