// Probe (gfx950): semantics of v_fmac_f64_dpp row_newbcast and v_permlane16/32_swap as used by the condense chains.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int L> __device__ __forceinline__ void fmac_bc(double& acc, double vec, double row) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(vec), "v"(row), "n"(L));
}
__global__ void k(double* out, int* sw) {
  const int lane = threadIdx.x;
  double v = 100.0 + lane;  // value held by each lane
  double a = 0.0, b = 0.0;
  asm volatile("s_nop 1");
  fmac_bc<3>(a, v, 1.0);   // expect 100 + 16*(lane/16) + 3
  fmac_bc<15>(b, v, 2.0);  // expect 2*(100 + 16*(lane/16) + 15)
  out[lane] = a;
  out[64 + lane] = b;
  auto r = __builtin_amdgcn_permlane16_swap(lane, lane + 1000, false, false);
  auto r2 = __builtin_amdgcn_permlane32_swap(lane, lane + 1000, false, false);
  sw[lane] = r[0]; sw[64 + lane] = r[1]; sw[128 + lane] = r2[0]; sw[192 + lane] = r2[1];
}
int main() {
  double* d; int* s;
  hipMalloc(&d, 128 * 8); hipMalloc(&s, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, s);
  double h[128]; int hs[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hs, s, sizeof(hs), hipMemcpyDeviceToHost);
  printf("newbcast:3 :"); for (int i = 0; i < 64; i += 5) printf(" [%d]=%.0f", i, h[i]); printf("\n");
  printf("newbcast:15:"); for (int i = 0; i < 64; i += 5) printf(" [%d]=%.0f", i, h[64 + i]); printf("\n");
  const char* nm[4] = {"p16 r0", "p16 r1", "p32 r0", "p32 r1"};
  for (int k2 = 0; k2 < 4; ++k2) { printf("%s:", nm[k2]); for (int i = 0; i < 64; i += 8) printf(" [%d]=%d", i, hs[64 * k2 + i]); printf("\n"); }
  return 0;
}
