// micro-benchmark: issue rate of v_fmac_f64_dpp (row_newbcast) vs plain v_fmac_f64, 8 independent chains per lane
#include <hip/hip_runtime.h>
#include <cstdio>
template <bool DPP> __global__ __launch_bounds__(256) void k(double* out, int iters) {
  double acc[8];
  for (int c = 0; c < 8; ++c) acc[c] = threadIdx.x * 1e-3 + c;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1e-7 * threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if constexpr (DPP) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[c]) : "v"(b), "v"(a));
      else asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc[c]) : "v"(b), "v"(a));
    }
  }
  double s = 0;
  for (int c = 0; c < 8; ++c) s += acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <bool DPP> void run(double* out, int blocks) {
  int iters = 4000;
  hipLaunchKernelGGL(k<DPP>, dim3(blocks), dim3(256), 0, 0, out, iters);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<DPP>, dim3(blocks), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double flops = 2.0 * 8 * iters * blocks * 256.0;
  printf("%s blocks %4d: %.2f TFLOP/s\n", DPP ? "v_fmac_f64_dpp" : "v_fmac_f64    ", blocks, flops / ms / 1e9);
}
int main() {
  double* out; (void)hipMalloc(&out, 8 * 256 * 4096);
  for (int rep = 0; rep < 2; ++rep) { run<false>(out, 1024); run<true>(out, 1024); run<false>(out, 2048); run<true>(out, 2048); }
  return 0;
}
