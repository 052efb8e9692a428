// micro-benchmark: cycles per v_mfma_f64_16x16x4_f64 (NCH independent accumulator chains, one wave per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NCH> __global__ __launch_bounds__(256) void k(double* out, long long* cyc, int iters) {
  d4 acc[NCH];
  for (int c = 0; c < NCH; ++c) acc[c] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NCH> void run(double* out, long long* cyc, int blocks) {
  int iters = 1000;
  hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  double flops = 2048.0 * NCH * iters * blocks * 4;
  printf("chains %d blocks %d: %.1f memtime-ticks per MFMA (per wave), %.2f TFLOP/s\n", NCH, blocks, (double)c / (iters * NCH), flops / ms / 1e9);
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 8 * 256 * 2048); hipMalloc(&cyc, 8);
  run<1>(out, cyc, 256); run<2>(out, cyc, 256); run<4>(out, cyc, 256); run<8>(out, cyc, 256);
  run<4>(out, cyc, 512); run<4>(out, cyc, 1);
  return 0;
}
