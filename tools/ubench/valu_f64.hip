// micro-benchmark: v_fma_f64 throughput (NCH independent chains per lane)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NCH> __global__ __launch_bounds__(256) void k(double* out, int iters) {
  double acc[NCH];
  for (int c = 0; c < NCH; ++c) acc[c] = threadIdx.x * 1e-3 + c;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1e-7;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = __builtin_fma(acc[c], a, b);
  }
  double s = 0;
  for (int c = 0; c < NCH; ++c) s += acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH> void run(double* out, int blocks) {
  int iters = 4000;
  hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(256), 0, 0, out, iters);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double flops = 2.0 * NCH * iters * blocks * 256.0;
  printf("chains %2d blocks %4d: %.2f TFLOP/s\n", NCH, blocks, flops / ms / 1e9);
}
int main() {
  double* out; (void)hipMalloc(&out, 8 * 256 * 4096);
  run<1>(out, 256); run<4>(out, 256); run<8>(out, 256); run<16>(out, 256); run<8>(out, 512); run<8>(out, 1024); run<8>(out, 2048);
  return 0;
}
