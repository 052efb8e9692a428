// micro-benchmark + layout probe: v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks per instruction)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned long long* out) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      unsigned long long m = __ballot(d != 0.0);
      if (lane == 0) out[la * 64 + lb] = m;
    }
}
template <int NCH> __global__ __launch_bounds__(256) void k(double* out, long long* cyc, int iters) {
  double acc[NCH];
  for (int c = 0; c < NCH; ++c) acc[c] = 0.0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int c = 0; c < NCH; ++c) s += acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
// one wave per SIMD issues MFMAs (16x16x4 if BIG else 4x4x4), a second wave per SIMD issues dependent-free f64 FMAs
typedef double d4 __attribute__((ext_vector_type(4)));
template <int BIG> __global__ __launch_bounds__(512) void mix(double* out, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  double s = 0;
  if (wave < 4 && (mode & 1)) {
    if (BIG) {
      d4 acc[4];
      for (int c = 0; c < 4; ++c) acc[c] = d4{0, 0, 0, 0};
      for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
      for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    } else {
      double acc[4] = {0, 0, 0, 0};
      for (int i = 0; i < 4 * iters; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
      for (int c = 0; c < 4; ++c) s += acc[c];
    }
  } else if (wave >= 4 && (mode & 2)) {
    double f[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
    for (int i = 0; i < iters * 8; ++i)
#pragma unroll
      for (int c = 0; c < 8; ++c) f[c] = __builtin_fma(f[c], 0.999, b);
    for (int c = 0; c < 8; ++c) s += f[c];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
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
  double flops = 512.0 * NCH * iters * blocks * 4;
  printf("4x4x4_4b chains %d blocks %d: %.1f memtime-ticks per MFMA (per wave), %.2f TFLOP/s, %.3f ms\n", NCH, blocks, (double)c / (iters * NCH), flops / ms / 1e9, ms);
}
template <int BIG> void runmix(double* out, int mode) {
  int iters = 2000;
  hipLaunchKernelGGL(mix<BIG>, dim3(256), dim3(512), 0, 0, out, iters, mode);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(mix<BIG>, dim3(256), dim3(512), 0, 0, out, iters, mode);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("mix big=%d mode=%d (1 = mfma waves, 2 = fma waves, 3 = both): %.3f ms  (mfma %d per wave, fma %d per wave)\n", BIG, mode, ms, BIG ? 4 * iters : 16 * iters, 64 * iters);
}
int main() {
  double* out; long long* cyc; unsigned long long* pm;
  hipMalloc(&out, 8 * 512 * 2048); hipMalloc(&cyc, 8); hipMalloc(&pm, 8 * 64 * 64);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, pm);
  static unsigned long long h[64 * 64];
  hipMemcpy(h, pm, sizeof(h), hipMemcpyDeviceToHost);
  // for each A lane: which B lanes pair with it and where the product lands
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb]) {
      printf(" B%d->", lb);
      for (int l = 0; l < 64; ++l) if ((h[la * 64 + lb] >> l) & 1ull) printf("D%d", l);
    }
    printf("\n");
  }
  run<1>(out, cyc, 256); run<2>(out, cyc, 256); run<4>(out, cyc, 256); run<8>(out, cyc, 256);
  for (int m = 1; m <= 3; ++m) runmix<1>(out, m);
  for (int m = 1; m <= 3; ++m) runmix<0>(out, m);
  return 0;
}
