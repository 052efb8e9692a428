// Round 6, MEASURED AND NOT SHIPPED: a symmetric matrix of up to 48 rows inverted by ONE wave with its rows in lanes (the form of qp_rl.h on
// all 64 lanes: a pivot is NR v_fmac_f64_dpp row_newbcast after the pivot column -- used as the pivot row, by symmetry -- has been handed
// to the four 16-lane rows by v_permlane16_swap / v_permlane32_swap).  Tried for the two Gram inverses of shared_model2_kernel (waves 0 and
// 1, no barrier): model kernel 39.2 -> 36.2 us, but the controls of test_shared_model_closed_loop_vs_oracle[32-40-12-2] went from inside to
// outside their 1e-6 bound (1.5e-5; model 2.3e-8): with the column standing in for the row the recurrence is no longer the Gauss-Jordan
// elimination of ONE matrix once rounding has made T_ij and T_ji differ, and cond(G + I/P0) ~ 1e10 amplifies it.  Reading the true row
// (2 NR v_readlane per pivot) costs what the gather saves.  This file is the stand-alone check of the routine (well-conditioned matrices:
// |A X - I| ~ 1e-13):   hipcc -O3 --offload-arch=gfx950 wave_inverse_test.hip && ./a.out      (profiles/r6_cfg4_model_kernel.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstdlib>
__device__ __forceinline__ double sm_fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  return r;
}
template <int LANE> __device__ __forceinline__ void sm_fmac_bcast(double& acc, double vec, double coef, bool first) {
  // (a VALU write of `vec` must be two wait states old before DPP reads it: the first use after the gather carries the nop)
  if (first) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(vec), "v"(coef), "n"(LANE));
  else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(vec), "v"(coef), "n"(LANE));
}
template <int NR, int l = 0> __device__ __forceinline__ void sm_row_update(double (&M)[NR], double w0, double w1, double w2, double coef) {
  if constexpr (l < NR) {
    if constexpr (l < 16) sm_fmac_bcast<l>(M[l], w0, coef, l == 0);
    else if constexpr (l < 32) sm_fmac_bcast<l - 16>(M[l], w1, coef, l == 16);
    else sm_fmac_bcast<l - 32>(M[l], w2, coef, l == 32);
    sm_row_update<NR, l + 1>(M, w0, w1, w2, coef);
  }
}
// lane i holds element i of a: every 16-lane row receives w0 = elements 0-15, w1 = 16-31, w2 = 32-47
__device__ __forceinline__ void sm_gather48(double a, double& w0, double& w1, double& w2) {
  const int lo = __double2loint(a), hi = __double2hiint(a);
  const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);  // within each half: [0] = its first row's elements, [1] = its second's
  const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  // lower half: (rl[0], rl[1]) = elements (0-15, 16-31); upper half: (32-47, 48-63).  v_permlane32_swap exchanges the upper half of its
  // first operand with the lower half of its second: (x, x) -> (lower everywhere, upper everywhere)
  const auto a0l = __builtin_amdgcn_permlane32_swap(rl[0], rl[0], false, false);
  const auto a0h = __builtin_amdgcn_permlane32_swap(rh[0], rh[0], false, false);
  const auto a1l = __builtin_amdgcn_permlane32_swap(rl[1], rl[1], false, false);
  const auto a1h = __builtin_amdgcn_permlane32_swap(rh[1], rh[1], false, false);
  w0 = __hiloint2double(a0h[0], a0l[0]);
  w2 = __hiloint2double(a0h[1], a0l[1]);
  w1 = __hiloint2double(a1h[0], a1l[0]);
}
template <int NR, int K = 0>
__device__ __forceinline__ void sm_wave_sweep_all(double (&M)[NR], double& rs, double& rsi, int n, int lane) {
  if constexpr (K < NR) {
    if (K < n) {  // (uniform)
      const double colk = M[K];
      const double tk = rs * colk;  // the true column K = row K
      const double d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tk), K), __builtin_amdgcn_readlane(__double2loint(tk), K));
      const double dinv = sm_fast_rcp(d);
      double w0, w1, w2;
      sm_gather48(tk, w0, w1, w2);
      const bool isk = lane == K;
      const double c = isk ? 0.0 : -colk * dinv;  // M_i[j] += c_i T_kj  (stored rows: the row scale cancels)
      sm_row_update<NR>(M, w0, w1, w2, c);
      M[K] = isk ? -rsi : colk * dinv;            // column K; (K, K): -1/d = rs_new * (-1 / rs_old)
      if (isk) { rs *= dinv; rsi *= d; }
    }
    sm_wave_sweep_all<NR, K + 1>(M, rs, rsi, n, lane);
  }
}
// A^-1 of the n x n symmetric positive definite matrix src (leading dimension lds) into dst, by the calling wave alone
template <int NR>
__device__ __forceinline__ void sm_wave_inverse(const double* src, double* dst, int lds, int n, int lane) {
  double M[NR];
  const bool mine = lane < n;
#pragma unroll
  for (int j = 0; j < NR; ++j) M[j] = (mine && j < n) ? src[lane * lds + j] : 0.0;
  double rs = 1.0, rsi = 1.0;
  sm_wave_sweep_all<NR>(M, rs, rsi, n, lane);
  if (mine) {
#pragma unroll
    for (int j = 0; j < NR; ++j)
      if (j < n) dst[lane * lds + j] = -(rs * M[j]);  // (the sweeps leave -(.)^-1)
  }
}


template <int NR> __global__ void k(const double* src, double* dst, int lds, int n) { sm_wave_inverse<NR>(src, dst, lds, n, threadIdx.x); }
int main() {
  for (int n : {5, 16, 17, 32, 33}) {
    const int lds = 49;
    std::vector<double> A(48 * lds, 0.0), X(48 * lds, 0.0);
    srand(n);
    std::vector<double> R(n * n);
    for (auto& v : R) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = (i == j) ? 1e-3 : 0.0; for (int k2 = 0; k2 < n; ++k2) s += R[i * n + k2] * R[j * n + k2]; A[i * lds + j] = s; }
    double *dA, *dX;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dX, X.size() * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemset(dX, 0, X.size() * 8);
    if (n <= 18) hipLaunchKernelGGL(k<18>, dim3(1), dim3(64), 0, 0, dA, dX, lds, n); else hipLaunchKernelGGL(k<34>, dim3(1), dim3(64), 0, 0, dA, dX, lds, n);
    hipDeviceSynchronize();
    hipMemcpy(X.data(), dX, X.size() * 8, hipMemcpyDeviceToHost);
    double worst = 0.0;  // |A X - I|
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0.0; for (int k2 = 0; k2 < n; ++k2) s += A[i * lds + k2] * X[k2 * lds + j]; worst = fmax(worst, fabs(s - (i == j ? 1.0 : 0.0))); }
    printf("n = %d: max |A X - I| = %.3e  (%s)\n", n, worst, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
