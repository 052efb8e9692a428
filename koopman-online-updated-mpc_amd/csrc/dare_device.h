// dare_device.h -- the reference's Riccati iteration (solve_DARE, duffing.py:583-598) by ONE wave on one trajectory's model, for the
// per-step terminal refresh inside the fused roll-out (rollout_kernel<.., TERM = true>; Koopman_update.m:215, 381).
//
//   X_0 = Q;   X_{k+1} = A'X_k A - A'X_k B (R + B'X_k B)^+ B'X_k A + Q;   stop after the first k with max|X_{k+1} - X_k| < eps
//   (the iterate X_{k+1} is kept) or after maxiter iterations;   W = Co X Co' - Qw I   (the step kernels take P_N - Qw I)
//
// Element for element the arithmetic of dare_kernel (dare_kernel.hip: same products, same order of summation), so that a block
// refreshed inside a launch equals the block kmpc_terminal_from_dare computes for the same model.  The work arrays live in a
// per-trajectory block of global memory (RolloutArgs::term_scratch; L1 / L2 resident while the wave works on it): the refresh is a
// rare, long operation (tens of iterations of three L^3 products) beside a step whose LDS budget is what sets the residency.
#pragma once
#include "step_body.h"

namespace kmpc {

// MODEL(i, j): element (i, j) of [A B] (j = L: B);  COUT(r, j): row r of the output map Co (r < q)
template <typename MODEL, typename COUT>
__device__ __forceinline__ int wave_dare(const int L, const int q, MODEL&& model, COUT&& cout, const double* __restrict__ Qm, const double Rw,
                                         const double eps, const int maxiter, const double sub_diag, double* const scr, double* const Wout) {
  const int tid = local_tid<64>(), LL = L * L, nt = 64;
  double* const sA = scr;
  double* const sX = sA + LL;
  double* const sM = sX + LL;
  double* const sg = sM + LL;   // X B
  double* const sw = sg + L;    // X'B
  double* const shu = sw + L;   // A'X B
  double* const shv = shu + L;  // (B'X A)'
  double* const sB = shv + L;
  double* const red = sB + L;   // [8]: s
  double* const sXn = red + 16;
  auto sync = [&]() { __threadfence_block(); block_sync<64>(); };
  for (int e = tid; e < LL; e += nt) {
    const int i = e / L, j = e - i * L;
    sA[e] = model(i, j);
    sX[e] = Qm[e];
  }
  for (int i = tid; i < L; i += nt) sB[i] = model(i, L);
  sync();
  int it = 0;
  double s = 0.0;
  auto products = [&]() {  // g = X B, w = X'B, s = R + B'g, hu = A'g, hv = A'w
    for (int i = tid; i < 2 * L; i += nt) {
      const int r = i < L ? i : i - L;
      double acc = 0.0;
      if (i < L) { for (int j = 0; j < L; ++j) acc += sX[r * L + j] * sB[j]; sg[r] = acc; }
      else { for (int j = 0; j < L; ++j) acc += sX[j * L + r] * sB[j]; sw[r] = acc; }
    }
    sync();
    for (int i = tid; i < 2 * L + 1; i += nt) {
      double acc = 0.0;
      if (i < L) { for (int j = 0; j < L; ++j) acc += sA[j * L + i] * sg[j]; shu[i] = acc; }
      else if (i < 2 * L) { const int r = i - L; for (int j = 0; j < L; ++j) acc += sA[j * L + r] * sw[j]; shv[r] = acc; }
      else { for (int j = 0; j < L; ++j) acc += sB[j] * sg[j]; red[8] = Rw + acc; }
    }
    sync();
    s = red[8];
  };
  for (; it < maxiter;) {
    products();
    const double sinv = s != 0.0 ? 1.0 / s : 0.0;  // pinv of a scalar
    for (int e = tid; e < LL; e += nt) {  // M = X A
      const int i = e / L, j = e - i * L;
      double acc = 0.0;
      for (int k = 0; k < L; ++k) acc += sX[i * L + k] * sA[k * L + j];
      sM[e] = acc;
    }
    sync();
    // Xn = A'M - hu hv' / s + Q ; diff = max |Xn - X|
    double dmax = 0.0;
    for (int e = tid; e < LL; e += nt) {
      const int i = e / L, j = e - i * L;
      double acc = 0.0;
      for (int k = 0; k < L; ++k) acc += sA[k * L + i] * sM[k * L + j];
      const double v = (acc - (shu[i] * sinv) * shv[j]) + Qm[e];
      const double d = fabs(v - sX[e]);
      dmax = d > dmax || !(d == d) ? d : dmax;  // (a NaN ends the iteration: it never satisfies < eps, maxiter does)
      sXn[e] = v;  // (the new iterate waits in a second buffer until every lane has read X and M)
    }
    for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(dmax, o, 64); dmax = (t > dmax || !(t == t)) ? t : dmax; }
    sync();  // everyone has read X and M
    for (int e = tid; e < LL; e += nt) sX[e] = sXn[e];
    sync();
    ++it;
    if (dmax < eps) break;
  }
  // W = Co X Co' - sub_diag I   (T = Co X -> sM)
  for (int e = tid; e < q * L; e += nt) {
    const int r = e / L, j = e - r * L;
    double acc = 0.0;
    for (int k = 0; k < L; ++k) acc += cout(r, k) * sX[k * L + j];
    sM[e] = acc;
  }
  sync();
  for (int e = tid; e < q * q; e += nt) {
    const int r = e / q, c2 = e - r * q;
    double acc = 0.0;
    for (int k = 0; k < L; ++k) acc += sM[r * L + k] * cout(c2, k);
    Wout[e] = acc - (r == c2 ? sub_diag : 0.0);
  }
  sync();
  return it;
}

}  // namespace kmpc
