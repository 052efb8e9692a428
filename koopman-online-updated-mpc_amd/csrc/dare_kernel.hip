// dare_kernel.hip -- terminal ingredients: the reference's discrete Riccati iteration and LQR gain
// (solve_DARE / dlqr, duffing.py:583-613; called on the lifted model at duffing.py:667-671 and, in the MATLAB
// controller, every step: Koopman_update.m:215) batched over models, one workgroup per model, float64.
//
//   X_0 = Q;   X_{k+1} = A'X_k A - A'X_k B (R + B'X_k B)^+ B'X_k A + Q;   stop after the first k with
//   max|X_{k+1} - X_k| < eps (the iterate X_{k+1} is returned) or after maxiter iterations          duffing.py:587-598
//   K = (B'XB + R)^+ (B'XA)                                                                          duffing.py:611
//   optional terminal block of Q_bar:  P_N = Co X Co'                                                 Koopman_update.m:381
// m = 1 (one input, as everywhere on this path: duffing.py:170-171), so the pseudo-inverse is a reciprocal (0 -> 0).
// The products are written without assuming that X is symmetric, as the reference evaluates them.
#include "kernels.h"

namespace kmpc {

// LDS: A, X, M (L*L each) + g, w, hu, hv (L each) + reduction scratch
__global__ __launch_bounds__(256) void dare_kernel(const DareArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dare_smem[];
  double* const sA = reinterpret_cast<double*>(dare_smem);
  const int L = a.L, LL = L * L, tid = threadIdx.x, nt = blockDim.x;
  double* const sX = sA + LL;
  double* const sM = sX + LL;
  double* const sg = sM + LL;   // X B
  double* const sw = sg + L;    // X'B
  double* const shu = sw + L;   // A'X B
  double* const shv = shu + L;  // (B'X A)'
  double* const red = shv + L;  // 8 partial maxima + s
  const int m = blockIdx.x;
  const double* A = a.A + (size_t)m * (a.shared_model ? 0 : a.strideA);
  const double* Bv = a.B + (size_t)m * (a.shared_model ? 0 : a.strideB);
  double* const sB = red + 16;
  for (int e = tid; e < LL; e += nt) {
    const int i = e / L, j = e - i * L;
    sA[e] = A[(size_t)i * a.ldA + j];
    sX[e] = a.Q[e];
  }
  for (int i = tid; i < L; i += nt) sB[i] = Bv[(size_t)i * a.incB];
  __syncthreads();

  int it = 0;
  double s = 0.0;
  auto products = [&]() {  // g = X B, w = X'B, s = R + B'g, hu = A'g, hv = A'w
    for (int i = tid; i < 2 * L; i += nt) {
      const int r = i < L ? i : i - L;
      double acc = 0.0;
      if (i < L) { for (int j = 0; j < L; ++j) acc += sX[r * L + j] * sB[j]; sg[r] = acc; }
      else { for (int j = 0; j < L; ++j) acc += sX[j * L + r] * sB[j]; sw[r] = acc; }
    }
    __syncthreads();
    for (int i = tid; i < 2 * L + 1; i += nt) {
      double acc = 0.0;
      if (i < L) { for (int j = 0; j < L; ++j) acc += sA[j * L + i] * sg[j]; shu[i] = acc; }
      else if (i < 2 * L) { const int r = i - L; for (int j = 0; j < L; ++j) acc += sA[j * L + r] * sw[j]; shv[r] = acc; }
      else { for (int j = 0; j < L; ++j) acc += sB[j] * sg[j]; red[8] = a.R + acc; }
    }
    __syncthreads();
    s = red[8];
  };
  for (; it < a.maxiter;) {
    products();
    const double sinv = s != 0.0 ? 1.0 / s : 0.0;  // pinv of a scalar
    // M = X A
    for (int e = tid; e < LL; e += nt) {
      const int i = e / L, j = e - i * L;
      double acc = 0.0;
      for (int k = 0; k < L; ++k) acc += sX[i * L + k] * sA[k * L + j];
      sM[e] = acc;
    }
    __syncthreads();
    // Xn = A'M - hu hv' / s + Q ; diff = max |Xn - X|
    double dmax = 0.0;
    double xn[16];  // L*L / 256 <= 16 for L <= 64
    int c = 0;
    for (int e = tid; e < LL; e += nt, ++c) {
      const int i = e / L, j = e - i * L;
      double acc = 0.0;
      for (int k = 0; k < L; ++k) acc += sA[k * L + i] * sM[k * L + j];
      const double v = (acc - (shu[i] * sinv) * shv[j]) + a.Q[e];
      const double d = fabs(v - sX[e]);
      dmax = d > dmax || !(d == d) ? d : dmax;  // (a NaN ends the iteration: it never satisfies < eps, maxiter does)
      xn[c] = v;
    }
    for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(dmax, o, 64); dmax = (t > dmax || !(t == t)) ? t : dmax; }
    __syncthreads();  // everyone has read X
    if ((tid & 63) == 0) red[tid >> 6] = dmax;
    c = 0;
    for (int e = tid; e < LL; e += nt, ++c) sX[e] = xn[c];
    __syncthreads();
    double dm = 0.0;
    for (int i = 0; i < (nt >> 6); ++i) dm = (red[i] > dm || !(red[i] == red[i])) ? red[i] : dm;
    ++it;
    if (dm < a.eps) break;
  }
  double* P = a.P + (size_t)m * LL;
  for (int e = tid; e < LL; e += nt) P[e] = sX[e];
  if (a.iters && tid == 0) a.iters[m] = it;
  if (a.K || a.PN) {
    products();  // with the returned X
    const double sinv = s != 0.0 ? 1.0 / s : 0.0;
    if (a.K) for (int j = tid; j < L; j += nt) a.K[(size_t)m * L + j] = sinv * shv[j];
    if (a.PN) {  // P_N = Co X Co'   (Co: q rows of C)
      const double* Co = a.Co + (size_t)m * (a.shared_model ? 0 : a.strideC);
      const int q = a.q;
      for (int e = tid; e < q * L; e += nt) {  // T = Co X  -> sM
        const int r = e / L, j = e - r * L;
        double acc = 0.0;
        for (int k = 0; k < L; ++k) acc += Co[r * L + k] * sX[k * L + j];
        sM[e] = acc;
      }
      __syncthreads();
      for (int e = tid; e < q * q; e += nt) {
        const int r = e / q, c2 = e - r * q;
        double acc = 0.0;
        for (int k = 0; k < L; ++k) acc += sM[r * L + k] * Co[c2 * L + k];
        a.PN[(size_t)m * q * q + e] = acc - (r == c2 ? a.pn_sub_diag : 0.0);
      }
    }
  }
}

hipError_t launch_dare(const DareArgs& a, hipStream_t s) {
  if (a.nb <= 0) return hipSuccess;
  if (a.L < 1 || a.L > 64) return hipErrorInvalidValue;
  const size_t lds = sizeof(double) * ((size_t)3 * a.L * a.L + 5 * a.L + 16);
  static size_t configured_dev[16] = {};  // (function attributes are per device)
  size_t& configured = configured_dev[device_slot()];
  if (lds > 64 * 1024 && lds > configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dare_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    configured = lds;
  }
  hipLaunchKernelGGL(dare_kernel, dim3(a.nb), dim3(256), lds, s, a);
  return hipGetLastError();
}

}  // namespace kmpc
