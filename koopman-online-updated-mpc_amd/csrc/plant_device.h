// plant_device.h -- the plants of the reference as device functions (shared by plant_kernel and the fused
// step kernel).  Duffing / Van der Pol: RK4, h = 0.05 (duffing.py:250-261, 991-992; vanderpol_RBF.py:113;
// vanderpol.py:923-931).  Cascaded tanks: a discrete map with clipping at 0 (Tank_System.m:9-10, 194-195, 211).
#pragma once
#include <hip/hip_runtime.h>

namespace kmpc {

template <typename T> __device__ __forceinline__ void plant_f(int plant, int sw, T x1, T x2, T u, T& d1, T& d2) {
  if (plant == 0) {
    if (sw) { d1 = x2; d2 = T(-10.0) * T(0.5) * x2 + T(2.0) * x1 - T(0.5) * x1 * x1 * x1 + u; }
    else    { d1 = x2; d2 = T(-0.5) * x2 + x1 - x1 * x1 * x1 + u; }
  } else {
    if (sw) { d1 = x2; d2 = T(-3.0) * x2 - T(10.0) * x1 * x1 * x2 - T(3.0) * x1 + u; }
    else    { d1 = T(2.0) * x2; d2 = T(2.0) * x2 - T(10.0) * x1 * x1 * x2 - T(0.8) * x1 + u; }
  }
}

// (x1, x2) <- plant(x1, x2, u).  Bit 4 of `plant` (KMPC_PLANT_RK4_MATLAB): the Runge-Kutta step as the MATLAB scripts write it,
// k4 = f(x + h k1) instead of f(x + h k3) (Koopman_update.m:24).
template <typename T> __device__ __forceinline__ void plant_apply(int plant, int sw, T h, T& x1, T& x2, T u) {
  const bool k4_from_k1 = (plant & 16) != 0;
  plant &= 15;
  if (plant == 2) {
    const T s1 = sqrt(x1 > T(0) ? x1 : T(0)), s2 = sqrt(x2 > T(0) ? x2 : T(0));
    T y1, y2;
    if (sw) { y1 = x1 - T(0.53) * s1 + T(0.3) * u; y2 = x2 + T(0.1) * s1 - T(0.35) * s2; }
    else    { y1 = x1 - T(0.5) * s1 + T(0.4) * u;  y2 = x2 + T(0.2) * s1 - T(0.3) * s2; }
    x1 = y1 > T(0) ? y1 : T(0);
    x2 = y2 > T(0) ? y2 : T(0);
    return;
  }
  T k1a, k1b, k2a, k2b, k3a, k3b, k4a, k4b;
  plant_f(plant, sw, x1, x2, u, k1a, k1b);
  plant_f(plant, sw, x1 + T(0.5) * h * k1a, x2 + T(0.5) * h * k1b, u, k2a, k2b);
  plant_f(plant, sw, x1 + T(0.5) * h * k2a, x2 + T(0.5) * h * k2b, u, k3a, k3b);
  plant_f(plant, sw, x1 + h * (k4_from_k1 ? k1a : k3a), x2 + h * (k4_from_k1 ? k1b : k3b), u, k4a, k4b);
  const T n1 = x1 + (h / T(6.0)) * (k1a + T(2.0) * k2a + T(2.0) * k3a + k4a);
  const T n2 = x2 + (h / T(6.0)) * (k1b + T(2.0) * k2b + T(2.0) * k3b + k4b);
  x1 = n1;
  x2 = n2;
}


// ---------------------------------------------------------------------------------------
// Natural logarithm for the thin-plate RBF dictionary (vanderpol_RBF.py:21-22, rbf.m:26-29).  Domain: positive FINITE arguments
// (normal or subnormal); +inf gives NaN (s = inf / inf) -- callers that can meet it guard it (kmpc_logT below).
// The library log() keeps its dozen polynomial coefficients in vector registers for the whole kernel once the step loop of a
// roll-out makes them loop-invariant (24 VGPRs of the 128 a trajectory has: the N = 30 roll-out spilled because of them and
// re-read them from scratch memory at every step).  Here every coefficient is written into its register where it is used, by
// an instruction the optimiser cannot move (two v_mov_b32 each: 32 instructions a step).  Algorithm: the classical reduction
// x = 2^k (1 + f), sqrt(1/2) <= 1 + f < sqrt(2), s = f / (2 + f), log(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2)) with the degree-7
// minimax polynomial R of fdlibm's e_log.c (error < 1 ulp).
// ---------------------------------------------------------------------------------------
template <unsigned HI, unsigned LO> __device__ __forceinline__ double kmpc_const() {
  int h, l;
  asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(l), "=v"(h) : "n"(LO), "n"(HI));
  return __hiloint2double(h, l);
}
__device__ __forceinline__ double kmpc_log(double x) {
  int k = __builtin_amdgcn_frexp_exp(x);          // x = m 2^k, 0.5 <= m < 1
  double m = __builtin_amdgcn_frexp_mant(x);
  if (m < kmpc_const<0x3fe6a09eu, 0x667f3bcdu>()) { m *= 2.0; k -= 1; }  // sqrt(1/2)
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s, w = z * z;
  double t1 = kmpc_const<0x3fc39a09u, 0xd078c69fu>();                    // Lg6
  t1 = __builtin_fma(w, t1, kmpc_const<0x3fcc71c5u, 0x1d8e78afu>());     // Lg4
  t1 = __builtin_fma(w, t1, kmpc_const<0x3fd99999u, 0x9997fa04u>());     // Lg2
  t1 *= w;
  double t2 = kmpc_const<0x3fc2f112u, 0xdf3e5244u>();                    // Lg7
  t2 = __builtin_fma(w, t2, kmpc_const<0x3fc74664u, 0x96cb03deu>());     // Lg5
  t2 = __builtin_fma(w, t2, kmpc_const<0x3fd24924u, 0x94229359u>());     // Lg3
  t2 = __builtin_fma(w, t2, kmpc_const<0x3fe55555u, 0x55555593u>());     // Lg1
  t2 *= z;
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)k;
  // k ln2_hi - ((hfsq - (s (hfsq + R) + k ln2_lo)) - f)
  return dk * kmpc_const<0x3fe62e42u, 0xfee00000u>() - ((hfsq - (s * (hfsq + R) + dk * kmpc_const<0x3dea39efu, 0x35793c76u>())) - f);
}
template <typename T> __device__ __forceinline__ T kmpc_logT(T x);
// (stand-alone lift: +inf -> +inf as the library log; NaN stays NaN.  Subnormal arguments need no rescaling here -- v_frexp_mant /
//  v_frexp_exp handle them natively, unlike the integer bit-twiddling of e_log.c.  The fused roll-out calls kmpc_log directly: a
//  trajectory whose state is not finite ends as status 2 there whichever non-finite value its lift carries.)
template <> __device__ __forceinline__ double kmpc_logT<double>(double x) { return x < __builtin_inf() ? kmpc_log(x) : x; }
template <> __device__ __forceinline__ float kmpc_logT<float>(float x) { return logf(x); }
// One thin-plate observable of a TWO-state plant in float64 (vanderpol_RBF.py:20-23: sklearn's euclidean distance, d^2 log(d + eps);
// rbf.m:24-29: r2 log(sqrt(r2)), NaN -> 0) as a FUNCTION OF ITS OWN, shared by the stand-alone lift and the fused roll-out: the
// logarithm's coefficients and its division are not the step loop's registers or code (cfg3: kernel 1.479 -> 1.435 ms), and both
// callers run the SAME machine code -- inlined, the two contexts contracted the sums differently in the last bit, which a saturated
// lifted-output controller turns into a flipped input.
static __device__ __attribute__((noinline)) double kmpc_rbf_psi2(double x0, double x1, double c0, double c1, double eps, int matlab) {
  if (matlab) {
    const double d0 = x0 - c0, d1 = x1 - c1;
    double r2 = 0.0;
    r2 += d0 * d0;
    r2 += d1 * d1;
    return r2 > 0.0 ? r2 * kmpc_logT<double>(sqrt(r2)) : 0.0;  // NaN -> 0 (rbf.m:28)
  }
  double xx = 0.0, cc = 0.0, xc = 0.0;
  xx += x0 * x0; cc += c0 * c0; xc += x0 * c0;
  xx += x1 * x1; cc += c1 * c1; xc += x1 * c1;
  double d2 = xx - 2.0 * xc + cc;
  d2 = d2 < 0.0 ? 0.0 : d2;  // (np.maximum(d2, 0) of sklearn's euclidean_distances: a NaN stays a NaN)
  const double d = sqrt(d2);
  return d * d * kmpc_logT<double>(d + eps);
}

}  // namespace kmpc
