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

}  // namespace kmpc
