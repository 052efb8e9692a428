// lift_kernel.hip -- Koopman lifting psi(x) for a batch of states on gfx950.
//
//   MLP encoder  psi = W_d relu(... relu(W_1 x + b_1) ...) + b_d      duffing.py:17-29,
//                Revise_2/Encoder_Duffing.m:3-6, Encoder_Tank.m:3-5 (two hidden layers)
//   RBF          psi_j = d_j^2 log(d_j + eps)                          vanderpol_RBF.py:20-23
//                psi_j = r2 log(sqrt(r2)), NaN -> 0                    rbf.m:24-29
//
// MLP mapping: the batch is the GEMM N dimension.  A workgroup of four waves owns a tile of
// 16 trajectories; the hidden rows (M) are split over the four waves, each wave keeps ITS
// weight rows of every layer resident in registers as MFMA A-fragments for the whole kernel
// (weights-stationary), activations go wave -> LDS -> all waves as MFMA B-fragments.
// v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32: A lane l holds A[l&15][l>>4],
// B lane l holds B[l>>4][l&15]; bias + ReLU are fused on the accumulator registers.
#include "kernels.h"

namespace kmpc {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mma;
template <> struct Mma<double> {
  typedef d4 acc_t;
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // accumulator register r of lane l holds row (l>>4) + 4r, column l&15
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <> struct Mma<float> {
  typedef f4 acc_t;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // accumulator register r of lane l holds row 4(l>>4) + r, column l&15
  static __device__ __forceinline__ int row(int lane, int r) { return 4 * (lane >> 4) + r; }
};

constexpr int LIFT_WAVES = 4;
constexpr int LIFT_TPB = 64 * LIFT_WAVES;

// KS = K-steps of 4 over the padded hidden width (Hp = 4*KS, multiple of 16 -> KS multiple of 4),
// MT = hidden M-tiles (16 rows) per wave, NHH = number of hidden->hidden layers (layers-1).
template <typename T, int KS, int MT, int NHH>
__global__ __launch_bounds__(LIFT_TPB, 1) void lift_mlp_kernel(const LiftArgs<T> a) {
  typedef typename Mma<T>::acc_t acc_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* const sm = reinterpret_cast<T*>(smem_raw);
  const int Hp = 4 * KS;
  T* const sAct0 = sm;                 // KS*64  B-fragments of the current activations
  T* const sAct1 = sAct0 + KS * 64;    // KS*64
  T* const sW1 = sAct1 + KS * 64;      // Hp*4 (input columns zero-padded to 4)
  T* const sb1 = sW1 + Hp * 4;         // Hp
  T* const sbh = sb1 + Hp;             // NHH*Hp
  T* const sbo = sbh + NHH * Hp;       // Lp

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int n = a.n, B = a.B, L = a.L;
  const int col = lane & 15;   // trajectory within the tile
  const int kq = lane >> 4;    // k offset within a K-step / row group of the accumulator
  const int mtiles_h = Hp / 16;
  const int mtiles_o = a.Lp / 16;

  // ---- stage the small shared operands
  for (int e = tid; e < Hp * 4; e += LIFT_TPB) sW1[e] = a.W1[e];
  for (int e = tid; e < Hp; e += LIFT_TPB) sb1[e] = a.b1[e];
  for (int k = 0; k < NHH; ++k)
    for (int e = tid; e < Hp; e += LIFT_TPB) sbh[k * Hp + e] = a.bh[k][e];
  for (int e = tid; e < a.Lp; e += LIFT_TPB) sbo[e] = a.bo[e];

  // ---- this wave's weight rows as A-fragments, resident for the whole kernel
  T wh[NHH][MT][KS];
#pragma unroll
  for (int k = 0; k < NHH; ++k)
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int tile = wave + LIFT_WAVES * t;
      const int row = 16 * tile + col;  // A[i = lane&15][k = lane>>4]
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        wh[k][t][ks] = (tile < mtiles_h) ? a.Wh[k][(size_t)row * Hp + 4 * ks + kq] : T(0);
    }
  T wo[KS];
  {
    const int tile = wave;  // Lp <= 64: at most one output tile per wave
    const int row = 16 * tile + col;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wo[ks] = (tile < mtiles_o) ? a.Wo[(size_t)row * Hp + 4 * ks + kq] : T(0);
  }
  __syncthreads();

  const int ntiles = (B + 15) / 16;
  for (int bt = blockIdx.x; bt < ntiles; bt += gridDim.x) {
    const int b = bt * 16 + col;
    const bool live = b < B;
    // ---- layer 1 (K = n, VALU): every wave builds the full B-fragment set in registers
    const T x0 = (live && n > 0) ? a.X[b] : T(0);
    const T x1 = (live && n > 1) ? a.X[(size_t)B + b] : T(0);
    const T x2 = (live && n > 2) ? a.X[(size_t)2 * B + b] : T(0);
    const T x3 = (live && n > 3) ? a.X[(size_t)3 * B + b] : T(0);
    T act[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = 4 * ks + kq;
      const T* wr = sW1 + 4 * k;  // zero-padded columns: no branch on n
      const T v = sb1[k] + wr[0] * x0 + wr[1] * x1 + wr[2] * x2 + wr[3] * x3;
      act[ks] = v > T(0) ? v : T(0);
    }
    // ---- hidden -> hidden layers on MFMA
#pragma unroll
    for (int k = 0; k < NHH; ++k) {
      T* const sAct = (k & 1) ? sAct1 : sAct0;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int tile = wave + LIFT_WAVES * t;
        if (tile < mtiles_h) {
          // K split over four independent accumulator chains (the f64 MFMA's dependent latency is
          // several times its issue interval)
          acc_t ac[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) ac[c] = acc_t{T(0), T(0), T(0), T(0)};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) ac[ks & 3] = Mma<T>::mma(wh[k][t][ks], act[ks], ac[ks & 3]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * tile + Mma<T>::row(lane, r);  // hidden unit index
            T v = ((ac[0][r] + ac[1][r]) + (ac[2][r] + ac[3][r])) + sbh[k * Hp + row];
            v = v > T(0) ? v : T(0);
            // becomes element k' = row of the next layer's B operand: K-step row/4, lane (row%4)*16 + col
            sAct[(row >> 2) * 64 + ((row & 3) << 4) + col] = v;
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) act[ks] = sAct[ks * 64 + lane];
    }
    // ---- output layer
    if (wave < mtiles_o) {
      acc_t ac[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) ac[c] = acc_t{T(0), T(0), T(0), T(0)};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) ac[ks & 3] = Mma<T>::mma(wo[ks], act[ks], ac[ks & 3]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * wave + Mma<T>::row(lane, r);
        if (row < L && live)
          a.Psi[(size_t)row * a.ps_l + (size_t)b * a.ps_b] = ((ac[0][r] + ac[1][r]) + (ac[2][r] + ac[3][r])) + sbo[row];
      }
    }
    // the next tile's first LDS write (sAct0) is ordered behind this tile's barriers: with
    // NHH == 1 the single buffer is re-written only after every wave passed the barrier above
    // and consumed it into registers, so one more barrier is needed in that case only.
    if (NHH == 1) __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// RBF dictionary: one thread per trajectory, centres in LDS
// ---------------------------------------------------------------------------------------
template <typename T> __global__ __launch_bounds__(256) void lift_rbf_kernel(const LiftArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* const scx = reinterpret_cast<T*>(smem_raw);
  const int n = a.n, L = a.L, B = a.B;
  for (int e = threadIdx.x; e < L * n; e += blockDim.x) scx[e] = a.cx[e];
  __syncthreads();
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
    T x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = (i < n) ? a.X[(size_t)i * B + b] : T(0);
    for (int j = 0; j < L; ++j) {
      T v;
      if (a.rbf_matlab) {
        T r2 = T(0);
        for (int i = 0; i < n; ++i) {
          const T d = x[i] - scx[j * n + i];
          r2 += d * d;
        }
        v = r2 > T(0) ? r2 * log(sqrt(r2)) : T(0);  // NaN -> 0 (rbf.m:28)
      } else {
        // sklearn euclidean_distances: sqrt(max(|x|^2 - 2 x.c + |c|^2, 0))
        T xx = T(0), cc = T(0), xc = T(0);
        for (int i = 0; i < n; ++i) {
          const T c = scx[j * n + i];
          xx += x[i] * x[i];
          cc += c * c;
          xc += x[i] * c;
        }
        T d2 = xx - T(2) * xc + cc;
        d2 = d2 > T(0) ? d2 : T(0);
        const T d = sqrt(d2);
        v = d * d * log(d + a.eps);
      }
      a.Psi[(size_t)j * a.ps_l + (size_t)b * a.ps_b] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
template <typename T, int KS, int MT, int NHH> static hipError_t launch_mlp_impl(const LiftArgs<T>& a, hipStream_t s) {
  const int Hp = 4 * KS;
  const size_t lds = (size_t)(2 * KS * 64 + Hp * 4 + Hp + NHH * Hp + a.Lp) * sizeof(T);
  const int ntiles = (a.B + 15) / 16;
  int grid = ntiles < 1024 ? ntiles : 1024;
  hipLaunchKernelGGL((lift_mlp_kernel<T, KS, MT, NHH>), dim3(grid), dim3(LIFT_TPB), lds, s, a);
  return hipGetLastError();
}

template <typename T> hipError_t launch_lift_mlp(const LiftArgs<T>& a, hipStream_t s) {
  if (a.B <= 0) return hipSuccess;
  if (a.n > 4 || a.Lp > 64 || a.Hp > 128 || (a.nlayers != 2 && a.nlayers != 3)) return hipErrorInvalidValue;
  const int nhh = a.nlayers - 1;
  if (a.Hp <= 112) {
    if (a.Hp != 112) return hipErrorInvalidValue;
    return nhh == 2 ? launch_mlp_impl<T, 28, 2, 2>(a, s) : launch_mlp_impl<T, 28, 2, 1>(a, s);
  }
  if (a.Hp != 128) return hipErrorInvalidValue;
  return nhh == 2 ? launch_mlp_impl<T, 32, 2, 2>(a, s) : launch_mlp_impl<T, 32, 2, 1>(a, s);
}

template <typename T> hipError_t launch_lift_rbf(const LiftArgs<T>& a, hipStream_t s) {
  if (a.B <= 0) return hipSuccess;
  if (a.n > 4) return hipErrorInvalidValue;
  int grid = (a.B + 255) / 256;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL((lift_rbf_kernel<T>), dim3(grid), dim3(256), (size_t)a.L * a.n * sizeof(T), s, a);
  return hipGetLastError();
}

template hipError_t launch_lift_mlp<float>(const LiftArgs<float>&, hipStream_t);
template hipError_t launch_lift_mlp<double>(const LiftArgs<double>&, hipStream_t);
template hipError_t launch_lift_rbf<float>(const LiftArgs<float>&, hipStream_t);
template hipError_t launch_lift_rbf<double>(const LiftArgs<double>&, hipStream_t);

}  // namespace kmpc
