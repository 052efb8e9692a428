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
#include "plant_device.h"
#include <cstdlib>

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

  // ---- this wave's weight rows as A-fragments, resident for the whole kernel.  Read from the packed form
  //      [tile][k-step][lane]: every fragment is one coalesced wave load and all of them are in flight together (the
  //      row-major form made each load 16 segments of 32 bytes: 27 us for B = 4096, nearly all of it this prologue).
  const int KSp = a.KSp;
  T wh[NHH][MT][KS];
#pragma unroll
  for (int k = 0; k < NHH; ++k)
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int tile = wave + LIFT_WAVES * t;
      const int tl = tile < mtiles_h ? tile : 0;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const T v = a.Whp[k][((size_t)tl * KSp + (ks < KSp ? ks : 0)) * 64 + lane];  // (unconditional loads, clamped indices)
        wh[k][t][ks] = (tile < mtiles_h && ks < KSp) ? v : T(0);
      }
    }
  T wo[KS];
  {
    const int tile = wave;  // Lp <= 64: at most one output tile per wave
    const int tl = tile < mtiles_o ? tile : 0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const T v = a.Wop[((size_t)tl * KSp + (ks < KSp ? ks : 0)) * 64 + lane];
      wo[ks] = (tile < mtiles_o && ks < KSp) ? v : T(0);
    }
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
// lift_coop_kernel (float64): the cooperative encoder of the fused roll-out (rollout_kernel.hip) as a kernel of its own.
// A workgroup of 16 waves lifts 16 trajectories: wave w owns hidden M tile w for the whole K range (two alternating
// accumulator chains), bias + ReLU on the accumulator registers, results written straight into the next layer's B-fragment
// layout (one barrier per layer); the weights are packed A-fragments, a fragment is one 512-byte wave load from L2, fetched
// in two alternating batches of eight k-steps, the first batch of a layer a layer ahead.  The weights-stationary kernel
// above keeps 180 KB of fragments in registers per workgroup and walks 140 MFMAs per wave one after the other: 27 us for a
// tile, whatever the batch (6.8 TFLOP/s at B = 4096); here a tile is ~6 us of the same arithmetic spread over 7 + 2 waves.
// ---------------------------------------------------------------------------------------
constexpr int LC_ACT = 32 * 64;  // B-fragments of one activation vector set (Hp <= 128)
constexpr int LC_KB = 8;
__device__ __forceinline__ void lc_load_afrags(const double* Wp, int KS, int tile, int ks0, int lane, double (&af)[LC_KB]) {
#pragma unroll
  for (int i = 0; i < LC_KB; ++i) {
    const int ks = ks0 + i;
    af[i] = ks < KS ? Wp[((size_t)tile * KS + ks) * 64 + lane] : 0.0;
  }
}
// GRAM (round 4): the Gram sums of the step's transitions (gram_kernel: rows [psi_prev; u_prev; psi_now; x_now] against [psi_prev;
// u_prev], Koopman_update.m:94-98) in the same launch -- the sixteen trajectories a workgroup has just lifted are four k-steps of the
// Gram tiles, wave t keeps output tile t in its accumulator across the workgroup's trajectory tiles and writes it as the workgroup's
// partial when the kernel ends (summed in a fixed order by gram_reduce_kernel, as before).  One launch and one round trip of psi through
// memory less per shared-model step (lift 12.2 + Gram 10.4 us -> one kernel).
// Round 6: CT column tiles per pass.  With CT = 2 a wave multiplies every weight fragment it has fetched with the activations of TWO
// tiles (32 trajectories per workgroup and pass): the same MFMAs, half the passes, half the fragment traffic -- used where a workgroup
// would otherwise walk two or more tiles one after the other (cfg4: 512 tiles on 256 workgroups; 16.9 -> 15.3 us,
// profiles/r6_cfg4_lift_gram.txt).  DEV_SKIP: measurement builds only (see the profile record).
template <int KS_, bool GRAM, int CT>
__global__ __launch_bounds__(1024) void lift_coop_kernel(const LiftArgs<double> a, const GramArgs<double> g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int NC = 16 * CT;          // trajectories per pass
  constexpr int GLD = NC + 1;          // leading dimension of the Gram panel [rows][NC trajectories]
  double* const sAct0 = reinterpret_cast<double*>(smem_raw);   // [CT][LC_ACT]
  double* const sAct1 = sAct0 + CT * LC_ACT;
  double* const sXn = sAct1 + CT * LC_ACT;  // [CT][16 x 4]
  double* const sR = sXn + CT * 64;         // GRAM: [Rp][GLD] panel of the transitions of the pass
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int B = a.B, n = a.n, L = a.L;
  const int KS = KS_ > 0 ? KS_ : a.KSp, Hp = a.Hp, MTH = Hp >> 4, MTO = a.Lp >> 4;
  const int nhh = a.nlayers - 1;
  const bool hid = wv < MTH, out = wv < MTO;
  // GRAM: p = L + 1 regressor rows, R = p + L + n rows in all; tile wv = (mt, nt) of the (R x p) block
  const int gp = L + 1, gR = gp + L + n, gMT = (gR + 15) >> 4, gNT = (gp + 15) >> 4;
  const int gmt = wv / gNT, gnt = wv - gmt * gNT;
  const bool gmine = GRAM && wv < gMT * gNT;
  d4 gacc = {0.0, 0.0, 0.0, 0.0};
  for (int bt = blockIdx.x; bt * NC < B; bt += gridDim.x) {
    const int b0 = bt * NC;
    __syncthreads();  // (the previous pass's activations have been consumed)
    if (tid < 64 * CT) {
      const int c = tid >> 2, i = tid & 3, b = b0 + c;  // (column c of the pass: tile c >> 4, its column c & 15)
      sXn[tid] = (b < B && i < n) ? a.X[(size_t)i * B + b] : 0.0;
    }
    if constexpr (GRAM) {
      // rows 0 .. L-1: psi_prev, row L: u_prev, rows p + L ..: x_now (psi_now is written by the output layer below); zeros beyond B / R
      for (int e = tid; e < gMT * 16 * NC; e += 1024) {
        const int r = e / NC, c = e - r * NC, b = b0 + c;
        double v = 0.0;
        if (b < B) {
          if (r < L) v = g.psi_prev[(size_t)r * g.pp_sl + (size_t)b * g.pp_sb];
          else if (r == L) v = g.u_prev[b];
          else if (r >= gp + L && r < gR) v = a.X[(size_t)(r - gp - L) * B + b];
        }
        if (r < gp || r >= gp + L) sR[r * GLD + c] = v;
      }
    }
    double af[2][LC_KB];
    if (nhh > 0) { if (hid) lc_load_afrags(a.Whp[0], KS, wv, 0, lane, af[0]); }
    else if (out) lc_load_afrags(a.Wop, KS, wv, 0, lane, af[0]);
    // layer 1 (K = n <= 4: one k-step, W1 zero-padded to 4 columns): one MFMA per hidden tile and column tile, bias as the accumulator input
    double a1 = 0.0;
    d4 c1 = {0.0, 0.0, 0.0, 0.0};
    if (hid) {
      a1 = a.W1[4 * (16 * wv + (lane & 15)) + (lane >> 4)];
#pragma unroll
      for (int r = 0; r < 4; ++r) c1[r] = a.b1[16 * wv + (lane >> 4) + 4 * r];
    }
    __syncthreads();
    if (hid) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const d4 c = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, sXn[ct * 64 + 4 * (lane & 15) + (lane >> 4)], c1, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * wv + (lane >> 4) + 4 * r;
          sAct0[ct * LC_ACT + (row >> 2) * 64 + ((row & 3) << 4) + (lane & 15)] = c[r] > 0.0 ? c[r] : 0.0;
        }
      }
    }
    __syncthreads();
    for (int h = 0; h <= nhh; ++h) {
      const bool last = h == nhh;
      const bool mine = last ? out : hid;
      const double* act = (h & 1) ? sAct1 : sAct0;
      double* actn = (h & 1) ? sAct0 : sAct1;
      const double* Wp = last ? a.Wop : a.Whp[h & 1];
      const double* bias = last ? a.bo : a.bh[h & 1];
      d4 acc0[CT], acc1[CT];
      if (mine) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          acc1[ct] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int r = 0; r < 4; ++r) acc0[ct][r] = bias[16 * wv + (lane >> 4) + 4 * r];
        }
#pragma unroll
        for (int bt2 = 0; bt2 < 32 / LC_KB; ++bt2) {
          const int kb = bt2 * LC_KB;
          if (kb + LC_KB < KS) lc_load_afrags(Wp, KS, wv, kb + LC_KB, lane, af[(bt2 + 1) & 1]);
#pragma unroll
          for (int i = 0; i < LC_KB; i += 2) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {  // (every fragment meets the activations of all the pass's column tiles)
              if (kb + i < KS) acc0[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[bt2 & 1][i], act[ct * LC_ACT + (kb + i) * 64 + lane], acc0[ct], 0, 0, 0);
              if (kb + i + 1 < KS) acc1[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[bt2 & 1][i + 1], act[ct * LC_ACT + (kb + i + 1) * 64 + lane], acc1[ct], 0, 0, 0);
            }
          }
        }
        const int col = lane & 15;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * wv + (lane >> 4) + 4 * r;
            const double v = acc0[ct][r] + acc1[ct][r];
            const int bc = b0 + 16 * ct + col;
            if (last) {
              if (row < L && bc < B) a.Psi[(size_t)row * a.ps_l + (size_t)bc * a.ps_b] = v;
              if constexpr (GRAM) { if (row < L) sR[(gp + row) * GLD + 16 * ct + col] = bc < B ? v : 0.0; }
            } else actn[ct * LC_ACT + (row >> 2) * 64 + ((row & 3) << 4) + col] = v > 0.0 ? v : 0.0;
          }
        }
      }
      if (!last) {  // the next layer's first fragments travel across the barrier
        if (h + 1 < nhh) { if (hid) lc_load_afrags(a.Whp[(h + 1) & 1], KS, wv, 0, lane, af[0]); }
        else if (out) lc_load_afrags(a.Wop, KS, wv, 0, lane, af[0]);
        __syncthreads();
      }
    }
    if constexpr (GRAM) {
      __syncthreads();  // (psi_now of the pass's trajectories is in the panel)
      if (gmine) {
        // out[i][j] += sum_k R[i][k] Z[j][k]: A lane l: R[16 mt + (l & 15)][4 ks + (l >> 4)], B lane l: Z[16 nt + (l & 15)][4 ks + (l >> 4)]
        // (the transitions are summed in the order of their index whatever CT is: k-steps 0 .. 4 CT - 1 of pass after pass)
        const double* const pa = sR + (16 * gmt + (lane & 15)) * GLD + (lane >> 4);
        const double* const pb = sR + (16 * gnt + (lane & 15)) * GLD + (lane >> 4);
#pragma unroll
        for (int ks = 0; ks < 4 * CT; ++ks) gacc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * ks], pb[4 * ks], gacc, 0, 0, 0);
      }
    }
  }
  if constexpr (GRAM) {
    if (gmine) {  // this workgroup's partial tile -> [block][Rp][16 NT] (gram_reduce_kernel)
      double* const outp = g.partial + (size_t)blockIdx.x * (gMT * 16) * (16 * gNT);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // (only what the reduction reads: the padding of the tiles is 42 % of the block's bytes at cfg4's 67 x 33, and every byte written
        //  here is written back through L2 when the kernel ends)
        const int gi = 16 * gmt + (lane >> 4) + 4 * r, gj = 16 * gnt + (lane & 15);
        if (gi < gR && gj < gp) outp[gi * (16 * gNT) + gj] = gacc[r];
      }
    }
  }
}
static hipError_t launch_lift_coop(const LiftArgs<double>& a, hipStream_t s) {
  const size_t lds = (size_t)(2 * LC_ACT + 64) * sizeof(double);
  const int ntiles = (a.B + 15) / 16;
  const int grid = ntiles < 2048 ? ntiles : 2048;
  const GramArgs<double> g{};
  if (a.KSp == 25 && a.Hp == 112) hipLaunchKernelGGL((lift_coop_kernel<25, false, 1>), dim3(grid), dim3(1024), lds, s, a, g);
  else hipLaunchKernelGGL((lift_coop_kernel<0, false, 1>), dim3(grid), dim3(1024), lds, s, a, g);
  return hipGetLastError();
}
// lift + Gram sums of the step's transitions in one launch (float64 MLP lift, the cooperative encoder): g.partial receives one partial
// block per workgroup ([*nblocks][Rp][Cp]); the caller reduces them (launch_gram_reduce)
bool lift_gram_available(const LiftArgs<double>& a) {
  const int p = a.L + 1, R = p + a.L + a.n;
  return a.n <= 4 && a.Lp <= 64 && a.Hp <= 128 && (a.Hp & 15) == 0 && a.KSp <= 32 && (a.nlayers == 2 || a.nlayers == 3) &&
         ((R + 15) / 16) * ((p + 15) / 16) <= 16;
}
hipError_t launch_lift_gram(const LiftArgs<double>& a, const GramArgs<double>& g, int* nblocks, hipStream_t s) {
  if (a.B <= 0 || !lift_gram_available(a)) return hipErrorInvalidValue;
  const int p = a.L + 1, R = p + a.L + a.n, MT = (R + 15) / 16;
  const int ntiles = (a.B + 15) / 16;
  // two column tiles per pass where a workgroup would otherwise walk two or more tiles one after the other (cfg4: 512 tiles, 256 blocks)
  static const bool one = dbg_env("KMPC_LIFT_GRAM_CT1") != nullptr;  // measurement / test aid: one tile per pass as in round 5
  const int ct = (ntiles > g.max_blocks && !one) ? 2 : 1;
  const size_t lds = (size_t)(ct * (2 * LC_ACT + 64) + MT * 16 * (16 * ct + 1)) * sizeof(double);
  const int npass = (ntiles + ct - 1) / ct;
  int grid = npass < g.max_blocks ? npass : g.max_blocks;
  if (grid < 1) grid = 1;
  *nblocks = grid;
  const bool ks25 = a.KSp == 25 && a.Hp == 112;
  auto go = [&](auto kern) -> hipError_t {
    static size_t configured_dev[16] = {};  // (function attributes are per device)
    size_t& configured = configured_dev[device_slot()];
    if (lds > 64 * 1024 && lds > configured) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      configured = lds;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, s, a, g);
    return hipGetLastError();
  };
  if (ct == 2) return ks25 ? go(&lift_coop_kernel<25, true, 2>) : go(&lift_coop_kernel<0, true, 2>);
  return ks25 ? go(&lift_coop_kernel<25, true, 1>) : go(&lift_coop_kernel<0, true, 1>);
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
      if constexpr (std::is_same<T, double>::value) {
        if (n == 2) {  // (the fused roll-out's function: bit for bit its observables)
          a.Psi[(size_t)j * a.ps_l + (size_t)b * a.ps_b] = kmpc_rbf_psi2(x[0], x[1], scx[j * 2], scx[j * 2 + 1], a.eps, a.rbf_matlab);
          continue;
        }
      }
      if (a.rbf_matlab) {
        T r2 = T(0);
        for (int i = 0; i < n; ++i) {
          const T d = x[i] - scx[j * n + i];
          r2 += d * d;
        }
        v = r2 > T(0) ? r2 * kmpc_logT<T>(sqrt(r2)) : T(0);  // NaN -> 0 (rbf.m:28)
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
        d2 = d2 < T(0) ? T(0) : d2;  // (np.maximum(d2, 0) of sklearn's euclidean_distances: a NaN stays a NaN)
        const T d = sqrt(d2);
        v = d * d * kmpc_logT<T>(d + a.eps);  // (the same logarithm as the fused roll-out's lift: plant_device.h)
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
  if constexpr (sizeof(T) == 8) {
    static const bool stationary = dbg_env("KMPC_LIFT_STATIONARY") != nullptr;  // measurement aid: the weights-stationary kernel
    if (!stationary && a.KSp <= 32 && (a.Hp & 15) == 0) return launch_lift_coop(reinterpret_cast<const LiftArgs<double>&>(a), s);
  }
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
