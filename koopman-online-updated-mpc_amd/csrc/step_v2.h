// step_v2.h -- the step of one trajectory with the persistent state in REGISTERS (round 3).
//
// For y = C x with L + 2 <= 32 (BASELINE cfg2 / cfg3 and the reference's own dimensions) the whole per-trajectory
// state of the online update -- inv_K_G (p x p), [A B] (L x p), bar_Q (L x L), C (n x L): 1301 doubles at L = 20 --
// is kept as ROWS IN LANES: one matrix row per lane, one register per column, in two register layers
//
//     layer 2   lanes 0 .. p-1 : rows of inv_K_G        lanes 32 .. 32+L-1 : rows of bar_Q
//     layer 1   lanes 0 .. L-1 and 32 .. 32+L-1 : rows of [A B] (both halves hold a copy)
//               lanes L, L+1 and 32+L, 32+L+1   : rows of C
//
// so that every product of the update is ONE v_fmac_f64_dpp per column for ALL rows of a layer
// (row_newbcast broadcasts element j of the regressor / of P z inside each 16-lane row; the vector itself is handed
// to every row of a half with one v_permlane16_swap per dword): M z, the rank-one downdates and the gain updates of
// duffing.py:927-953 are 2 x 21 + 21 + 21 + 20 instructions on full waves, and no element of the state passes through
// LDS.  The layer-1 registers then ARE the rows of [A; C_o] the condensed-QP recursion needs (v chain in lanes 0-31,
// w chain in lanes 32-63), so the model is never re-read either.  In HBM a trajectory's state is the image of these
// registers ("wave image"): [column pair][slot][2] doubles, a wave reads or writes 16 bytes per lane, fully coalesced
// (v2_* below; kmpc_* entry points that want the dense row-major blocks convert, aux_kernels.hip).
//
// Arithmetic: the same estimator and the same order of operations per element as step_body.h (gain form,
// K += (y - K z) g', P -= (P z)(P z)' / d); sums over a row run in four partial sums.
#pragma once
#include "step_body.h"
#include <type_traits>
#include "qp_rl.h"

namespace kmpc {

// ---------------------------------------------------------------------------------------
// wave image of one trajectory (host + device)
// ---------------------------------------------------------------------------------------
// (N <= 32: qp_rl and the H / f pass keep one variable per lane of a 32-lane half)
// (y = C x with q <= 2 rows of C.  The lifted-output form y = psi, q = L -- vanderpol.py:456-459 -- runs on the 8 x 8-grid kernels of
//  step_body.h: a register-state variant of it was built and measured slower in rounds 4 and 5 and is NOT in this header any more;
//  tools/experiments/step_v2_lifted_output_variant.h.txt, profiles/r5_lifted_output_step_v2_ab.txt)
static constexpr bool step_v2_dims(int L, int N, int q) { return q != L && q > 0 && q <= 2 && L + 2 <= 32 && N <= 32 && L >= 2; }
// LDS of one trajectory (elements):
//   R    N x N   H while a solve runs, the tableau of the last solve between two solves (qp_rl.h)          persistent
//   cs   130     row scales of that tableau (2 x 32), its variable set / validity (2 ints), the gains of a covariance
//                update done ahead (64)                                                                       persistent
//   zp   N q     zeros: e_j beyond the horizon (the f lanes of the H / f pass read past the end)             persistent
//   vec  red 16 | f N | predicted bounds N | e (one q-block in front, N q) ... zp ... | g (N+1) q | dump 64 + (N+1) q
// (the fall-back solver's vectors alias g and the dump).  Nothing here is overlaid by the lift scratch.
static constexpr int v2_region1(int N) { return (N * rl_stride(N) + 1) & ~1; }  // (padded row stride: qp_rl.h)
static constexpr int v2_cov_elems(int L) { return 32 + ((L + 1) & ~1); }  // gains of a covariance update done ahead: lanes 0-31, lanes 32 .. 32+L-1
static constexpr int v2_carry_elems(int L) { return 66 + v2_cov_elems(L); }
// long horizons (N >= 30): the recursion's stores are masked instead of sending the lanes without an output to a dump area --
// the kilobyte is the difference between 14 and 16 trajectories per CU there
static constexpr bool v2_chain_dump(int N) { return N < 30; }
static constexpr int v2_vec_elems(int q, int N) { return 16 + 2 * ((N + 1) & ~1) + (q + 2 * N * q) + imax(2 * N, (N + 1) * q + (v2_chain_dump(N) ? (N + 1) * q + 64 : 0)) + 2; }
static constexpr size_t v2_lds_elems(int L, int q, int N) { return ((size_t)v2_region1(N) + v2_carry_elems(L) + v2_vec_elems(q, N) + 1) & ~(size_t)1; }

// ---------------------------------------------------------------------------------------
// row products on DPP
// ---------------------------------------------------------------------------------------
template <int LANE, bool NOP, bool LOWHALF>
__device__ __forceinline__ void fmac_rowbcast_m(double& acc, double vec, double coef) {
  // LOWHALF: only the 16-lane rows 0 and 1 (lanes 0-31) take part (row_mask 0x3)
  if constexpr (LOWHALF) {
    if constexpr (NOP)
      asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0x3 bank_mask:0xf" : "+v"(acc) : "v"(vec), "v"(coef), "n"(LANE));
    else
      asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0x3 bank_mask:0xf" : "+v"(acc) : "v"(vec), "v"(coef), "n"(LANE));
  } else {
    fmac_rowbcast<LANE, NOP>(acc, vec, coef);
  }
}
// a2[l & 3] += R2[l] * vec[l], a1[l & 3] += R1[l] * vec[l], l < CNT; vec[l] = lane l of v0 (l < 16) or lane l - 16 of v1 of the
// reader's 16-lane row.  The two layers alternate: eight independent chains.
template <int CNT, int NCOL, int l = 0>
__device__ __forceinline__ void rowdot2(double (&a2)[4], double (&a1)[4], double v0, double v1, const double (&R2)[NCOL], const double (&R1)[NCOL]) {
  if constexpr (l < CNT) {
    if constexpr (l < 16) {
      fmac_rowbcast<l, l == 0>(a2[l & 3], v0, R2[l]);
      fmac_rowbcast<l, false>(a1[l & 3], v0, R1[l]);
    } else {
      fmac_rowbcast<l - 16, l == 16>(a2[l & 3], v1, R2[l]);
      fmac_rowbcast<l - 16, false>(a1[l & 3], v1, R1[l]);
    }
    rowdot2<CNT, NCOL, l + 1>(a2, a1, v0, v1, R2, R1);
  }
}
template <int CNT, int NCOL, int l = 0>
__device__ __forceinline__ void rowdot1(double (&ac)[4], double v0, double v1, const double (&R)[NCOL]) {
  if constexpr (l < CNT) {
    if constexpr (l < 16) fmac_rowbcast<l, l == 0>(ac[l & 3], v0, R[l]);
    else fmac_rowbcast<l - 16, l == 16>(ac[l & 3], v1, R[l]);
    rowdot1<CNT, NCOL, l + 1>(ac, v0, v1, R);
  }
}
// ac += R[l] * vec[l], l < CNT: ONE chain of multiply-adds (its latency is covered by the other waves of the SIMD; four partial sums
// cost four register clears and three adds per product)
template <int CNT, int NCOL, int l = 0>
__device__ __forceinline__ void rowdot_one(double& ac, double v0, double v1, const double (&R)[NCOL]) {
  if constexpr (l < CNT) {
    if constexpr (l < 16) fmac_rowbcast<l, l == 0>(ac, v0, R[l]);
    else fmac_rowbcast<l - 16, l == 16>(ac, v1, R[l]);
    rowdot_one<CNT, NCOL, l + 1>(ac, v0, v1, R);
  }
}
// R[l] += vec[l] * coef for l < CNT (coef is the lane's own); column LOWCOL only in lanes 0-31 (-1: none)
template <int CNT, int NCOL, int LOWCOL, int l = 0>
__device__ __forceinline__ void rowupd(double (&R)[NCOL], double v0, double v1, double coef) {
  if constexpr (l < CNT) {
    if constexpr (l < 16) fmac_rowbcast_m<l, l == 0, l == LOWCOL>(R[l], v0, coef);
    else fmac_rowbcast_m<l - 16, l == 16, l == LOWCOL>(R[l], v1, coef);
    rowupd<CNT, NCOL, LOWCOL, l + 1>(R, v0, v1, coef);
  }
}
// compile-time loop: f(std::integral_constant<int, I>) for I = B .. E-1
template <int B, int E, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}
// two LDS stores of v (addresses p1 + OFF, p2 + OFF bytes) by the lanes 0 .. CNT-1 only
template <int CNT, int OFF> __device__ __forceinline__ void lds_store2_lanes(double* p1, double* p2, double v) {
  typedef __attribute__((address_space(3))) double* lds_t;
  const unsigned a1 = (unsigned)(size_t)(lds_t)p1, a2 = (unsigned)(size_t)(lds_t)p2;
  constexpr unsigned LO = CNT >= 32 ? 0xffffffffu : ((1u << CNT) - 1u), HI = CNT > 32 ? ((1u << (CNT - 32)) - 1u) : 0u;
  unsigned long long saved;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b32 exec_lo, %4\n\ts_mov_b32 exec_hi, %5\n\tds_write_b64 %1, %3 offset:%6\n\t"
               "ds_write_b64 %2, %3 offset:%6\n\ts_mov_b64 exec, %0"
               : "=&s"(saved) : "v"(a1), "v"(a2), "v"(v), "n"(LO), "n"(HI), "n"(OFF) : "memory");
}
// half_gather for a vector whose elements AND readers all sit in the first 16-lane row of each half (ROWS <= 16: the small lifts --
// p = 9 at L = 8): row_newbcast reads inside the reader's own row, so the lane's own register is the operand and nothing is
// exchanged (two v_permlane16_swap and their moves per product: a tenth of the vector instructions of the cfg3 step)
template <int ROWS> __device__ __forceinline__ void gather_rows(double a, double& v0, double& v1) {
  if constexpr (ROWS <= 16) { v0 = a; v1 = a; }
  else half_gather(a, v0, v1);
}
// lanes t and 32 + t exchange: lo = the value of lane t (lanes 0-31) in both halves, hi = that of lane 32 + t
__device__ __forceinline__ void halves_both(double a, double& lo, double& hi) {
  const int al = __double2loint(a), ah = __double2hiint(a);
  const auto rl = __builtin_amdgcn_permlane32_swap(al, al, false, false);
  const auto rh = __builtin_amdgcn_permlane32_swap(ah, ah, false, false);
  lo = __hiloint2double(rh[0], rl[0]);
  hi = __hiloint2double(rh[1], rl[1]);
}

// ---------------------------------------------------------------------------------------
// the step
// ---------------------------------------------------------------------------------------
// The covariance half of the RLS update on the wave image: inv_K_G <- (inv_K_G - Pz Pz' / d) / lam, bar_Q <- bar_Q - (Q psi)(Q psi)' / dc
// (duffing.py:931-932, 947-951), z = [psi; u] in the lanes of both halves.  Returns lanes < p: g_i = (P z)_i / d, lanes 32 + i:
// h_i = (bar_Q psi)_i / dc (the gains of the model half).  SYNC: the caller is a wave of a roll-out workgroup that has no tile
// of the cooperative encoder -- it does this work in the encoder's shadow and meets the other waves at the encoder's second
// and third barrier from inside (request | barrier | products, d | barrier | downdate, write back).
template <int L_, bool SYNC>
__device__ __forceinline__ double v2_rls_cov(double* const img, const double z, const double lam) {
  typedef double d2_t __attribute__((ext_vector_type(2)));
  constexpr int P_ = L_ + 1, CP = (L_ + 2) / 2, NC = 2 * CP, S2 = 2 * L_ + 1;
  const int tid = local_tid<64>(), half = tid >> 5, t = tid & 31;
  const d2_t* const im = reinterpret_cast<const d2_t*>(img);
  d2_t* const imw = reinterpret_cast<d2_t*>(img);
  const int slot2 = half ? P_ + (t < L_ ? t : L_ - 1) : (t < P_ ? t : P_ - 1);
  double R2[NC];
#pragma unroll
  for (int c = 0; c < CP; ++c) {
    const d2_t v = im[c * S2 + slot2];
    R2[2 * c] = v.x; R2[2 * c + 1] = v.y;
  }
  if constexpr (SYNC) __syncthreads();
  double zv0, zv1;
  gather_rows<P_>(z, zv0, zv1);
  double acc2 = 0.0;  // lanes < p: (P z)_i ; lanes 32 + i: (bar_Q psi)_i
  rowdot_one<P_, NC>(acc2, zv0, zv1, R2);
  // d = lam + z'Pz (lanes 0-31), dc = 1 + psi' bar_Q psi (lanes 32-63): one sum per half
  double s = ((half && t >= L_) ? 0.0 : z) * acc2;
  s += dpp_shr(s, 1);
  s += dpp_shr(s, 2);
  s += dpp_shr(s, 4);
  s += dpp_shr(s, 8);
  double rt = 0.0;
  fmac_rowbcast<15, true>(rt, s, 1.0);  // the total of this lane's 16-lane row
  double r0, r1 = 0.0;
  if constexpr (P_ <= 16) r0 = rt;  // (the second row of the half holds no element)
  else half_gather(rt, r0, r1);
  const double dd = (half ? 1.0 : lam) + (r0 + r1);
  const double dinv = 1.0 / dd;
  const double c2 = -acc2 * dinv;
  if constexpr (SYNC) __syncthreads();
  double w0, w1;
  gather_rows<P_>(acc2, w0, w1);
  rowupd<P_, NC, L_>(R2, w0, w1, c2);
  if (lam != 1.0) {
    const double sc = half ? 1.0 : 1.0 / lam;
#pragma unroll
    for (int c = 0; c < P_; ++c) R2[c] *= sc;
  }
  if (half ? t < L_ : t < P_) {
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      d2_t v;
      v.x = R2[2 * c]; v.y = R2[2 * c + 1];
      imw[c * S2 + slot2] = v;
    }
  }
  return acc2 * dinv;
}
// LDS slots of a wave's region that the roll-out kernel touches itself: the gains of a covariance update done ahead, the
// first move of the last solve
template <int N_> __device__ __forceinline__ double* v2_cov_slot(double* sm) { return sm + v2_region1(N_) + 66; }
// (its element of lane `lane`: lanes 0-31 as they are, lanes 32 .. 32+L-1 behind them; the other lanes of the upper half share the last slot)
template <int L_> __device__ __forceinline__ int v2_cov_index(int lane) { return lane < 32 ? lane : (lane - 32 < L_ ? lane : 32 + L_ - 1 + (L_ & 1)); }

// sv.psi_now_v / psi_prev_v: lane with (lane & 31) = i < L carries psi_i (BOTH halves).  img: this trajectory's wave image.
// PART (the roll-out with the per-step terminal refresh calls the step in two parts with the Riccati iteration between them):
//   0 the whole step;  1 the RLS phase only (the updated model is in the image when it returns);  2 condensed QP and solve only (the model
//   comes from the image)
template <int L_, int N_, int Q_, bool LOWREG, bool ASREG, typename IOT = double, int PART = 0>
__device__ __forceinline__ void step_v2(const StepArgs<double>& a, const StepVar<double>& sv, const int b, double* const sm, double* const img) {
  typedef double d2_t __attribute__((ext_vector_type(2)));
  constexpr int P_ = L_ + 1, CP = (L_ + 2) / 2, NC = 2 * CP, NX = 2, S2 = 2 * L_ + 1, S1 = L_ + NX;
  constexpr int N = N_, q = Q_;
  static_assert(step_v2_dims(L_, N_, Q_), "step_v2: dimension set");
  static_assert(N_ <= 32 && L_ + 2 <= 32, "step_v2: one variable / one state row per lane of a 32-lane half");
  const int tid = local_tid<64>();
  const int half = tid >> 5, t = tid & 31;
  const int B = a.B;
  // LDS map (v2_lds_elems)
  double* const sR = sm;                          // H / carried tableau
  double* const sCs = sR + v2_region1(N_);        // rs[32] | rsi[32] | {smask, valid} | gains of rls_cov [64]
  double* const sCov = sCs + 66;
  double* const vec = sCs + v2_carry_elems(L_);
  double* const red = vec;
  constexpr int NE = (N_ + 1) & ~1;               // (even offsets: g and e are read as 16-byte vectors)
  double* const sf = red + 16;
  double* const qxo = sf + NE;                    // predicted bounds of the solve / its hand-over point
  double* const sEr = qxo + NE + q;                // e_1 .. e_N at sEr[(j - 1) q + r]; one q-block in front; N q zeros behind
  double* const sG = sEr + 2 * N * q;             // g_0 .. g_N        (N + 1) q
  double* const dump = sG + (N + 1) * q;          // 64 + (N + 1) q: where lanes without an output write
  double* const qxa = sG;                         // (fall-back solver: aliases g and the dump)
  double* const qg = qxa + N;

  KTRACE(0);
  const double up = uniform_value(a.u_prev[b]);  // (one trajectory per wave: scalar registers; K = 200 - 2 %)
  double xw_pre = 0.0;
  constexpr int REFN = (Q_ * N_ + 63) / 64;
  double refp[REFN];
  if (sv.phases & PH_CONDENSE) {
    const double* refg = io_at<IOT>(a.ref, a.ref_per_traj ? (size_t)b * q * N : 0);
#pragma unroll
    for (int i = 0; i < REFN; ++i) {
      const int e = tid + i * 64, ec = e < q * N ? e : 0, k = ec / q, r = ec - k * q;
      refp[i] = io_ld<IOT>(refg, r * N + k);
    }
  }

  // =====================================================================================
  // phase 1: recursive least squares in registers.  The covariance half -- inv_K_G and bar_Q, which only need the regressor
  // z = [psi(x_{k-1}); u_{k-1}] -- normally ran at the END of the previous step (rls_cov below, sv.cov_done): its state then
  // travels while the waves of the workgroup are out of step with each other, not all at once behind the lift's barrier,
  // and only the model half (16 x 3.9 KB per CU) is read here.
  // =====================================================================================
  double R1[NC];
  const int slot1 = t < S1 ? t : S1 - 1;
  const d2_t* const im = reinterpret_cast<const d2_t*>(img);
  d2_t* const imw = reinterpret_cast<d2_t*>(img);
  const double psin = sv.psi_now_v;
  if (PART != 2 && (sv.phases & PH_RLS)) {
    const bool fu = sv.first_update != 0;
    if (!fu) {
#pragma unroll
      for (int c = 0; c < CP; ++c) {
        const d2_t v = im[CP * S2 + c * S1 + slot1];
        R1[2 * c] = v.x; R1[2 * c + 1] = v.y;
      }
    } else {  // K_A = 0, bar_X = 0 (duffing.py:927-928, 944-945)
#pragma unroll
      for (int c = 0; c < NC; ++c) R1[c] = 0.0;
    }
    // x_{k+1} for the rows of C (the roll-out keeps it in LDS)
    const int xr = t - L_ < 0 ? 0 : (t - L_ < NX ? t - L_ : NX - 1);
    double xn;  // (x_next, when given, is the roll-out's LDS slot: step_body.h lds_ld)
    if (sv.x_next) xn = lds_ld(sv.x_next + xr);
    else xn = io_ld<IOT>(a.x_now, (size_t)xr * B + b);
    // z = [psi(x_{k-1}); u_{k-1}] in the lanes of both halves
    const double z = t < L_ ? sv.psi_prev_v : (t == L_ ? up : 0.0);
    KTRACE(1);
    const double u2 = sv.cov_done ? sCov[v2_cov_index<L_>(tid)] : v2_rls_cov<L_, false>(img, z, a.lam);
    KTRACE(2);
    double zv0, zv1;
    gather_rows<(S1 > P_ ? S1 : P_)>(z, zv0, zv1);
    double acc1 = 0.0;  // [A B] rows: (K z)_r ; C rows: (C psi)_r
    rowdot_one<P_, NC>(acc1, zv0, zv1, R1);
    KTRACE(3);
    // [A B] <- ([A B] - K z g') / lam + y g' = K / lam + (e / lam + y (1 - 1/lam)) g',  g = Pz / d   (Koopman_update.m:270-274;
    // lam = 1: K + e g', duffing.py:927-938);  C <- C + (x_{k+1} - C psi) h',  h = bar_Q psi / dc          duffing.py:943-953
    {
      double gall, hall;
      halves_both(u2, gall, hall);
      double g0, g1, h0, h1;
      gather_rows<(S1 > P_ ? S1 : P_)>(gall, g0, g1);
      gather_rows<(S1 > P_ ? S1 : P_)>(hall, h0, h1);
      const bool isK = t < L_, isC = t >= L_ && t < S1;
      double eK = psin - acc1;
      if (a.lam != 1.0) {
        const double linv = 1.0 / a.lam;
        eK = eK * linv + psin * (1.0 - linv);
        const double sc = isK ? linv : 1.0;
#pragma unroll
        for (int c = 0; c < P_; ++c) R1[c] *= sc;
      }
      const double cK = isK ? eK : 0.0;
      const double cC = (isC && !(a.c_skip_first && fu)) ? xn - acc1 : 0.0;
      rowupd<P_, NC, -1>(R1, g0, g1, cK);
      rowupd<L_, NC, -1>(R1, h0, h1, cC);
      if (!half && t < S1) {
#pragma unroll
        for (int c = 0; c < CP; ++c) {
          d2_t v;
          v.x = R1[2 * c]; v.y = R1[2 * c + 1];
          imw[CP * S2 + c * S1 + slot1] = v;
        }
      }
    }
    KTRACE(4);
  } else if (PART != 1) {
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      const d2_t v = im[CP * S2 + c * S1 + slot1];
      R1[2 * c] = v.x; R1[2 * c + 1] = v.y;
    }
  }
  if constexpr (PART == 1) return;

  // =====================================================================================
  // phase 2: condensed QP.  v_{j+1} = A v_j (v_0 = B) in lanes 0-31, w_{j+1} = A w_j (w_0 = psi) in lanes 32-63; the rows of C
  // in the same registers give g_j = C_o v_j and C_o w_j.  Every lane writes its value of the step to LDS -- the output rows
  // where H and f are built from, all others into a dump -- so that the store needs no mask.
  // =====================================================================================
  {
    const bool isA = t < L_;
    double v0, v1;
    gather_rows<S1>(isA ? (half ? psin : R1[L_]) : 0.0, v0, v1);
    // delta-u form (Tank_System.m:110-113): x+ = A x + B s, s = 1 on the v chain and u_prev on the w chain
    const double bs = (a.du_mode && isA) ? R1[L_] * (half ? up : 1.0) : 0.0;
    // the rows of C_o give g_j = C_o v_j and C_o w_j at step j (j = 0 .. N)
    const int ro = t - L_ - a.cy0;
    const bool isO = ro >= 0 && ro < q;
    double* const optr = isO ? (half ? sEr - q + ro : sG + ro) : dump + tid;
    KTRACE(5);
    constexpr int JN = N_;
#pragma unroll
    for (int j = 0; j <= JN; ++j) {
      double acc = bs;
      rowdot_one<L_, NC>(acc, v0, v1, R1);
      if constexpr (v2_chain_dump(N_)) optr[j * q] = acc;
      else if (isO) optr[j * q] = acc;
      if (j < JN) gather_rows<S1>(acc, v0, v1);
    }
    block_sync<64>();
    // e_j = C_o w_j - r_{j-1}
#pragma unroll
    for (int i = 0; i < REFN; ++i) {
      const int e = tid + i * 64;
      if (e < q * N) sEr[e] -= refp[i];
    }
    block_sync<64>();
  }
  KTRACE(6);
  if ((sv.phases & PH_QP) && a.x_warm) xw_pre = a.x_warm[(size_t)(t < N_ ? t : 0) * a.B + b];
  // the tableau of the last solve leaves LDS before H takes its place
  double M[N_];
  double rs = 1.0, rsi = 1.0;
  QpCarry cs;
  {
    const int* const ci = reinterpret_cast<const int*>(sCs + 64);
    cs.smask = (unsigned)__builtin_amdgcn_readfirstlane(ci[0]);  // (wave-uniform: the solve branches on them)
    cs.valid = __builtin_amdgcn_readfirstlane(ci[1]);
    cs.umask = (unsigned)__builtin_amdgcn_readfirstlane(ci[2]);
    cs.trust = __builtin_amdgcn_readfirstlane(ci[3]);
    if (cs.valid) {
      const double* const trow = sR + (t < N_ ? t : N_ - 1) * rl_stride(N_);
      if constexpr ((N_ & 1) == 0) {
        const d2_t* t2 = reinterpret_cast<const d2_t*>(__builtin_assume_aligned(trow, 16));
#pragma unroll
        for (int j = 0; j < N_ / 2; ++j) { const d2_t v = t2[j]; M[2 * j] = v.x; M[2 * j + 1] = v.y; }
      } else {
#pragma unroll
        for (int j = 0; j < N_; ++j) M[j] = trow[j];
      }
      rs = sCs[t];
      rsi = sCs[32 + t];
    }
    block_sync<64>();
  }
  // H[a][b] = Qw S(b - a, N-1-b) (+ Rw on the diagonal), S(d, t) = sum_{s <= t} g_{s+d} . g_s;  f[a] = 2 Qw sum_t g_t . e_{t+a}:
  // lane d < N walks diagonal d of H and writes both triangles, lane 32 + a accumulates f[a] -- one instruction stream; the f
  // lanes read zeros behind e_N, the H lanes leave a diagonal under a compile-time lane mask
  {
    typedef double dq_t __attribute__((ext_vector_type(Q_ == 2 ? 2 : 1)));
    const int hf = tid >> 5, idx = tid & 31;
    // (even element offsets in the static layout: g and e are read as 16-byte vectors when q = 2)
    const dq_t* const gq = reinterpret_cast<const dq_t*>(__builtin_assume_aligned(sG, Q_ == 2 ? 16 : 8));
    const dq_t* const wq = reinterpret_cast<const dq_t*>(__builtin_assume_aligned((hf ? sEr : sG) + idx * Q_, Q_ == 2 ? 16 : 8));
    double acc = 0.0;
    constexpr int NS = rl_stride(N_);
    double* const h1 = sR - idx * NS;
    double* const h2 = sR - idx;
    const double Qw = a.Qw, rdiag = idx == 0 ? a.Rw : 0.0;  // (read once: inside the masked regions below every use was a scalar load of its own)
    // Round 4: HCH diagonal steps at a time -- their values first, then their stores.  One step at a time, every step was a complete
    // LDS round trip: the stores are volatile assembly (the immediate exec masks), nothing moves across them, so the two reads
    // of a step could only be requested behind the previous step's stores and waited for in full (20 x ~230 cycles = 1.9 of the
    // 33 us of a cfg2 step).  In a chunk the reads of all its steps are in flight together.
    constexpr int HCH = 10;
    static_for<0, (N_ + HCH - 1) / HCH>([&](auto CC) {
      constexpr int c0 = decltype(CC)::value * HCH, c1 = c0 + HCH < N_ ? c0 + HCH : N_;
      double hv[HCH];
      static_for<c0, c1>([&](auto TT) {
        constexpr int tt = decltype(TT)::value;
        if constexpr (Q_ == 2) {
          const dq_t u = gq[tt], w = wq[tt];
          acc = tfma(u[0], w[0], acc);
          acc = tfma(u[1], w[1], acc);
        } else {
#pragma unroll
          for (int r = 0; r < Q_; ++r) acc = tfma(sG[tt * Q_ + r], ((hf ? sEr : sG) + idx * Q_)[tt * Q_ + r], acc);
        }
        hv[tt - c0] = Qw * acc + rdiag;
      });
      // lanes tid < N - tt: (aa, bb) = (N-1-tt-idx, N-1-tt) is inside H.  The lane set is a compile-time constant: the two stores run
      // under an exec mask written as an immediate (a compare, a saved mask and a branch per diagonal step otherwise)
      static_for<c0, c1>([&](auto TT) {
        constexpr int tt = decltype(TT)::value;
        lds_store2_lanes<N_ - tt, (N_ - 1 - tt) * (NS + 1) * 8>(h1, h2, hv[tt - c0]);
      });
    });
    if (hf && idx < N_) sf[idx] = 2.0 * Qw * acc;
  }
  if (a.Wterm) {
    // terminal block of Q_bar is PN instead of Qw I (Koopman_update.m:381); Wterm = PN - Qw I:
    //   H[a][b] += g_{N-1-a}' sym(W) g_{N-1-b},   f[a] += 2 g_{N-1-a}' W e_N
    block_sync<64>();
    const double* const Wt = a.Wterm + (a.wterm_per_traj ? (size_t)b * q * q : (size_t)0);
    for (int e = tid; e < N * N; e += 64) {
      const int aa = e / N, bb = e - aa * N;
      const double* ga = sG + (N - 1 - aa) * q;
      const double* gb = sG + (N - 1 - bb) * q;
      double acc = 0.0;
#pragma unroll 1
      for (int r = 0; r < q; ++r)
#pragma unroll 2
        for (int s2 = 0; s2 < q; ++s2) acc += ga[r] * (0.5 * (Wt[r * q + s2] + Wt[s2 * q + r])) * gb[s2];
      sR[aa * rl_stride(N_) + bb] += acc;
    }
    for (int aa = tid; aa < N; aa += 64) {
      const double* ga = sG + (N - 1 - aa) * q;
      const double* eN = sEr + (N - 1) * q;
      double acc = 0.0;
#pragma unroll 1
      for (int r = 0; r < q; ++r)
#pragma unroll 2
        for (int s2 = 0; s2 < q; ++s2) acc += ga[r] * Wt[r * q + s2] * eN[s2];
      sf[aa] += 2.0 * acc;
    }
  }
  block_sync<64>();
  KTRACE(7);

  // =====================================================================================
  // phase 3: box QP (rows in lanes, carried tableau: qp_rl.h)
  // =====================================================================================
  if (sv.phases & PH_QP) {
    if (qp_rl<N_, IOT>(sR, sf, a, sv, b, qxo, red + 15, M, rs, rsi, cs, up, xw_pre)) {
      // crawling solve (rare): H moves to this trajectory's global scratch block, the active-set loop of qp_lds works with an
      // LDS tableau in its place
#ifdef KMPC_TRACE
      if (tid == 0 && b < 8192) kmpc_trace_buf[b * 32 + 28] += 1ull;
#endif
      block_sync<64>();
      double* const Hg = a.qp_scratch + (size_t)b * N * N;
      for (int e = tid; e < N * N; e += 64) Hg[e] = sR[(e / N) * rl_stride(N_) + (e % N)];
      __threadfence_block();
      block_sync<64>();
      qp_lds<double, 64, IOT>(Hg, sf, sR, qxo, qxa, qg, red, a, sv, b, N, true);
      block_sync<64>();
      for (int e = tid; e < N * q; e += 64) sEr[N * q + e] = 0.0;  // (the solver's vectors may have covered the zeros behind e_N)
      if (tid == 0) red[15] = qxo[0];
    }
    if (!half) {
      sCs[t] = rs;
      sCs[32 + t] = rsi;
    }
    if (tid == 0) {
      int* const ci = reinterpret_cast<int*>(sCs + 64);
      ci[0] = (int)cs.smask;
      ci[1] = cs.valid;
      ci[2] = (int)cs.umask;
      ci[3] = cs.trust;
    }
    block_sync<64>();
    if (sv.cov_ahead) {
      // the covariance half of the NEXT step's update: its regressor [psi(x_k); u_k] is complete now
      const double uk = (a.du_mode ? up : 0.0) + red[15];  // (the solve leaves its first move there)
      const double gains = v2_rls_cov<L_, false>(img, t < L_ ? psin : (t == L_ ? uk : 0.0), a.lam);
      if (!half || t < L_) sCov[v2_cov_index<L_>(tid)] = gains;
    }
  }
}

// once per launch, before the first step: no carried tableau, zeros behind e_N
template <int L_, int N_, int Q_> __device__ __forceinline__ void step_v2_init(double* const sm) {
  const int tid = local_tid<64>();
  double* const sCs = sm + v2_region1(N_);
  double* const sEr = sCs + v2_carry_elems(L_) + 16 + 2 * ((N_ + 1) & ~1) + Q_;
  if (tid == 0) {
    int* const ci = reinterpret_cast<int*>(sCs + 64);
    ci[0] = 0;
    ci[1] = 0;
    ci[2] = 0;
    ci[3] = 0;
  }
  for (int e = tid; e < N_ * Q_; e += 64) sEr[N_ * Q_ + e] = 0.0;
}

}  // namespace kmpc
