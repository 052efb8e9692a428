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

namespace kmpc {

// ---------------------------------------------------------------------------------------
// wave image of one trajectory (host + device)
// ---------------------------------------------------------------------------------------
static constexpr bool step_v2_dims(int L, int N, int q) { return q != L && q > 0 && L + 2 <= 32 && N <= 40 && L >= 2; }
// LDS of one trajectory (elements): H | fall-back tableau (short horizons) | red, f | chain outputs / QP vectors
static constexpr int v2_region1(int N) { return (N * N + 1) & ~1; }
static constexpr int v2_region2(int N, int L) { return tableau_saves_lds(N, L) ? 0 : ((N * N + 1) & ~1); }
static constexpr int v2_vec_elems(int q, int N) { return 16 + N + imax(3 * N, 3 * (N + 1) * q + q + 64) + 2; }
static constexpr size_t v2_lds_elems(int L, int q, int N) { return ((size_t)v2_region1(N) + v2_region2(N, L) + v2_vec_elems(q, N) + 1) & ~(size_t)1; }

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
// R[l] += vec[l] * coef for l < CNT (coef is the lane's own); column LOWCOL only in lanes 0-31 (-1: none)
template <int CNT, int NCOL, int LOWCOL, int l = 0>
__device__ __forceinline__ void rowupd(double (&R)[NCOL], double v0, double v1, double coef) {
  if constexpr (l < CNT) {
    if constexpr (l < 16) fmac_rowbcast_m<l, l == 0, l == LOWCOL>(R[l], v0, coef);
    else fmac_rowbcast_m<l - 16, l == 16, l == LOWCOL>(R[l], v1, coef);
    rowupd<CNT, NCOL, LOWCOL, l + 1>(R, v0, v1, coef);
  }
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
// sv.psi_now_v / psi_prev_v: lane with (lane & 31) = i < L carries psi_i (BOTH halves).  img: this trajectory's wave image.
template <int L_, int N_, int Q_, bool LOWREG, bool ASREG>
__device__ __forceinline__ void step_v2(const StepArgs<double>& a, const StepVar<double>& sv, const int b, double* const sm, double* const img) {
  typedef double d2_t __attribute__((ext_vector_type(2)));
  constexpr int P_ = L_ + 1, CP = (L_ + 2) / 2, NC = 2 * CP, NX = 2, S2 = 2 * L_ + 1, S1 = L_ + NX;
  constexpr int N = N_, q = Q_;
  static_assert(step_v2_dims(L_, N_, Q_), "step_v2: dimension set");
  const int tid = local_tid<64>();
  const int half = tid >> 5, t = tid & 31;
  const int B = a.B;
  // LDS map
  double* const sH = sm;
  double* const sM = sH + a.r1;  // fall-back tableau of qp_lds (short horizons; else the fall-back works in global scratch)
  double* const vec = sM + a.r2;
  double* const red = vec;
  double* const sf = red + 16;
  double* const va = sf + N;
  double* const sG = va;                          // g_0 .. g_N        (N + 1) q
  double* const sEr = sG + (N + 1) * q + q;       // e_0 .. e_N  at  sEr[(j - 1) q + r]: one q-block in front
  double* const dump = sEr + N * q;               // 64 + (N + 1) q: where lanes without an output write
  double* const qx = va;
  double* const qxa = qx + N;
  double* const qg = qxa + N;

  KTRACE(0);
  const double up = a.u_prev[b];
  double xw_pre = 0.0;
  constexpr int REFN = (Q_ * N_ + 63) / 64;
  double refp[REFN];
  if (sv.phases & PH_CONDENSE) {
    const double* refg = a.ref + (a.ref_per_traj ? (size_t)b * q * N : 0);
#pragma unroll
    for (int i = 0; i < REFN; ++i) {
      const int e = tid + i * 64, ec = e < q * N ? e : 0, k = ec / q, r = ec - k * q;
      refp[i] = refg[r * N + k];
    }
  }

  // =====================================================================================
  // phase 1: recursive least squares in registers
  // =====================================================================================
  double R1[NC];
  const int slot1 = t < S1 ? t : S1 - 1;
  const d2_t* const im = reinterpret_cast<const d2_t*>(img);
  d2_t* const imw = reinterpret_cast<d2_t*>(img);
  const double psin = sv.psi_now_v;
  if (sv.phases & PH_RLS) {
    const bool fu = sv.first_update != 0;
    const int slot2 = half ? P_ + (t < L_ ? t : L_ - 1) : (t < P_ ? t : P_ - 1);
    double R2[NC];
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      const d2_t v = im[c * S2 + slot2];
      R2[2 * c] = v.x; R2[2 * c + 1] = v.y;
    }
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
    const double xn = sv.x_next ? sv.x_next[xr] : a.x_now[(size_t)xr * B + b];
    // z = [psi(x_{k-1}); u_{k-1}] in the lanes of both halves
    const double z = t < L_ ? sv.psi_prev_v : (t == L_ ? up : 0.0);
    double zv0, zv1;
    half_gather(z, zv0, zv1);
    KTRACE(1);
    double a2[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0};
    rowdot2<P_, NC>(a2, a1, zv0, zv1, R2, R1);
    const double acc2 = (a2[0] + a2[1]) + (a2[2] + a2[3]);  // lanes < p: (P z)_i ; lanes 32 + i: (bar_Q psi)_i
    const double acc1 = (a1[0] + a1[1]) + (a1[2] + a1[3]);  // [A B] rows: (K z)_r ; C rows: (C psi)_r
    // d = lam + z'Pz (lanes 0-31), dc = 1 + psi' bar_Q psi (lanes 32-63): one sum per half
    double s = ((half && t >= L_) ? 0.0 : z) * acc2;
    s += dpp_shr(s, 1);
    s += dpp_shr(s, 2);
    s += dpp_shr(s, 4);
    s += dpp_shr(s, 8);
    double rt = 0.0;
    fmac_rowbcast<15, true>(rt, s, 1.0);  // the total of this lane's 16-lane row
    double r0, r1;
    half_gather(rt, r0, r1);
    const double dd = (half ? 1.0 : a.lam) + (r0 + r1);
    const double dinv = 1.0 / dd;
    KTRACE(2);
    // inv_K_G <- (inv_K_G - Pz Pz' / d) / lam ; bar_Q <- bar_Q - (Q psi)(Q psi)' / dc        duffing.py:931-932, 947-951
    {
      const double c2 = -acc2 * dinv;
      double w0, w1;
      half_gather(acc2, w0, w1);
      rowupd<P_, NC, L_>(R2, w0, w1, c2);
      if (a.lam != 1.0) {
        const double sc = half ? 1.0 : 1.0 / a.lam;
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
    }
    KTRACE(3);
    // [A B] <- ([A B] - K z g') / lam + y g' = K / lam + (e / lam + y (1 - 1/lam)) g',  g = Pz / d   (Koopman_update.m:270-274;
    // lam = 1: K + e g', duffing.py:927-938);  C <- C + (x_{k+1} - C psi) h',  h = bar_Q psi / dc          duffing.py:943-953
    {
      const double u2 = acc2 * dinv;
      double gall, hall;
      halves_both(u2, gall, hall);
      double g0, g1, h0, h1;
      half_gather(gall, g0, g1);
      half_gather(hall, h0, h1);
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
  } else {
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      const d2_t v = im[CP * S2 + c * S1 + slot1];
      R1[2 * c] = v.x; R1[2 * c + 1] = v.y;
    }
  }

  // =====================================================================================
  // phase 2: condensed QP.  v_{j+1} = A v_j (v_0 = B) in lanes 0-31, w_{j+1} = A w_j (w_0 = psi) in lanes 32-63; the rows of C
  // in the same registers give g_j = C_o v_j and C_o w_j.  Every lane writes its value of the step to LDS -- the output rows
  // where H and f are built from, all others into a dump -- so that the store needs no mask.
  // =====================================================================================
  {
    const bool isA = t < L_;
    double v0, v1;
    half_gather(isA ? (half ? psin : R1[L_]) : 0.0, v0, v1);
    // delta-u form (Tank_System.m:110-113): x+ = A x + B s, s = 1 on the v chain and u_prev on the w chain
    const double bs = (a.du_mode && isA) ? R1[L_] * (half ? up : 1.0) : 0.0;
    const int ro = t - L_ - a.cy0;
    double* const optr = (ro >= 0 && ro < q) ? (half ? sEr - q + ro : sG + ro) : dump + tid;
    KTRACE(5);
#pragma unroll
    for (int j = 0; j <= N_; ++j) {
      double ac4[4] = {bs, 0.0, 0.0, 0.0};
      rowdot1<L_, NC>(ac4, v0, v1, R1);
      const double acc = (ac4[0] + ac4[1]) + (ac4[2] + ac4[3]);
      optr[j * q] = acc;
      if (j < N_) half_gather(acc, v0, v1);
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
  if ((sv.phases & PH_QP) && a.x_warm) {
    const int mv = (tid >> 3) + 8 * (tid & 7);
    xw_pre = a.x_warm[(size_t)(mv < N_ ? mv : 0) * a.B + b];
  }
  // H[a][b] = Qw S(b - a, N-1-b) (+ Rw on the diagonal), S(d, t) = sum_{s <= t} g_{s+d} . g_s;  f[a] = 2 Qw sum_t g_t . e_{t+a}
  if constexpr (N_ <= 32) {
    // lane d < N walks diagonal d of H, lane 32 + a accumulates f[a]: one instruction stream (see step_body.h)
    const int hf = tid >> 5, idx = tid & 31;
    const double* const wb = (hf ? sEr : sG) + idx * Q_;
    const bool on = idx < N_;
    double acc = 0.0;
    double* const h1 = hf ? red + 14 : sH - idx * N_;
    double* const h2 = hf ? red + 15 : sH - idx;
    const int hstep = hf ? 0 : N_ + 1;
    const double rdiag = (idx == 0 && !hf) ? a.Rw : 0.0;
#pragma unroll
    for (int tt = 0; tt < N_; ++tt) {
      if (on && tt + idx < N_) {
        double s0 = 0.0;
#pragma unroll
        for (int r = 0; r < Q_; ++r) s0 += sG[tt * Q_ + r] * wb[tt * Q_ + r];
        acc += s0;
        const double hv = a.Qw * acc + rdiag;
        h1[(N_ - 1 - tt) * hstep] = hv;
        h2[(N_ - 1 - tt) * hstep] = hv;
      }
    }
    if (hf && on) sf[idx] = 2.0 * a.Qw * acc;
  } else {
    for (int d = tid; d < N; d += 64) {
      double acc = 0.0;
#pragma unroll
      for (int tt = 0; tt < N; ++tt) {
        if (tt + d < N) {
          double s0 = 0.0;
#pragma unroll
          for (int r = 0; r < q; ++r) s0 += sG[(tt + d) * q + r] * sG[tt * q + r];
          acc += s0;
          const int bb = N - 1 - tt, aa = bb - d;
          const double hv = a.Qw * acc + (d == 0 ? a.Rw : 0.0);
          sH[aa * N + bb] = hv;
          sH[bb * N + aa] = hv;
        }
      }
    }
    for (int aa = tid; aa < N; aa += 64) {
      double acc = 0.0;
#pragma unroll
      for (int tt = 0; tt < N; ++tt)
        if (tt + aa < N) {
#pragma unroll
          for (int r = 0; r < q; ++r) acc += sG[tt * q + r] * sEr[(tt + aa) * q + r];
        }
      sf[aa] = 2.0 * a.Qw * acc;
    }
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
      for (int r = 0; r < q; ++r)
        for (int s2 = 0; s2 < q; ++s2) acc += ga[r] * (0.5 * (Wt[r * q + s2] + Wt[s2 * q + r])) * gb[s2];
      sH[e] += acc;
    }
    for (int aa = tid; aa < N; aa += 64) {
      const double* ga = sG + (N - 1 - aa) * q;
      const double* eN = sEr + (N - 1) * q;
      double acc = 0.0;
      for (int r = 0; r < q; ++r)
        for (int s2 = 0; s2 < q; ++s2) acc += ga[r] * Wt[r * q + s2] * eN[s2];
      sf[aa] += 2.0 * acc;
    }
  }
  block_sync<64>();
  KTRACE(7);

  // =====================================================================================
  // phase 3: box QP (register tableau; step_body.h)
  // =====================================================================================
  if (sv.phases & PH_QP) {
    if (qp_regs<double, N_, LOWREG, ASREG>(sH, sf, a, sv, b, red, qx, up, xw_pre)) {
      block_sync<64>();
      if constexpr (!tableau_saves_lds(N_, L_)) {
        qp_lds<double, 64>(sH, sf, sM, qx, qxa, qg, red, a, sv, b, N, true);
      } else {
        double* const Hg = a.qp_scratch + (size_t)b * N * N;
        for (int e = tid; e < N * N; e += 64) Hg[e] = sH[e];
        __threadfence_block();
        block_sync<64>();
        qp_lds<double, 64>(Hg, sf, sH, qx, qxa, qg, red, a, sv, b, N, true);
      }
    }
  }
}

}  // namespace kmpc
