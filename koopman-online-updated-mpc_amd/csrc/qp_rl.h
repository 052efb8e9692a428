// qp_rl.h -- box QP of the register-state step (step_v2.h): the swept tableau as ROWS IN LANES, carried from step to step.
//
//   min u'Hu + f'u,  lb <= u <= ub      (replaces optimize.minimize(..., bounds) duffing.py:857-861 / quadprog Koopman_update.m:214)
//
// Same method and the same decisions as qp_regs (step_body.h): projected Newton on the swept tableau T = sweep_F(2H),
// active-set prediction rounds, Armijo search on the true cost, termination on the componentwise KKT test evaluated with H
// itself.  What differs is the data layout and what a solve starts from:
//  * Lane i < N keeps row i of the tableau, lane 32 + i row i of H, in the SAME N registers (the two halves of the wave).  A
//    product with either matrix is N v_fmac_f64_dpp on the whole wave (row_newbcast broadcasts element j of the vector inside a
//    16-lane row; the vector reaches every row of a half through one v_permlane16_swap per dword) and needs no reduction
//    across lanes; the 8 x 8 lane grid of qp_regs paid 6 ds_bpermute, 9 multiply-adds and three 8-lane DPP all-reduces for it.
//  * A symmetric sweep on variable k is N such instructions as well: the pivot column T(., k) is register k of every lane --
//    already a vector in the lanes.  Row k is not rescaled element by element: every row carries a scale (T_ij = rs_i M_i[j]),
//    so "row k <- row k / d" is one multiply in lane k.
//  * The tableau of the last solve stays in LDS (the H region: H is only needed while a solve runs, T only between two
//    solves) together with its variable set.  In closed loop H moves by a rank-one model update per step and the free set
//    rarely changes, so the old tableau is an approximate inverse for the new H: the projected-Newton iteration then converges
//    linearly (a factor |I - T 2H|, 1e-3 .. 5e-2 per iteration on the bench workloads) instead of in one step, at ~140
//    instructions per iteration -- against the N sweeps (~50 instructions each) that built a fresh tableau at every step.
//    The answer does not depend on it: the KKT test is evaluated with H.  A tableau that contracts too slowly is rebuilt from
//    2H (N sweeps, as before); a launch always starts with a rebuild, so results do not depend on how a roll-out is split.
#pragma once
#include "step_body.h"

namespace kmpc {

// sum over the lanes of each half (lanes 0-31 / 32-63); every lane gets its half's sum
__device__ __forceinline__ double half_sum(double s) {
  s += dpp_shr(s, 1);
  s += dpp_shr(s, 2);
  s += dpp_shr(s, 4);
  s += dpp_shr(s, 8);
  double rt = 0.0;
  fmac_rowbcast<15, true>(rt, s, 1.0);  // the total of this lane's 16-lane row
  double r0, r1;
  half_gather(rt, r0, r1);
  return r0 + r1;
}
// sums over lanes 0-31 and over lanes 32-63 as WAVE-UNIFORM values (scalar registers: what the solve branches on)
__device__ __forceinline__ void half_sums_u(double s, double& lo, double& hi) {
  s += dpp_shr(s, 1);
  s += dpp_shr(s, 2);
  s += dpp_shr(s, 4);
  s += dpp_shr(s, 8);
  lo = lane_bcast(s, 15) + lane_bcast(s, 31);
  hi = lane_bcast(s, 47) + lane_bcast(s, 63);
}
__device__ __forceinline__ double dpp_shr_keep(double v, int ctrl) {  // row_shr with the lane's own value where no lane is read
  const int lo = __double2loint(v), hi = __double2hiint(v);
  switch (ctrl) {
    case 1: return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x111, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(lo, lo, 0x111, 0xf, 0xf, false));
    case 2: return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x112, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(lo, lo, 0x112, 0xf, 0xf, false));
    case 4: return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x114, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(lo, lo, 0x114, 0xf, 0xf, false));
    default: return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x118, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(lo, lo, 0x118, 0xf, 0xf, false));
  }
}
// maximum over the lanes of each half
__device__ __forceinline__ double half_max(double s) {
  double w;
  w = dpp_shr_keep(s, 1); s = w > s ? w : s;
  w = dpp_shr_keep(s, 2); s = w > s ? w : s;
  w = dpp_shr_keep(s, 4); s = w > s ? w : s;
  w = dpp_shr_keep(s, 8); s = w > s ? w : s;
  double rt = 0.0;
  fmac_rowbcast<15, true>(rt, s, 1.0);
  double r0, r1;
  half_gather(rt, r0, r1);
  return r0 > r1 ? r0 : r1;
}
// lanes t and 32 + t: lo = the value of lane t in both, hi = that of lane 32 + t
__device__ __forceinline__ void halves_both_q(double a, double& lo, double& hi) {
  const int al = __double2loint(a), ah = __double2hiint(a);
  const auto rl = __builtin_amdgcn_permlane32_swap(al, al, false, false);
  const auto rh = __builtin_amdgcn_permlane32_swap(ah, ah, false, false);
  lo = __hiloint2double(rh[0], rl[0]);
  hi = __hiloint2double(rh[1], rl[1]);
}
// ac[l & 3] += M[l] * vec[l], l < N (vec as v0 / v1 of half_gather)
template <int N_, int l = 0>
__device__ __forceinline__ void rl_dot(double (&ac)[4], double v0, double v1, const double (&M)[N_]) {
  if constexpr (l < N_) {
    if constexpr (l < 16) fmac_rowbcast<l, l == 0>(ac[l & 3], v0, M[l]);
    else fmac_rowbcast<l - 16, l == 16>(ac[l & 3], v1, M[l]);
    rl_dot<N_, l + 1>(ac, v0, v1, M);
  }
}
template <int N_, int l = 0>
__device__ __forceinline__ void rl_upd(double (&M)[N_], double v0, double v1, double coef) {
  if constexpr (l < N_) {
    if constexpr (l < 16) fmac_rowbcast<l, l == 0>(M[l], v0, coef);
    else fmac_rowbcast<l - 16, l == 16>(M[l], v1, coef);
    rl_upd<N_, l + 1>(M, v0, v1, coef);
  }
}
// y = M v for the rows in this lane's half (v: one element per lane of the half)
template <int N_, int l = 0>
__device__ __forceinline__ void rl_dot1(double& ac, double v0, double v1, const double (&M)[N_]) {
  if constexpr (l < N_) {
    if constexpr (l < 16) fmac_rowbcast<l, l == 0>(ac, v0, M[l]);
    else fmac_rowbcast<l - 16, l == 16>(ac, v1, M[l]);
    rl_dot1<N_, l + 1>(ac, v0, v1, M);
  }
}
template <int N_> __device__ __forceinline__ double rl_matvec(const double (&M)[N_], double v) {
  double v0, v1;
  half_gather(v, v0, v1);
  if constexpr (N_ > 0) {
    // (long horizons run sixteen trajectories per CU on a saturated vector pipe: ONE chain of multiply-adds -- the other waves cover
    //  its latency -- instead of four partial sums with their four register clears and three adds)
    double ac = 0.0;
    rl_dot1<N_>(ac, v0, v1, M);
    return ac;
  } else {
    double ac[4] = {0.0, 0.0, 0.0, 0.0};
    rl_dot<N_>(ac, v0, v1, M);
    return (ac[0] + ac[1]) + (ac[2] + ac[3]);
  }
}

// Symmetric sweep of the tableau (lanes 0-31; lanes 32-63 hold H and must not change: their coefficient is zero) on variable K.
//   true T_ij = rs_i M_i[j];  d = T_kk;  s = +1 (K enters the set) / -1 (K leaves it)
//   T_ij -= T_ik T_kj / d,   T_kj <- s T_kj / d,   T_ik <- s T_ik / d,   T_kk <- -1 / d
template <int N_, int K>
__device__ __forceinline__ void rl_sweep(double (&M)[N_], double& rs, double& rsi, bool rev, double d, int t, int half) {
  asm volatile("" : "+v"(half));
  asm volatile("" : "+v"(t));  // (opaque: the lane tests of the N sweep blocks are not hoisted out of the solve's loops)
  const double dinv = fast_rcp(d);
  const double colk = M[K];
  double v0, v1;
  half_gather(rs * colk, v0, v1);                      // the true column K = row K (symmetric)
  const bool isk = t == K;
  const double c = (half || isk) ? 0.0 : -colk * dinv;  // M_i[j] += c_i T_kj   (stored rows: the row scale cancels)
  rl_upd<N_>(M, v0, v1, c);
  const double sg = rev ? -1.0 : 1.0;
  if (!half) {
    M[K] = isk ? -sg * rsi : sg * colk * dinv;          // column K; (K, K): -1/d = rs_new * (-s / rs_old)
    if (isk) { rs *= sg * dinv; rsi *= sg * d; }
  }
}
// the sweeps that take the tableau from the set Smask to the set Fmask: one straight-line block per variable (the pivot
// column is register K: the index has to be a constant), blocks of variables that stay on their side are skipped by a
// uniform branch.  A pivot of the wrong sign (numerical breakdown) is reported; `drop` then also takes the variable out of F.
template <int N_, int K = 0>
__device__ __forceinline__ void rl_sweep_set(double (&M)[N_], double& rs, double& rsi, unsigned diff, unsigned& Smask, unsigned& Fmask,
                                             bool& broke, bool drop, int t, int half) {
  if constexpr (K < N_) {
    if ((diff >> K) & 1u) {
      const bool rev = (Smask >> K) & 1u;
      const double d = lane_bcast(rs * M[K], K);  // T_kk
      if (!((rev ? -d : d) > 0.0)) {
        broke = true;
        if (drop) Fmask &= ~(1u << K);
      } else {
        rl_sweep<N_, K>(M, rs, rsi, rev, d, t, half);
        Smask ^= (1u << K);
      }
    }
    rl_sweep_set<N_, K + 1>(M, rs, rsi, diff, Smask, Fmask, broke, drop, t, half);
  }
}

// row stride of H / the tableau in LDS: even (16-byte row accesses) with an odd half, so that 16 consecutive rows start in 16
// different groups of four banks -- N = 20 becomes 22 (with stride 20 the rows of lanes 8 apart collide), N = 30 and 10 stay
constexpr int rl_stride(int N) { return ((((N + 1) & ~1) / 2) & 1) ? ((N + 1) & ~1) : ((N + 1) & ~1) + 2; }

struct QpCarry {  // what a solve leaves for the next one (registers of the step loop; the rows themselves wait in LDS)
  unsigned smask;   // variables swept into the carried tableau
  int valid;        // 0: no tableau (rebuild), 1: carried
  unsigned umask;   // of the others (held at a bound by that solve): those at the UPPER bound
  // how the carried tableaux of this trajectory have fared: bits 0-7 consecutive solves whose carried tableau did not contract
  // (stale), bits 8-15 solves since the last one that was given the full pass budget.  A trajectory whose model still moves a lot
  // from step to step (2 stale solves in a row) gets 3 refinement passes instead of 8 before the tableau is given up -- it paid the
  // full budget AND the N sweeps at every step --, and the full budget again every fourth solve (so that it finds its way back).
  // (Round 6: 8 / 3 passes and "kept unless a direction needed 7" instead of 6 / 2 and 5 -- the three numbers were swept with the
  //  plug-in A/B tool, every setting its own kernel on one box: the window moves by - 3.5 % ... + 4 % with them, the code object is the
  //  same to the instruction; profiles/r6_cfg2_pass_budget_ab.txt)
  int trust;
};

// M: on entry lanes t < N of half 0 hold the carried tableau rows (cs.valid) -- loaded by the caller BEFORE H was written into
// sR --, sR holds H (N x N, row-major).  On exit (return false) the rows of the final tableau are in sR and cs describes them.
// Return true: the solve has to continue in the active-set loop of qp_lds from qx_out (H is still in sR; nothing carried).
// (An active-set continuation in registers for crawling solves was built and measured in round 5 -- 14 more spilled vector registers and
//  3.3 % of the settled cfg2 window for solves that are 1 in 10 000 there, profiles/r5_cfg2_active_set_ab.txt -- and is not in this header:
//  tools/experiments/qp_rl_active_set_variant.h.txt.  Crawling solves are finished by the active-set loop of qp_lds, step_v2.h.)
template <int N_, typename IOT = double>
__device__ __forceinline__ bool qp_rl(double* const sR, const double* sf, const StepArgs<double>& a, const StepVar<double>& sv, const int b,
                                      double* qx_out, double* u_slot, double (&M)[N_], double& rs, double& rsi, QpCarry& cs, double up, double xw_pre) {
  typedef double d2_t __attribute__((ext_vector_type(2)));
  static_assert(N_ <= 32, "qp_rl: one variable per lane of a 32-lane half (ownmask, row_newbcast index, diagonal lane)");
  const int tid = local_tid<64>(), half = tid >> 5, t = tid & 31;
  const bool own = t < N_;
  const int B = a.B;
  const double uprev = a.du_mode ? up : 0.0;
  double lb = a.lb, ub = a.ub;
  const double tol = Tol<double>::kkt();
  const double eact = Tol<double>::act() * (ub - lb);
  const double xmaxb = tabs(lb) > tabs(ub) ? tabs(lb) : tabs(ub);
  if (a.du_mode && t == 0) {  // first increment: absolute input range folded in (Tank_System.m:182-188)
    lb = (a.umin - uprev) > lb ? (a.umin - uprev) : lb;
    ub = (a.umax - uprev) < ub ? (a.umax - uprev) : ub;
  }
  const double c0 = tclip(0.0, lb, ub);
  const bool warm = a.x_warm != nullptr, predict_on = (a.qp_predict & 1) != 0;  // (read once: inside the loops every use was a scalar load)
  const int max_iter = a.max_iter;
  constexpr unsigned ownmask = N_ >= 32 ? 0xffffffffu : ((1u << N_) - 1u);
  const int rowi = own ? t : N_ - 1;
  constexpr int NS = rl_stride(N_);
  const double* const hrow = sR + rowi * NS;
  auto load_h_rows = [&]() {
    if constexpr ((N_ & 1) == 0) {
      const d2_t* h2 = reinterpret_cast<const d2_t*>(__builtin_assume_aligned(hrow, 16));
#pragma unroll
      for (int j = 0; j < N_ / 2; ++j) { const d2_t v = h2[j]; M[2 * j] = v.x; M[2 * j + 1] = v.y; }
    } else {
#pragma unroll
      for (int j = 0; j < N_; ++j) M[j] = hrow[j];
    }
  };
  bool carried = cs.valid != 0;  // the tableau in lanes 0-31 is last step's
  const int nstale0 = cs.trust & 0xff, probe_age = (cs.trust >> 8) & 0xff;
  const bool full_budget = nstale0 < 2 || probe_age >= 3;
  const int kr_max = full_budget ? 8 : 3;
  bool went_stale = false;
  const bool carried_at_start = carried;
  unsigned Smask = carried ? cs.smask : 0u;
  if (half || !carried) load_h_rows();  // H rows; without a carried tableau also in lanes 0-31 (T = 2H below)
  if (!carried) {
    if (!half) {
#pragma unroll
      for (int j = 0; j < N_; ++j) M[j] *= 2.0;
    }
    rs = 1.0; rsi = 1.0;
  }
  const double fi = own ? sf[t] : 0.0;
  // magnitude of the terms of the gradient (scale of the KKT tolerance): |f_i| + 2 sum_j |H_ij| max|x|
  double ra = 0.0;
#pragma unroll
  for (int j = 0; j < N_; ++j) ra += tabs(M[j]);
  {
    double lo, hi;
    halves_both_q(ra, lo, hi);
    ra = hi;
  }
  const double gs = tabs(fi) + 2.0 * ra * xmaxb;
  // A solve without a primal warm start (the reference's pastRes = zeros, duffing.py:634-635) starts at clip(0) -- moved onto the
  // face the last minimiser lay on when a tableau is carried: the inputs the last solve held at a bound start at that bound again.
  // The carried tableau is that face's, so the first Newton point costs no sweeps; a wrong guess is an input at a bound with an
  // inward gradient, which the first KKT test frees again.
  const bool held = carried && !warm && !((Smask >> t) & 1u);
  double x = own ? (warm ? tclip(xw_pre, lb, ub) : (held ? (((cs.umask >> t) & 1u) ? ub : lb) : c0)) : 0.0;
  double hx;
  {
    double lo, hi;
    halves_both_q(rl_matvec<N_>(M, x), lo, hi);
    hx = own ? hi : 0.0;
  }
  // the cost at x (wave-uniform: the line search branches on it) is summed when a line search first needs it: the usual iteration of a
  // running loop -- the refined Newton point of the face, unclipped -- cannot raise the cost of a convex quadratic and skips the sums
  double J0 = 0.0, Jhi;
  bool Jok = false;
  int it = 0, status = 1, refresh = 0, polish = 0, rtot = 0, ncrawl = 0, nref = 0;
  bool nopredict = false, rebuild = false;
#ifdef KMPC_TRACE
  int tc_sweeps = 0, tc_rebuild = 0, tc_pass = 0, tc_ls = 0, tc_mv = 1;  // (work counters of the trace build: slots 22-27, summed over the launch)
  const bool tc_carried0 = carried;
#define KCOUNT(x) (x)
#else
#define KCOUNT(x)
#endif
  KTRACE(8);

  while (true) {
    double g = 0.0;
    bool bad = false, inI = false, loose = false;
    if (own) {
      g = 2.0 * hx + fi;
      const bool atl = x <= lb + eact, atu = x >= ub - eact;
      inI = (atl && (g > 0.0)) || (atu && (g < 0.0));
      const double viol = inI ? 0.0 : tabs(g);
      const double res = tabs(x - tclip(x - g, lb, ub));
      const double xs = tabs(x) > 1.0 ? tabs(x) : 1.0;
      bad = !((viol <= tol * gs) || (res <= tol * xs));
      loose = viol > Tol<double>::tight() * gs;
    }
    const unsigned Bmask = (unsigned)__ballot(bad);
    const bool refine = (unsigned)__ballot(loose) != 0u;
    const unsigned Imask = (unsigned)__ballot(inI);
    if ((unsigned)__ballot(own && !(tabs(g) <= 1e300)) != 0u || (Jok && (!(J0 == J0) || tabs(J0) > 1e300))) { status = 2; break; }
    if (Bmask == 0u && (it > 0 || !warm)) {
      if (!refine || polish >= 2) { status = 0; break; }
      ++polish;
    }
    if (it >= max_iter || refresh > 4) { status = 1; break; }
    // (hand-over point: N + 14 iterations / 2 N + 8 refused steps until round 5.  A solve that has not converged in N iterations is
    //  crawling and costs its workgroup ~2 us per further iteration at every lift barrier; handed over at N: post-reset window + 6.6 %,
    //  settled window unchanged; at 12 the fall-back's 70 us are paid by solves that would have finished: no gain --
    //  profiles/r5_cfg2_crawl_threshold_ab.txt)
    if (it >= N_ || ncrawl >= N_ + 4) { status = 3; break; }  // crawling: the caller finishes with the active-set loop of qp_lds
    // a carried tableau that has not brought the point inside the tolerance in four iterations is replaced by a fresh one
    if (carried && it >= 4 && Bmask != 0u) rebuild = true;
    unsigned Fmask = ~Imask & ownmask;
    if (it == 0) KTRACE(9);

    const bool predict = !nopredict && predict_on;
    int rounds = 0;
    bool broke = false, isF = false, exact = false;  // exact: the direction's residual gradient on F is at rounding level
    double pdir = 0.0, hp = 0.0;
    while (true) {
      broke = false;
      for (int pass = 0; pass < 2; ++pass) {
        if (rebuild) {  // T = 2H, nothing swept in
          KCOUNT(++tc_rebuild);
          if (!half) {
            load_h_rows();
#pragma unroll
            for (int j = 0; j < N_; ++j) M[j] *= 2.0;
          }
          rs = 1.0; rsi = 1.0;
          Smask = 0u;
          carried = false;
          rebuild = false;
        }
        if (Smask != Fmask) {  // (one test instead of N when nothing changes sides)
          // more variables change sides than F has members: sweeping F into a fresh 2H is the shorter way
          // (a saturated solution met from the all-free tableau, or the other way round)
          if (__builtin_popcount(Smask ^ Fmask) > __builtin_popcount(Fmask) + 2) {
            KCOUNT(++tc_rebuild);
              if (!half) {
              load_h_rows();
#pragma unroll
              for (int j = 0; j < N_; ++j) M[j] *= 2.0;
            }
            rs = 1.0; rsi = 1.0;
            Smask = 0u;
            carried = false;
          }
          KCOUNT(tc_sweeps += __builtin_popcount(Smask ^ Fmask));
          rl_sweep_set<N_>(M, rs, rsi, Smask ^ Fmask, Smask, Fmask, broke, pass == 1, t, half);
        }
        if (!broke || pass == 1) break;
        if (!carried) ++refresh;  // (a carried tableau that breaks down is simply replaced; only fresh ones count towards giving up)
        rebuild = true;  // pass 1: from 2H, dropping a variable whose pivot fails
      }
      Fmask = Smask;
      if (it == 0 && rounds == 0) KTRACE(10);
      // Newton direction on F: p = T g_F, then refined against H itself -- p <- p + T (g + 2 H p)_F until the gradient the step
      // leaves behind is at rounding level.  With a tableau built in this solve the first residual already is; with a carried
      // one every pass gains a factor |I - T 2H|.  (Both products are ONE instruction stream: lanes 0-31 multiply T, lanes
      // 32-63 multiply H.)
      isF = own && ((Fmask >> t) & 1u);
      {
        double lo, hi;
        halves_both_q(rs * rl_matvec<N_>(M, isF ? g : 0.0), lo, hi);
        pdir = lo;
        bool stale = false;
        exact = false;
        for (int kr = 0;; ++kr) {
          halves_both_q(rl_matvec<N_>(M, isF ? pdir : 0.0), lo, hi);
          hp = hi;  // (H p)_i
          const double r = isF ? g + 2.0 * hp : 0.0;
          if ((unsigned)__ballot(!(tabs(r) <= 1e-13 * gs)) == 0u) { exact = true; break; }
          if (kr >= kr_max || !carried) { stale = carried; break; }  // (a fresh tableau is not refined: rounding in T costs an iteration, as before)
          halves_both_q(rs * rl_matvec<N_>(M, r), lo, hi);
          pdir += lo;
          KCOUNT(++tc_pass);
          nref = kr + 1 > nref ? kr + 1 : nref;  // (the most passes one direction of this solve needed)
        }
        if (stale) {  // the carried tableau does not contract: this direction again from 2H
          went_stale = true;
          rebuild = true;
          continue;
        }
      }
      if (!predict || broke || rounds >= N_) break;
      const double cand = x + pdir;
      const double over = isF ? (cand - ub > lb - cand ? cand - ub : lb - cand) : -1.0;
      const unsigned offenders = (unsigned)__ballot(over > 0.0);
      if (offenders == 0u) break;  // (the usual case: the Newton point is feasible)
      // prediction is for the one or two inputs that cross a bound from step to step; a Newton point with many offenders
      // (a solve started far from its solution, e.g. at zeros with a saturated minimiser) is projected as a whole instead:
      // one iteration moves them all, where the rounds would take one sweep and one Newton point each
      if (rounds == 0 && __builtin_popcount(offenders) > 2) break;
      const double worst = half_max(over);
      const int jl = __ffs((int)(unsigned)__ballot(over == worst)) - 1;  // the variable (= its lane in the lower half)
      const double bj = cand > ub ? ub : lb;
      const double delta = lane_bcast(bj - x, jl);
      if (tid == jl) qx_out[jl] = bj;
      if (own) g += 2.0 * sR[t * NS + jl] * delta;  // gradient at the new base point (H is intact in LDS)
      Fmask &= ~(1u << jl);
      ++rounds;
    }
    if (it == 0) KTRACE(11);

    double alpha = 1.0, xa = x, hxa = 0.0, Ja = J0;
    bool redo = false;
    while (true) {
      double dstep = pdir;
      if (!isF) {
        const double g0 = 2.0 * hx + fi;
        dstep = own ? ((g0 > 0.0 ? lb : (g0 < 0.0 ? ub : x)) - x) : 0.0;
        if (rounds > 0 && own && !inI) dstep = qx_out[t] - x;
      }
      const double xu = x + alpha * dstep;
      xa = own ? tclip(xu, lb, ub) : 0.0;
      // the full, unclipped Newton step: H xa = H x + H p, and H p is what the refinement ended with
      const bool plain = alpha == 1.0 && rounds == 0 && (unsigned)__ballot(own && (isF ? xa != xu : xa != x)) == 0u;
      if (plain) {
        hxa = own ? hx + hp : 0.0;
      } else {
        double lo, hi;
        halves_both_q(rl_matvec<N_>(M, xa), lo, hi);
        hxa = own ? hi : 0.0;
      }
      if (plain && exact) {  // (the exact Newton point of the face, not clipped: accepted without the sums)
        Jok = false;
        break;
      }
      if (!Jok) {
        half_sums_u(own ? x * (hx + fi) : 0.0, J0, Jhi);
        Jok = true;
      }
      const double pJa = own ? xa * (hxa + fi) : 0.0;
      if (rounds > 0) {
        half_sums_u(pJa, Ja, Jhi);
        if (!(Ja <= J0)) redo = true;
        break;
      }
      const double pdec = own ? (isF ? alpha * (-g * pdir) : g * (x - xa)) : 0.0;
      double sD;
      half_sums_u(half ? pdec : pJa, Ja, sD);  // lanes 0-31 sum the cost, lanes 32-63 the predicted decrease
      const double mag = tabs(J0) > tabs(Ja) ? tabs(J0) : tabs(Ja);
      if ((J0 - Ja >= 1e-4 * sD - Tol<double>::slack() * mag) || alpha < 1e-10) break;
      if (carried) { redo = true; break; }  // a stale tableau gave a poor direction: this iteration again with a fresh one
      alpha *= 0.25;
      KCOUNT(++tc_ls);
      ++ncrawl;
    }
    if (redo) {
      if (carried && rounds == 0) { rebuild = true; continue; }
      nopredict = true;  // as a plain projected-Newton step
      ncrawl += 3;
      continue;
    }
    nopredict = false;
    rtot += rounds;
    x = xa;
    hx = hxa;
    if (Jok) J0 = Ja;
    if (it == 0) KTRACE(12);
    ++it;
  }
  KTRACE(13);
#ifdef KMPC_TRACE
  if (tid == 0 && b < 8192) {
    kmpc_trace_buf[b * 32 + 15] = (unsigned long long)(it + rtot);
    kmpc_trace_buf[b * 32 + 22] += (unsigned long long)tc_sweeps;
    kmpc_trace_buf[b * 32 + 23] += (unsigned long long)tc_rebuild;
    kmpc_trace_buf[b * 32 + 24] += (unsigned long long)tc_pass;
    kmpc_trace_buf[b * 32 + 25] += (unsigned long long)it;
    kmpc_trace_buf[b * 32 + 26] += (unsigned long long)(tc_ls + rtot * 1000);
    kmpc_trace_buf[b * 32 + 27] += (unsigned long long)((tc_carried0 ? 1 : 0) + ((it == 1 && tc_sweeps == 0 && tc_rebuild == 0 && rtot == 0) ? 1000 : 0));
    (void)tc_mv;
  }
#endif
  if (status == 2) x = own ? c0 : 0.0;
  if (status == 3) {
    if (own && !half) qx_out[t] = x;
    cs.valid = 0;
    return true;
  }
  if (own && !half) {
    if (a.Useq) io_st<IOT>(a.Useq, (size_t)t * B + b, x);
    if (a.x_warm) a.x_warm[(size_t)t * B + b] = x;
  }
  if (tid == 0) {
    const double uout = a.du_mode ? uprev + x : x;  // U0 = U0 + dU*(1)   (Tank_System.m:192)
    *u_slot = x;  // (the first move, for a covariance update done ahead: step_v2.h)
    if (sv.U0) io_st<IOT>(sv.U0, b, uout);
    if (a.u_store) a.u_store[b] = uout;
    if (a.plant >= 0) {  // x_loc = f_update(0, x_loc, u_loc)   (duffing.py:871)
      double x1, x2;  // (x_next, when given, is the roll-out's LDS slot -- step_body.h lds_ld: as a select of two addresses these were
      // flat loads, which wait for both counters: behind the step's write-back a drain of every outstanding store, K = 20 - 4 %)
      if (sv.x_next) { x1 = lds_ld(sv.x_next); x2 = lds_ld(sv.x_next + 1); }
      else { x1 = io_ld<IOT>(a.X_rw, b); x2 = io_ld<IOT>(a.X_rw, (size_t)B + b); }
      plant_apply<double>(a.plant, sv.plant_switched, a.plant_h, x1, x2, uout);
      io_st<IOT>(a.X_rw, b, x1);
      io_st<IOT>(a.X_rw, (size_t)B + b, x2);
      if (sv.x_next) { lds_st(sv.x_next, x1); lds_st(sv.x_next + 1, x2); }
    }
    if (sv.x_next) {
      lds_i32* const acc = (lds_i32*)reinterpret_cast<int*>(sv.x_next + 2);
      acc[0] = acc[0] > status ? acc[0] : status;
      acc[1] += it + rtot;
      // what this solve cost beyond the easy case, in units of ~0.08 us (the time regression of tools/dbg/qp_work.py: 0.23 us per
      // sweep, 0.38 per refinement pass, ~0.6 per further Newton point): the roll-out deals its trajectories to the waves by it
      acc[2] += 8 * (it + rtot - 1) + 5 * nref + (carried ? 3 * __builtin_popcount(cs.smask ^ Smask) : 3 * N_);
    } else {
      if (a.status) a.status[b] = a.accumulate ? (a.status[b] > status ? a.status[b] : status) : status;
      if (a.iters) a.iters[b] = a.accumulate ? a.iters[b] + it + rtot : it + rtot;
    }
  }
  // the tableau takes H's place in LDS until the next solve.  A solve that worked with the carried tableau and needed many
  // iterations, or seven refinement passes for one direction (the contraction |I - T 2H| has worn), leaves nothing: the next one
  // starts from 2H.
  const bool keep = status == 0 && !(carried && (it >= 4 || nref >= 7));
  if (keep && own && !half) {
    double* const trow = sR + t * NS;
    if constexpr ((N_ & 1) == 0) {
      d2_t* t2 = reinterpret_cast<d2_t*>(__builtin_assume_aligned(trow, 16));
#pragma unroll
      for (int j = 0; j < N_ / 2; ++j) { d2_t v; v.x = M[2 * j]; v.y = M[2 * j + 1]; t2[j] = v; }
    } else {
#pragma unroll
      for (int j = 0; j < N_; ++j) trow[j] = M[j];
    }
  }
  cs.valid = keep ? 1 : 0;
  if (carried_at_start) {
    const int ns = went_stale ? (nstale0 < 255 ? nstale0 + 1 : 255) : 0;
    const int age = full_budget ? 0 : probe_age + 1;
    cs.trust = ns | (age << 8);
  }
  cs.smask = Smask;
  cs.umask = (unsigned)__ballot(own && x >= ub - eact);
  KTRACE(14);
  return false;
}

}  // namespace kmpc
