// rollout_dyn.h -- the fused roll-out WITHOUT a workgroup barrier per step: lift groups are formed at run time.
//
// The lock-step kernel (rollout_kernel) makes the 16 trajectories of a workgroup wait for the slowest box QP of the
// 16 at every step, because the encoder of all 16 is one cooperative MFMA product (in-kernel stamps: 16-18 us of a
// 43-47 us step are spent waiting at that barrier; a single QP that needs 10-30 Newton solves stalls 15 waves).
// Here a wave that has finished step k joins a QUEUE; as soon as `G` waves are queued (or the queue head has waited
// `timeout`, or nobody else is left to come) they form a group, lift their G states together on
// v_mfma_f64_4x4x4_4b_f64 (16 rows x 4 trajectories x 4 k per instruction: four columns cost a quarter of sixteen) and
// go on with their own steps.  Groups are "the next G to arrive": fast trajectories never wait for slow ones, a
// straggler only delays itself and catches up when the others have finished.  Nothing in the kernel uses s_barrier
// after the prologue; waves synchronise through LDS words (a spin lock around the queue, per-wave mail boxes, a
// per-group arrival counter), every spin is bounded and gives the SIMD away with s_sleep.
//
// Reference loop body: duffing.py:823-1012 (847 lift, 900-967 update, 857-861 solve, 871 plant).
#pragma once
#include "step_body.h"

namespace kmpc {

// control words (32-bit) in LDS, behind the keep area
enum { DW_LOCK = 0, DW_NWAIT = 1, DW_ACTIVE = 2, DW_ERR = 3, DW_Q = 4, DW_EPOCH = 20, DW_DESC = 36, DW_MEMB = 52,
       DW_BAR = 68, DW_BASE = 84, DW_KMAX = 100, DW_WORDS = 102 };
constexpr int DYN_CTL_ELEMS = (DW_WORDS + 1) / 2 + 2;  // in doubles
constexpr int DYN_SPIN_LIMIT = 1 << 21;                // bounded spins: ~0.2 s of s_sleep, then the error word is set

typedef __attribute__((address_space(3))) int lds_int_t;

__device__ __forceinline__ int dyn_ld(lds_int_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void dyn_st(lds_int_t* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// the queue is touched by one lane of one wave at a time
__device__ __forceinline__ void dyn_lock(lds_int_t* ctl) {
  int spins = 0;
  while (true) {
    int expected = 0;
    if (__hip_atomic_compare_exchange_strong(&ctl[DW_LOCK], &expected, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_WORKGROUP))
      break;
    __builtin_amdgcn_s_sleep(1);
    if (++spins > DYN_SPIN_LIMIT || dyn_ld(&ctl[DW_ERR])) { dyn_st(&ctl[DW_ERR], 1); break; }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void dyn_unlock(lds_int_t* ctl) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  dyn_st(&ctl[DW_LOCK], 0);
}
// (lock held) the queued waves become a group: position, size and leader go to every member's mail box
__device__ __forceinline__ void dyn_dispatch(lds_int_t* ctl) {
  const int g = dyn_ld(&ctl[DW_NWAIT]);
  if (g <= 0) return;
  const int leader = dyn_ld(&ctl[DW_Q]);
  int pack = 0;
  for (int i = 0; i < g; ++i) pack |= dyn_ld(&ctl[DW_Q + i]) << (8 * i);
  dyn_st(&ctl[DW_MEMB + leader], pack);
  // the leader's arrival counter only ever grows (a member of an earlier group that is slow to see its last value must
  // not find it reset): this group counts from where it stands -- every earlier add to it has happened, its leader is here
  dyn_st(&ctl[DW_BASE + leader], dyn_ld(&ctl[DW_BAR + leader]));
  for (int i = 0; i < g; ++i) dyn_st(&ctl[DW_DESC + ((pack >> (8 * i)) & 255)], i | (g << 4) | (leader << 8));
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  for (int i = 0; i < g; ++i) {
    lds_int_t* e = &ctl[DW_EPOCH + ((pack >> (8 * i)) & 255)];
    dyn_st(e, dyn_ld(e) + 1);
  }
  dyn_st(&ctl[DW_NWAIT], 0);
}
// wave `w` (lane 0 acts) queues up and returns its group descriptor: pos | size << 4 | leader << 8
__device__ __forceinline__ int dyn_join(lds_int_t* ctl, int w, int lane, int G, unsigned timeout, int& epoch, int k, int& lag) {
  int desc = 0, lg = 0;
  if (lane == 0) {
    dyn_lock(ctl);
    const int km = dyn_ld(&ctl[DW_KMAX]);  // how far the workgroup's first trajectory has come
    if (k > km) dyn_st(&ctl[DW_KMAX], k);
    lg = km > k ? km - k : 0;
    const int q = dyn_ld(&ctl[DW_NWAIT]);
    dyn_st(&ctl[DW_Q + q], w);
    dyn_st(&ctl[DW_NWAIT], q + 1);
    if (q + 1 >= G || q + 1 >= dyn_ld(&ctl[DW_ACTIVE])) dyn_dispatch(ctl);
    dyn_unlock(ctl);
    const unsigned long long t0 = wall_clock64();
    int spins = 0;
    while (dyn_ld(&ctl[DW_EPOCH + w]) == epoch) {
      __builtin_amdgcn_s_sleep(1);
      if (q == 0 && (unsigned)(wall_clock64() - t0) > timeout) {  // the queue head does not wait for a full group for ever
        dyn_lock(ctl);
        if (dyn_ld(&ctl[DW_EPOCH + w]) == epoch) dyn_dispatch(ctl);
        dyn_unlock(ctl);
      }
      if (++spins > DYN_SPIN_LIMIT || dyn_ld(&ctl[DW_ERR])) { dyn_st(&ctl[DW_ERR], 1); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    desc = dyn_ld(&ctl[DW_DESC + w]);
  }
  ++epoch;
  lag = __builtin_amdgcn_readfirstlane(lg);
  return __builtin_amdgcn_readfirstlane(desc);
}
// a wave that has run all its steps leaves; the waves still queued may be the last ones
__device__ __forceinline__ void dyn_leave(lds_int_t* ctl, int lane) {
  if (lane == 0) {
    dyn_lock(ctl);
    const int a = dyn_ld(&ctl[DW_ACTIVE]) - 1;
    dyn_st(&ctl[DW_ACTIVE], a);
    const int nq = dyn_ld(&ctl[DW_NWAIT]);
    if (nq > 0 && nq >= a) dyn_dispatch(ctl);
    dyn_unlock(ctl);
  }
}
// the g members of a group meet: everything they wrote to LDS before is visible to all of them afterwards
__device__ __forceinline__ void dyn_group_sync(lds_int_t* ctl, int leader, int target, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) {
    __hip_atomic_fetch_add(&ctl[DW_BAR + leader], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    int spins = 0;
    while (dyn_ld(&ctl[DW_BAR + leader]) < target) {
      __builtin_amdgcn_s_sleep(0);
      if (++spins > DYN_SPIN_LIMIT || dyn_ld(&ctl[DW_ERR])) { dyn_st(&ctl[DW_ERR], 1); break; }
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// LDS elements (doubles) of the dynamic kernel: 16 per-wave regions | psi slots (Lp x 16) | x slots (16 x 4) | control
static size_t rollout_dyn_lds_elems(int n, int L, int q, int N, int Lp, int* wstride) {
  const size_t per_wave = (step_lds_bytes(n, L, q, N, sizeof(double), nullptr, nullptr, !tableau_saves_lds(N, L)) / sizeof(double) + 1) & ~(size_t)1;
  if (wstride) *wstride = (int)per_wave;
  return per_wave * 16 + (size_t)Lp * 16 + 64 + DYN_CTL_ELEMS;
}

template <int L_, int N_, int Q_, int KS_>
__global__ __launch_bounds__(1024) void rollout_dyn_kernel(const RolloutArgs<double> ra) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* const smem = reinterpret_cast<double*>(smem_raw);
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int B = ra.s.B, n = ra.s.n, L = L_;
  const int b0 = blockIdx.x * 16, b = b0 + wave;
  const bool live = b < B;
  double* const sPsi = smem + ra.keep_off;       // psi slots: element (row, wave) at row * 16 + wave
  double* const sXn = sPsi + ra.Lp * 16;         // x slots: wave w at 4 w (+2: status / iteration counters of the launch)
  lds_int_t* const ctl = (lds_int_t*)(sXn + 64);
  if (tid0 < DW_WORDS) ctl[tid0] = 0;
  if (tid0 < 64) sXn[tid0] = 0.0;
  __syncthreads();
  if (tid0 == 0) {
    const int nlive = B - b0 < 16 ? B - b0 : 16;
    ctl[DW_ACTIVE] = nlive;
  }
  if ((tid0 & 63) < 4)
    sXn[wave * 4 + (tid0 & 63)] = (live && (int)(tid0 & 63) < n) ? ra.s.X_rw[(size_t)(tid0 & 63) * B + b] : 0.0;
  __syncthreads();  // the only workgroup barriers of the kernel
  if (!live) return;

  bool have_prev = ra.have_prev != 0, fresh = ra.rls_fresh != 0;
  int cur = ra.cur;
  int epoch = 0;
  double psi_prev_reg = 0.0;
  if (ra.have_prev && (int)(tid0 & 63) < L) psi_prev_reg = ra.psi[ra.cur ^ 1][(size_t)b * L + (tid0 & 63)];
  typedef const RolloutArgs<double> __attribute__((address_space(4))) * kernarg_ptr_t;
  for (int k = 0; k < ra.steps; ++k) {
    // (see rollout_kernel: step arguments are re-read from the kernel-argument segment, lane-derived values recomputed)
    kernarg_ptr_t kp = (kernarg_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    const RolloutArgs<double> __attribute__((address_space(4)))& R = *kp;
    const StepArgs<double>& a = *(const StepArgs<double>*)(&kp->s);
    int wv = wave;
    asm volatile("" : "+s"(wv));
    const int lane = local_tid<64>();
#ifdef KMPC_TRACE
    if (lane == 0 && b < 8192) kmpc_trace_buf[b * 32 + 16] = wall_clock64();  // this wave is ready for step k
    if (lane == 0 && b < 8192 && k == 0) for (int i = 24; i < 30; ++i) kmpc_trace_buf[b * 32 + i] = 0;
#endif
    // ---- queue up; the group lifts its states together
    int lag = 0;
    const int desc = dyn_join(ctl, wv, lane, R.dyn_group, (unsigned)R.dyn_timeout, epoch, k, lag);
    // Instruction issue is oldest-first: without this the younger waves of a SIMD lose every arbitration, fall behind
    // step after step (in-kernel stamps: 24 vs 35 us per step body) and the launch lasts as long as the unluckiest one.
    // A trajectory that is behind the workgroup's first one runs at a higher priority until it has caught up.
    __builtin_amdgcn_s_setprio(3);  // the lift is the part others wait for: it runs ahead of the step bodies
    const int pos = desc & 15, g = (desc >> 4) & 15, leader = (desc >> 8) & 31;
    const int mm = __builtin_amdgcn_readfirstlane(dyn_ld(&ctl[DW_MEMB + leader]));
    const int bar0 = __builtin_amdgcn_readfirstlane(dyn_ld(&ctl[DW_BASE + leader]));
#ifdef KMPC_TRACE
    if (lane == 0 && b < 8192) kmpc_trace_buf[b * 32 + 22] = wall_clock64();  // group formed
#endif
    double psi_i = 0.0;
    if (g > 0) {
      constexpr int NC = 4;
      const int KS = KS_ > 0 ? KS_ : R.KS, Hp = KS_ > 0 ? (KS_ <= 28 ? 112 : 128) : R.Hp, MTH = Hp >> 4, MTO = R.Lp >> 4;
      double* const sA0 = smem + leader * R.wstride;  // activations (Hp x 4) x 2 in the leader's region, dead between steps
      double* const sA1 = sA0 + Hp * NC;
      const int boff = (lane >> 4) * NC + (lane & 3);        // B operand: element (k, j) of a k-step
      const int drow = ((lane >> 2) & 3) * 4 + (lane >> 4);  // D: row within the tile
      const int dcol = lane & 3;
      const int mcol = (mm >> (8 * (dcol < g ? dcol : 0))) & 255;  // wave that owns column dcol (columns >= g: a copy of column 0)
      int phase = 0;
      // first layer: one k-step (K = n <= 4, W1 zero-padded to 4 columns), bias as the accumulator input
      {
        const double xb = sXn[mcol * 4 + (lane >> 4)];
        for (int th = pos; th < MTH; th += g) {
          const double a1 = R.W1[4 * (16 * th + (lane & 15)) + (lane >> 4)];
          const double c1 = R.b1[16 * th + drow];
          const double v = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, xb, c1, 0, 0, 0);
          sA0[(16 * th + drow) * NC + dcol] = v > 0.0 ? v : 0.0;
        }
      }
      if constexpr (KS_ > 0) {
        // Whole tiles of A-fragments in flight (two register sets of KS_ fragments): a tile's KS_ loads are one L2 round
        // trip instead of KS_ / 8 exposed ones, the next tile -- of this layer or the first of the next, weights do not
        // depend on the activations -- is requested before the current one is multiplied, i.e. it travels across the
        // group's synchronisation.  (With batches of 8 the group's lift took 12 us under load, 20 exposed round trips.)
        // Half tiles: two register sets of HA = ceil(KS_/2) fragments; while one half is multiplied the next one is in
        // flight (the second half of the tile, the first half of the wave's next tile, or of its first tile of the next
        // layer: that one travels across the group's synchronisation).  Two whole tiles in flight would need 100 VGPRs.
        constexpr int HA = (KS_ + 1) / 2, HB = KS_ - HA;
        double fa[HA], fb[HA];
        auto load_a = [&](const double* Wp, int t) {
#pragma unroll
          for (int i = 0; i < HA; ++i) fa[i] = Wp[((size_t)t * KS_ + i) * 64 + lane];
        };
        auto load_b = [&](const double* Wp, int t) {
#pragma unroll
          for (int i = 0; i < HB; ++i) fb[i] = Wp[((size_t)t * KS_ + HA + i) * 64 + lane];
        };
        // k-step ks goes to chain ks & 1, as in the lock-step kernel (same summation order)
        auto mul_a = [&](const double* act, double& acc0, double& acc1) {
#pragma unroll
          for (int i = 0; i < HA; ++i) {
            double& ac = (i & 1) ? acc1 : acc0;
            ac = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[i], act[i * 4 * NC + boff], ac, 0, 0, 0);
          }
        };
        auto mul_b = [&](const double* act, double& acc0, double& acc1) {
#pragma unroll
          for (int i = 0; i < HB; ++i) {
            double& ac = ((HA + i) & 1) ? acc1 : acc0;
            ac = __builtin_amdgcn_mfma_f64_4x4x4f64(fb[i], act[(HA + i) * 4 * NC + boff], ac, 0, 0, 0);
          }
        };
        // first tile of a layer for this position (output tiles are handed out from the last position down)
        auto first_tile = [&](bool last) -> int { return last ? (g - 1 - pos) : pos; };
        {
          const bool last0 = R.nhh == 0;
          const int t0 = first_tile(last0);
          if (t0 < (last0 ? MTO : MTH)) load_a(last0 ? R.Wop : R.Whp[0], t0);
        }
#ifdef KMPC_TRACE
        if (lane == 0 && b < 8192) kmpc_trace_buf[b * 32 + 24] += wall_clock64() - kmpc_trace_buf[b * 32 + 22];  // layer 1 computed
#endif
        dyn_group_sync(ctl, leader, bar0 + g * (++phase), lane);
#ifdef KMPC_TRACE
        if (lane == 0 && b < 8192) kmpc_trace_buf[b * 32 + 25] += wall_clock64() - kmpc_trace_buf[b * 32 + 22];  // ... and synchronised
#endif
        for (int h = 0; h <= R.nhh; ++h) {
          const bool last = h == R.nhh;
          const double* act = (h & 1) ? sA1 : sA0;
          double* actn = (h & 1) ? sA0 : sA1;
          const double* Wp = last ? R.Wop : R.Whp[h & 1];
          const double* bias = last ? R.bo : R.bh[h & 1];
          const int ntile = last ? MTO : MTH;
          const bool nl = h + 1 == R.nhh;
          const double* Wn = last ? nullptr : (nl ? R.Wop : R.Whp[(h + 1) & 1]);
          const int tnext = first_tile(nl), ntn = last ? 0 : (nl ? MTO : MTH);
          for (int t = first_tile(last); t < ntile; t += g) {  // first half of tile t is in fa / on its way
            load_b(Wp, t);
            double acc0 = bias[16 * t + drow], acc1 = 0.0;
            mul_a(act, acc0, acc1);
            if (t + g < ntile) load_a(Wp, t + g);
            else if (tnext < ntn) load_a(Wn, tnext);  // the next layer's first half tile travels across the synchronisation
            mul_b(act, acc0, acc1);
            const double v = acc0 + acc1;
            if (last) { if (dcol < g) sPsi[(16 * t + drow) * 16 + mcol] = v; }
            else actn[(16 * t + drow) * NC + dcol] = v > 0.0 ? v : 0.0;
          }
          if (!last && !(first_tile(last) < ntile) && tnext < ntn) load_a(Wn, tnext);  // (a position without a tile in this layer)
#ifdef KMPC_TRACE
          if (lane == 0 && b < 8192) kmpc_trace_buf[b * 32 + 26 + 2 * (h > 0)] += wall_clock64() - kmpc_trace_buf[b * 32 + 22];  // layer computed (26: h = 0, 28: h >= 1 summed)
#endif
          dyn_group_sync(ctl, leader, bar0 + g * (++phase), lane);
#ifdef KMPC_TRACE
          if (lane == 0 && b < 8192) kmpc_trace_buf[b * 32 + 27 + 2 * (h > 0)] += wall_clock64() - kmpc_trace_buf[b * 32 + 22];
#endif
        }
      } else {
      dyn_group_sync(ctl, leader, bar0 + g * (++phase), lane);
      for (int h = 0; h <= R.nhh; ++h) {
        const bool last = h == R.nhh;
        const double* act = (h & 1) ? sA1 : sA0;
        double* actn = (h & 1) ? sA0 : sA1;
        const double* Wp = last ? R.Wop : R.Whp[h & 1];
        const double* bias = last ? R.bo : R.bh[h & 1];
        const int ntile = last ? MTO : MTH;
        // output tiles are handed out from the last position down (the first positions hold more hidden tiles)
        for (int t = last ? (g - 1 - pos) : pos; t < ntile; t += g) {
          double acc0 = bias[16 * t + drow], acc1 = 0.0;
          double af[2][RO_KB2];
          ro_load_afrags(Wp, KS, t, 0, lane, af[0]);
#pragma unroll
          for (int bt = 0; bt < 32 / RO_KB2; ++bt) {
            const int kb = bt * RO_KB2;
            if (kb + RO_KB2 < KS) ro_load_afrags(Wp, KS, t, kb + RO_KB2, lane, af[(bt + 1) & 1]);
            if (kb < KS) {
#pragma unroll
              for (int i = 0; i < RO_KB2; i += 2) {
                if (kb + i < KS) acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(af[bt & 1][i], act[(kb + i) * 4 * NC + boff], acc0, 0, 0, 0);
                if (kb + i + 1 < KS) acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(af[bt & 1][i + 1], act[(kb + i + 1) * 4 * NC + boff], acc1, 0, 0, 0);
              }
            }
          }
          const double v = acc0 + acc1;
          if (last) { if (dcol < g) sPsi[(16 * t + drow) * 16 + mcol] = v; }
          else actn[(16 * t + drow) * NC + dcol] = v > 0.0 ? v : 0.0;
        }
        dyn_group_sync(ctl, leader, bar0 + g * (++phase), lane);
      }
      }
      if (lane < L) psi_i = sPsi[lane * 16 + wv];
    }
    if (lag >= 2) __builtin_amdgcn_s_setprio(2);
    else if (lag == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
#ifdef KMPC_TRACE
    if (lane == 0 && b < 8192) kmpc_trace_buf[b * 32 + 17] = wall_clock64();  // lift done
    if (lane == 0 && b < 8192 && k == 0) kmpc_trace_buf[b * 32 + 19] = wall_clock64();
#endif
    {
      int woff = wv * R.wstride, bk = b;
      asm volatile("" : "+s"(woff), "+s"(bk));
      double* const wsm = smem + woff;
      double* const psi_now = R.psi[cur];
      if (lane < L) psi_now[(size_t)b * L + lane] = psi_i;
      StepVar<double> sv;
      sv.psi_now = psi_now;
      sv.psi_prev = R.psi[cur ^ 1];
      sv.psi_in_regs = 1;
      sv.psi_now_v = psi_i;
      sv.psi_prev_v = psi_prev_reg;
      sv.phases = PH_CONDENSE | PH_QP | (have_prev ? PH_RLS : 0);
      sv.first_update = fresh ? 1 : 0;
      sv.plant_switched = (R.switch_step >= 0 && R.step0 + k >= R.switch_step) ? 1 : 0;
      sv.U0 = R.U_log ? R.U_log + (size_t)k * B : a.U0;
      sv.x_next = sXn + wv * 4;
      step_body<double, 64, L_, N_, Q_>(a, sv, bk, wsm);
      if (R.X_log) {
        __threadfence_block();
        if (lane < n) R.X_log[((size_t)k * n + lane) * B + b] = a.X_rw[(size_t)lane * B + b];
      }
    }
#ifdef KMPC_TRACE
    if (lane == 0 && b < 8192) {
      const unsigned long long t18 = wall_clock64();
      kmpc_trace_buf[b * 32 + 18] = t18;  // step k done
      const unsigned long long dl = kmpc_trace_buf[b * 32 + 17] - kmpc_trace_buf[b * 32 + 16], db = t18 - kmpc_trace_buf[b * 32 + 17];
      const unsigned long long dq = kmpc_trace_buf[b * 32 + 22] - kmpc_trace_buf[b * 32 + 16];
      kmpc_trace_buf[b * 32 + 20] = (k == 0 ? 0ull : kmpc_trace_buf[b * 32 + 20]) + dl;
      kmpc_trace_buf[b * 32 + 21] = (k == 0 ? 0ull : kmpc_trace_buf[b * 32 + 21]) + db;
      kmpc_trace_buf[b * 32 + 23] = (k == 0 ? 0ull : kmpc_trace_buf[b * 32 + 23]) + dq;
    }
#endif
    if (have_prev) fresh = false;
    have_prev = true;
    cur ^= 1;
    psi_prev_reg = psi_i;
  }
  dyn_leave(ctl, (int)(tid0 & 63));
  if ((tid0 & 63) == 0) {  // (the host zeroed status / iters before the launch)
    const int* const acc = reinterpret_cast<const int*>(sXn + wave * 4 + 2);
    const int err = dyn_ld(&ctl[DW_ERR]) ? 9 : 0;  // a bounded spin ran out: the results of this workgroup are not valid
    if (ra.s.status) ra.s.status[b] = acc[0] > err ? acc[0] : err;
    if (ra.s.iters) ra.s.iters[b] = acc[1];
  }
}

}  // namespace kmpc
