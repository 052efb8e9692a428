// rollout_kernel.hip -- the fused roll-out: `steps` iterations of the reference loop body (duffing.py:823-1012) in ONE
// launch, step_body (step_body.h) inside, encoder on MFMA.  Compiled as its own translation unit (in parallel with
// step_kernel.hip); the trace build includes it into step_kernel.hip instead.
//
// Three kinds of translation unit include this file:
//   rollout_kernel.o        the instantiations compiled into libkoopmpc.so (the dimension sets BASELINE.json and the reference's
//                           scripts use) and the launchers / dispatch
//   rollout_kernel_io32.o   the same sets behind float32 panels (KMPC_ROLLOUT_IO32_TU)
//   a PLUG-IN               ONE dimension set, workgroup size and lift variant, compiled when a handle of a dimension set without a
//                           built-in instantiation is created (KMPC_ROLLOUT_JIT_TU, rollout_jit.hip; rollout_plugin.hip drives hipcc
//                           and keeps the code objects on disk): the fused path is a property of the library, not of a list
#include "step_body.h"
#include "step_v2.h"
#include "dare_device.h"
#ifdef KMPC_ROLLOUT_JIT_TU
// a plug-in is self-contained: the library decides the workgroup size and hands it over, no debug switches are read
#define dbg_env(name) ((const char*)nullptr)
#endif

namespace kmpc {

// ---------------------------------------------------------------------------------------
// Fused roll-out: `steps` iterations of the reference loop body (duffing.py:823-1012) in ONE launch.
//
// A workgroup of 16 waves owns 16 trajectories and walks them through all the steps; workgroups never
// synchronise with each other, so the launch no longer waits for the slowest QP of the whole batch at every
// step (a per-step launch is one round of waves: it lasts as long as its slowest trajectory), only the 16
// trajectories of a workgroup meet -- at the lift, which they compute together:
//   MLP encoder on v_mfma_f64_16x16x4_f64 with the 16 trajectories as the N dimension; wave w owns hidden M
//   tile w over the whole K range, bias + ReLU are applied on the accumulator registers and written straight
//   into the next layer's B-fragment layout (one barrier per layer).  The weights are pre-packed A-fragments.
//   The lift scratch overlays the per-wave LDS regions, which are dead between two steps.
// RBF lift: every wave lifts its own state, the waves of a workgroup never meet.
// ---------------------------------------------------------------------------------------
typedef double d4_t __attribute__((ext_vector_type(4)));
#ifdef KMPC_TRACE
// lift timeline of the trace build: wave 0 of a workgroup adds (now - barrier-0 release) into slot 32 * (b0 + 1 + i) + 31 ... kept simple:
// stamps 0..8 of the LAST step are written to the slots 22..30 of trajectory b0 + 1 (whose wave does not stamp these slots itself when
// the work counters of qp_rl are off); read with tools/dbg/lift_timeline.py
#define LSTAMP(i)                                                                                      \
  do {                                                                                                 \
    if (wv == 0 && lane == 0 && b0 < 8192) kmpc_lift_buf[(b0 >> 4) * 16 + (i)] = wall_clock64();      \
  } while (0)
__device__ unsigned long long kmpc_lift_buf[512 * 16];
extern "C" int kmpc_lift_trace_read(void* host, size_t bytes) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(kmpc_lift_buf), bytes, 0, hipMemcpyDeviceToHost);
}
// per trajectory (< 4096) and step (< 32) of the last launch: low words of the clock when the wave was ready for the step, when the lift
// was done, when the step body ended (10 ns ticks; tools/dbg/step_timeline.py)
__device__ unsigned kmpc_step_buf[4096 * 96];
extern "C" int kmpc_step_trace_read(void* host, size_t bytes) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(kmpc_step_buf), bytes, 0, hipMemcpyDeviceToHost);
}
#else
#define LSTAMP(i)
#endif
// NW = waves (= trajectories) per workgroup: 16 (one workgroup per CU) or 8 (two per CU, MFMA tiles half empty:
// used when the batch would otherwise leave CUs without a workgroup).
constexpr int RO_ACT = 32 * 64;                     // B-fragments of one activation vector set (Hp <= 128)
constexpr int ro_scratch() { return 2 * RO_ACT; }     // overlays the per-wave regions between two steps
// eight-trajectory workgroups keep eight activation columns (the other eight MFMA columns repeat them): half the scratch, so that
// TWO such workgroups fit on a CU beside their per-wave regions
constexpr int ro_act8() { return 32 * 32; }
static size_t ro_scratch_of(int waves) { return waves == 4 ? RO_ACT + 128 * 4 : (waves == 8 ? 2 * ro_act8() : ro_scratch()); }
constexpr int RO_KB2 = 8;                             // A-fragments are fetched and multiplied in batches of 8 k-steps
// not overlaid: psi (Lp x columns), x_{k+1} of the trajectories (columns x 4); 16 MFMA columns for 8 / 16 waves, 4 for 4
static int ro_cols(int waves) { return waves == 4 ? 4 : 16; }
static int ro_keep(int Lp, int waves) { return Lp * ro_cols(waves) + 4 * ro_cols(waves); }

// A-fragments of tile `tile`, k-steps ks0 .. ks0+15 (zero beyond KS)
__device__ __forceinline__ void ro_load_afrags(const double* Wp, int KS, int tile, int ks0, int lane, double (&af)[RO_KB2]) {
#pragma unroll
  for (int i = 0; i < RO_KB2; ++i) {
    const int ks = ks0 + i;
    af[i] = ks < KS ? Wp[((size_t)tile * KS + ks) * 64 + lane] : 0.0;
  }
}

// KS_ < 0: RBF lift (no cooperation between the waves; the MLP code and its registers are not in that kernel, and the
// log / sqrt constants of the RBF not in the MLP kernels).
// KS_ > 0: k-steps of the encoder's hidden width fixed at compile time (25 = the reference's 100 hidden units): the
// fragment loads and the MFMAs become straight-line code (with a run-time count every one of them sat behind its own
// uniform branch and waited for its own LDS read); 0: run-time width.
// Register budget: the LDS decides how many trajectories (= waves) a CU holds; up to four waves per SIMD get 128
// VGPRs each (cfg2: 16 trajectories per CU), dimension sets with large per-trajectory regions leave room for more.
// RBF roll-outs with a long horizon and y = C x keep inv_K_G, [A B], bar_Q and H in ONE LDS region (step_body's ONE64, as the
// four-wave kernels do): their residency is set by the LDS alone
template <int L_, int N_, int Q_, int KS_> constexpr bool ro_one_region() { return N_ > 24 && Q_ != L_ && Q_ > 0; }
static bool ro_one_region_rt(int L, int N, int q, bool) { return N > 24 && q != L && q > 0; }
// dimension sets whose step keeps the state in registers (step_v2.h): their roll-out reads and writes the handle's wave image
template <int L_, int N_, int Q_> constexpr bool ro_v2() { return step_v2_dims(L_, N_, Q_); }
// RBF sets whose step does not fit 128 registers by far (the (32, 40) sets: 758 spilled with sixteen waves per CU): eight waves, 256 registers
static constexpr bool ro_rbf_eight(int L, int N, int q) { return !step_v2_dims(L, N, q) && L >= 24 && N > 32; }
template <int L_, int N_, int Q_, int NW, int KS_> constexpr int ro_max_threads() {
  if constexpr (ro_v2<L_, N_, Q_>()) {  // (LDS per trajectory <= 10 KB: sixteen trajectories per CU, 128 registers)
    constexpr size_t pw2 = v2_lds_elems(L_, Q_, N_), cap2 = 160 * 1024 / sizeof(double), wgs2 = cap2 / (pw2 * NW);
    constexpr size_t fit2 = wgs2 * NW > 16 ? 16 : (wgs2 * NW < (size_t)NW ? (size_t)NW : wgs2 * NW);
    return fit2 > 8 ? 1024 : (fit2 > 4 ? 512 : 256);
  }
  constexpr size_t pw = ro_one_region<L_, N_, Q_, KS_>()
                            ? (((size_t)step_region1(L_, N_) + (2 * L_ <= N_ * Q_ ? 0 : ((2 * L_ + 1) & ~1)) + vec_elems_one_region(2, L_, Q_, N_) + 1) & ~(size_t)1)
                            : ((step_lds_elems(2, L_, Q_, N_, step_tableau_in_lds<64, N_, L_>()) + 1) & ~(size_t)1);
  constexpr size_t cap = 160 * 1024 / sizeof(double);
  if constexpr (KS_ < 0) {
    // RBF lift: the waves of a workgroup never meet and rollout_waves launches as many as the LDS holds (NW is a placeholder here): the
    // register budget follows THAT number.  (Until late in round 6 the placeholder's sixteen set it: the (32, 40) sets, four waves per
    // CU, were compiled for 128 registers -- 758 of them spilled -- while three quarters of the register file stood empty.)
    constexpr size_t wf = cap / pw, wr0 = wf > 16 ? 16 : (wf < 4 ? 4 : wf), wr = (ro_rbf_eight(L_, N_, Q_) && wr0 > 8) ? 8 : wr0;
    return wr > 8 ? 1024 : (wr > 4 ? 512 : 256);
  }
  constexpr size_t wgs = cap / (pw * NW);
  constexpr size_t fit = wgs * NW > 16 ? 16 : (wgs * NW < (size_t)NW ? (size_t)NW : wgs * NW);
  // long horizons with the MLP lift: at most 8 trajectories per CU, 256 registers each (the N = 30 solver keeps H and
  // the tableau in registers; L = 8, N = 30: 24.6 M steps/s like this, 18.7 M with 16 trajectories at 128 registers and
  // 173 of them spilled.  With the RBF lift the same step measures the other way round -- 90 against 60 M steps/s)
  constexpr size_t waves = (KS_ >= 0 && N_ > 24 && fit > 8 && !ro_one_region<L_, N_, Q_, KS_>()) ? 8 : fit;
  return waves > 8 ? 1024 : (waves > 4 ? 512 : 256);  // 4 / 2 / 1 waves per SIMD
}
// Which dimension sets call the encoder as a function (ro_encoder_tiles) and which keep it inline.  Measured (old / new library on one box,
// bench.py --config cfg2 --L .. --N .., K = 20, kernel ms): (20, 30) 1.515 -> 1.225 (84 -> 19 spilled registers), (8, 30) 2.24 -> 2.19,
// (20, 20) 0.618 -> 0.611 (15 -> 2; 6.35 -> 6.26 at K = 200) -- but (8, 10) 0.495 -> 0.737: with a horizon of ten the step keeps its
// state in registers from step to step, and across a call those have to be saved.  Hence: inline for N < 16.
template <int L_, int N_, int Q_> constexpr bool ro_encoder_call() { return N_ >= 16; }
// The tile waves' part of the cooperative encoder -- layer 1, then every further layer as tile = wave over the full K range, the A-fragments in
// two alternating batches of 8 k-steps (the first batch of a layer is requested a layer ahead and travels across the barrier, every further
// batch while the previous one is being multiplied) -- as a FUNCTION OF ITS OWN (round 6): inlined, the encoder's 72 registers were part of
// the kernel's register allocation problem and the kernel spilled 15 vector registers (60 bytes of scratch per lane) around the box QP; as a
// call the kernel spills 2, the function needs no stack at all (it lives in the call-clobbered registers; the 6 - 15 values the kernel keeps
// across the lift sit in the preserved ones), and the window is 1.5 - 2.5 % shorter at K = 20, 3.5 - 4 % at K = 200
// (profiles/r6_cfg2_lift_ksplit.txt).  The arguments arrive in vector registers: the wave-uniform ones go back to scalar registers, the
// pointers to their address spaces (generic pointers are flat loads, which wait for both counters).
template <int KS_, int NW>
__device__ __attribute__((noinline)) void ro_encoder_tiles(const int wv_, const int lane, const int b0_, const int KSr_, const bool hid_, const bool out_, const int nhh_,
                                                        const double* W1_, const double* b1_, const double* Wh0_, const double* Wh1_, const double* Wo_,
                                                        const double* bh0_, const double* bh1_, const double* bo_, const double* sXn_, double* sAct0_,
                                                        double* sAct1_, double* sPsi_) {
  typedef double d4_t __attribute__((ext_vector_type(4)));
  typedef const double __attribute__((address_space(1))) * gp_t;
  typedef double __attribute__((address_space(3))) * lp_t;
  auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  auto unip = [&](const double* q) {
    const unsigned long long a = (unsigned long long)q;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    return (gp_t)(((unsigned long long)hi << 32) | lo);
  };
  const int wv = uni(wv_), KSr = uni(KSr_), nhh = uni(nhh_), b0 = uni(b0_);
  (void)b0;  // (the stamps of the trace build)
  const bool hid = uni(hid_ ? 1 : 0) != 0, out = uni(out_ ? 1 : 0) != 0;
  const gp_t W1 = unip(W1_), b1 = unip(b1_), Wh0 = unip(Wh0_), Wh1 = unip(Wh1_), Wo = unip(Wo_), bh0 = unip(bh0_), bh1 = unip(bh1_), bo = unip(bo_);
  const lp_t sXn = (lp_t)sXn_, sAct0 = (lp_t)sAct0_, sAct1 = (lp_t)sAct1_, sPsi = (lp_t)sPsi_;
  // B-fragment layout of an activation set: [k-step][4 k x 16 columns]; eight-trajectory workgroups store eight columns
  // ([k-step][4 k x 8]) and the lanes of columns 8-15 read columns 0-7 again (their outputs are never stored)
  constexpr int AR = NW == 8 ? 32 : 64;
  const int bl = NW == 8 ? ((lane >> 4) << 3) + (lane & 7) : lane;
  const int KS = KS_ > 0 ? KS_ : KSr;
  double af[2][RO_KB2];
  auto loadf = [&](gp_t Wp_, int ks0, double (&dst)[RO_KB2]) {
#pragma unroll
    for (int i = 0; i < RO_KB2; ++i) dst[i] = ks0 + i < KS ? Wp_[((size_t)wv * KS + ks0 + i) * 64 + lane] : 0.0;
  };
  if (nhh > 0) { if (hid) loadf(Wh0, 0, af[0]); }
  else if (out) loadf(Wo, 0, af[0]);
  double a1 = 0.0;
  d4_t c1 = {0.0, 0.0, 0.0, 0.0};
  if (hid) {
    a1 = W1[4 * (16 * wv + (lane & 15)) + (lane >> 4)];
#pragma unroll
    for (int r = 0; r < 4; ++r) c1[r] = b1[16 * wv + (lane >> 4) + 4 * r];
  }
  __syncthreads();  // every wave is done with its LDS region (previous step); x_{k} of all trajectories is in sXn
  LSTAMP(0);
  if (hid) {
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, sXn[4 * (lane & 15) + (lane >> 4)], c1, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * wv + (lane >> 4) + 4 * r;
      if (NW != 8 || (lane & 15) < 8) sAct0[(row >> 2) * AR + (row & 3) * (AR / 4) + (lane & 15)] = c1[r] > 0.0 ? c1[r] : 0.0;
    }
  }
  LSTAMP(1);
  __syncthreads();
  LSTAMP(2);
  // ---- hidden -> hidden layers, then the output layer, all as: tile = wave, full K
  for (int h = 0; h <= nhh; ++h) {
    const bool last = h == nhh;
    const bool mine = last ? out : hid;
    const lp_t act = (h & 1) ? sAct1 : sAct0;
    const lp_t actn = (h & 1) ? sAct0 : sAct1;
    const gp_t Wp = last ? Wo : ((h & 1) ? Wh1 : Wh0);
    const gp_t bias = last ? bo : ((h & 1) ? bh1 : bh0);
    d4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    if (mine) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc0[r] = bias[16 * wv + (lane >> 4) + 4 * r];
#pragma unroll
      for (int bt = 0; bt < 32 / RO_KB2; ++bt) {
        const int kb = bt * RO_KB2;
        if (kb + RO_KB2 < KS) loadf(Wp, kb + RO_KB2, af[(bt + 1) & 1]);
#pragma unroll
        for (int i = 0; i < RO_KB2; i += 2) {
          if (kb + i < KS) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[bt & 1][i], act[(kb + i) * AR + bl], acc0, 0, 0, 0);
          if (kb + i + 1 < KS) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[bt & 1][i + 1], act[(kb + i + 1) * AR + bl], acc1, 0, 0, 0);
        }
      }
      const int col = lane & 15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * wv + (lane >> 4) + 4 * r;
        const double v = acc0[r] + acc1[r];
        if (last) sPsi[row * 16 + col] = v;
        else if (NW != 8 || col < 8) actn[(row >> 2) * AR + (row & 3) * (AR / 4) + col] = v > 0.0 ? v : 0.0;
      }
    }
    // the next layer's first fragments travel across the barrier
    if (!last) {
      if (h + 1 < nhh) { if (hid) loadf(((h + 1) & 1) ? Wh1 : Wh0, 0, af[0]); }
      else if (out) loadf(Wo, 0, af[0]);
    }
    LSTAMP(3 + 2 * h);
    __syncthreads();  // (after the last layer: psi is outside the overlay, the waves go their own way)
    LSTAMP(4 + 2 * h);
  }
}
// IOT: element type of the caller-owned panels (X, ref, U0, Useq, U_log, X_log): double, or float for the float32-I/O roll-out of a
// KMPC_F32 handle (row g2; register-state dimension sets only) -- the state, the handle's own vectors and all arithmetic stay float64,
// and x_{k+1} is carried from step to step in LDS in float64: only what crosses the boundary is rounded (step_body.h io_ld / io_st)
// TERM: with the per-step terminal refresh (RolloutArgs::term_*; Koopman_update.m:215, 381): the step runs in two parts -- RLS, then
// condensed QP and solve -- with the reference's Riccati iteration on the freshly updated model between them (dare_device.h).  These
// instantiations are always plug-ins (kmpc_set_terminal_refresh loads them); the kernels without it are what they were.
template <int L_, int N_, int Q_, int NW, int KS_, typename IOT = double, bool TERM = false>
__global__ __launch_bounds__((ro_max_threads<L_, N_, Q_, NW, KS_>())) void rollout_kernel(const RolloutArgs<double> ra) {
  static_assert(sizeof(IOT) == 8 || ro_v2<L_, N_, Q_>(), "float32 I/O: register-state dimension sets only");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* const smem = reinterpret_cast<double*>(smem_raw);
  constexpr int NC = NW == 4 ? 4 : 16;  // trajectory columns of the cooperative encoder
  constexpr bool RBF = KS_ < 0;         // the lift kind is a compile-time property (KS_ = -1: thin-plate RBF, per wave)
  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);  // wave-uniform: trajectory index and LDS base stay scalar
  const int B = ra.s.B, n = ra.s.n, L = L_;
  const int b0 = blockIdx.x * (int)(blockDim.x >> 6);  // NW waves (MLP lift); RBF: as many as fit in LDS
  // lift scratch (overlays the per-wave regions between two steps)
  double* const sAct0 = smem;
  double* const sAct1 = sAct0 + (NW == 8 ? ro_act8() : RO_ACT);
  double* const sPsi = smem + ra.keep_off;  // behind the per-wave regions: survives into the step
  double* const sXn = sPsi + ra.Lp * NC;
  int slot_traj = wave;  // which of the workgroup's trajectories this wave takes
  if constexpr (!RBF && NW == 16) {
    if (ra.perm) {  // (RolloutArgs::perm: the whole batch dealt out by work)
      slot_traj = b0 + wave < B ? ra.perm[b0 + wave] - b0 : wave;
    } else if (ra.work) {  // (RolloutArgs::work: by last launch's solver work, heaviest to the oldest waves)
      int* const sHint = reinterpret_cast<int*>(sPsi);
      int* const sRank = sHint + 16;
      if (tid0 < 16) sHint[tid0] = b0 + (int)tid0 < B ? ra.work[b0 + tid0] : -1;
      __syncthreads();
      if (tid0 < 16) {
        const int ht = sHint[tid0];
        int r = 0;
        for (int v = 0; v < 16; ++v) { const int hv = sHint[v]; r += (hv > ht || (hv == ht && v < (int)tid0)) ? 1 : 0; }
        sRank[r] = tid0;
      }
      __syncthreads();
      slot_traj = __builtin_amdgcn_readfirstlane(sRank[wave]);
      __syncthreads();
    }
  }
  const int b = b0 + slot_traj;
  const bool live = b < B;
  if (!RBF && tid0 < 4 * NC) sXn[tid0] = 0.0;  // (columns of trajectories this workgroup does not have)
  __syncthreads();
  if (!RBF && (tid0 & 63) < 4)
    sXn[wave * 4 + (tid0 & 63)] = (live && (int)(tid0 & 63) < n) ? io_ld<IOT>(ra.s.X_rw, (size_t)(tid0 & 63) * B + b) : 0.0;

  if constexpr (ro_v2<L_, N_, Q_>()) {
    if (live) step_v2_init<L_, N_, Q_>(smem + ra.wbase + wave * ra.wstride);
  }
  bool have_prev = ra.have_prev != 0, fresh = ra.rls_fresh != 0;
  int cur = ra.cur;
  // lane i < L: psi_i(x_{k-1}) of this wave's trajectory, carried from step to step (a launch that continues an earlier
  // one starts from the handle's copy)
  // (register-state step: lanes i and 32 + i both carry psi_i -- the two halves of the wave run the v and the w chain)
  constexpr bool V2 = ro_v2<L_, N_, Q_>();
  constexpr int PSI_MASK = V2 ? 31 : 63;
  double psi_prev_reg = 0.0;
  if (live && ra.have_prev && (int)(tid0 & PSI_MASK) < L) psi_prev_reg = ra.psi[ra.cur ^ 1][(size_t)b * L + (tid0 & PSI_MASK)];
  typedef const RolloutArgs<double> __attribute__((address_space(4))) * kernarg_ptr_t;
  for (int k = 0; k < ra.steps; ++k) {
    // The step arguments stay in the kernel-argument segment and are re-read where they are used: hoisted out
    // of this loop they would pin ~150 scalar registers for the whole kernel (the asm hides the loop invariance).
    kernarg_ptr_t kp = (kernarg_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    const RolloutArgs<double> __attribute__((address_space(4)))& R = *kp;
    const StepArgs<double>& a = *(const StepArgs<double>*)(&kp->s);  // psi strides (1, L), accumulate = 1: host
    // (as local_tid: lane- and wave-derived addresses and the lift's tiling constants are recomputed in every
    //  iteration instead of being carried across the step in registers)
    int wv = wave;
    asm volatile("" : "+s"(wv));
    const int lane = local_tid<64>();
    (void)wv;
#ifdef KMPC_TRACE
    if (lane == 0 && b < 8192) kmpc_trace_buf[b * 32 + 16] = wall_clock64();  // this wave is ready for step k
#endif
    if constexpr (!RBF) {  // (RolloutArgs::work_tail: the solver work that ranks the next launch is that of the last steps)
      if (R.work_tail > 0 && k == R.steps - R.work_tail && lane == 0) ((lds_i32*)reinterpret_cast<int*>(sXn + wv * 4 + 2))[2] = 0;
    }
    double psi_i = 0.0;  // lane i < L: psi_i(x_k) of this wave's trajectory
    if constexpr (RBF) {
      if (live && (lane & PSI_MASK) < L) {  // (n = 2: the plants of the roll-outs are two-state systems)
        const double x0 = io_ld<IOT>(a.X_rw, (size_t)b), x1 = io_ld<IOT>(a.X_rw, (size_t)B + b);
        const double* c = R.cx + (size_t)(lane & PSI_MASK) * n;
        psi_i = kmpc_rbf_psi2(x0, x1, c[0], c[1], R.eps, R.rbf_matlab);  // (plant_device.h: the stand-alone lift's code)
      }
    } else if constexpr (NW == 4) {  // (written for 4 or 8 columns; with 8 the 16x16x4 path below measures better: 89 vs 87 M steps/s)
      // Four or eight trajectories per workgroup (four / two workgroups per CU, which drift apart: a SIMD then holds
      // waves in different phases of the step, and a barrier only makes a few trajectories wait for each other).
      // Encoder on v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 blocks per instruction = 16 output rows x 4
      // trajectories x 4 k.  Operand lanes (probed, tools/ubench/mfma_f64_4x4.hip): A lane 16k + 4blk + i, B lane
      // 16k + 4blk + j, D lane 16i + 4blk + j -- so the A-fragments are the SAME packed tiles the 16x16x4 path reads
      // (lane = 16k + row in tile), B is the activation (k, j) replicated over the blocks (broadcast LDS read), and a
      // tile's output comes back as row 4blk + i, column j.  On gfx950 this shape runs at the flop rate of the
      // 16x16x4 one, so the lift costs the same f64 pipe time per trajectory as with 16 columns (the 16x16x4 shape with
      // 8 columns wastes half of it).  NW = 4: wave w owns hidden tiles w and w + 4; NW = 8: tile w, multiplied with
      // both groups of four columns (the fragment is loaded once).  Either way four independent accumulator chains
      // per wave; the output tiles go to the waves from the top (the last wave has less hidden work when Hp = 112).
      constexpr int CG = NW / 4;      // groups of four trajectory columns
      constexpr int NT = 8 / NW;      // hidden tiles per wave (Hp <= 128: eight tiles)
      const int KS = KS_ > 0 ? KS_ : R.KS, Hp = KS_ > 0 ? (KS_ <= 28 ? 112 : 128) : R.Hp, MTH = Hp >> 4, MTO = R.Lp >> 4;
      const int boff = (lane >> 4) * NC + (lane & 3);         // B operand: element (k, j) of a k-step (+ 4 cg)
      const int drow = ((lane >> 2) & 3) * 4 + (lane >> 4);   // D: row within the tile
      const int dcol = lane & 3;
      const int th0 = wv, th1 = wv + NW;                      // hidden tiles of this wave
      const bool vh0 = th0 < MTH, vh1 = NT > 1 && th1 < MTH;
      const int to0 = NW - 1 - wv;                            // output tile of this wave
      const bool vo0 = to0 < MTO;
      double af[2][RO_KB2];  // double-buffered batches of A-fragments (one tile at a time: registers are scarce here)
      // first layer: one k-step (K = n <= 4, W1 zero-padded to 4 columns), bias as the accumulator input
      double a1[NT], c1[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int th = wv + t * NW;
        const bool v = th < MTH;
        a1[t] = v ? R.W1[4 * (16 * th + (lane & 15)) + (lane >> 4)] : 0.0;
        c1[t] = v ? R.b1[16 * th + drow] : 0.0;
      }
      if (R.nhh > 0) { if (vh0) ro_load_afrags(R.Whp[0], KS, th0, 0, lane, af[0]); }
      else if (vo0) ro_load_afrags(R.Wop, KS, to0, 0, lane, af[0]);
      __syncthreads();  // every wave is done with its LDS region (previous step); x_k of the trajectories is in sXn
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int th = wv + t * NW;
        if (th < MTH) {
#pragma unroll
          for (int cg = 0; cg < CG; ++cg) {
            const double xb = sXn[(cg * 4 + (lane & 3)) * 4 + (lane >> 4)];
            const double v = __builtin_amdgcn_mfma_f64_4x4x4f64(a1[t], xb, c1[t], 0, 0, 0);
            sAct0[(16 * th + drow) * NC + cg * 4 + dcol] = v > 0.0 ? v : 0.0;
          }
        }
      }
      __syncthreads();
      for (int h = 0; h <= R.nhh; ++h) {
        const bool last = h == R.nhh;
        const int t0 = last ? to0 : th0, t1 = th1;
        const bool v0 = last ? vo0 : vh0, v1 = last ? false : vh1;
        const double* act = (h & 1) ? sAct1 : sAct0;
        double* actn = (h & 1) ? sAct0 : sAct1;
        const double* Wp = last ? R.Wop : R.Whp[h & 1];
        const double* bias = last ? R.bo : R.bh[h & 1];
        double acc[NT][CG][2];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const double bv = (t ? v1 : v0) ? bias[16 * (t ? t1 : t0) + drow] : 0.0;
#pragma unroll
          for (int cg = 0; cg < CG; ++cg) { acc[t][cg][0] = bv; acc[t][cg][1] = 0.0; }
        }
        constexpr int NB = 32 / RO_KB2;  // batches per tile (KS <= 32 k-steps)
#pragma unroll
        for (int g = 0; g < NT * NB; ++g) {  // batch g: tile g / NB, k-steps (g % NB) * 8 ..; batch g + 1 is requested first
          const int tl = g / NB, kb = (g % NB) * RO_KB2;
          if (g + 1 < NT * NB) {
            const int tl2 = (g + 1) / NB, kb2 = ((g + 1) % NB) * RO_KB2;
            if ((tl2 ? v1 : v0) && kb2 < KS) ro_load_afrags(Wp, KS, tl2 ? t1 : t0, kb2, lane, af[(g + 1) & 1]);
          }
          if ((tl ? v1 : v0) && kb < KS) {
#pragma unroll
            for (int i = 0; i < RO_KB2; ++i)
              if (kb + i < KS) {
#pragma unroll
                for (int cg = 0; cg < CG; ++cg)
                  acc[tl][cg][i & 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[g & 1][i], act[(kb + i) * 4 * NC + boff + 4 * cg],
                                                                         acc[tl][cg][i & 1], 0, 0, 0);
              }
          }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (t ? v1 : v0) {
            const int tt = t ? t1 : t0;
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
              const double v = acc[t][cg][0] + acc[t][cg][1];
              if (last) sPsi[(16 * tt + drow) * NC + cg * 4 + dcol] = v;
              else actn[(16 * tt + drow) * NC + cg * 4 + dcol] = v > 0.0 ? v : 0.0;
            }
          }
        }
        // the next layer's first fragments travel across the barrier
        if (!last) {
          if (h + 1 < R.nhh) { if (vh0) ro_load_afrags(R.Whp[h + 1], KS, th0, 0, lane, af[0]); }
          else if (vo0) ro_load_afrags(R.Wop, KS, to0, 0, lane, af[0]);
        }
        __syncthreads();  // (after the last layer: psi is outside the overlay, the waves go their own way)
      }
      if ((lane & PSI_MASK) < L) psi_i = sPsi[(lane & PSI_MASK) * NC + wv];
    } else {
      // Cooperative encoder.  Wave w < Hp/16 owns hidden M tile w for the whole K range (two alternating
      // accumulator chains), so bias + ReLU are applied on the accumulator registers and the result is written
      // straight into the next layer's B-fragment layout: one barrier per layer, no partial sums.  The other
      // waves only take part in the barriers.
      const int KS = KS_ > 0 ? KS_ : R.KS, Hp = KS_ > 0 ? (KS_ <= 28 ? 112 : 128) : R.Hp, MTH = Hp >> 4, MTO = R.Lp >> 4;
      const bool hid = wv < MTH, out = wv < MTO;
      if (V2 && !hid && !out) {
        // A wave without a tile of the encoder only takes part in its barriers (2 + layers of them).  Register-state step: it
        // does the covariance half of its trajectory's RLS update meanwhile -- inv_K_G, bar_Q only need [psi(x_{k-1}); u_{k-1}]
        // -- and meets the others at the second and third barrier from inside v2_rls_cov.  (The waves WITH a tile did theirs at
        // the end of the previous step: they are the oldest of their SIMDs and finish the step body first.)
        __syncthreads();
        if constexpr (V2) {
          if (live && k > 0 && !R.no_update) {
            int woff0 = R.wbase + wv * R.wstride, bk0 = b;
            asm volatile("" : "+s"(woff0), "+s"(bk0));
            double* const wsm0 = smem + woff0;
            const double uk = a.u_prev[bk0];  // u_{k-1}: the previous step stored it (u_store) when its solve ended
            const int t0 = lane & 31;
            const double z0 = t0 < L_ ? psi_prev_reg : (t0 == L_ ? uk : 0.0);
            const double gains0 = v2_rls_cov<L_, true>(R.img + (size_t)bk0 * R.img_stride, z0, a.lam);
            if (lane < 32 || t0 < L_) v2_cov_slot<N_>(wsm0)[v2_cov_index<L_>(lane)] = gains0;
          } else {
            __syncthreads();
            __syncthreads();
          }
        }
        for (int i = 0; i < R.nhh; ++i) __syncthreads();
      } else if constexpr (ro_encoder_call<L_, N_, Q_>()) {
        // the tile waves: the products of all layers (ro_encoder_tiles, a function of its own)
        ro_encoder_tiles<KS_, NW>(wv, lane, b0, KS, hid, out, R.nhh, R.W1, R.b1, R.Whp[0], R.Whp[1], R.Wop, R.bh[0], R.bh[1], R.bo, sXn, sAct0, sAct1, sPsi);
      } else {
      // (the same products inline -- the short-horizon sets, see ro_encoder_call)
      constexpr int AR = NW == 8 ? 32 : 64;
      const int bl = NW == 8 ? ((lane >> 4) << 3) + (lane & 7) : lane;
      // A-fragments in two alternating batches of 8 k-steps: the first batch of a layer is requested a layer ahead
      // (it travels across the barrier), every further batch while the previous one is being multiplied
      double af[2][RO_KB2];
      if (R.nhh > 0) { if (hid) ro_load_afrags(R.Whp[0], KS, wv, 0, lane, af[0]); }
      else if (out) ro_load_afrags(R.Wop, KS, wv, 0, lane, af[0]);
      // ---- layer 1 (K = n <= 4: one k-step, W1 zero-padded to 4 columns): one MFMA per hidden tile, bias as the
      //      accumulator input, straight into B-fragment layout.  Operands are requested before the barrier.
      double a1 = 0.0;
      d4_t c1 = {0.0, 0.0, 0.0, 0.0};
      if (hid) {
        a1 = R.W1[4 * (16 * wv + (lane & 15)) + (lane >> 4)];
#pragma unroll
        for (int r = 0; r < 4; ++r) c1[r] = R.b1[16 * wv + (lane >> 4) + 4 * r];
      }
      __syncthreads();  // every wave is done with its LDS region (previous step); x_{k} of all trajectories is in sXn
      LSTAMP(0);
      if (hid) {
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, sXn[4 * (lane & 15) + (lane >> 4)], c1, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * wv + (lane >> 4) + 4 * r;
          if (NW != 8 || (lane & 15) < 8) sAct0[(row >> 2) * AR + (row & 3) * (AR / 4) + (lane & 15)] = c1[r] > 0.0 ? c1[r] : 0.0;
        }
      }
      LSTAMP(1);
      __syncthreads();
      LSTAMP(2);
      // ---- hidden -> hidden layers, then the output layer, all as: tile = wave, full K
      for (int h = 0; h <= R.nhh; ++h) {
        const bool last = h == R.nhh;
        const bool mine = last ? out : hid;
        const double* act = (h & 1) ? sAct1 : sAct0;
        double* actn = (h & 1) ? sAct0 : sAct1;
        const double* Wp = last ? R.Wop : R.Whp[h & 1];
        const double* bias = last ? R.bo : R.bh[h & 1];
        // accumulator register r of lane l holds row (l >> 4) + 4 r, column l & 15 of the tile
        d4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        if (mine) {  // the bias is the accumulator input of the first chain
#pragma unroll
          for (int r = 0; r < 4; ++r) acc0[r] = bias[16 * wv + (lane >> 4) + 4 * r];
        }
        if (mine) {
#pragma unroll
          for (int bt = 0; bt < 32 / RO_KB2; ++bt) {  // KS <= 32 k-steps in batches
            const int kb = bt * RO_KB2;
            if (kb + RO_KB2 < KS) ro_load_afrags(Wp, KS, wv, kb + RO_KB2, lane, af[(bt + 1) & 1]);
#pragma unroll
            for (int i = 0; i < RO_KB2; i += 2) {
              if (kb + i < KS) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[bt & 1][i], act[(kb + i) * AR + bl], acc0, 0, 0, 0);
              if (kb + i + 1 < KS)
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[bt & 1][i + 1], act[(kb + i + 1) * AR + bl], acc1, 0, 0, 0);
            }
          }
        }
        if (mine) {
          const int col = lane & 15;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * wv + (lane >> 4) + 4 * r;
            const double v = acc0[r] + acc1[r];
            if (last) sPsi[row * 16 + col] = v;
            else if (NW != 8 || col < 8) actn[(row >> 2) * AR + (row & 3) * (AR / 4) + col] = v > 0.0 ? v : 0.0;
          }
        }
        // the next layer's first fragments travel across the barrier
        if (!last) {
          if (h + 1 < R.nhh) { if (hid) ro_load_afrags(R.Whp[h + 1], KS, wv, 0, lane, af[0]); }
          else if (out) ro_load_afrags(R.Wop, KS, wv, 0, lane, af[0]);
        }
        LSTAMP(3 + 2 * h);
        __syncthreads();  // (after the last layer: psi is outside the overlay, the waves go their own way)
        LSTAMP(4 + 2 * h);
      }
      }
      if ((lane & PSI_MASK) < L) psi_i = sPsi[(lane & PSI_MASK) * 16 + wv];
    }

#ifdef KMPC_TRACE
    if (lane == 0 && b < 8192) kmpc_trace_buf[b * 32 + 17] = wall_clock64();  // lift done
    if (lane == 0 && b < 8192 && k == 0) {
      // where this wave runs: XCC_ID (hwreg 20, bits 3:0) | HW_ID (hwreg 4: wave, simd, pipe, cu, sh, se) << 8
      kmpc_trace_buf[b * 32 + 29] = (unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 8);
      kmpc_trace_buf[b * 32 + 19] = wall_clock64();
      for (int sl = 22; sl < 29; ++sl) kmpc_trace_buf[b * 32 + sl] = 0ull;  // (qp_rl's work counters: summed over the launch)
    }
#endif
    if (live) {
      int woff = R.wbase + wv * R.wstride, bk = b;
      asm volatile("" : "+s"(woff), "+s"(bk));  // (as local_tid: keeps the step's address arithmetic inside the loop)
      double* const wsm = smem + woff;
      double* const psi_now = R.psi[cur];
      if (lane < L) psi_now[(size_t)b * L + lane] = psi_i;  // (state of the handle; the step takes psi from the registers)
      StepVar<double> sv;
      sv.psi_now = psi_now;
      sv.psi_prev = R.psi[cur ^ 1];
      sv.psi_in_regs = 1;
      sv.psi_now_v = psi_i;
      sv.psi_prev_v = psi_prev_reg;
      sv.phases = PH_CONDENSE | PH_QP | ((have_prev && !R.no_update) ? PH_RLS : 0);
      sv.first_update = fresh ? 1 : 0;
      sv.plant_switched = (R.switch_step >= 0 && R.step0 + k >= R.switch_step) ? 1 : 0;
      sv.U0 = R.U_log ? io_at<IOT>(R.U_log, (size_t)k * B) : a.U0;
      sv.x_next = RBF ? nullptr : sXn + wv * 4;
      sv.cov_done = (k > 0 && !R.no_update) ? 1 : 0;
      // (waves without an encoder tile do the next step's covariance half inside the next lift instead: see above)
      const bool tile_wave = RBF || NW == 4 || wv < ((KS_ > 0 ? (KS_ <= 28 ? 112 : 128) : R.Hp) >> 4) || wv < (R.Lp >> 4);
      sv.cov_ahead = (k + 1 < R.steps && !R.no_update && tile_wave) ? 1 : 0;
      // (16 trajectories per CU with a long horizon -- the RBF roll-out of cfg3 -- : H re-read from LDS, and the active-set
      //  safeguard stays the fall-back on the global-scratch tableau instead of living in registers: 67 -> 51 spilled
      //  registers, 109.7 -> 117.8 M steps/s; its crawling solves are 14 of 327 680 on that workload)
      constexpr bool LOWREG = ro_max_threads<L_, N_, Q_, NW, KS_>() == 1024 && N_ > 24;
      // (y = psi, Q_ == L_: ill-conditioned H, crawling solves are common -- those instantiations keep the register safeguard)
      if constexpr (TERM) {
        // part 1: the RLS update (the model it leaves is in global memory: wave image / dense blocks)
        StepVar<double> sv1 = sv, sv2 = sv;
        sv1.phases = sv.phases & PH_RLS;
        sv1.cov_ahead = 0;
        sv2.phases = sv.phases & ~PH_RLS;
        double* imgb = V2 ? R.img + (size_t)bk * R.img_stride : nullptr;
        if (sv1.phases) {
          if constexpr (V2) step_v2<L_, N_, Q_, LOWREG, !LOWREG, IOT, 1>(a, sv1, bk, wsm, imgb);
          else step_body<double, 64, L_, N_, Q_, LOWREG, !LOWREG || Q_ == L_, ro_one_region<L_, N_, Q_, KS_>()>(a, sv1, bk, wsm);
        }
        __threadfence_block();
        block_sync<64>();
        // the Riccati iteration on this trajectory's current [A B] and its block Co P Co' - Qw I, every term_every-th step
        if (R.term_every > 0 && ((R.term_count0 + k) % R.term_every) == 0) {
          double* const scr = R.term_scratch + (size_t)bk * R.term_scratch_stride;
          double* const Wb = R.term_W + (size_t)bk * Q_ * Q_;
          int itd;
          if constexpr (V2) {
            const V2Dims vd(L_, 2);
            const int cy0 = a.cy0;
            itd = wave_dare(L_, Q_, [&](int i, int j) { return imgb[vd.off1(i, j)]; }, [&](int r, int j) { return imgb[vd.off1(L_ + cy0 + r, j)]; },
                            R.term_Q, R.term_R, R.term_eps, R.term_maxiter, a.Qw, scr, Wb);
          } else {
            const double* const Kb = a.K + (size_t)bk * a.strideK;
            const double* const Cb = a.C + (size_t)bk * a.strideC;
            const int cy0 = a.cy0;
            itd = wave_dare(L_, Q_, [&](int i, int j) { return Kb[i * (L_ + 1) + j]; },
                            [&](int r, int j) { return Q_ == L_ ? (r == j ? 1.0 : 0.0) : Cb[(cy0 + r) * L_ + j]; },
                            R.term_Q, R.term_R, R.term_eps, R.term_maxiter, a.Qw, scr, Wb);
          }
          if (R.term_iters && lane == 0) R.term_iters[bk] = itd;
        }
        // part 2: condensed QP (its terminal block is term_W[b]: StepArgs::Wterm, per trajectory) and the solve
        if constexpr (V2) step_v2<L_, N_, Q_, LOWREG, !LOWREG, IOT, 2>(a, sv2, bk, wsm, imgb);
        else step_body<double, 64, L_, N_, Q_, LOWREG, !LOWREG || Q_ == L_, ro_one_region<L_, N_, Q_, KS_>()>(a, sv2, bk, wsm);
      } else if constexpr (V2) {
        double* imgb = R.img + (size_t)bk * R.img_stride;
        step_v2<L_, N_, Q_, LOWREG, !LOWREG, IOT>(a, sv, bk, wsm, imgb);
      } else {
        step_body<double, 64, L_, N_, Q_, LOWREG, !LOWREG || Q_ == L_, ro_one_region<L_, N_, Q_, KS_>()>(a, sv, bk, wsm);
      }
      if (R.X_log) {
        __threadfence_block();
        if (lane < n) io_st<IOT>(R.X_log, ((size_t)k * n + lane) * B + b, io_ld<IOT>(a.X_rw, (size_t)lane * B + b));
      }
    }
#ifdef KMPC_TRACE
    if (lane == 0 && b < 8192) {
      const unsigned long long t18 = wall_clock64();
      kmpc_trace_buf[b * 32 + 18] = t18;  // step k done
      // whole-launch sums: barrier wait + lift, step body (k == 0 resets)
      const unsigned long long dl = kmpc_trace_buf[b * 32 + 17] - kmpc_trace_buf[b * 32 + 16], db = t18 - kmpc_trace_buf[b * 32 + 17];
      if (b < 4096 && k < 32) {
        kmpc_step_buf[b * 96 + 3 * k] = (unsigned)kmpc_trace_buf[b * 32 + 16];
        kmpc_step_buf[b * 96 + 3 * k + 1] = (unsigned)kmpc_trace_buf[b * 32 + 17];
        kmpc_step_buf[b * 96 + 3 * k + 2] = (unsigned)t18;
      }
      kmpc_trace_buf[b * 32 + 20] = (k == 0 ? 0ull : kmpc_trace_buf[b * 32 + 20]) + dl;
      kmpc_trace_buf[b * 32 + 21] = (k == 0 ? 0ull : kmpc_trace_buf[b * 32 + 21]) + db;
    }
#endif
    if (have_prev) fresh = false;
    have_prev = true;
    cur ^= 1;
    psi_prev_reg = psi_i;
  }
  if (!RBF && live && (tid0 & 63) == 0) {  // (the host zeroed status / iters before the launch)
    const int* const acc = reinterpret_cast<const int*>(sXn + wave * 4 + 2);
    if (ra.s.status) ra.s.status[b] = acc[0];
    if (ra.s.iters) ra.s.iters[b] = acc[1];
    if (ra.work) ra.work[b] = acc[2];
  }
}


// waves (= trajectories) per workgroup of the fused roll-out.  MLP lift: 16 (one workgroup per CU) or 8 (two per
// CU); the RBF lift needs no cooperation, so the workgroup is as large as the per-trajectory LDS regions allow.
// 0: does not fit.
static int ro_one_region_r2(int n, int L, int q, int N) { return n * L <= N * q ? 0 : ((n * L + 1) & ~1); }  // (C where g goes, else its own region)
// (step_lds_bytes of step_kernel.hip, restated on the constexpr layout functions so that a plug-in needs no symbol of the library)
static size_t ro_step_lds_elems(int n, int L, int q, int N, int* r1, int* r2, bool lds_tableau) {
  if (r1) *r1 = step_region1(L, N);
  if (r2) *r2 = step_region2(n, L, N, lds_tableau);
  return step_lds_elems(n, L, q, N, lds_tableau);
}
static size_t rollout_lds_elems(int n, int L, int q, int N, bool rbf, int waves, int Lp, int* wstride) {
  size_t per_wave = (ro_step_lds_elems(n, L, q, N, nullptr, nullptr, !tableau_saves_lds(N, L)) + 1) & ~(size_t)1;
  if (ro_one_region_rt(L, N, q, rbf))
    per_wave = ((size_t)step_region1(L, N) + ro_one_region_r2(n, L, q, N) + vec_elems_one_region(n, L, q, N) + 1) & ~(size_t)1;
  const bool v2 = step_v2_dims(L, N, q);
  if (v2) per_wave = v2_lds_elems(L, q, N);  // register-state step: H / carried tableau and vectors only
  if (wstride) *wstride = (int)per_wave;
  size_t elems = per_wave * waves;
  if (rbf) return elems;
  const size_t scratch = ro_scratch_of(waves);  // (sAct1 sits one activation set behind sAct0)
  if (v2) return scratch + elems + ro_keep(Lp, waves);  // (the regions carry the tableau from step to step: no overlay)
  if (elems < scratch) elems = scratch;
  return elems + ro_keep(Lp, waves);
}
#if defined(KMPC_ROLLOUT_JIT_TU)
static int g_rollout_workgroup = 0;  // (unused: the library hands the workgroup size over)
#elif defined(KMPC_ROLLOUT_IO32_TU)
extern int g_rollout_workgroup;
#else
int g_rollout_workgroup = 0;  // kmpc_set_rollout_workgroup
void set_rollout_workgroup(int trajectories) { g_rollout_workgroup = trajectories; }
#endif
static int rollout_waves(int n, int L, int q, int N, bool rbf, int Lp, int B = 1 << 30) {
  const size_t cap = 160 * 1024 / sizeof(double);
  if (!rbf) {
    static const char* env = dbg_env("KMPC_ROLLOUT_WAVES");  // measurement aid: force 4, 8 or 16
    // workgroups of w trajectories that fit on one CU (LDS is handed out in 512-byte granules; 16 waves per CU)
    auto wgs = [&](int w) -> int {
      const size_t e = (rollout_lds_elems(n, L, q, N, false, w, Lp, nullptr) + 63) & ~(size_t)63;
      const int k = (int)(cap / e);
      const int maxw = (N > 24 && !ro_one_region_rt(L, N, q, false) && !step_v2_dims(L, N, q)) ? 8 : 16;  // (long horizons: see ro_max_threads)
      return k * w > maxw ? maxw / w : k;
    };
    if (g_rollout_workgroup) return wgs(g_rollout_workgroup) > 0 ? g_rollout_workgroup : 0;
    if (env && (atoi(env) == 4 || atoi(env) == 8 || atoi(env) == 16)) return wgs(atoi(env)) > 0 ? atoi(env) : 0;
    static int cus_dev[16] = {};
    int& cus = cus_dev[device_slot()];
    if (!cus) {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    // Most trajectories per CU wins; ties go to the larger workgroup (cfg2 at B = 4096: 91.3 / 89.0 / 84.5 M steps/s
    // with 16 / 8 / 4 trajectories per workgroup).  Batches that leave CUs without a 16-trajectory workgroup are
    // spread as smaller ones (B = 2048: 43.5 vs 35.7 M steps/s with 8, B = 1024: 22.6 vs 18.4).
    // (four-trajectory workgroups only where they at least double the residency: their lifts expose the L2 round
    //  trips of the weight fragments, L = 8, N = 30: 17.8 M steps/s with 3 x 4 against 24.6 M with 1 x 8)
    int best = 0, best_traj = 0;
    for (int w = 16; w >= 8; w >>= 1) {
      const int t = wgs(w) * w;
      if (t > best_traj) { best = w; best_traj = t; }
    }
    if (wgs(4) * 4 >= 2 * best_traj && wgs(4) > 0) { best = 4; best_traj = wgs(4) * 4; }
    if (best == 16 && wgs(8) * 8 >= 16 && (B + 15) / 16 < cus) return 8;
    return best;
  }
  for (int w = ro_rbf_eight(L, N, q) ? 8 : 16; w >= 4; --w)  // (RBF lift: the waves of a workgroup never meet, any number of them will do)
    if (rollout_lds_elems(n, L, q, N, true, w, Lp, nullptr) <= cap) return w;
  return 0;
}
template <int L_, int N_, int Q_, int NW, int KS_, typename IOT = double, bool TERM = false>
static hipError_t launch_rollout_nw(const RolloutArgs<double>& k, int waves, size_t lds, hipStream_t s) {
  static size_t configured_dev[16] = {};  // (function attributes are per device)
  size_t& configured = configured_dev[device_slot()];
  if (lds > 64 * 1024 && lds > configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rollout_kernel<L_, N_, Q_, NW, KS_, IOT, TERM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    configured = lds;
  }
  const int grid = (k.s.B + waves - 1) / waves;
  hipLaunchKernelGGL((rollout_kernel<L_, N_, Q_, NW, KS_, IOT, TERM>), dim3(grid), dim3(64 * waves), lds, s, k);
  return hipGetLastError();
}

// waves_in > 0: the workgroup size is the caller's decision (plug-ins: the library made it with rollout_waves)
template <int L_, int N_, int Q_, typename IOT = double> static hipError_t launch_rollout_impl(const RolloutArgs<double>& a, hipStream_t s, int waves_in = 0) {
  RolloutArgs<double> k = a;
  const bool rbf = a.lift_rbf != 0;
  constexpr bool V2 = ro_v2<L_, N_, Q_>();
  int waves = waves_in > 0 ? waves_in : rollout_waves(a.s.n, a.s.L, a.s.q, a.s.N, rbf, a.Lp, a.s.B);
  // (float32 panels: workgroups of sixteen and eight trajectories are instantiated; a request for four -- kmpc_set_rollout_workgroup(4),
  //  KMPC_ROLLOUT_WAVES=4 -- is served by eight: the register-state sets always fit eight per CU.  ADVICE r5: the handle reported a fused
  //  roll-out and the launch then failed)
  if (sizeof(IOT) == 4 && !rbf && waves == 4) waves = 8;
  if (waves == 0) return hipErrorInvalidValue;
  ro_step_lds_elems(a.s.n, a.s.L, a.s.q, a.s.N, &k.s.r1, &k.s.r2, !tableau_saves_lds(a.s.N, a.s.L));
  if (!a.s.qp_scratch) return hipErrorInvalidValue;
  if (ro_one_region_rt(a.s.L, a.s.N, a.s.q, rbf)) k.s.r2 = ro_one_region_r2(a.s.n, a.s.L, a.s.q, a.s.N);
  if constexpr (V2) {
    if (!a.img || a.s.n != 2) return hipErrorInvalidValue;
    k.s.r1 = v2_region1(N_);
    k.s.r2 = 0;
  }
  const size_t elems = rollout_lds_elems(a.s.n, a.s.L, a.s.q, a.s.N, rbf, waves, a.Lp, &k.wstride);
  k.keep_off = rbf ? 0 : (int)(elems - ro_keep(a.Lp, waves));
  k.wbase = (V2 && !rbf) ? (int)ro_scratch_of(waves) : 0;
  const size_t lds = elems * sizeof(double);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  // (the RBF lift never uses the template's tiling: one instantiation serves every workgroup size)
  const bool ks25 = !rbf && a.KS == 25 && a.Hp == 112;  // the reference's encoders: 100 hidden units
#ifdef KMPC_ROLLOUT_JIT_TU
  // a plug-in holds ONE kernel: KMPC_JIT_KS = -1 (RBF lift, any workgroup size), 25 or 0 (MLP lift) with KMPC_JIT_NW trajectories per workgroup
  constexpr int JKS = KMPC_JIT_KS, JNW = JKS < 0 ? 16 : KMPC_JIT_NW;
  if (rbf != (JKS < 0)) return hipErrorInvalidValue;
  if (!rbf && (waves != JNW || ks25 != (JKS == 25))) return hipErrorInvalidValue;
  if ((a.term_every > 0) != (KMPC_JIT_TERM != 0)) return hipErrorInvalidValue;
  if (a.term_every > 0 && (!a.term_Q || !a.term_W || !a.term_scratch || a.term_scratch_stride < term_scratch_elems(L_) || !a.s.Wterm || !a.s.wterm_per_traj))
    return hipErrorInvalidValue;
  return launch_rollout_nw<L_, N_, Q_, JNW, JKS, IOT, KMPC_JIT_TERM != 0>(k, waves, lds, s);
#else
  if (rbf) return launch_rollout_nw<L_, N_, Q_, 16, -1, IOT>(k, waves, lds, s);
  if constexpr (sizeof(IOT) == 4) {  // (float32 I/O: workgroups of sixteen and eight trajectories -- what rollout_waves picks for these sets)
    if (waves == 16) return ks25 ? launch_rollout_nw<L_, N_, Q_, 16, 25, IOT>(k, waves, lds, s) : launch_rollout_nw<L_, N_, Q_, 16, 0, IOT>(k, waves, lds, s);
    if (waves == 8) return ks25 ? launch_rollout_nw<L_, N_, Q_, 8, 25, IOT>(k, waves, lds, s) : launch_rollout_nw<L_, N_, Q_, 8, 0, IOT>(k, waves, lds, s);
    return hipErrorInvalidValue;
  } else {
  // (workgroup sizes whose per-wave regions alone exceed the LDS are not instantiated)
  constexpr size_t pw = V2 ? v2_lds_elems(L_, Q_, N_)
                        : ro_one_region<L_, N_, Q_, 0>()
                            ? (((size_t)step_region1(L_, N_) + (2 * L_ <= N_ * Q_ ? 0 : ((2 * L_ + 1) & ~1)) + vec_elems_one_region(2, L_, Q_, N_) + 1) & ~(size_t)1)
                            : ((step_lds_elems(2, L_, Q_, N_, step_tableau_in_lds<64, N_, L_>()) + 1) & ~(size_t)1);
  constexpr size_t cap = 160 * 1024 / sizeof(double);
  if constexpr (4 * pw <= cap)
    if (waves == 4) return ks25 ? launch_rollout_nw<L_, N_, Q_, 4, 25>(k, waves, lds, s) : launch_rollout_nw<L_, N_, Q_, 4, 0>(k, waves, lds, s);
  if constexpr (8 * pw <= cap)
    if (waves == 8) return ks25 ? launch_rollout_nw<L_, N_, Q_, 8, 25>(k, waves, lds, s) : launch_rollout_nw<L_, N_, Q_, 8, 0>(k, waves, lds, s);
  if constexpr (16 * pw <= cap)
    if (waves == 16) return ks25 ? launch_rollout_nw<L_, N_, Q_, 16, 25>(k, waves, lds, s) : launch_rollout_nw<L_, N_, Q_, 16, 0>(k, waves, lds, s);
  return hipErrorInvalidValue;
  }
#endif  // KMPC_ROLLOUT_JIT_TU
}

#if defined(KMPC_ROLLOUT_JIT_TU)
// (the plug-in's entry point: rollout_jit.hip)
#elif defined(KMPC_ROLLOUT_IO32_TU)
// the float32-I/O instantiations (register-state dimension sets), compiled as their own translation unit (rollout_kernel_io32.hip)
hipError_t launch_rollout_io32(const RolloutArgs<double>& a, hipStream_t s) {
  if (a.s.B <= 0 || a.steps <= 0) return hipSuccess;
  if (!a.lift_rbf && (a.Hp > 128 || (a.Hp & 15) || a.Lp > 64 || a.KS > 32 || a.nhh < 0 || a.nhh > 2 || a.s.n > 4))
    return hipErrorInvalidValue;
  if (a.s.L == 20 && a.s.N == 20 && a.s.q == 2) return launch_rollout_impl<20, 20, 2, float>(a, s);  // BASELINE cfg2 ("fp32")
#ifndef KMPC_DEV_CFG2_ONLY
  if (a.s.L == 8 && a.s.N == 30 && a.s.q == 2) return launch_rollout_impl<8, 30, 2, float>(a, s);
  if (a.s.L == 8 && a.s.N == 10 && a.s.q == 2) return launch_rollout_impl<8, 10, 2, float>(a, s);
  if (a.s.L == 10 && a.s.N == 20 && a.s.q == 1) return launch_rollout_impl<10, 20, 1, float>(a, s);
  if (a.s.L == 20 && a.s.N == 30 && a.s.q == 2) return launch_rollout_impl<20, 30, 2, float>(a, s);
#endif
  return hipErrorInvalidValue;
}
#else
#ifdef KMPC_TRACE  // (the measurement build has no float32-I/O instantiations)
hipError_t launch_rollout_io32(const RolloutArgs<double>&, hipStream_t) { return hipErrorInvalidValue; }
#else
hipError_t launch_rollout_io32(const RolloutArgs<double>& a, hipStream_t s);
#endif
// the dimension sets whose instantiations libkoopmpc.so holds itself (BASELINE.json's configurations, the reference's scripts); every
// other set of rollout_plugin_dims gets its kernel as a plug-in when a handle is created (rollout_plugin.hip)
bool rollout_builtin(int L, int N, int q, bool io32) {
  static const bool force = dbg_env("KMPC_FORCE_PLUGIN") != nullptr;  // measurement aid: every set on a plug-in compiled from the shipped sources
  if (force) return false;
#ifdef KMPC_DEV_CFG2_ONLY
  if (io32) return L == 20 && N == 20 && q == 2;
  return (L == 20 && N == 20 && q == 2) || (L == 8 && N == 30 && q == 2);
#endif
  if (io32)
    return (L == 20 && N == 20 && q == 2) || (L == 8 && N == 30 && q == 2) || (L == 8 && N == 10 && q == 2) || (L == 10 && N == 20 && q == 1) ||
           (L == 20 && N == 30 && q == 2);
  return (L == 20 && N == 20 && q == 2) || (L == 8 && N == 10 && q == 2) || (L == 8 && N == 10 && q == 8) ||
         (L == 8 && N == 30 && q == 8) || (L == 8 && N == 30 && q == 2) || (L == 10 && N == 20 && q == 1) ||
         (L == 20 && N == 30 && q == 2) || (L == 32 && N == 40 && q == 2) || (L == 32 && N == 40 && q == 1);
}
// true: a KMPC_F32 handle of this dimension set has the fused roll-out with float32 panels around the float64 state
bool rollout_io32_available(int n, int L, int N, int q, bool rbf) {
#ifdef KMPC_TRACE
  return false;
#endif
  return n == 2 && step_v2_dims(L, N, q) && rollout_fused_available<double>(n, L, N, q, 64, rbf);
}
template <typename T> bool rollout_fused_available(int n, int L, int N, int q, int threads, bool rbf) {
  if (sizeof(T) != 8 || threads == 256 || n > 4) return false;
  const bool inst = rollout_builtin(L, N, q, false) || rollout_plugin_dims(n, L, N, q);
  return inst && rollout_waves(n, L, q, N, rbf, 64) > 0;  // (Lp <= 64)
}
bool rollout_plugin_key(int n, int L, int N, int q, bool rbf, int Lp, int KS, int Hp, int B, bool io32, RolloutPluginKey* out, bool term) {
  if ((!term && rollout_builtin(L, N, q, io32)) || !rollout_plugin_dims(n, L, N, q)) return false;
  if (term && io32) return false;  // (the refresh is a float64 feature, as kmpc_terminal_from_dare)
  if (io32 && !step_v2_dims(L, N, q)) return false;
  int waves = rollout_waves(n, L, q, N, rbf, Lp, B);
  if (io32 && !rbf && waves == 4) waves = 8;  // (float32 panels: launch_rollout_impl)
  if (waves == 0) return false;
  out->L = L; out->N = N; out->q = q; out->io32 = io32 ? 1 : 0; out->term = term ? 1 : 0;
  out->ks = rbf ? -1 : ((KS == 25 && Hp == 112) ? 25 : 0);
  out->nw = rbf ? 16 : waves;  // (the RBF kernel never uses the template's tiling: one object serves every workgroup size)
  return true;
}
static hipError_t launch_rollout_plugin(const RolloutArgs<double>& a, hipStream_t s) {
  RolloutPluginKey k{};
  const bool rbf = a.lift_rbf != 0;
  if (!rollout_plugin_key(a.s.n, a.s.L, a.s.N, a.s.q, rbf, a.Lp, a.KS, a.Hp, a.s.B, a.io_f32 != 0, &k, a.term_every > 0)) return hipErrorInvalidValue;
  // (a handle loads its plug-in when it is created; a launch only compiles when the workgroup size was changed afterwards)
  const rollout_plugin_fn fn = rollout_plugin_get(k, nullptr);
  if (!fn) return hipErrorInvalidValue;
  int waves = rollout_waves(a.s.n, a.s.L, a.s.q, a.s.N, rbf, a.Lp, a.s.B);
  if (a.io_f32 && !rbf && waves == 4) waves = 8;
  return fn(&a, waves, s);
}
// true: the fused roll-out of this dimension set works on the wave image of the state (step_v2.h) instead of the dense blocks
bool rollout_uses_image(int n, int L, int N, int q) { return n == 2 && step_v2_dims(L, N, q); }
template <> hipError_t launch_rollout_fused<double>(const RolloutArgs<double>& a, hipStream_t s) {
  if (a.s.B <= 0 || a.steps <= 0) return hipSuccess;
  if (!a.lift_rbf && (a.Hp > 128 || (a.Hp & 15) || a.Lp > 64 || a.KS > 32 || a.nhh < 0 || a.nhh > 2 || a.s.n > 4))
    return hipErrorInvalidValue;
  if (a.term_every > 0 || !rollout_builtin(a.s.L, a.s.N, a.s.q, a.io_f32 != 0)) return launch_rollout_plugin(a, s);
  if (a.io_f32) return launch_rollout_io32(a, s);
  if (a.s.L == 20 && a.s.N == 20 && a.s.q == 2) return launch_rollout_impl<20, 20, 2>(a, s);
  if (a.s.L == 8 && a.s.N == 30 && a.s.q == 2) return launch_rollout_impl<8, 30, 2>(a, s);  // BASELINE cfg3 (RBF lift, y = Cx)
#if !defined(KMPC_DEV_CFG2_ONLY) || defined(KMPC_DEV_LIFT)  // (development builds compile the cfg2 / cfg3 instantiations only)
  if (a.s.L == 8 && a.s.N == 10 && a.s.q == 8) return launch_rollout_impl<8, 10, 8>(a, s);
  if (a.s.L == 8 && a.s.N == 30 && a.s.q == 8) return launch_rollout_impl<8, 30, 8>(a, s);
#endif
#ifndef KMPC_DEV_CFG2_ONLY
  if (a.s.L == 8 && a.s.N == 10 && a.s.q == 2) return launch_rollout_impl<8, 10, 2>(a, s);
  if (a.s.L == 10 && a.s.N == 20 && a.s.q == 1) return launch_rollout_impl<10, 20, 1>(a, s);  // Tank_System.m dimensions
  if (a.s.L == 20 && a.s.N == 30 && a.s.q == 2) return launch_rollout_impl<20, 30, 2>(a, s);  // BASELINE cfg3 sizes, y = Cx
  if (a.s.L == 32 && a.s.N == 40 && a.s.q == 2) return launch_rollout_impl<32, 40, 2>(a, s);  // BASELINE cfg4 sizes
  if (a.s.L == 32 && a.s.N == 40 && a.s.q == 1) return launch_rollout_impl<32, 40, 1>(a, s);
#endif
  return hipErrorInvalidValue;
}
template <> hipError_t launch_rollout_fused<float>(const RolloutArgs<float>&, hipStream_t) { return hipErrorInvalidValue; }
template bool rollout_fused_available<float>(int, int, int, int, int, bool);
template bool rollout_fused_available<double>(int, int, int, int, int, bool);

#endif  // KMPC_ROLLOUT_IO32_TU

}  // namespace kmpc
