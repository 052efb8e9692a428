// step_body.h -- the per-trajectory step (RLS-EDMD update -> condensed QP build -> box-QP solve) as device
// templates, shared by the per-step kernel (step_kernel.hip) and the fused roll-out kernel (rollout_kernel.hip).
// The arithmetic restated here and the LDS map are described at the top of step_kernel.hip.
#pragma once
#include "kernels.h"
#include "plant_device.h"

#ifdef KMPC_TRACE
// Measurement build only (make trace -> libkoopmpc_trace.so, tools/trace_phases.py): lane 0 of every
// workgroup stamps the 100 MHz wall clock at the phase boundaries.  Never compiled into libkoopmpc.so.
__device__ unsigned long long kmpc_trace_buf[8192 * 32];
#define KTRACE(slot)                                                                                      \
  do {                                                                                                    \
    if (tid == 0 && b < 8192) kmpc_trace_buf[b * 32 + (slot)] = wall_clock64();                          \
  } while (0)
extern "C" int kmpc_trace_read(void* host, size_t bytes) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(kmpc_trace_buf), bytes, 0, hipMemcpyDeviceToHost);
}
#else
#define KTRACE(slot)
#endif

namespace kmpc {

// ---------------------------------------------------------------------------------------
// host-side LDS sizing (shared with the launcher)
// ---------------------------------------------------------------------------------------
static constexpr int imax(int a, int b) { return a > b ? a : b; }

static constexpr int vec_elems(int n, int L, int q, int N) {
  const int p = L + 1;
  const int setA = 2 * p + 6 * L + n + 2 * N * q;  // sz sPz | sy sE sV(2) sW(2) | sx | sG sEr
  const int setB = 3 * N;                          // qx qxa qg
  return imax(setA, setB) + N /*sf*/ + 16 /*reduction scratch*/;
}

static constexpr int vec_elems_one_region(int n, int L, int q, int N) {  // (step_one_region kernels: see the pointer map in step_body)
  return 5 * L + n + 2 * N * q + N + 16;
}
static constexpr int step_region1(int L, int N) { return (imax(imax((L + 1) * (L + 1), L * L), N * N) + 1) & ~1; }
// lds_tableau: the region also has to hold the N x N tableau of the LDS solver (run-time dimensions).  The register
// solvers (static N) only need that tableau in their rare fall-back, which then works in a per-trajectory global
// scratch block instead: for long horizons this region was most of a trajectory's LDS (N = 30, L = 8: 19 -> 12.7 KB,
// 8 -> 12 trajectories per CU).
static constexpr int step_region2(int n, int L, int N, bool lds_tableau = true) {
  return (imax(L * (L + 1) + n * L, lds_tableau ? N * N : 0) + 1) & ~1;
}
static constexpr size_t step_lds_elems(int n, int L, int q, int N, bool lds_tableau = true) {
  return (size_t)step_region1(L, N) + step_region2(n, L, N, lds_tableau) + vec_elems(n, L, q, N);
}

// ---------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------
// One trajectory is solved by TPB threads.  TPB == 64: a single wave -- its lanes exchange data through LDS in
// program order, so a "barrier" is only a compiler/memory fence at wavefront scope and the code may run as one
// wave of a larger workgroup (the fused roll-out kernel puts 16 trajectories in a workgroup).  TPB == 256:
// a real workgroup barrier.
template <int TPB> __device__ __forceinline__ void block_sync() {
  if constexpr (TPB == 64) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}
// Round 4 -- a barrier for data exchanged through LDS only.  __syncthreads() also waits for every outstanding vector-memory
// operation of the wave (s_waitcnt vmcnt(0)): in the four-wave step that made the first barrier wait for ALL of the state
// requested up front (bar_Q, C, inv_K_G, [A B]: the loads are issued in the order of use so that each phase waits only for its
// own block) and every phase of the update wait until its write-back had reached HBM.  Here only the LDS counter is drained;
// registers that receive loads are waited for where they are used (the compiler's own counters), stores complete in the
// background.  NOT for data that threads of a workgroup hand each other through global memory.
template <int TPB> __device__ __forceinline__ void block_sync_lds() {
  if constexpr (TPB == 64) {
    block_sync<TPB>();
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
}
template <int TPB> __device__ __forceinline__ int local_tid() {
  // (single wave: the lane index from mbcnt, so that no register has to keep threadIdx alive)
  int t = TPB == 64 ? (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) : (int)threadIdx.x;
  // opaque on purpose: inside the step loop of the fused roll-out every address and mask derived from the lane
  // index is loop-invariant, and hoisting them all out of the loop costs far more registers than recomputing
  asm volatile("" : "+v"(t));
  return t;
}

// the arguments that change from step to step inside a fused roll-out (everything else stays in the kernel
// argument segment); the per-step kernel fills it from its StepArgs
template <typename T> struct StepVar {
  int phases, first_update, plant_switched;
  const T* psi_prev;
  const T* psi_now;
  T* U0;
  T* x_next;  // fused roll-out: LDS slot that receives x_{k+1} for the workgroup's next lift (else null)
  // fused roll-out: lane i < L carries psi_i(x_k) and psi_i(x_{k-1}) in registers -- the step does not read them back
  // from memory (the read-back sat behind the round trip of the store that had just written them)
  int psi_in_regs;
  T psi_now_v, psi_prev_v;
  // solve-only launches of the shared-model step: H is ONE matrix for the whole batch and is read where it lies (global memory,
  // L1 / L2 hits) instead of being staged into every trajectory's LDS -- the launch then needs LDS for its vectors only
  bool h_global = false;
  // register-state step (step_v2.h): the covariance half of this step's RLS update was done at the end of the previous step /
  // is to be done for the next one at the end of this step
  int cov_done = 0, cov_ahead = 0;
};

// e = tid, tid + TPB, ... < count.  With a compile-time COUNT the loop is fully unrolled, so that the loads of all
// its iterations are in flight before the first one is waited for (as a run-time loop each iteration is a
// complete memory round trip: 14 of them made up the 6 us the RLS phase spent fetching P and K).
template <int TPB, int COUNT, typename F> __device__ __forceinline__ void for_strided(int tid, int count, F&& f) {
  if constexpr (COUNT > 0) {
    constexpr int IT = (COUNT + TPB - 1) / TPB;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int e = tid + i * TPB;
      if ((i + 1) * TPB <= COUNT || e < COUNT) f(e, i);  // only the last round needs the lane test (tid < TPB)
    }
  } else {
    int i = 0;
    for (int e = tid; e < count; e += TPB, ++i) f(e, i);
  }
}

// true: the tableau region of this instantiation is in LDS -- the LDS solver (run-time dimensions), or a register
// solver whose model block [K, C] is at least as large as N^2 anyway (cfg2: nothing to save, and the fall-back code
// with a global pointer costs the fused kernel registers: 95 -> 91 M steps/s when tried); false: a register solver with
// a long horizon, whose fall-back tableau lives in global scratch (StepArgs::qp_scratch)
static constexpr bool tableau_saves_lds(int N, int L) { return N * N > L * (L + 1) + 4 * L; }
template <int TPB, int N_, int L_> constexpr bool step_tableau_in_lds() {
  return !(N_ > 0 && L_ > 0 && ((TPB == 64 && N_ <= 40) || (TPB == 256 && N_ <= 64)) && tableau_saves_lds(N_, L_));
}

// Four-wave trajectories with static dimensions (cfg5 sizes: L = 64, N = 50) keep inv_K_G, [A B], bar_Q and H in ONE LDS
// region, one after the other (bar_Q / C update first, then inv_K_G, then [A B], whose rows go to registers for the
// recursion, then H): 74 -> 41 KB per trajectory at L = 64, three workgroups on a CU instead of two (measured: two instead
// of one is 1.76 x).  Region 2 only holds C.
template <int TPB, int L_, int N_, int Q_> constexpr bool step_one_region() {
  return TPB == 256 && L_ > 0 && N_ > 0 && Q_ > 0 && Q_ != L_ && Q_ <= 2 && (L_ & 3) == 0 && L_ <= 64 &&
         ((L_ + 1) * (L_ + 1) + TPB - 1) / TPB <= 17;
}

template <typename T> struct Tol;
template <> struct Tol<double> {
  static __device__ __forceinline__ double kkt() { return 1e-9; }
  static __device__ __forceinline__ double tight() { return 1e-12; }
  static __device__ __forceinline__ double act() { return 1e-8; }
  static __device__ __forceinline__ double slack() { return 1e-14; }
};
template <> struct Tol<float> {
  static __device__ __forceinline__ float kkt() { return 2e-5f; }
  static __device__ __forceinline__ float tight() { return 2e-6f; }
  static __device__ __forceinline__ float act() { return 1e-5f; }
  static __device__ __forceinline__ float slack() { return 1e-6f; }
};

__device__ __forceinline__ double tfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float tfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// (|v| as the source modifier of the instruction that uses it: a compare and a select per absolute value -- three vector instructions
//  where none is needed -- were a twentieth of the cfg3 step)
__device__ __forceinline__ double tabs(double v) { return __builtin_fabs(v); }
__device__ __forceinline__ float tabs(float v) { return __builtin_fabsf(v); }
// Accesses through a generic pointer that is KNOWN to point into LDS (StepVar::x_next in the fused roll-out).  Where the other arm of
// the choice is a global pointer the compiler selects between the two ADDRESSES and issues a flat access, which waits for both
// counters -- behind the step's write-back that is a drain of every outstanding store.
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) int lds_i32;
typedef __attribute__((address_space(3))) float lds_f32;
__device__ __forceinline__ double lds_ld(const double* p) { return *(const lds_f64*)p; }
__device__ __forceinline__ void lds_st(double* p, double v) { *(lds_f64*)p = v; }
__device__ __forceinline__ float lds_ld(const float* p) { return *(const lds_f32*)p; }
__device__ __forceinline__ void lds_st(float* p, float v) { *(lds_f32*)p = v; }
__device__ __forceinline__ double uniform_value(double v) {  // a value every lane holds alike, moved to scalar registers
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ float uniform_value(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
template <typename T> __device__ __forceinline__ T tclip(T v, T lo, T hi) { return v < lo ? lo : (v > hi ? hi : v); }
// Caller-owned panels of a roll-out with float32 I/O (IOT = float, row g2 of the scope table: BASELINE configs[1] names fp32, SURVEY G6
// asks for fp32 panels around a float64 state): the StepArgs / RolloutArgs fields keep their T* type, the element is addressed and
// converted as IOT.  With IOT = T these are plain loads and stores.
template <typename IOT, typename T> __device__ __forceinline__ T io_ld(const T* base, size_t i) { return (T) reinterpret_cast<const IOT*>(base)[i]; }
template <typename IOT, typename T> __device__ __forceinline__ void io_st(T* base, size_t i, T v) { reinterpret_cast<IOT*>(base)[i] = (IOT)v; }
template <typename IOT, typename T> __device__ __forceinline__ T* io_at(T* base, size_t i) { return reinterpret_cast<T*>(reinterpret_cast<IOT*>(base) + i); }
template <typename IOT, typename T> __device__ __forceinline__ const T* io_at(const T* base, size_t i) { return reinterpret_cast<const T*>(reinterpret_cast<const IOT*>(base) + i); }

// ---- register-resident mat-vec chain (condense, static path): the current vector lives in the lanes and is
// read with the f64 DPP row broadcast of gfx90a+ -- v_fmac_f64_dpp ... row_newbcast:n multiplies by lane n of
// the reader's own 16-lane row -- so a chain step touches neither LDS nor a barrier.
template <int LANE, bool NOP>
__device__ __forceinline__ void fmac_rowbcast(double& acc, double vec, double coef) {
  // a VALU write of `vec` must be 2 wait states old before DPP reads it: the first use of a step carries the nop
  if constexpr (NOP)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc) : "v"(vec), "v"(coef), "n"(LANE));
  else
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc) : "v"(vec), "v"(coef), "n"(LANE));
}
// acc[l & 3] += row[l] * vec[l], vec[l] = lane l of v0 (l < 16) or lane l-16 of v1, within the reader's row
template <int L_, int l = 0>
__device__ __forceinline__ void chain_dot(double (&ac)[4], double v0, double v1, const double (&row)[L_]) {
  if constexpr (l < L_) {
    if constexpr (l < 16) fmac_rowbcast<l, l == 0>(ac[l & 3], v0, row[l]);
    else fmac_rowbcast<l - 16, l == 16>(ac[l & 3], v1, row[l]);
    chain_dot<L_, l + 1>(ac, v0, v1, row);
  }
}
// Lane t of a 32-lane half holds element t.  v_permlane16_swap (gfx950) exchanges the odd 16-lane rows of its
// first operand with the even rows of the second: with both = a, every row of a half receives
// v0 = elements 0..15 and v1 = elements 16..31 of that half, the layout chain_dot reads.
__device__ __forceinline__ void half_gather(double a, double& v0, double& v1) {
  const int lo = __double2loint(a), hi = __double2hiint(a);
  const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  v0 = __hiloint2double(rh[0], rl[0]);
  v1 = __hiloint2double(rh[1], rl[1]);
}

// ---- wave-wide sum on DPP (no LDS crossbar): Hillis-Steele prefix inside each 16-lane row with
// row_shr 1/2/4/8 (bound_ctrl zero-fills), then the four row totals are read from lanes 15/31/47/63.
__device__ __forceinline__ int dpp_shr(int v, int ctrl) {
  switch (ctrl) {
    case 1: return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    case 2: return __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    case 4: return __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    default: return __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
  }
}
__device__ __forceinline__ double dpp_shr(double v, int ctrl) {
  return __hiloint2double(dpp_shr(__double2hiint(v), ctrl), dpp_shr(__double2loint(v), ctrl));
}
__device__ __forceinline__ float dpp_shr(float v, int ctrl) { return __int_as_float(dpp_shr(__float_as_int(v), ctrl)); }
__device__ __forceinline__ double lane_bcast(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                          __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ float lane_bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
template <typename T> __device__ __forceinline__ T wave_sum(T v) {
  v += dpp_shr(v, 1);
  v += dpp_shr(v, 2);
  v += dpp_shr(v, 4);
  v += dpp_shr(v, 8);
  return (lane_bcast(v, 15) + lane_bcast(v, 31)) + (lane_bcast(v, 47) + lane_bcast(v, 63));
}

// FOUR wave sums at once: lane l ends with the sum of value (l & 3) over the 64 lanes.  Two butterfly steps fold the four values
// into one per lane (a lane keeps the value of its own class and sends the other), two row rotations and two row swaps add up
// the sixteen lanes of a class: ~40 instructions instead of 4 x 23 for four wave_sum.
__device__ __forceinline__ int dpp_ror(int v, int n) {
  return n == 4 ? __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, true) : __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, true);
}
__device__ __forceinline__ double dpp_ror(double v, int n) {
  return __hiloint2double(dpp_ror(__double2hiint(v), n), dpp_ror(__double2loint(v), n));
}
__device__ __forceinline__ double rows_sum(double v) {  // sum over the four 16-lane rows, lane by lane; every row gets it
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double s = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  const int slo = __double2loint(s), shi = __double2hiint(s);
  const auto c = __builtin_amdgcn_permlane32_swap(slo, slo, false, false);
  const auto d = __builtin_amdgcn_permlane32_swap(shi, shi, false, false);
  return __hiloint2double(d[0], c[0]) + __hiloint2double(d[1], c[1]);
}

// sum over the whole block; every thread gets the result.  `red` holds >= 8 elements.
template <typename T, int TPB, bool LDS_ONLY = false> __device__ __forceinline__ T block_sum(T v, T* red) {
  v = wave_sum(v);
  if (TPB == 64) return v;
  const int w = threadIdx.x >> 6;
  if constexpr (LDS_ONLY) block_sync_lds<TPB>(); else block_sync<TPB>();  // protect red[] from the previous use
  if ((threadIdx.x & 63) == 0) red[w] = v;
  if constexpr (LDS_ONLY) block_sync_lds<TPB>(); else block_sync<TPB>();
  T s = T(0);
#pragma unroll
  for (int i = 0; i < TPB / 64; ++i) s += red[i];
  return s;
}

// two sums at once (their shuffle chains interleave); results returned in place
template <typename T, int TPB, bool LDS_ONLY = false> __device__ __forceinline__ void block_sum2(T& v0, T& v1, T* red) {
  v0 += dpp_shr(v0, 1); v1 += dpp_shr(v1, 1);
  v0 += dpp_shr(v0, 2); v1 += dpp_shr(v1, 2);
  v0 += dpp_shr(v0, 4); v1 += dpp_shr(v1, 4);
  v0 += dpp_shr(v0, 8); v1 += dpp_shr(v1, 8);
  v0 = (lane_bcast(v0, 15) + lane_bcast(v0, 31)) + (lane_bcast(v0, 47) + lane_bcast(v0, 63));
  v1 = (lane_bcast(v1, 15) + lane_bcast(v1, 31)) + (lane_bcast(v1, 47) + lane_bcast(v1, 63));
  if (TPB == 64) return;
  const int w = threadIdx.x >> 6;
  if constexpr (LDS_ONLY) block_sync_lds<TPB>(); else block_sync<TPB>();
  if ((threadIdx.x & 63) == 0) { red[w] = v0; red[4 + w] = v1; }
  if constexpr (LDS_ONLY) block_sync_lds<TPB>(); else block_sync<TPB>();
  T s0 = T(0), s1 = T(0);
#pragma unroll
  for (int i = 0; i < TPB / 64; ++i) { s0 += red[i]; s1 += red[4 + i]; }
  v0 = s0; v1 = s1;
}

// ---------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------
// Register-tableau box QP for compile-time N <= 40, one wave per trajectory.
//
// The 64 lanes form an 8 x 8 grid (ti = lane>>3, tj = lane&7).  Lane (ti, tj) keeps the blocks
// Tm[r][c] = T(ti+8r, tj+8c) and Hm[r][c] = H(ti+8r, tj+8c) of the swept tableau and of H in
// REGISTERS for the whole solve; variable i is owned by lane (i&7, i>>3).  A sweep on variable k
// needs column k for the lane's rows and row k for its columns: two register fetches across
// lanes (ds_bpermute, no LDS storage, no barrier), then RM*RM fused multiply-adds
// T_ij -= T_ik T_kj / d, with T_kj <- s T_kj / d, T_ik <- s T_ik / d, T_kk <- -1/d on the lanes that hold
// row / column k (s = +1 sweep in, -1 sweep out).
// Mat-vecs (Newton direction with T, line-search products with H) are partial sums over the
// lane's columns followed by an 8-lane DPP all-reduce.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int dpp_q(int v, int sel) {
  switch (sel) {
    case 0: return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    case 1: return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    default: return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true); // row_half_mirror
  }
}
__device__ __forceinline__ double dpp_q(double v, int sel) {
  return __hiloint2double(dpp_q(__double2hiint(v), sel), dpp_q(__double2loint(v), sel));
}
__device__ __forceinline__ float dpp_q(float v, int sel) { return __int_as_float(dpp_q(__float_as_int(v), sel)); }
template <typename T> __device__ __forceinline__ T allreduce8(T v) {
  v += dpp_q(v, 0);
  v += dpp_q(v, 1);
  v += dpp_q(v, 2);
  return v;
}
__device__ __forceinline__ double wave_sum4(double a, double b, double c, double d, int lane) {
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
  const double r1 = (b0 ? b : a) + dpp_q(b0 ? a : b, 0);  // lane ^ 1: even lanes collect a, odd lanes b
  const double r2 = (b0 ? d : c) + dpp_q(b0 ? c : d, 0);  //           even lanes collect c, odd lanes d
  double r = (b1 ? r2 : r1) + dpp_q(b1 ? r1 : r2, 1);     // lane ^ 2: lane & 3 = 0, 1, 2, 3 collect a, b, c, d
  r += dpp_ror(r, 4);                                     // the four lanes of a class inside a 16-lane row
  r += dpp_ror(r, 8);
  return rows_sum(r);
}

// fast reciprocal: hardware estimate + two Newton steps (full double / float accuracy for the
// normalised pivots met here; an IEEE-exact division costs ~10 dependent f64 ops at ~40 cycles each)
__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ float fast_rcp(float d) {
  float r = __builtin_amdgcn_rcpf(d);
  r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
  return r;
}
// register fetch from another lane with a precomputed byte address (ds_bpermute_b32)
__device__ __forceinline__ double bperm(int addr, double v) {
  return __hiloint2double(__builtin_amdgcn_ds_bpermute(addr, __double2hiint(v)),
                          __builtin_amdgcn_ds_bpermute(addr, __double2loint(v)));
}
__device__ __forceinline__ float bperm(int addr, float v) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v)));
}

template <typename T, int N_, int KR>
__device__ __forceinline__ void sweep_regs(T (&Tm)[(N_ + 7) / 8][(N_ + 7) / 8], int kt, bool rev, T d, int ti, int tj) {
  constexpr int RM = (N_ + 7) / 8;
  const T dinv = fast_rcp(d);
  const T s = rev ? T(-1) : T(1);
  const bool rowk = (ti == kt), colk = (tj == kt);  // this lane holds row k / column k in block KR
  const int arow = (ti * 8 + kt) << 2, acol = (kt * 8 + tj) << 2;
  T ct[RM], rt[RM];
#pragma unroll
  for (int r = 0; r < RM; ++r) ct[r] = bperm(arow, Tm[r][KR]);  // T(ti+8r, k)
#pragma unroll
  for (int c = 0; c < RM; ++c) rt[c] = bperm(acol, Tm[KR][c]) * dinv;  // T(k, tj+8c) / d
  // Uniform update v = T_ij - ct_i * rt_j.  Row and column k come out of the SAME expression by
  // editing its inputs on the lanes that hold them (exact, no cancellation):
  //   row k    : T := 0, ct := -s             ->  v =  s T_kj / d
  //   column k : T := 0, rt := -s / d         ->  v =  s T_ik / d
  //   (k, k)   : both                         ->  v = -1 / d
  if (rowk) {
    ct[KR] = -s;
#pragma unroll
    for (int c = 0; c < RM; ++c) Tm[KR][c] = T(0);
  }
  if (colk) {
    rt[KR] = -s * dinv;
#pragma unroll
    for (int r = 0; r < RM; ++r) Tm[r][KR] = T(0);
  }
#pragma unroll
  for (int r = 0; r < RM; ++r)
#pragma unroll
    for (int c = 0; c < RM; ++c) Tm[r][c] -= ct[r] * rt[c];
}

// (kr is a constant after unrolling: the switch folds to one call)
template <typename T, int N_>
__device__ __forceinline__ void sweep_regs_at(T (&Tm)[(N_ + 7) / 8][(N_ + 7) / 8], int kr, int kt, bool rev, T d, int ti, int tj) {
  constexpr int RM = (N_ + 7) / 8;
  switch (kr) {
    case 0: sweep_regs<T, N_, 0>(Tm, kt, rev, d, ti, tj); break;
    case 1: if constexpr (RM > 1) sweep_regs<T, N_, 1>(Tm, kt, rev, d, ti, tj); break;
    case 2: if constexpr (RM > 2) sweep_regs<T, N_, 2>(Tm, kt, rev, d, ti, tj); break;
    case 3: if constexpr (RM > 3) sweep_regs<T, N_, 3>(Tm, kt, rev, d, ti, tj); break;
    default: if constexpr (RM > 4) sweep_regs<T, N_, 4>(Tm, kt, rev, d, ti, tj); break;
  }
}

template <typename T> __device__ __forceinline__ T wave_max_x(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const T w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
  return v;
}
template <typename T> __device__ __forceinline__ T wave_min_x(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const T w = __shfl_xor(v, o, 64); v = w < v ? w : v; }
  return v;
}

template <typename T, int N_, bool LOWREG = false, bool ASREG = true>
__device__ __forceinline__ bool qp_regs(const T* sH, const T* sf, const StepArgs<T>& a, const StepVar<T>& sv, int b,
                                        T* red, T* qx_out, T up, T xw_pre) {
  constexpr int RM = (N_ + 7) / 8;
  const int tid = local_tid<64>(), ti = tid >> 3, tj = tid & 7;
  const int myvar = ti + 8 * tj;
  const bool own = (tj < RM) && (myvar < N_);
  // per-variable box: in the delta-u form the first increment also keeps the absolute input inside
  // [umin, umax]:  lb_1 = max(lb, umin - u_prev), ub_1 = min(ub, umax - u_prev)   (Tank_System.m:182-188)
  const T uprev = a.du_mode ? up : T(0);  // (requested at the top of the step)
  T lb = a.lb, ub = a.ub;
  const T tol = (T)Tol<T>::kkt();
  const T eact = (T)Tol<T>::act() * (ub - lb);
  const T xmaxb = tabs(lb) > tabs(ub) ? tabs(lb) : tabs(ub);
  if (a.du_mode && tid == 0) {
    lb = (a.umin - uprev) > lb ? (a.umin - uprev) : lb;
    ub = (a.umax - uprev) < ub ? (a.umax - uprev) : ub;
  }
  const T c0 = tclip(T(0), lb, ub);

  // The tableau lives in registers.  N <= 24 (cfg1/2, reference dims): H itself stays in LDS (this lane's blocks at
  // hb[r] + 8c) and is re-read for the few mat-vecs with H -- a second register copy costs 2 RM^2 VGPRs of the 128
  // that four waves per SIMD allow, and the fused roll-out spilled because of it.  Longer horizons evaluate more
  // line-search products per solve and keep the register copy (L = 8, N = 30: 209 vs 254 us per step).
  // (LOWREG: instantiations that run four waves per SIMD with a long horizon -- the RBF roll-out of cfg3 -- also re-read H:
  //  with both copies in registers that kernel spilled 65 dwords and wrote 15 KB of scratch per trajectory-step)
  constexpr bool HREG = N_ > 24 && !LOWREG;
  T Tm[RM][RM], Hm[HREG ? RM : 1][HREG ? RM : 1];
  int hb[RM];    // LDS element offset of H(ti+8r, tj), or -1 beyond N
  bool cok[RM];  // column tj+8c exists
#pragma unroll
  for (int r = 0; r < RM; ++r) hb[r] = (ti + 8 * r < N_) ? (ti + 8 * r) * N_ + tj : -1;
#pragma unroll
  for (int c = 0; c < RM; ++c) cok[c] = tj + 8 * c < N_;
  if constexpr (HREG) {
#pragma unroll
    for (int r = 0; r < RM; ++r)
#pragma unroll
      for (int c = 0; c < RM; ++c) Hm[r][c] = (hb[r] >= 0 && cok[c]) ? sH[hb[r] + 8 * c] : T(0);
  }
  auto Hel = [&](int r, int c) -> T {
    if constexpr (HREG) return Hm[r][c];
    else return (hb[r] >= 0 && cok[c]) ? sH[hb[r] + 8 * c] : T(0);
  };
  // shared-model mode: H is the same for every trajectory, and so is the tableau of the free set "all inputs",
  // T0 = -(2H)^-1 (built once per step by shared_model_kernel): the solve starts from it and only sweeps OUT the inputs
  // that sit on a bound, instead of sweeping every free input IN (N sweeps per trajectory and step)
  const bool t0 = a.T_in != nullptr;
#pragma unroll
  for (int r = 0; r < RM; ++r)
#pragma unroll
    for (int c = 0; c < RM; ++c)
      Tm[r][c] = t0 ? ((hb[r] >= 0 && cok[c]) ? a.T_in[hb[r] + 8 * c] : T(0)) : T(2) * Hel(r, c);
  const T fi = own ? sf[myvar] : T(0);
  // row sums of |H| for the owner's variable: partial over my columns, 8-lane all-reduce,
  // the owner of variable ti + 8*tj picks block row r = tj
  T ra = T(0);
  {
    T pa[RM];
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      T s1 = T(0);
#pragma unroll
      for (int c = 0; c < RM; ++c) s1 += tabs(Hel(r, c));
      pa[r] = allreduce8(s1);
    }
#pragma unroll
    for (int r = 0; r < RM; ++r) if (tj == r) ra = pa[r];
  }
  const T gs = tabs(fi) + T(2) * ra * xmaxb;
  // start: the previous minimiser when the handle keeps one, else clip(0) -- the reference's start (its
  // pastRes_loc is never updated: zeros at every step, duffing.py:634-635, 859).  The minimiser is unique: the
  // start only changes the work.
  T x = own ? (a.x_warm ? tclip(xw_pre, lb, ub) : c0) : T(0);  // (xw_pre: requested before the H / f pass)
  T hx = T(0);  // H x at the start (x need not be uniform: the first variable's box may differ)
  {
    T xc[RM];
#pragma unroll
    for (int c = 0; c < RM; ++c) xc[c] = __shfl(x, tj * 8 + c, 64);
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      T s0 = T(0);
#pragma unroll
      for (int c = 0; c < RM; ++c) s0 += Hel(r, c) * xc[c];
      s0 = allreduce8(s0);
      if (tj == r) hx = s0;
    }
    if (!own) hx = T(0);
  }
  T p0 = own ? x * (hx + fi) : T(0);
  T J0 = wave_sum(p0);
  const unsigned long long ownmask = __ballot(own);
  unsigned long long Smask = t0 ? ownmask : 0ull;  // lane-space mask of the variables swept into T
  int it = 0, status = 1, refresh = 0, polish = 0;
  int rtot = 0;            // prediction rounds of the solve: each is a Newton solve on a smaller face (reported with `it`)
  // Active-set mode (the safeguard, in registers): when projected Newton crawls -- three predicted points refused, eight
  // Armijo backtracks, or N + 10 iterations (ill-conditioned, almost fully saturated problems) -- the solve continues on
  // the same tableau as a PRIMAL ACTIVE-SET method: Newton direction on the free set, ratio test to the first blocking
  // bound (which joins the working set W), and at a minimiser of the face the worst wrong-signed multiplier leaves W --
  // monotone and finite for a strictly convex QP.  (Round 1 left the register solver for an LDS / global-memory tableau
  // here: 30 ms for a step of 8192 Delta-u tank QPs after the plant switch, tools/cfg4_transient.py.)
  // Compiled in for the long horizons (N > 24: their kernels have 256+ registers per lane, and their LDS has no room for a
  // tableau, so the round-1 fall-back worked on H in global memory); the short-horizon kernels sit at the 128-register limit
  // of four waves per SIMD (this code costs the cfg2 roll-out 13 %) and keep the LDS fall-back of qp_lds.
  constexpr bool AS_CT = N_ > 24 && ASREG;  // (ASREG = false: the fused RBF roll-out of cfg3, see rollout_kernel.hip)
  const bool AS_REGS = AS_CT && (a.qp_predict & 2) == 0;  // (bit 1 of qp_predict: measurement aid, the LDS fall-back instead)
  bool mode_as = false, enter_as = false, at_min = false, nopredict = false;
  int ncrawl = 0;  // refused predictions (x 3) + Armijo backtracks of this solve
  unsigned long long Wmask = 0ull;  // active-set mode: variables held at a bound (lane space)
  KTRACE(8);

  while (true) {
    T g = T(0);
    bool bad = false, inI = false, loose = false;
    if (own) {
      g = T(2) * hx + fi;
      const bool atl = x <= lb + eact, atu = x >= ub - eact;
      inI = (atl && (g > T(0))) || (atu && (g < T(0)));
      // KKT violation in gradient units: |g| inside the box, the wrong-signed part of g on a bound
      const T viol = inI ? T(0) : tabs(g);
      const T res = tabs(x - tclip(x - g, lb, ub));  // what a projected gradient step would still move
      const T xs = tabs(x) > T(1) ? tabs(x) : T(1);
      bad = !((viol <= tol * gs) || (res <= tol * xs));
      loose = viol > (T)Tol<T>::tight() * gs;
    }
    const unsigned long long Bmask = __ballot(bad);
    const bool refine = __ballot(loose) != 0ull;
    const unsigned long long Imask = __ballot(inI);
    if (!(J0 == J0) || tabs(J0) > (T)1e300) { status = 2; break; }
    // The KKT test certifies, it does not define the answer.  A point that passes it only loosely (after a
    // damped or clipped step, or after a Newton step with a tableau that many set changes have worn) gets up
    // to two more Newton solves on its face -- iterative refinement -- so that the result does not depend on
    // the path (cold or warm start) beyond rounding.  A warm start itself is never returned unsolved.
    if (Bmask == 0ull && (it > 0 || !a.x_warm)) {
      if (!refine || (AS_REGS && mode_as) || polish >= 2) { status = 0; break; }
      ++polish;
    }
    if (it >= a.max_iter || refresh > 4) { status = 1; break; }
    if (!AS_REGS) {
      if (it >= N_ + 10) { status = 3; break; }  // crawling: the caller finishes the solve with the active-set loop of qp_lds
    }
    if (AS_REGS && !mode_as && (enter_as || it >= N_ + 10)) {
      mode_as = true;
      at_min = false;
      Wmask = Imask;
    }
    if (AS_REGS && mode_as && at_min) {  // x minimises the cost on the free set: release the worst wrong-signed multiplier
      const bool inW = (Wmask >> tid) & 1ull;
      const T vr = (own && inW && !inI) ? tabs(g) / gs : T(0);
      const T vmax = wave_max_x(vr);
      if (vmax > tol) Wmask &= ~(1ull << (__ffsll((long long)__ballot(own && vr == vmax)) - 1));
      at_min = false;
    }
    unsigned long long Fmask = ((AS_REGS && mode_as) ? ~Wmask : ~Imask) & ownmask;
    if (it == 0) KTRACE(9);

    // Active-set prediction (on top of Bertsekas' iteration): when the Newton point of the free set F carries free
    // variables beyond a bound, the worst offender is fixed at that bound, swept out of the tableau (one O(N^2) sweep) and the
    // Newton point of the smaller face is recomputed -- repeated until the Newton point is feasible.  Plain projected Newton
    // finds such variables one ITERATION at a time (projection, Armijo, a new KKT pass: ~4x the work of a round); on the
    // bench workload that is 2.1 -> 1.06 iterations per solve right after the RLS reset, and the slowest of 16 trajectories --
    // whom the workgroup waits for -- needs 1.7 instead of 7.8 (tools/qp_study.py on QPs collected from the device).  The
    // predicted point is taken only if it lowers the true cost; otherwise this iteration falls back to the projected Armijo
    // step, so convergence is that of the plain method and termination is still the KKT test at the top.
    // (a predicted variable's bound waits in LDS, qx_out[variable]; nothing is written or kept when no round happens)
    const bool predict = !(AS_REGS && mode_as) && !nopredict && (a.qp_predict & 1) != 0;
    int rounds = 0;
    bool broke = false;
    T pdir = T(0);
    bool isF = false;
    while (true) {
      broke = false;
      for (int pass = 0; pass < 2; ++pass) {
        // The build of a solve's first tableau -- nothing swept in yet, the free set goes in -- is what every step pays
        // (N sweeps; set changes later in the solve are one sweep each), so it has its own straight-line code: the
        // direction of the sweep is known (no sign logic, no per-sweep update of the set mask) and the pivots are checked
        // together afterwards; a non-positive pivot leaves garbage behind, which the rebuild below discards exactly as it
        // discards the partial tableau of the general pass.  Same sweeps in the same order: bitwise the same tableau.
        if (!t0 && pass == 0 && Smask == 0ull) {
          const unsigned flo = (unsigned)Fmask, fhi = (unsigned)(Fmask >> 32);
          bool pos = true;
#pragma unroll
          for (int kt = 0; kt < 8; ++kt) {
#pragma unroll
            for (int kr = 0; kr < RM; ++kr) {
              if (kt + 8 * kr < N_) {
                const int kl = kt * 8 + kr;
                if (((kl < 32 ? flo >> kl : fhi >> (kl - 32)) & 1u) != 0u) {
                  const T d = lane_bcast(Tm[kr][kr], kt * 9);
                  pos = pos && (d > T(0));
                  sweep_regs_at<T, N_>(Tm, kr, kt, false, d, ti, tj);
                }
              }
            }
          }
          if (pos) {
            Smask = Fmask;
            break;
          }
          broke = true;
          ++refresh;
#pragma unroll
          for (int r = 0; r < RM; ++r)
#pragma unroll
            for (int c = 0; c < RM; ++c) Tm[r][c] = T(2) * Hel(r, c);
          continue;  // pass 1: the general sweeps, which drop a variable whose pivot fails
        }
        const unsigned long long diff = Smask ^ Fmask;
        // One straight-line block per variable, in owner-lane order ((k&7)*8 + (k>>3), the order of the bit scan):
        // with k a compile-time constant the block index, the pivot lane and the owner tests fold, and the tableau
        // stays in the same registers from block to block (a run-time switch over the block index cost ~20 register
        // moves per sweep).  Blocks of variables that do not change sides are skipped by a uniform branch.
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
#pragma unroll
          for (int kr = 0; kr < RM; ++kr) {
            if (kt + 8 * kr < N_) {
              const int kl = kt * 8 + kr;
              if ((diff >> kl) & 1ull) {
                const bool rev = (Smask >> kl) & 1ull;
                const T d = lane_bcast(Tm[kr][kr], kt * 9);
                if (!((rev ? -d : d) > T(0))) {  // numerical breakdown of the tableau
                  broke = true;
                  if (pass == 1) Fmask &= ~(1ull << kl);
                } else {
                  sweep_regs_at<T, N_>(Tm, kr, kt, rev, d, ti, tj);
                  Smask ^= (1ull << kl);
                }
              }
            }
          }
        }
        if (!broke || pass == 1) break;
        ++refresh;  // rebuild T = 2H and sweep the free set in from scratch
#pragma unroll
        for (int r = 0; r < RM; ++r)
#pragma unroll
          for (int c = 0; c < RM; ++c) Tm[r][c] = T(2) * Hel(r, c);
        Smask = 0ull;
      }
      Fmask = Smask;
      if (it == 0 && rounds == 0) KTRACE(10);

      // Newton direction on F: p_i = sum_{j in F} T_ij g_j
      isF = own && ((Fmask >> tid) & 1ull);
      const T gm = isF ? g : T(0);
      {
        T gc[RM];
#pragma unroll
        for (int c = 0; c < RM; ++c) gc[c] = __shfl(gm, tj * 8 + c, 64);  // owner of variable tj + 8c
#pragma unroll
        for (int r = 0; r < RM; ++r) {
          T s0 = T(0);
#pragma unroll
          for (int c = 0; c < RM; ++c) s0 += Tm[r][c] * gc[c];
          s0 = allreduce8(s0);
          if (tj == r) pdir = s0;
        }
      }
      if (!predict || broke || rounds >= N_) break;
      // the free variable whose Newton value lies furthest outside the box (free variables still sit at x)
      const T cand = x + pdir;
      const T over = isF ? (cand - ub > lb - cand ? cand - ub : lb - cand) : T(-1);
      if (__ballot(over > T(0)) == 0ull) break;  // (the usual case: the Newton point is feasible)
      const T worst = wave_max_x(over);
      const int jl = __ffsll((long long)__ballot(over == worst)) - 1;  // its owner lane
      const T bj = cand > ub ? ub : lb;
      const T delta = lane_bcast(bj - x, jl);
      if (tid == jl) qx_out[myvar] = bj;
      // gradient at the new base point: g += 2 H[:, j] delta (H is intact in LDS)
      const int jv = (jl >> 3) + 8 * (jl & 7);
      if (own) g += T(2) * sH[myvar * N_ + jv] * delta;
      Fmask &= ~(1ull << jl);
      ++rounds;
    }
    if (it == 0) KTRACE(11);

    T alpha = T(1), xa = x, hxa = T(0), Ja = J0;
    bool redo = false;
    int jb = -1;
    if (AS_REGS && mode_as) {
      // ratio test: the largest step along the Newton direction that keeps the free variables inside the box
      T al = (T)1e300;
      if (isF && pdir < T(0) && x + pdir < lb) al = (lb - x) / pdir;
      if (isF && pdir > T(0) && x + pdir > ub) al = (ub - x) / pdir;
      const T amin = wave_min_x(al);
      alpha = amin < T(1) ? (amin > T(0) ? amin : T(0)) : T(1);
      if (amin < T(1)) {
        jb = __ffsll((long long)__ballot(own && al == amin)) - 1;
        Wmask |= (1ull << jb);
      } else {
        at_min = true;
      }
    }
    while (true) {
      // free variables move along the Newton direction of the (predicted) face, the bound set of the KKT pass goes onto its
      // bounds (sign of the gradient at x, which the rounds have not kept), predicted variables onto theirs; active-set
      // mode: the working set stays where it is, the blocking variable lands exactly on its bound
      T dstep = pdir;
      if (!isF) {
        const T g0 = T(2) * hx + fi;
        dstep = (own && !(AS_REGS && mode_as)) ? ((g0 > T(0) ? lb : (g0 < T(0) ? ub : x)) - x) : T(0);
        if (rounds > 0 && own && !inI) dstep = qx_out[myvar] - x;
      }
      xa = own ? tclip(x + alpha * dstep, lb, ub) : T(0);
      if (AS_REGS && tid == jb) xa = pdir < T(0) ? lb : ub;
      T xc[RM];
#pragma unroll
      for (int c = 0; c < RM; ++c) xc[c] = __shfl(xa, tj * 8 + c, 64);
#pragma unroll
      for (int r = 0; r < RM; ++r) {
        T s0 = T(0);
#pragma unroll
        for (int c = 0; c < RM; ++c) s0 += Hel(r, c) * xc[c];
        s0 = allreduce8(s0);
        if (tj == r) hxa = s0;
      }
      T pJa = own ? xa * (hxa + fi) : T(0);
      if (rounds > 0 || (AS_REGS && mode_as)) {
        // a predicted point: accepted if it lowers the cost, else the solve goes on in active-set mode from x;
        // an active-set step (exact line search along a Newton direction of a convex quadratic) is always taken
        Ja = wave_sum(pJa);
        if (!(AS_REGS && mode_as) && !(Ja <= J0)) redo = true;
        break;
      }
      T pdec = own ? (isF ? alpha * (-g * pdir) : g * (x - xa)) : T(0);
      block_sum2<T, 64>(pJa, pdec, red);
      Ja = pJa;
      const T mag = tabs(J0) > tabs(Ja) ? tabs(J0) : tabs(Ja);
      if ((J0 - Ja >= T(1e-4) * pdec - (T)Tol<T>::slack() * mag) || alpha < T(1e-10)) break;
      alpha *= T(0.25);
      ++ncrawl;
    }
    if (AS_REGS && ncrawl >= 9) enter_as = true;
    if (redo) {  // this iteration again from its KKT pass (the sweeps put the predicted variables back)
      nopredict = true;  // as a plain projected-Newton step; a solve that keeps doing this goes on in active-set mode
      ncrawl += 3;
      if (AS_REGS && ncrawl >= 9) enter_as = true;
      continue;
    }
    nopredict = false;
    rtot += rounds;
    x = xa;
    hx = hxa;
    J0 = Ja;
    if (it == 0) KTRACE(12);
    ++it;
  }
  KTRACE(13);
#ifdef KMPC_TRACE
  if (tid == 0 && b < 8192) kmpc_trace_buf[b * 32 + 15] = (unsigned long long)(it + rtot);
#endif

  // non-finite problem data (status 2): the point handed back is the feasible start clip(0), never a NaN -- the
  // plant state and the next warm start stay finite and the caller sees the status
  if (status == 2) x = own ? c0 : T(0);
  if (status == 3) {  // hand the current point to the active-set solver
    if (own) qx_out[myvar] = x;
    return true;
  }
  const int B = a.B;
  if (own) {
    if (a.Useq) a.Useq[(size_t)myvar * B + b] = x;
    if (a.x_warm) a.x_warm[(size_t)myvar * B + b] = x;
  }
  if (tid == 0) {  // lane 0 owns variable 0
    const T uout = a.du_mode ? uprev + x : x;  // U0 = U0 + dU*(1)   (Tank_System.m:192)
    if (sv.U0) sv.U0[b] = uout;
    if (a.u_store) a.u_store[b] = uout;
    if (a.plant >= 0) {  // x_loc = f_update(0, x_loc, u_loc)   (duffing.py:871)
      // fused roll-out: x_k waits in the workgroup's LDS slot (no HBM round trip at the end of the step)
      T x1, x2;  // (x_next, when given, is the roll-out's LDS slot: lds_ld above)
      if (sv.x_next) { x1 = lds_ld(sv.x_next); x2 = lds_ld(sv.x_next + 1); }
      else { x1 = a.X_rw[b]; x2 = a.X_rw[(size_t)B + b]; }
      plant_apply<T>(a.plant, sv.plant_switched, a.plant_h, x1, x2, uout);
      a.X_rw[b] = x1;
      a.X_rw[(size_t)B + b] = x2;
      if (sv.x_next) { lds_st(sv.x_next, x1); lds_st(sv.x_next + 1, x2); }
    }
    if (sv.x_next) {  // ... and the status / iteration counters are accumulated next to it, written once at the end
      lds_i32* const acc = (lds_i32*)reinterpret_cast<int*>(sv.x_next + 2);
      acc[0] = acc[0] > status ? acc[0] : status;
      acc[1] += it + rtot;
    } else {
      if (a.status) a.status[b] = a.accumulate ? (a.status[b] > status ? a.status[b] : status) : status;
      if (a.iters) a.iters[b] = a.accumulate ? a.iters[b] + it + rtot : it + rtot;
    }
  }
  KTRACE(14);
  return false;
}

// ---------------------------------------------------------------------------------------
// Register-tableau box QP for four-wave trajectories (compile-time N <= 64, 256 threads; cfg5 sizes N = 50).
// Same method and the same decisions as qp_regs; the 256 threads form a 16 x 16 grid (ti = tid >> 4, tj = tid & 15),
// thread (ti, tj) keeps T(ti+16r, tj+16c) and H(ti+16r, tj+16c) in registers, variable i is owned by thread
// (i & 15, i >> 4).  The threads of a grid row are the 16 lanes of a DPP row, so mat-vec partial sums are all-reduced
// with DPP; what crosses waves goes through LDS: the pivot row / column of a sweep (double-buffered: one barrier per
// sweep instead of the two barriers and the N^2 LDS read-modify-writes of the LDS tableau), the vector of a mat-vec,
// the variable sets (64-bit masks assembled with LDS atomics).
// ---------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T dpp_row_mirror(T v);
template <> __device__ __forceinline__ double dpp_row_mirror<double>(double v) {
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x140, 0xf, 0xf, true),
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x140, 0xf, 0xf, true));
}
template <> __device__ __forceinline__ float dpp_row_mirror<float>(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));
}
template <typename T> __device__ __forceinline__ T allreduce16(T v) {
  v = allreduce8(v);
  v += dpp_row_mirror<T>(v);
  return v;
}

template <typename T, int N_, int KR>
__device__ __forceinline__ void sweep_put256(const T (&Tm)[(N_ + 15) / 16][(N_ + 15) / 16], int kt, int ti, int tj, T* col, T* row) {
  constexpr int RM = (N_ + 15) / 16;
  if (tj == kt) {
#pragma unroll
    for (int r = 0; r < RM; ++r) col[ti + 16 * r] = Tm[r][KR];
  }
  if (ti == kt) {
#pragma unroll
    for (int c = 0; c < RM; ++c) row[tj + 16 * c] = Tm[KR][c];
  }
}
template <typename T, int N_, int KR>
__device__ __forceinline__ void sweep_regs256(T (&Tm)[(N_ + 15) / 16][(N_ + 15) / 16], int kt, bool rev, T d, int ti, int tj,
                                              const T* col, const T* row) {
  constexpr int RM = (N_ + 15) / 16;
  const T dinv = fast_rcp(d);
  const T s = rev ? T(-1) : T(1);
  T ct[RM], rt[RM];
#pragma unroll
  for (int r = 0; r < RM; ++r) ct[r] = col[ti + 16 * r];            // T(ti+16r, k)
#pragma unroll
  for (int c = 0; c < RM; ++c) rt[c] = row[tj + 16 * c] * dinv;     // T(k, tj+16c) / d
  if (ti == kt) {  // this thread holds row k in block row KR (see sweep_regs for the edited-input form)
    ct[KR] = -s;
#pragma unroll
    for (int c = 0; c < RM; ++c) Tm[KR][c] = T(0);
  }
  if (tj == kt) {
    rt[KR] = -s * dinv;
#pragma unroll
    for (int r = 0; r < RM; ++r) Tm[r][KR] = T(0);
  }
#pragma unroll
  for (int r = 0; r < RM; ++r)
#pragma unroll
    for (int c = 0; c < RM; ++c) Tm[r][c] -= ct[r] * rt[c];
}

// HREG: H also in registers (two workgroups per CU, 256 registers); false: re-read from LDS for the mat-vecs with H, where it
// stays intact for the whole solve (the one-region kernels: more workgroups per CU, fewer registers each)
template <typename T, int N_, bool HREG = true>
__device__ __forceinline__ bool qp_regs256(const T* sH, const T* sf, const StepArgs<T>& a, const StepVar<T>& sv, int b,
                                           T* red, T* work, T* qx_out, T up, T xw_pre, const int32_t (&cs_pre)[3]) {
  constexpr int RM = (N_ + 15) / 16;
  const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  const int myvar = ti + 16 * tj;
  const bool own = (tj < RM) && (myvar < N_);
  T* const qv = work;             // 64: the vector of a mat-vec, by variable
  T* const colb = work + 64;      // 2 x 64: pivot column of a sweep (double-buffered)
  T* const rowb = work + 192;     // 2 x 64: pivot row
  unsigned long long* const smk = reinterpret_cast<unsigned long long*>(work + 320);  // 3 variable sets
  const T uprev = a.du_mode ? up : T(0);
  T lb = a.lb, ub = a.ub;
  const T tol = (T)Tol<T>::kkt();
  const T eact = (T)Tol<T>::act() * (ub - lb);
  const T xmaxb = tabs(lb) > tabs(ub) ? tabs(lb) : tabs(ub);
  if (a.du_mode && tid == 0) {  // first increment: absolute input range folded in (Tank_System.m:182-188)
    lb = (a.umin - uprev) > lb ? (a.umin - uprev) : lb;
    ub = (a.umax - uprev) < ub ? (a.umax - uprev) : ub;
  }
  const T c0 = tclip(T(0), lb, ub);

  T Tm[RM][RM], Hm[HREG ? RM : 1][HREG ? RM : 1];
  auto Hel = [&](int r, int c) -> T {
    if constexpr (HREG) return Hm[r][c];
    else { const int i = ti + 16 * r, j = tj + 16 * c; return (i < N_ && j < N_) ? sH[i * N_ + j] : T(0); }
  };
#pragma unroll
  for (int r = 0; r < RM; ++r)
#pragma unroll
    for (int c = 0; c < RM; ++c) {
      const int i = ti + 16 * r, j = tj + 16 * c;
      const T h = (i < N_ && j < N_) ? sH[i * N_ + j] : T(0);
      if constexpr (HREG) Hm[r][c] = h;
      Tm[r][c] = T(2) * h;
    }
  // Round 3: the tableau of the trajectory's last solve (global memory, a.qp_carry) instead of 2H.  The model moves by a rank-one
  // update per step and the free set rarely changes, so it is an approximate inverse of the new 2 H_FF: the Newton direction
  // below is refined against H itself (two products per pass, each pass gains a factor |I - T 2H|) instead of paying the N
  // sweeps -- N barrier rounds -- that build a fresh tableau.  The answer does not depend on it: the KKT test uses H.
  bool carried = false;
  unsigned long long Smask0 = 0ull;
  if (a.qp_carry) {
    const int32_t (&cs)[3] = cs_pre;  // (requested before the H / f pass)
    if (__builtin_amdgcn_readfirstlane(cs[2]) != 0) {
      const T* const Tg = a.qp_carry + (size_t)b * N_ * N_;
#pragma unroll
      for (int r = 0; r < RM; ++r)
#pragma unroll
        for (int c = 0; c < RM; ++c) {
          const int i = ti + 16 * r, j = tj + 16 * c;
          Tm[r][c] = (i < N_ && j < N_) ? Tg[i * N_ + j] : T(0);
        }
      Smask0 = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(cs[0]) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(cs[1]) << 32);
      carried = true;
    }
  }
#ifdef KMPC_TRACE
  const bool carried0 = carried;
#endif
  const T fi = own ? sf[myvar] : T(0);
  if (tid < 64) qv[tid] = T(0);  // (entries beyond N stay zero)
  // y_i = sum_j M_ij v_j for the owner of variable i (v given by the owners)
  auto matvec = [&](auto&& Mel, T vin) -> T {
    block_sync_lds<256>();  // the previous vector has been read
    if (own) qv[myvar] = vin;
    block_sync_lds<256>();
    T xc[RM];
#pragma unroll
    for (int c = 0; c < RM; ++c) xc[c] = qv[tj + 16 * c];
    T out = T(0);
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      T s0 = T(0);
#pragma unroll
      for (int c = 0; c < RM; ++c) s0 += Mel(r, c) * xc[c];
      s0 = allreduce16(s0);
      if (tj == r) out = s0;
    }
    return own ? out : T(0);
  };
  auto Tel = [&](int r, int c) -> T { return Tm[r][c]; };
  T ra = T(0);
  {
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      T s1 = T(0);
#pragma unroll
      for (int c = 0; c < RM; ++c) s1 += tabs(Hel(r, c));
      s1 = allreduce16(s1);
      if (tj == r) ra = s1;
    }
  }
  const T gs = tabs(fi) + T(2) * ra * xmaxb;
  T x = own ? (a.x_warm ? tclip(xw_pre, lb, ub) : c0) : T(0);
  T hx = matvec(Hel, x);
  T J0 = block_sum<T, 256, true>(own ? x * (hx + fi) : T(0), red);
  unsigned long long Smask = Smask0;
  int it = 0, status = 1, refresh = 0, polish = 0, nsw = 0, rtot = 0, nref = 0;
  (void)nref;  // (read by the trace build only)
  bool rebuild = false;
  bool nopredict = false;
  KTRACE(8);

  while (true) {
    T g = T(0);
    bool bad = false, inI = false, loose = false;
    if (own) {
      g = T(2) * hx + fi;
      const bool atl = x <= lb + eact, atu = x >= ub - eact;
      inI = (atl && (g > T(0))) || (atu && (g < T(0)));
      const T viol = inI ? T(0) : tabs(g);
      const T res = tabs(x - tclip(x - g, lb, ub));
      const T xs = tabs(x) > T(1) ? tabs(x) : T(1);
      bad = !((viol <= tol * gs) || (res <= tol * xs));
      loose = viol > (T)Tol<T>::tight() * gs;
    }
    block_sync_lds<256>();
    if (tid < 3) smk[tid] = 0ull;
    block_sync_lds<256>();
    if (own) {
      if (bad) atomicOr(&smk[0], 1ull << myvar);
      if (loose) atomicOr(&smk[1], 1ull << myvar);
      if (inI) atomicOr(&smk[2], 1ull << myvar);
    }
    block_sync_lds<256>();
    const unsigned long long Bmask = smk[0], Imask = smk[2];
    const bool refine = smk[1] != 0ull;
    constexpr unsigned long long allmask = (N_ >= 64) ? ~0ull : ((1ull << N_) - 1ull);
    if (!(J0 == J0) || tabs(J0) > (T)1e300) { status = 2; break; }
    if (Bmask == 0ull && (it > 0 || !a.x_warm)) {  // (see qp_regs: a warm start is never returned unsolved)
      if (!refine || polish >= 2) { status = 0; break; }
      ++polish;
    }
    if (it >= a.max_iter || refresh > 4) { status = 1; break; }
    if (it >= N_ + 10) { status = 3; break; }  // crawling: the active-set loop of qp_lds finishes from here
    if (carried && it >= 4 && Bmask != 0ull) rebuild = true;  // (a carried tableau that has not converged in four iterations)
    unsigned long long Fmask = ~Imask & allmask;
    if (it == 0) KTRACE(9);

    // active-set prediction rounds inside the iteration: see qp_regs (same rule, the block-wide maximum goes through LDS)
    const bool predict = !nopredict && (a.qp_predict & 1) != 0;
    int rounds = 0;
    bool broke = false, isF = false;
    T pdir = T(0);
    while (true) {
      broke = false;
      for (int pass = 0; pass < 2; ++pass) {
        // from 2H again: asked for above, or more variables change sides than F has members
        if (rebuild || (carried && __builtin_popcountll(Smask ^ Fmask) > __builtin_popcountll(Fmask) + 2)) {
#pragma unroll
          for (int r = 0; r < RM; ++r)
#pragma unroll
            for (int c = 0; c < RM; ++c) Tm[r][c] = T(2) * Hel(r, c);
          Smask = 0ull;
          carried = false;
          rebuild = false;
        }
        unsigned long long diff = Smask ^ Fmask;
        while (diff) {
          const int k = __ffsll((long long)diff) - 1;
          diff &= diff - 1ull;
          const int kt = k & 15, kr = k >> 4;
          const bool rev = (Smask >> k) & 1ull;
          T* const col = colb + (nsw & 1) * 64;
          T* const row = rowb + (nsw & 1) * 64;
          ++nsw;
          switch (kr) {
            case 0: sweep_put256<T, N_, 0>(Tm, kt, ti, tj, col, row); break;
            case 1: if constexpr (RM > 1) sweep_put256<T, N_, 1>(Tm, kt, ti, tj, col, row); break;
            case 2: if constexpr (RM > 2) sweep_put256<T, N_, 2>(Tm, kt, ti, tj, col, row); break;
            default: if constexpr (RM > 3) sweep_put256<T, N_, 3>(Tm, kt, ti, tj, col, row); break;
          }
          block_sync_lds<256>();
          const T d = col[k];  // T(k, k)
          if (!((rev ? -d : d) > T(0))) {  // numerical breakdown of the tableau
            broke = true;
            if (pass == 1) Fmask &= ~(1ull << k);
            continue;
          }
          switch (kr) {
            case 0: sweep_regs256<T, N_, 0>(Tm, kt, rev, d, ti, tj, col, row); break;
            case 1: if constexpr (RM > 1) sweep_regs256<T, N_, 1>(Tm, kt, rev, d, ti, tj, col, row); break;
            case 2: if constexpr (RM > 2) sweep_regs256<T, N_, 2>(Tm, kt, rev, d, ti, tj, col, row); break;
            default: if constexpr (RM > 3) sweep_regs256<T, N_, 3>(Tm, kt, rev, d, ti, tj, col, row); break;
          }
          Smask ^= (1ull << k);
        }
        if (!broke || pass == 1) break;
        if (!carried) ++refresh;  // rebuild T = 2H and sweep the free set in from scratch
#pragma unroll
        for (int r = 0; r < RM; ++r)
#pragma unroll
          for (int c = 0; c < RM; ++c) Tm[r][c] = T(2) * Hel(r, c);
        Smask = 0ull;
        carried = false;
      }
      Fmask = Smask;
      if (it == 0 && rounds == 0) KTRACE(10);

      // Newton direction on F; with a carried tableau refined against H: p <- p + T (g + 2 H p)_F
      isF = own && ((Fmask >> myvar) & 1ull);
      pdir = matvec(Tel, isF ? g : T(0));
      if (carried) {
        // Round 6 -- every refinement pass also CORRECTS the tableau (symmetric rank-one secant update).  A pass applies d_k = T r_k and
        // finds the next residual r_{k+1} = r_k + 2 H d_k: the pair (y, d_k), y = r_{k+1} - r_k = 2 H d_k, is exact information about
        // H -- the true tableau T* = -(2 H_FF)^-1 maps y to -d_k.  With c = T r_{k+1} (the product the pass computes anyway),
        // T y = c - d_k, so the error on y is v = -d_k - T y = -c and
        //     T <- T + v v' / (v' y) = T - c c' / (c' y)
        // makes T exact on y at the cost of two dot products and RM^2 multiply-adds per thread -- no further product -- and the
        // correction the UPDATED tableau gives for r_{k+1} is beta c with beta = 1 - (c' r_{k+1}) / (c' y).  What a plain pass
        // (T unchanged, p += c) gains is a factor |I - T 2H| per pass, the same factor again at the next step: the tableau a
        // trajectory carries was an exact inverse once and falls behind H by one model update per step (round-4 trace: 4 - 6 passes for
        // a carried solve, 16 % of the solves gave up after six and paid the N sweeps of a rebuild on top, 31 % started from 2 H because
        // their last solve had needed five passes).  With the correction the error is deflated in exactly the directions the
        // closed loop's gradients excite, and it STAYS corrected in the tableau the next step starts from.  The answer does not depend on
        // any of it: the KKT test is evaluated with H.  (profiles/r6_cfg5_secant.txt)
        bool stale = false;
        T rprev = isF ? g : T(0);
        for (int kr = 0;; ++kr) {
          const T hp = matvec(Hel, isF ? pdir : T(0));
          const T rr = isF ? g + T(2) * hp : T(0);
          if (!__syncthreads_or(!(tabs(rr) <= (T)1e-13 * gs) && isF)) break;
          if (kr >= 6) { stale = true; break; }
          const T cn = matvec(Tel, rr);
          const T cF = isF ? cn : T(0);
          // (vector of the update by variable, for the rows and the columns of every thread's block: the sweeps' column buffer is idle
          //  here; the barriers of the sums below publish it)
          if (own) colb[myvar] = cF;
          else if (tid >= N_ && tid < 64) colb[tid] = T(0);  // (the padding of the blocks stays zero)
          T da = cF * rr, db = cF * (rr - rprev);
          block_sum2<T, 256, true>(da, db, red);
          T beta = T(1);
          if (tabs(db) > T(0) && tabs(da) < T(4) * tabs(db)) {  // (safeguard: a pair without curvature information corrects nothing)
            const T coef = -T(1) / db;
            beta = T(1) - da / db;
            T cr[RM], cc[RM];
#pragma unroll
            for (int r = 0; r < RM; ++r) cr[r] = colb[ti + 16 * r] * coef;
#pragma unroll
            for (int c = 0; c < RM; ++c) cc[c] = colb[tj + 16 * c];
#pragma unroll
            for (int r = 0; r < RM; ++r)
#pragma unroll
              for (int c = 0; c < RM; ++c) Tm[r][c] += cr[r] * cc[c];
            ++nsw;  // (the tableau in registers is no longer the one global memory holds)
          }
          pdir += beta * cn;
          rprev = rr;
          ++nref;
        }
        if (stale) {  // the carried tableau does not contract: this direction again from 2H
          rebuild = true;
          continue;
        }
      }
      if (!predict || broke || rounds >= N_) break;
      // the free variable whose Newton value lies furthest outside the box is fixed at that bound
      const T cand = x + pdir;
      const T over = isF ? (cand - ub > lb - cand ? cand - ub : lb - cand) : T(-1);
      const T wmax = wave_max_x(over);
      block_sync_lds<256>();
      if ((tid & 63) == 0) red[tid >> 6] = wmax;
      if (tid == 0) smk[0] = ~0ull;
      block_sync_lds<256>();
      const T w01 = red[0] > red[1] ? red[0] : red[1], w23 = red[2] > red[3] ? red[2] : red[3];
      const T worst = w01 > w23 ? w01 : w23;
      if (!(worst > T(0))) break;
      if (isF && over == worst) atomicMin(&smk[0], (unsigned long long)myvar);
      block_sync_lds<256>();
      const int jv = (int)smk[0];
      if (own && myvar == jv) {
        const T bj = cand > ub ? ub : lb;
        qx_out[jv] = bj;
        red[12] = bj - x;
      }
      block_sync_lds<256>();
      const T delta = red[12];
      if (own) g += T(2) * sH[myvar * N_ + jv] * delta;  // gradient at the new base point
      Fmask &= ~(1ull << jv);
      ++rounds;
    }

    if (it == 0) KTRACE(11);
    // trial point (see qp_regs): projected Armijo search on the true cost, or the predicted point if it lowers the cost
    T alpha = T(1), xa = x, hxa = T(0), Ja = J0;
    bool redo = false;
    while (true) {
      T dstep = pdir;
      if (!isF) {
        const T g0 = T(2) * hx + fi;
        dstep = own ? ((g0 > T(0) ? lb : (g0 < T(0) ? ub : x)) - x) : T(0);
        if (rounds > 0 && own && !inI) dstep = qx_out[myvar] - x;
      }
      xa = own ? tclip(x + alpha * dstep, lb, ub) : T(0);
      hxa = matvec(Hel, xa);
      T pJa = own ? xa * (hxa + fi) : T(0);
      if (rounds > 0) {
        Ja = block_sum<T, 256, true>(pJa, red);
        if (!(Ja <= J0)) redo = true;
        break;
      }
      T pdec = own ? (isF ? alpha * (-g * pdir) : g * (x - xa)) : T(0);
      block_sum2<T, 256, true>(pJa, pdec, red);
      Ja = pJa;
      const T mag = tabs(J0) > tabs(Ja) ? tabs(J0) : tabs(Ja);
      if ((J0 - Ja >= T(1e-4) * pdec - (T)Tol<T>::slack() * mag) || alpha < T(1e-10)) break;
      if (carried) { redo = true; break; }  // a stale tableau gave a poor direction: this iteration again with a fresh one
      alpha *= T(0.25);
    }
    if (redo) {
      if (carried && rounds == 0) { rebuild = true; continue; }
      nopredict = true;
      continue;
    }
    nopredict = false;
    rtot += rounds;
    x = xa;
    hx = hxa;
    J0 = Ja;
    if (it == 0) KTRACE(12);
    ++it;
  }
  KTRACE(13);
  if (status == 2) x = own ? c0 : T(0);  // (see qp_regs: non-finite data never leaves as a NaN input)
  if (status == 3) {
    block_sync_lds<256>();
    if (own) qx_out[myvar] = x;
    if (a.qp_carry && tid == 0) a.qp_carry_set[4 * (size_t)b + 2] = 0;
    return true;
  }
  const int B = a.B;
  // (the plant's state is requested before the write-back below: behind it, thread 0 sat out a memory round trip at the very end)
  T x1p = T(0), x2p = T(0);
  if (tid == 0 && a.plant >= 0) { x1p = a.X_rw[b]; x2p = a.X_rw[(size_t)B + b]; }
  if (own) {
    if (a.Useq) a.Useq[(size_t)myvar * B + b] = x;
    if (a.x_warm) a.x_warm[(size_t)myvar * B + b] = x;
  }
  if (a.qp_carry) {  // the tableau stays for the next solve (a carried one that needed many iterations does not: 2H next time)
    // (round 6: the refinement passes correct the tableau they use, so the number of passes a solve needed is no reason to drop it any more)
    const bool keep = status == 0 && !(carried && it >= 4);
    // (a carried tableau that saw no sweep is what global memory already holds: half of the settled solves, 20 KB each at N = 50)
    if (keep && !(carried && nsw == 0)) {
      T* const Tg = a.qp_carry + (size_t)b * N_ * N_;
#pragma unroll
      for (int r = 0; r < RM; ++r)
#pragma unroll
        for (int c = 0; c < RM; ++c) {
          const int i = ti + 16 * r, j = tj + 16 * c;
          if (i < N_ && j < N_) Tg[i * N_ + j] = Tm[r][c];
        }
    }
    if (tid == 0) {
      int32_t* const cs = a.qp_carry_set + 4 * (size_t)b;
      cs[0] = (int32_t)(unsigned)(Smask & 0xffffffffull);
      cs[1] = (int32_t)(unsigned)(Smask >> 32);
      cs[2] = keep ? 1 : 0;
    }
  }
  if (tid == 0) {  // thread 0 owns variable 0
    const T uout = a.du_mode ? uprev + x : x;
    if (sv.U0) sv.U0[b] = uout;
    if (a.u_store) a.u_store[b] = uout;
    if (a.plant >= 0) {  // x_loc = f_update(0, x_loc, u_loc)   (duffing.py:871)
      T x1 = x1p, x2 = x2p;
      plant_apply<T>(a.plant, sv.plant_switched, a.plant_h, x1, x2, uout);
      a.X_rw[b] = x1;
      a.X_rw[(size_t)B + b] = x2;
    }
    if (a.status) a.status[b] = a.accumulate ? (a.status[b] > status ? a.status[b] : status) : status;
    if (a.iters) a.iters[b] = a.accumulate ? a.iters[b] + it + rtot : it + rtot;
#ifdef KMPC_TRACE
    if (b < 8192) {
      kmpc_trace_buf[b * 32 + 15] = (unsigned long long)(it + rtot);
      kmpc_trace_buf[b * 32 + 16] = (unsigned long long)nref;
      kmpc_trace_buf[b * 32 + 17] = (unsigned long long)nsw;
      kmpc_trace_buf[b * 32 + 18] = (unsigned long long)((carried0 ? 1 : 0) | (carried ? 2 : 0));
      kmpc_trace_buf[b * 32 + 19] = (unsigned long long)it;
    }
#endif
  }
  KTRACE(14);
  return false;
}

// ---------------------------------------------------------------------------------------
// LDS-tableau box QP (run-time N <= 64, 64 or 256 threads): the generic solver, and the safeguard the
// register solver falls back to.  Projected Newton on the swept tableau in LDS; when it crawls
// (N+10 iterations or 2N Armijo backtracks) or when `as_from_start` is set the loop continues as a PRIMAL
// ACTIVE-SET method on the same tableau: Newton direction on the free set, ratio test to the first blocking
// bound, one wrong-signed multiplier released per minimiser -- monotone and finite for a strictly convex QP.
// ---------------------------------------------------------------------------------------
template <typename T, int TPB, typename IOT = T>
__device__ __forceinline__ void qp_lds(const T* sH, const T* sf, T* sM, T* qx, T* qxa, T* qg, T* red,
                                       const StepArgs<T>& a, const StepVar<T>& sv, int b, int N, bool as_from_start) {
  const int tid = local_tid<TPB>(), B = a.B;
  const T uprev = a.du_mode ? a.u_prev[b] : T(0);
  T lb = a.lb, ub = a.ub;
  const T tol = (T)Tol<T>::kkt();
  const T eact = (T)Tol<T>::act() * (ub - lb);
  const T xmaxb = tabs(lb) > tabs(ub) ? tabs(lb) : tabs(ub);
  if (a.du_mode && tid == 0) {  // first increment: absolute input range folded in (Tank_System.m:182-188)
    lb = (a.umin - uprev) > lb ? (a.umin - uprev) : lb;
    ub = (a.umax - uprev) < ub ? (a.umax - uprev) : ub;
  }
  const T c0 = tclip(T(0), lb, ub);
  constexpr int TS = (TPB == 64) ? 8 : 16;  // 2-D thread tile of the sweeps
  constexpr int RMAX = 64 / TS;             // rows / columns per thread (N <= 64)
  const int ti = tid / TS, tj = tid % TS;
  const bool mine = tid < N;  // thread i < N owns variable i (N <= 64 <= TPB)
  const unsigned long long allmask = (N >= 64) ? ~0ull : ((1ull << N) - 1ull);
  unsigned long long* const smask = reinterpret_cast<unsigned long long*>(red + 8);
  T* const sval = red + 12;  // broadcast slot for wave-0 scalars (256-thread blocks)

  T gs = T(1);
  if (mine) {
    T ra = T(0);
    for (int j = 0; j < N; ++j) ra += tabs(sH[j * N + tid]);
    gs = tabs(sf[tid]) + T(2) * ra * xmaxb;  // magnitude of the terms of the gradient
    // start: clip(0) as the reference (duffing.py:634-635), or the point the register solver handed over
    qx[tid] = as_from_start ? tclip(qx[tid], lb, ub) : (a.x_warm ? tclip(a.x_warm[(size_t)tid * B + b], lb, ub) : c0);
  }
  for (int i = ti; i < N; i += TS)
    for (int j = tj; j < N; j += TS) sM[i * N + j] = T(2) * sH[i * N + j];
  block_sync<TPB>();
  T x = mine ? qx[tid] : T(0), hx = T(0);
  if (mine) {  // H x at the start (x need not be uniform: the first variable's box may differ)
    for (int j = 0; j < N; ++j) hx += sH[j * N + tid] * qx[j];
  }
  T J0 = block_sum<T, TPB>(mine ? x * (hx + sf[tid]) : T(0), red);

  unsigned long long Smask = 0ull;  // variables currently swept into T
  unsigned long long Wmask = 0ull;  // active-set mode: variables held at a bound
  int it = 0, status = 1, refresh = 0, polish = 0;
  bool mode_as = false, at_min = false;

  // every thread gets wave 0's value of a block-uniform quantity
  auto bcast_mask = [&](unsigned long long m) -> unsigned long long {
    if (TPB == 64) return m;
    block_sync<TPB>();
    if (tid == 0) smask[0] = m;
    block_sync<TPB>();
    return smask[0];
  };
  auto bcast_val = [&](T v) -> T {
    if (TPB == 64) return v;
    block_sync<TPB>();
    if (tid == 0) sval[0] = v;
    block_sync<TPB>();
    return sval[0];
  };

  while (true) {
    // ---- gradient, KKT residual, natural bound set
    T g = T(0);
    bool bad = false, inI = false, loose = false;
    if (mine) {
      g = T(2) * hx + sf[tid];
      const bool atl = x <= lb + eact, atu = x >= ub - eact;
      inI = (atl && (g > T(0))) || (atu && (g < T(0)));
      const T viol = inI ? T(0) : tabs(g);  // KKT violation in gradient units
      const T res = tabs(x - tclip(x - g, lb, ub));
      const T xs = tabs(x) > T(1) ? tabs(x) : T(1);
      bad = !((viol <= tol * gs) || (res <= tol * xs));
      loose = viol > (T)Tol<T>::tight() * gs;
    }
    const unsigned long long Bmask = bcast_mask(__ballot(bad));  // wave 0 holds every variable (N <= 64)
    const bool refine = bcast_mask(__ballot(loose)) != 0ull;
    const unsigned long long Imask = bcast_mask(__ballot(inI));
    if (!(J0 == J0) || tabs(J0) > (T)1e300) { status = 2; break; }
    if (Bmask == 0ull && (it > 0 || as_from_start || !a.x_warm)) {  // see qp_regs
      if (!refine || mode_as || polish >= 2) { status = 0; break; }
      ++polish;
    }
    if (it >= a.max_iter || refresh > 4) { status = 1; break; }
    if (!mode_as && (as_from_start || it >= N + 10)) {
      mode_as = true;
      at_min = false;
      Wmask = Imask;
    }
    if (mode_as && at_min) {  // x minimises the cost on the free set: release the worst wrong-signed multiplier
      const bool inW = (Wmask >> tid) & 1ull;
      const T vr = (mine && inW && !inI) ? tabs(g) / gs : T(0);
      const T vmax = bcast_val(wave_max_x(vr));
      const unsigned long long cand = bcast_mask(__ballot(mine && vr == vmax));
      if (vmax > tol) Wmask &= ~(1ull << (__ffsll((long long)cand) - 1));
      at_min = false;
    }
    unsigned long long Fmask = (mode_as ? ~Wmask : ~Imask) & allmask;

    // ---- bring T to the free set: one symmetric sweep per changed variable
    bool broke = false;
    for (int pass = 0; pass < 2; ++pass) {
      unsigned long long diff = Smask ^ Fmask;
      while (diff) {
        const int k = __ffsll((long long)diff) - 1;
        diff &= diff - 1ull;
        const bool rev = (Smask >> k) & 1ull;
        const T piv = sM[k * N + k];
        if (!((rev ? -piv : piv) > T(0))) {  // numerical breakdown of the tableau
          broke = true;
          if (pass == 1) Fmask &= ~(1ull << k);  // clean rebuild: keep this variable fixed this iteration
          continue;
        }
        const T dinv = T(1) / piv;
        T rj[RMAX];  // this thread's columns of pivot row k
#pragma unroll
        for (int c = 0; c < RMAX; ++c) {
          const int j = tj + c * TS;
          rj[c] = (j < N && j != k) ? sM[k * N + j] : T(0);
        }
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
          const int i = ti + r * TS;
          if (i < N && i != k) {
            const T ci = sM[k * N + i] * dinv;
#pragma unroll
            for (int c = 0; c < RMAX; ++c) {
              const int j = tj + c * TS;
              if (j < N && j != k) sM[i * N + j] -= ci * rj[c];
            }
          }
        }
        block_sync<TPB>();
        if (mine) {
          if (tid == k) {
            sM[k * N + k] = -dinv;
          } else {
            const T v = sM[k * N + tid] * (rev ? -dinv : dinv);
            sM[k * N + tid] = v;
            sM[tid * N + k] = v;
          }
        }
        block_sync<TPB>();
        Smask ^= (1ull << k);
      }
      if (!broke || pass == 1) break;
      ++refresh;  // rebuild T = 2H and sweep the free set in from scratch
      for (int i = ti; i < N; i += TS)
        for (int j = tj; j < N; j += TS) sM[i * N + j] = T(2) * sH[i * N + j];
      block_sync<TPB>();
      Smask = 0ull;
    }
    Fmask = Smask;  // what is actually swept (a variable whose pivot broke down stays fixed)

    // ---- direction: Newton on F (T_FF = -(2H_FF)^-1) as a mat-vec with the masked gradient
    const bool isF = mine && ((Fmask >> tid) & 1ull);
    if (mine) qg[tid] = isF ? g : T(0);
    block_sync<TPB>();
    T pdir = T(0);
    if (isF) {
      for (int j = 0; j < N; ++j) pdir += sM[j * N + tid] * qg[j];
    } else if (mine && !mode_as) {
      pdir = (g > T(0) ? lb : (g < T(0) ? ub : x)) - x;  // projected Newton: straight to the bound on I
    }

    T alpha = T(1), xa = x, hxa = T(0), Ja = J0;
    bool one_shot = false;
    int jb = -1;
    if (mode_as) {
      // ratio test: the largest step along the Newton direction that keeps the free variables inside the box
      T al = (T)1e300;
      if (isF && pdir < T(0) && x + pdir < lb) al = (lb - x) / pdir;
      if (isF && pdir > T(0) && x + pdir > ub) al = (ub - x) / pdir;
      const T amin = bcast_val(wave_min_x(al));
      alpha = amin < T(1) ? (amin > T(0) ? amin : T(0)) : T(1);
      if (amin < T(1)) {
        const unsigned long long cand = bcast_mask(__ballot(mine && al == amin));
        jb = __ffsll((long long)cand) - 1;
        Wmask |= (1ull << jb);
      } else {
        at_min = true;
      }
      one_shot = true;
    }
    // ---- projected Armijo search on the true cost (active-set mode: one evaluation at the ratio-test step)
    while (true) {
      if (mine) {
        xa = tclip(x + alpha * pdir, lb, ub);
        if (tid == jb) xa = pdir < T(0) ? lb : ub;  // the blocking variable sits exactly on its bound
        qxa[tid] = xa;
      }
      block_sync<TPB>();
      T pJa = T(0), pdec = T(0);
      if (mine) {
        hxa = T(0);
        for (int j = 0; j < N; ++j) hxa += sH[j * N + tid] * qxa[j];
        pJa = xa * (hxa + sf[tid]);
        pdec = isF ? alpha * (-g * pdir) : g * (x - xa);
      }
      block_sum2<T, TPB>(pJa, pdec, red);
      Ja = pJa;
      const T mag = tabs(J0) > tabs(Ja) ? tabs(J0) : tabs(Ja);
      if (one_shot || (J0 - Ja >= T(1e-4) * pdec - (T)Tol<T>::slack() * mag) || alpha < T(1e-10)) break;
      alpha *= T(0.25);
      block_sync<TPB>();
    }
    x = xa;
    hx = hxa;
    J0 = Ja;
    ++it;
  }
  if (status == 2) x = c0;  // (see qp_regs: non-finite data never leaves as a NaN input)
  if (mine) qx[tid] = x;
  block_sync<TPB>();

  if (mine) {
    if (a.Useq) io_st<IOT>(a.Useq, (size_t)tid * B + b, qx[tid]);
    if (a.x_warm) a.x_warm[(size_t)tid * B + b] = qx[tid];
  }
  if (tid == 0) {
    const T uout = a.du_mode ? uprev + qx[0] : qx[0];
    if (sv.U0) io_st<IOT>(sv.U0, b, uout);
    if (a.u_store) a.u_store[b] = uout;
    if (a.plant >= 0) {  // x_loc = f_update(0, x_loc, u_loc)   (duffing.py:871)
      // fused roll-out: x_k waits in the workgroup's LDS slot (no HBM round trip at the end of the step)
      T x1, x2;  // (x_next, when given, is the roll-out's LDS slot: lds_ld above)
      if (sv.x_next) { x1 = lds_ld(sv.x_next); x2 = lds_ld(sv.x_next + 1); }
      else { x1 = io_ld<IOT>(a.X_rw, b); x2 = io_ld<IOT>(a.X_rw, (size_t)B + b); }
      plant_apply<T>(a.plant, sv.plant_switched, a.plant_h, x1, x2, uout);
      io_st<IOT>(a.X_rw, b, x1);
      io_st<IOT>(a.X_rw, (size_t)B + b, x2);
      if (sv.x_next) { lds_st(sv.x_next, x1); lds_st(sv.x_next + 1, x2); }
    }
    if (sv.x_next) {  // ... and the status / iteration counters are accumulated next to it, written once at the end
      lds_i32* const acc = (lds_i32*)reinterpret_cast<int*>(sv.x_next + 2);
      acc[0] = acc[0] > status ? acc[0] : status;
      acc[1] += it;
    } else {
      if (a.status) a.status[b] = a.accumulate ? (a.status[b] > status ? a.status[b] : status) : status;
      if (a.iters) a.iters[b] = a.accumulate ? a.iters[b] + it : it;
    }
  }
}

// L_, N_, Q_ != 0: dimensions fixed at compile time (every inner loop unrolls, index math folds);
// 0: taken from the arguments at run time (generic fallback, same source).
// Four waves per SIMD (<= 128 VGPRs) for the small static configurations: BASELINE cfg2 puts exactly 4096
// trajectories = 4 waves per SIMD on the chip, so one register too many costs a whole second round.
template <typename T, int TPB, int L_, int N_, int Q_, bool LOWREG = false, bool ASREG = true, bool ONE64 = false>
__device__ __forceinline__ void step_body(const StepArgs<T>& a, const StepVar<T>& sv, const int b, T* const sm) {
  const int tid = local_tid<TPB>();
  // y = C x has q = rows of C <= n < L outputs, y = psi has q = L: with static dimensions the output kind is known
  // at compile time and the other variant's code disappears (Q_ == L_ only in the lifted-output instantiations)
  const bool out_cx = (Q_ > 0 && L_ > 0) ? (Q_ != L_) : (a.out_kind == OUT_CX);
  const int n = a.n, L = L_ ? L_ : a.L, p = L + 1, q = Q_ ? Q_ : a.q, N = N_ ? N_ : a.N, B = a.B;

  // (ONE64: the fused RBF roll-outs -- one wave per trajectory with the regions merged the same way: L = 20, N = 30 puts 16 instead
  //  of 11 trajectories on a CU)
  constexpr bool ONE_REGION = step_one_region<TPB, L_, N_, Q_>() || (ONE64 && TPB == 64 && L_ > 0 && N_ > 0 && Q_ > 0 && Q_ != L_);
  T* const sX = sm;            // P / bar_Q / H
  T* const sY = sX + a.r1;     // K, C / elimination matrix
  T* const sK = ONE_REGION ? sX : sY;  // (one region: [A B] follows inv_K_G in region 1, region 2 is C alone)
  T* const sC_y = ONE_REGION ? sY : sY + L * p;
  T* const sH = sX;
  T* const sM = sY;  // LDS solver; the register solvers' fall-back works in global scratch instead (computed there)
  T* const vec = sY + a.r2;
  T* const red = vec;          // 16  reduction scratch (+ the 64-bit set mask at red[8])
  T* const sf = red + 16;      // N   (lives condense -> QP)
  T* const va = sf + N;        // aliased vector sets
  // set A (RLS + condense)
  // (one region: the vectors of the update -- z, Pz, the innovation -- are dead before the recursion's v and w exist, and
  //  lie where those will: 194 doubles less at L = 64, the margin the fourth workgroup of a CU needs)
  T* const sz = ONE_REGION ? va + L : va;   // p
  T* const sPz = sz + p;                     // p
  T* const sy = ONE_REGION ? va : sPz + p;   // L   psi_now
  T* const sE = ONE_REGION ? sPz + p : sy + L;  // L
  T* const sV = ONE_REGION ? va + L : sE + L;   // 2L
  T* const sW = sV + 2 * L;    // 2L
  T* const sx = sW + 2 * L;    // n
  T* const sG = sx + n;        // N*q
  T* const sEr = sG + N * q;   // N*q
  // (one region, r2 == 0: C waits in the place of g_0 .. g_{N-1}, which the recursion writes only after it has taken its
  //  rows of C_o into registers -- 128 doubles that decide between three and four workgroups per CU at L = 64)
  T* const sC = (ONE_REGION && a.r2 == 0) ? sG : sC_y;
  // set B (QP)
  T* const qx = va;
  T* const qxa = qx + N;
  T* const qg = qxa + N;

    KTRACE(0);
  // requested before anything waits: the previous input (RLS regressor, delta-u form) and -- static sizes -- the
  // reference, which the condense phase needs only after the state has been written back (requested there, the
  // load sat behind the round trip of those stores)
  // (four-wave kernel, round 4: u_{k-1} is the same for the whole workgroup -- held in scalar registers: as a vector register that
  //  lives until the solve it was the first thing the compiler spilled, and its reload made wave 0 wait for every outstanding
  //  load and store in the middle of the update)
  constexpr bool UP_SCALAR = step_one_region<TPB, L_, N_, Q_>();
  T up = a.u_prev[b];
  if constexpr (UP_SCALAR) up = uniform_value(up);
  T xw_pre = T(0);
  int32_t cs_pre[3] = {0, 0, 0};  // (four-wave solver: sets and validity of the carried tableau)
  constexpr int REFN = (Q_ > 0 && N_ > 0 && Q_ * N_ <= 4 * TPB) ? (Q_ * N_ + TPB - 1) / TPB : 0;
  T refp[REFN > 0 ? REFN : 1];
  if constexpr (REFN > 0) {
    if (sv.phases & PH_CONDENSE) {
      const T* refg = a.ref + (a.ref_per_traj ? (size_t)b * q * N : 0);
#pragma unroll
      for (int i = 0; i < REFN; ++i) {
        const int e = tid + i * TPB, ec = e < q * N ? e : 0, k = ec / q, r = ec - k * q;
        refp[i] = refg[r * N + k];
      }
    }
  }

  // =====================================================================================
  // phase 1: recursive least squares (gain form; algebraically K_A inv_K_G of the reference)
  // =====================================================================================
  if constexpr (ONE_REGION) {
   if (sv.phases & PH_RLS) {
    // The same arithmetic as the general block below, element for element, in another ORDER: the output-map update
    // (bar_Q, C: it needs neither inv_K_G nor [A B]) comes first, then inv_K_G, then [A B] -- each takes the region
    // over from its predecessor.  Every HBM request is issued before the first wait, in the order of use.
    const T* Pg = a.P + (size_t)b * a.strideP;
    T* Kg = a.K + (size_t)b * a.strideK;
    const T* Qg = a.Qb + (size_t)b * a.strideQ;
    T* Cg = a.C + (size_t)b * a.strideC;
    constexpr int PP_ = (L_ + 1) * (L_ + 1), LP_ = L_ * (L_ + 1), LL_ = L_ * L_;
    constexpr int NPR = (PP_ + TPB - 1) / TPB, NKR = (LP_ + TPB - 1) / TPB, NQR = (LL_ + TPB - 1) / TPB;
    const int il = tid < L ? tid : L - 1, in = tid < n ? tid : n - 1;
    const T zp = sv.psi_in_regs ? sv.psi_prev_v : sv.psi_prev[il * a.pp_sl + b * a.pp_sb];
    const T yp = sv.psi_in_regs ? sv.psi_now_v : sv.psi_now[il * a.pn_sl + b * a.pn_sb];
    const T xp = a.x_now[(size_t)in * B + b];
    T qr[NQR], cr[2], pr[NPR], kr[NKR];
#pragma unroll
    for (int i = 0; i < NQR; ++i) { const int e = tid + i * TPB; qr[i] = Qg[e < LL_ ? e : LL_ - 1]; }
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int e = tid + i * TPB; cr[i] = Cg[e < n * L ? e : 0]; }
#pragma unroll
    for (int i = 0; i < NPR; ++i) { const int e = tid + i * TPB; pr[i] = Pg[e < PP_ ? e : PP_ - 1]; }
    if (tid < L) { sz[tid] = zp; sy[tid] = yp; }
    if (tid == 0) sz[L] = up;
    if (tid < n) sx[tid] = xp;
    const bool fu = sv.first_update != 0;
    // ---- C = bar_X bar_Q, target x_{k+1}, regressor psi(x_k)      duffing.py:943-953
#pragma unroll
    for (int i = 0; i < NQR; ++i) { const int e = tid + i * TPB; if (e < LL_) sX[e] = qr[i]; }
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int e = tid + i * TPB; if (e < n * L) sC[e] = fu ? T(0) : cr[i]; }
    // (round 4: [A B] is requested only now, in the registers bar_Q has just left: with all four blocks requested up front the 104
    //  registers they need did not exist -- the compiler waited for EVERY load to arrive, to spill 37 of them to scratch)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < NKR; ++i) { const int e = tid + i * TPB; kr[i] = Kg[e < LP_ ? e : LP_ - 1]; }
    block_sync_lds<TPB>();
    KTRACE(1);
    // Round 4, four waves and L = 64: the three matrix-vector products of the update (bar_Q z, inv_K_G z, [A B] z) were ONE wave
    // walking 64 or 65 dependent LDS round trips while three waited at the barrier (2.7 us each, a tenth of the trajectory-step).
    // Wave w now sums a quarter of the columns for all 64 rows, the partial sums meet in LDS (where the recursion's outputs will
    // go), wave 0 adds them and the idle waves take the odd ends (the rows of C, element 65 of inv_K_G z, the two dot products).
    constexpr bool SPLIT4 = TPB == 256 && L_ == 64;
    const int wv = tid >> 6, ln = tid & 63;
    T* const pt = sG;  // 3 x (L + 1)   (region 2 holds C: the launcher never folds it into sG for this kernel)
    T dc;
    if constexpr (SPLIT4) {
      T acc = T(0);
#pragma unroll
      for (int jj = 0; jj < L_ / 4; ++jj) { const int j = wv * (L_ / 4) + jj; acc += sX[j * L_ + ln] * sz[j]; }
      if (wv) pt[(wv - 1) * (L_ + 1) + ln] = acc;
      block_sync_lds<TPB>();
      if (wv == 0) {
        const T pz = (acc + pt[ln]) + (pt[(L_ + 1) + ln] + pt[2 * (L_ + 1) + ln]);
        sPz[ln] = pz;
        const T zz = wave_sum(sz[ln] * pz);
        if (ln == 0) red[0] = zz;
      } else {
        for (int r = wv - 1; r < n; r += 3) {
          const T cz = wave_sum(sC[r * L_ + ln] * sz[ln]);
          if (ln == 0) sE[r] = sx[r] - cz;
        }
      }
      block_sync_lds<TPB>();
      dc = T(1) + red[0];
    } else {
    for (int i = tid; i < L; i += TPB) {
      T acc = T(0);
#pragma unroll
      for (int j = 0; j < L; ++j) acc += sX[j * L + i] * sz[j];
      sPz[i] = acc;
    }
    for (int r = tid; r < n; r += TPB) {
      T acc = sx[r];
      for (int j = 0; j < L; ++j) acc -= sC[r * L + j] * sz[j];
      sE[r] = acc;
    }
    block_sync_lds<TPB>();
      T part2 = T(0);
      for (int i = tid; i < L; i += TPB) part2 += sz[i] * sPz[i];
      dc = T(1) + block_sum<T, TPB, true>(part2, red);
    }
    {
      const T dcinv = T(1) / dc;
      T* Qw = a.Qb + (size_t)b * a.strideQ;
      for_strided<TPB, LL_>(tid, L * L, [&](int e, int) {
        const int i = e / L, j = e - i * L;
        Qw[e] = sX[e] - (sPz[i] * sPz[j]) * dcinv;
      });
      for (int e = tid; e < n * L; e += TPB) {
        const int r = e / L, j = e - r * L;
        const T v = (a.c_skip_first && fu) ? T(0) : sC[e] + sE[r] * (sPz[j] * dcinv);
        sC[e] = v;
        Cg[e] = v;
      }
    }
    block_sync_lds<TPB>();  // everyone is done with bar_Q in the region and with sPz / sE
    KTRACE(2);
    // ---- inv_K_G                                                     duffing.py:931-932
#pragma unroll
    for (int i = 0; i < NPR; ++i) { const int e = tid + i * TPB; if (e < PP_) sX[e] = pr[i]; }
    block_sync_lds<TPB>();
    T d;
    if constexpr (SPLIT4) {
      constexpr int CH = (L_ + 4) / 4;  // 17, 17, 17, 14 columns
      T acc = T(0);
#pragma unroll
      for (int jj = 0; jj < CH; ++jj) { const int j = wv * CH + jj; if (j <= L_) acc += sX[j * (L_ + 1) + ln] * sz[j]; }
      if (wv) pt[(wv - 1) * (L_ + 1) + ln] = acc;
      block_sync_lds<TPB>();
      if (wv == 0) {
        const T pz = (acc + pt[ln]) + (pt[(L_ + 1) + ln] + pt[2 * (L_ + 1) + ln]);
        sPz[ln] = pz;
        const T zz = wave_sum(sz[ln] * pz);
        if (ln == 0) red[0] = zz;
      } else if (wv == 1) {  // element L of inv_K_G z (the matrix is symmetric: its row L), and its term of z' inv_K_G z
        T t = sX[L_ * (L_ + 1) + ln] * sz[ln];
        if (ln == 0) t += sX[L_ * (L_ + 1) + L_] * sz[L_];
        t = wave_sum(t);
        if (ln == 0) { sPz[L_] = t; red[1] = sz[L_] * t; }
      }
      block_sync_lds<TPB>();
      d = a.lam + (red[0] + red[1]);
    } else {
    for (int i = tid; i < p; i += TPB) {
      T acc = T(0);
#pragma unroll
      for (int j = 0; j < p; ++j) acc += sX[j * p + i] * sz[j];
      sPz[i] = acc;
    }
    block_sync_lds<TPB>();
    T part = T(0);
    for (int i = tid; i < p; i += TPB) part += sz[i] * sPz[i];
    d = a.lam + block_sum<T, TPB, true>(part, red);
    }
    const T dinv = T(1) / d;
    const T linv = T(1) / a.lam;
    T* Pw = a.P + (size_t)b * a.strideP;
    for_strided<TPB, PP_>(tid, p * p, [&](int e, int) {
      const int i = e / p, j = e - i * p;
      Pw[e] = (sX[e] - (sPz[i] * sPz[j]) * dinv) * linv;
    });
    block_sync_lds<TPB>();  // everyone is done with inv_K_G in the region
    KTRACE(3);
    // ---- [A B]: innovation y - K z and K <- (K - K z g') / lam + y g'   (Koopman_update.m:270-274; see the general block)
#pragma unroll
    for (int i = 0; i < NKR; ++i) { const int e = tid + i * TPB; if (e < LP_) sK[e] = fu ? T(0) : kr[i]; }
    block_sync_lds<TPB>();
    if constexpr (SPLIT4) {
      constexpr int CH = (L_ + 4) / 4;
      T acc = T(0);
#pragma unroll
      for (int jj = 0; jj < CH; ++jj) { const int j = wv * CH + jj; if (j <= L_) acc += sK[ln * (L_ + 1) + j] * sz[j]; }
      if (wv) pt[(wv - 1) * (L_ + 1) + ln] = acc;
      block_sync_lds<TPB>();
      if (wv == 0) {
        const T kz = (acc + pt[ln]) + (pt[(L_ + 1) + ln] + pt[2 * (L_ + 1) + ln]);
        sE[ln] = (sy[ln] - kz) * linv + sy[ln] * (T(1) - linv);
      }
    } else {
    for (int r = tid; r < L; r += TPB) {
      T acc = sy[r];
#pragma unroll
      for (int j = 0; j < p; ++j) acc -= sK[r * p + j] * sz[j];
      sE[r] = acc * linv + sy[r] * (T(1) - linv);
    }
    }
    block_sync_lds<TPB>();
    for_strided<TPB, LP_>(tid, L * p, [&](int e, int) {
      const int r = e / p, j = e - r * p;
      const T v = tfma(sE[r], sPz[j] * dinv, sK[e] * linv);
      sK[e] = v;
      Kg[e] = v;
    });
    block_sync_lds<TPB>();
    KTRACE(4);
   } else if (sv.phases & PH_CONDENSE) {
    const T* Kg = a.K + (size_t)b * a.strideK;
    for (int e = tid; e < L * p; e += TPB) sK[e] = Kg[e];
    const T* Cg = a.C + (size_t)b * a.strideC;
    for (int e = tid; e < n * L; e += TPB) sC[e] = Cg[e];
    if (sv.psi_in_regs) { if (tid < L) sy[tid] = sv.psi_now_v; }
    else for (int i = tid; i < L; i += TPB) sy[i] = sv.psi_now[i * a.pn_sl + b * a.pn_sb];
    block_sync<TPB>();
   }
  } else
  if (sv.phases & PH_RLS) {
    const T* Pg = a.P + (size_t)b * a.strideP;
    T* Kg = a.K + (size_t)b * a.strideK;
    // Every HBM request of the phase is issued before the first wait, in the order the data is needed (vector memory
    // returns in order): the step's small inputs, P and K, then -- static sizes -- bar_Q and C, which wait in registers
    // until the first half of the update is done with the LDS region they go to.  All of them are UNCONDITIONAL loads
    // with clamped indices: a load that is skipped on some lanes and replaced by a zero there makes the compiler wait
    // for every outstanding load before it may write that zero (the phase used to start with three serialised memory
    // round trips because of this), and costs a masked region of four scalar instructions per load.
    const int il = tid < L ? tid : L - 1, in = tid < n ? tid : n - 1;  // (L <= 64 <= TPB: one element per thread)
    const T zp = sv.psi_in_regs ? sv.psi_prev_v : sv.psi_prev[il * a.pp_sl + b * a.pp_sb];
    const T yp = sv.psi_in_regs ? sv.psi_now_v : sv.psi_now[il * a.pn_sl + b * a.pn_sb];
    const T xp = a.x_now[(size_t)in * B + b];
    constexpr int PP_ = L_ > 0 ? (L_ + 1) * (L_ + 1) : 0, LP_ = L_ > 0 ? L_ * (L_ + 1) : 0, LL_ = L_ * L_;
    constexpr bool STATE_REGS = L_ > 0 && (PP_ + TPB - 1) / TPB <= 17;
    T pr[STATE_REGS ? (PP_ + TPB - 1) / TPB : 1], kr[STATE_REGS ? (LP_ + TPB - 1) / TPB : 1];
    if constexpr (STATE_REGS) {
      for_strided<TPB, PP_>(tid, p * p, [&](int e, int i) { pr[i] = Pg[e]; });
      for_strided<TPB, LP_>(tid, L * p, [&](int e, int i) { kr[i] = Kg[e]; });
    }
    constexpr int QPRE = (L_ > 0) ? (L_ * L_ + TPB - 1) / TPB : 0;
    constexpr bool PREFETCH = (L_ > 0) && (QPRE <= 17);
    T qpre[PREFETCH ? QPRE : 1], cpre[PREFETCH ? 2 : 1];
    if constexpr (PREFETCH) {
      if (out_cx) {
        const T* Qg0 = a.Qb + (size_t)b * a.strideQ;
        const T* Cg0 = a.C + (size_t)b * a.strideC;
#pragma unroll
        for (int i = 0; i < QPRE; ++i) {
          const int e = tid + i * TPB;
          qpre[i] = Qg0[e < L * L ? e : L * L - 1];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int e = tid + i * TPB;
          cpre[i] = Cg0[e < n * L ? e : 0];  // (first update: replaced by zero when it goes to LDS)
        }
      }
    }
    if (tid < L) { sz[tid] = zp; sy[tid] = yp; }
    if (tid == 0) sz[L] = up;
    if (tid < n) sx[tid] = xp;
    if constexpr (STATE_REGS) {
      const bool fu = sv.first_update != 0;
      for_strided<TPB, PP_>(tid, p * p, [&](int e, int i) { sX[e] = pr[i]; });
      for_strided<TPB, LP_>(tid, L * p, [&](int e, int i) { sK[e] = fu ? T(0) : kr[i]; });
    } else {
      for (int e = tid; e < p * p; e += TPB) sX[e] = Pg[e];
      if (sv.first_update) {
        for (int e = tid; e < L * p; e += TPB) sK[e] = T(0);
      } else {
        for (int e = tid; e < L * p; e += TPB) sK[e] = Kg[e];
      }
    }
    block_sync<TPB>();
    KTRACE(1);

    // Pz (P symmetric: column walk is conflict-free in LDS)
    for (int i = tid; i < p; i += TPB) {
      T acc = T(0);
#pragma unroll
      for (int j = 0; j < p; ++j) acc += sX[j * p + i] * sz[j];
      sPz[i] = acc;
    }
    block_sync<TPB>();
    T part = T(0);
    for (int i = tid; i < p; i += TPB) part += sz[i] * sPz[i];
    const T d = a.lam + block_sum<T, TPB>(part, red);
    const T dinv = T(1) / d;
    const T linv = T(1) / a.lam;
    KTRACE(2);

    // P <- (P - Pz Pz' / d) / lam                                   duffing.py:931-932
    T* Pw = a.P + (size_t)b * a.strideP;
    for_strided<TPB, PP_>(tid, p * p, [&](int e, int) {
      const int i = e / p, j = e - i * p;
      Pw[e] = (sX[e] - (sPz[i] * sPz[j]) * dinv) * linv;
    });
    // innovation  y - K z.  The reference keeps K_A = sum y z' undiscounted and discounts inv_K_G only
    // (Koopman_update.m:270-274); with K = K_A inv_K_G that is  K <- (K - K z g') / lam + y g',  g = Pz / d, i.e.
    // K / lam + (e / lam + y (1 - 1/lam)) g'  with e = y - K z  (lam = 1: K + e g', duffing.py:927-938)
    for (int r = tid; r < L; r += TPB) {
      T acc = sy[r];
#pragma unroll
      for (int j = 0; j < p; ++j) acc -= sK[r * p + j] * sz[j];
      sE[r] = acc * linv + sy[r] * (T(1) - linv);
    }
    block_sync<TPB>();
    for_strided<TPB, LP_>(tid, L * p, [&](int e, int) {
      const int r = e / p, j = e - r * p;
      const T v = tfma(sE[r], sPz[j] * dinv, sK[e] * linv);  // (one rounding, as K + e g' always was at lam = 1)
      sK[e] = v;
      Kg[e] = v;
    });

    KTRACE(3);
    if (out_cx) {
      // ---- C = bar_X bar_Q, target x_{k+1}, regressor psi(x_k)      duffing.py:943-953
      const T* Qg = a.Qb + (size_t)b * a.strideQ;
      T* Cg = a.C + (size_t)b * a.strideC;
      block_sync<TPB>();  // everyone is done with P in sX and with sE / sPz
      if (PREFETCH && n * L <= 2 * TPB) {
        if constexpr (PREFETCH) {
#pragma unroll
          for (int i = 0; i < QPRE; ++i) {
            const int e = tid + i * TPB;
            if (e < L * L) sX[e] = qpre[i];
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int e = tid + i * TPB;
            if (e < n * L) sC[e] = sv.first_update ? T(0) : cpre[i];
          }
        }
      } else {
        for (int e = tid; e < L * L; e += TPB) sX[e] = Qg[e];
        if (sv.first_update) {
          for (int e = tid; e < n * L; e += TPB) sC[e] = T(0);
        } else {
          for (int e = tid; e < n * L; e += TPB) sC[e] = Cg[e];
        }
      }
      block_sync<TPB>();
      for (int i = tid; i < L; i += TPB) {
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < L; ++j) acc += sX[j * L + i] * sz[j];
        sPz[i] = acc;
      }
      for (int r = tid; r < n; r += TPB) {
        T acc = sx[r];
        for (int j = 0; j < L; ++j) acc -= sC[r * L + j] * sz[j];
        sE[r] = acc;
      }
      block_sync<TPB>();
      T part2 = T(0);
      for (int i = tid; i < L; i += TPB) part2 += sz[i] * sPz[i];
      // (bar_Q carries no forgetting factor: duffing.py:947-951 is the only form of this estimator the reference has)
      const T dc = T(1) + block_sum<T, TPB>(part2, red);
      const T dcinv = T(1) / dc;
      T* Qw = a.Qb + (size_t)b * a.strideQ;
      for_strided<TPB, LL_>(tid, L * L, [&](int e, int) {
        const int i = e / L, j = e - i * L;
        Qw[e] = sX[e] - (sPz[i] * sPz[j]) * dcinv;
      });
      for (int e = tid; e < n * L; e += TPB) {
        const int r = e / L, j = e - r * L;
        const T v = (a.c_skip_first && sv.first_update) ? T(0) : sC[e] + sE[r] * (sPz[j] * dcinv);
        sC[e] = v;
        Cg[e] = v;
      }
    }
    block_sync<TPB>();
    KTRACE(4);
  } else if (sv.phases & PH_CONDENSE) {
    const T* Kg = a.K + (size_t)b * a.strideK;
    for (int e = tid; e < L * p; e += TPB) sK[e] = Kg[e];
    if (out_cx) {
      const T* Cg = a.C + (size_t)b * a.strideC;
      for (int e = tid; e < n * L; e += TPB) sC[e] = Cg[e];
    }
    if (sv.psi_in_regs) { if (tid < L) sy[tid] = sv.psi_now_v; }
    else for (int i = tid; i < L; i += TPB) sy[i] = sv.psi_now[i * a.pn_sl + b * a.pn_sb];
    block_sync<TPB>();
  }

  // =====================================================================================
  // phase 2: condensed QP  H = Qw Phi'Phi + Rw I,  f = 2 Qw Phi'(Gamma psi - r)
  // =====================================================================================
  if (sv.phases & PH_CONDENSE) {
    const T* ref = a.ref + (a.ref_per_traj ? (size_t)b * q * N : 0);
    const bool cx = out_cx;
    for (int i = tid; i < L; i += TPB) {
      sV[i] = sK[i * p + L];  // v_0 = B
      sW[i] = sy[i];          // w_0 = psi(x_k)
    }
    if constexpr (REFN > 0) {  // sEr[k][r] starts as -r[r][k]
#pragma unroll
      for (int i = 0; i < REFN; ++i) {
        const int e = tid + i * TPB;
        if (e < q * N) sEr[e] = -refp[i];
      }
    } else {
      for (int e = tid; e < q * N; e += TPB) {
        const int k = e / q, r = e - k * q;
        sEr[e] = -ref[r * N + k];
      }
    }
    block_sync_lds<TPB>();
    KTRACE(5);
    // v_{j+1} = A v_j, w_{j+1} = A w_j;  g_j = Co v_j;  e_j = Co w_j - r_{j-1}
    if constexpr (L_ > 0 && TPB == 64 && (L_ + Q_ <= 32)) {
      // Static path: lanes 0-31 run the v-chain, lanes 32-63 the w-chain.  Lane t of a half keeps row t
      // of the stacked matrix [A; Co] in REGISTERS for the whole recursion; per step it only reads
      // the current vector (broadcast LDS reads) -- the matrix is never re-read from LDS.
      const int half = tid >> 5, t = tid & 31;
      const int nco = cx ? q : 0;
      const bool isA = t < L, isC = (t >= L) && (t < L + nco);
      T row[L_];
#pragma unroll
      for (int l = 0; l < L_; ++l) row[l] = isA ? sK[t * p + l] : (isC ? sC[(a.cy0 + t - L) * L + l] : T(0));
      // delta-u form (Tank_System.m:110-113): the augmented state [x; u_prev] propagates as
      // x+ = A x + B s with s = 1 on the v-chain (B~ = [B; 1]) and s = u_prev on the w-chain
      const T bs = (a.du_mode && isA) ? sK[t * p + L] * (half ? up : T(1)) : T(0);
      if (!cx && half == 0 && isA) sG[t] = sV[t];  // y = lifted state: g_0 = B
      if constexpr (sizeof(T) == 8) {
        // float64: the vector stays in the lanes (DPP row broadcast + one cross-row swap per step); LDS only
        // receives the outputs g_j, e_j.  11 -> 4 us for the 21 steps of cfg2 (tools/trace_phases.py): the
        // broadcast LDS reads of the version below kept the LDS pipe of the CU busy for the whole recursion.
        double v0, v1;
        half_gather(isA ? (double)(half ? sW[t] : sV[t]) : 0.0, v0, v1);
        double* const ocol = half ? sEr - q + (t - L) : sG + (t - L);  // output column of an isC lane
#pragma unroll
        for (int j = 0; j <= N_; ++j) {  // (unrolled: the step-dependent store conditions and LDS offsets fold)
          double ac4[4] = {0.0, 0.0, 0.0, 0.0};
          chain_dot<L_>(ac4, v0, v1, row);
          const double acc = ((ac4[0] + ac4[1]) + (ac4[2] + ac4[3])) + (isA ? bs : 0.0);
          if (cx) {
            // g_j = Co v_j (half 0, j < N) and e_j = Co w_j - r_{j-1} (half 1, j >= 1; sEr holds -r) in ONE masked
            // region: both are "element j of this lane's output column"
            if (isC && (half ? j >= 1 : j < N_)) {
              const double old = ocol[j * q];
              ocol[j * q] = half ? acc + old : acc;
            }
          } else if (half == 0) {
            if (isA && j + 1 < N) sG[(j + 1) * q + t] = acc;  // g_{j+1} = v_{j+1}
          } else {
            if (isA && j < N) sEr[j * q + t] += acc;          // e_{j+1} = w_{j+1} - r_j
          }
          half_gather(acc, v0, v1);  // v_{j+1} / w_{j+1} (lanes >= L are never read back)
        }
        block_sync<TPB>();
      } else {
      int cur = 0;
      for (int j = 0; j <= N; ++j) {
        const T* vec = (half ? sW : sV) + cur * L;
        // four independent FMA chains: a dependent f64 FMA costs ~40 cycles on gfx950 (tools/ubench), but with
        // four waves per SIMD the kernel is issue-bound, so products and adds stay fused (20 FMA + 3 adds)
        T ac4[4] = {T(0), T(0), T(0), T(0)};
        if constexpr ((L_ & 1) == 0) {
          // 16-byte broadcast reads (ds_read_b128: half the LDS cycles of the ds_read2_b64 the compiler
          // picks when it cannot prove the alignment); all LDS offsets are even in the static layout
          typedef T T2 __attribute__((ext_vector_type(2)));
          const T2* v2 = reinterpret_cast<const T2*>(__builtin_assume_aligned(vec, 2 * sizeof(T)));
#pragma unroll
          for (int l = 0; l < L_ / 2; ++l) {
            const T2 x2 = v2[l];
            ac4[(2 * l) & 3] += row[2 * l] * x2.x;
            ac4[(2 * l + 1) & 3] += row[2 * l + 1] * x2.y;
          }
        } else {
#pragma unroll
          for (int l = 0; l < L_; ++l) ac4[l & 3] += row[l] * vec[l];
        }
        T pr[1] = {(ac4[0] + ac4[1]) + (ac4[2] + ac4[3])};
        const T acc = pr[0] + (isA ? bs : T(0));
        if (half == 0) {
          if (isA && j < N) {
            sV[(cur ^ 1) * L + t] = acc;                      // v_{j+1}
            if (!cx && j + 1 < N) sG[(j + 1) * q + t] = acc;  // g_{j+1} = v_{j+1}
          }
          if (isC && j < N) sG[j * q + (t - L)] = acc;        // g_j = Co v_j
        } else {
          if (isA && j < N) {
            sW[(cur ^ 1) * L + t] = acc;                      // w_{j+1}
            if (!cx) sEr[j * q + t] += acc;                   // e_{j+1} = w_{j+1} - r_j
          }
          if (isC && j >= 1) sEr[(j - 1) * q + (t - L)] += acc;  // e_j = Co w_j - r_{j-1}
        }
        block_sync<TPB>();
        cur ^= 1;
      }
      }
    } else if constexpr (L_ > 0 && TPB == 64 && (L_ + Q_ <= 64) && ((L_ & 1) == 0)) {
      // Static path for 32 < L + q <= 64 (cfg4 sizes): lane t keeps row t of [A; Co] in registers and
      // advances BOTH recursions (v and w) each step.
      const int t = tid;
      const int nco = cx ? q : 0;
      const bool isA = t < L, isC = (t >= L) && (t < L + nco);
      T row[L_];
#pragma unroll
      for (int l = 0; l < L_; ++l) row[l] = isA ? sK[t * p + l] : (isC ? sC[(a.cy0 + t - L) * L + l] : T(0));
      const T bcol = (a.du_mode && isA) ? sK[t * p + L] : T(0);
      const T upv = a.du_mode ? up : T(0);
      if (!cx && isA) sG[t] = sV[t];  // y = lifted state: g_0 = B
      typedef T T2 __attribute__((ext_vector_type(2)));
      int cur = 0;
      for (int j = 0; j <= N; ++j) {
        const T2* v2 = reinterpret_cast<const T2*>(__builtin_assume_aligned(sV + cur * L, 2 * sizeof(T)));
        const T2* w2 = reinterpret_cast<const T2*>(__builtin_assume_aligned(sW + cur * L, 2 * sizeof(T)));
        T av[2] = {T(0), T(0)}, aw[2] = {T(0), T(0)};
#pragma unroll
        for (int l = 0; l < L_ / 2; ++l) {
          const T2 xv = v2[l], xw = w2[l];
          av[0] += row[2 * l] * xv.x;
          av[1] += row[2 * l + 1] * xv.y;
          aw[0] += row[2 * l] * xw.x;
          aw[1] += row[2 * l + 1] * xw.y;
        }
        const T accv = av[0] + av[1] + bcol, accw = aw[0] + aw[1] + bcol * upv;
        if (isA && j < N) {
          sV[(cur ^ 1) * L + t] = accv;                      // v_{j+1}
          sW[(cur ^ 1) * L + t] = accw;                      // w_{j+1}
          if (!cx) {
            if (j + 1 < N) sG[(j + 1) * q + t] = accv;       // g_{j+1} = v_{j+1}
            sEr[j * q + t] += accw;                          // e_{j+1} = w_{j+1} - r_j
          }
        }
        if (isC) {
          if (j < N) sG[j * q + (t - L)] = accv;             // g_j = Co v_j
          if (j >= 1) sEr[(j - 1) * q + (t - L)] += accw;    // e_j = Co w_j - r_{j-1}
        }
        block_sync<TPB>();
        cur ^= 1;
      }
    } else if constexpr (ONE_REGION && sizeof(T) == 8 && L_ == 64) {
      // Round 3 -- four-wave trajectories, y = C x (cfg5: L = 64), float64: threads 0-127 run the v-chain, 128-255 the w-chain; a
      // 16-lane DPP row holds HALF rows of A for 16 consecutive rows (rows 16 m .. 16 m + 15, columns 32 hh .. 32 hh + 31 with
      // hh = the row's parity within its wave), 32 registers a lane.  A step multiplies with v_fmac_f64_dpp ... row_newbcast: the
      // 32 vector elements a row needs are two registers of its lanes (element 32 hh + l and 32 hh + 16 + l in lane l: two 8-byte LDS
      // reads per lane and step).  The version below read them as 16 broadcast ds_read_b128 per lane and step -- 64 KB of LDS
      // traffic per trajectory and step, which is what the recursion's 0.58 us per step was: with four trajectories on a CU the
      // LDS pipe, not the multiply-adds, set its pace.  The two halves of a row meet through one v_permlane16_swap per dword.
      const int chain = tid >> 7, g16 = (tid & 127) >> 4, l16 = tid & 15;
      const int hh = g16 & 1, rr = (g16 >> 1) * 16 + l16;
      const int ln = tid & 63, corow = (tid >> 6) & 1;
      double row[32];
#pragma unroll
      for (int l = 0; l < 32; ++l) row[l] = sK[rr * p + hh * 32 + l];
      const double co = (ln < L && corow < q) ? sC[(a.cy0 + corow) * L + ln] : 0.0;
      const double bs = a.du_mode ? sK[rr * p + L] * (chain ? up : 1.0) : 0.0;
      block_sync_lds<TPB>();  // (C may sit where g is written from now on)
      int cur = 0;
      for (int j0 = 0; j0 <= N; j0 += 4) {  // (the outputs of four steps are reduced together: wave_sum4)
        double pc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = j0 + u;
          if (j <= N) {
            double* const vb = (chain ? sW : sV) + cur * L;
            const double x0 = vb[hh * 32 + l16], x1 = vb[hh * 32 + 16 + l16];
            pc[u] = co * vb[ln < L ? ln : 0];
            double ac4[4] = {0.0, 0.0, 0.0, 0.0};
            chain_dot<32>(ac4, x0, x1, row);
            const double part = (ac4[0] + ac4[1]) + (ac4[2] + ac4[3]);
            double pa, pb;
            half_gather(part, pa, pb);  // the two 16-lane rows of a pair: this row's other half sits in the neighbouring one
            const double acc = (pa + pb) + bs;
            if (hh == 0 && j < N) (chain ? sW : sV)[(cur ^ 1) * L + rr] = acc;  // v_{j+1} / w_{j+1}
            block_sync_lds<TPB>();
            cur ^= 1;
          }
        }
        const double g4 = wave_sum4(pc[0], pc[1], pc[2], pc[3], ln);
        const int j = j0 + ln;  // lane u < 4 of the wave holds the output of step j0 + u
        if (ln < 4 && corow < q && j <= N) {
          if (chain == 0) { if (j < N) sG[j * q + corow] = g4; }       // g_j = Co v_j
          else if (j >= 1) sEr[(j - 1) * q + corow] += g4;             // e_j = Co w_j - r_{j-1}
        }
      }
      block_sync_lds<TPB>();  // (the last outputs)
    } else if constexpr (ONE_REGION) {
      // Four-wave trajectories, y = C x (cfg5 sizes, L = 64): threads 0-127 run the v-chain, 128-255 the w-chain; a PAIR of
      // threads keeps one row of A in registers for the whole recursion, half a row each (32 of the 64 columns: every
      // thread multiplies, and a row costs 64 registers instead of the 128 of one-row-per-thread, which left two of the
      // four waves idle); the two output rows of C_o are one multiply per thread and a wave sum: wave w = (chain, row).
      constexpr int HL = L_ / 2;
      const int chain = tid >> 7, rr = (tid & 127) >> 1, hh = tid & 1;
      const int ln = tid & 63, corow = (tid >> 6) & 1;
      const bool isA = rr < L;
      T row[HL];
#pragma unroll
      for (int l = 0; l < HL; ++l) row[l] = isA ? sK[rr * p + hh * HL + l] : T(0);
      const T co = (ln < L && corow < q) ? sC[(a.cy0 + corow) * L + ln] : T(0);
      const T bs = (a.du_mode && isA) ? sK[rr * p + L] * (chain ? up : T(1)) : T(0);
      block_sync<TPB>();  // (C may sit where g is written from now on)
      typedef T T2 __attribute__((ext_vector_type(2)));
      int cur = 0;
      for (int j0 = 0; j0 <= N; j0 += 4) {  // (the outputs of four steps are reduced together: wave_sum4)
        T pc[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = j0 + u;
          if (j <= N) {
            T* const vb = (chain ? sW : sV) + cur * L;
            const T2* v2 = reinterpret_cast<const T2*>(__builtin_assume_aligned(vb + hh * HL, 2 * sizeof(T)));
            T ac4[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
            for (int l = 0; l < HL / 2; ++l) {
              const T2 x2 = v2[l];
              ac4[(2 * l) & 3] += row[2 * l] * x2.x;
              ac4[(2 * l + 1) & 3] += row[2 * l + 1] * x2.y;
            }
            T acc = (ac4[0] + ac4[1]) + (ac4[2] + ac4[3]);
            acc += dpp_q(acc, 0);  // the other half of the row (the neighbouring lane)
            acc += isA ? bs : T(0);
            pc[u] = co * vb[ln < L ? ln : 0];
            if (hh == 0 && isA && j < N) (chain ? sW : sV)[(cur ^ 1) * L + rr] = acc;  // v_{j+1} / w_{j+1}
            block_sync<TPB>();
            cur ^= 1;
          }
        }
        const T g4 = (T)wave_sum4((double)pc[0], (double)pc[1], (double)pc[2], (double)pc[3], ln);
        const int j = j0 + ln;  // lane u < 4 of the wave holds the output of step j0 + u
        if (ln < 4 && corow < q && j <= N) {
          if (chain == 0) { if (j < N) sG[j * q + corow] = g4; }       // g_j = Co v_j
          else if (j >= 1) sEr[(j - 1) * q + corow] += g4;             // e_j = Co w_j - r_{j-1}
        }
      }
      block_sync<TPB>();  // (the last outputs)
    } else if constexpr (L_ > 0 && TPB == 256 && (L_ + Q_ <= 128) && ((L_ & 1) == 0)) {
      // Static path for four-wave trajectories (cfg5 sizes, L = 64): waves 0-1 run the v-chain, waves 2-3 the w-chain;
      // thread t of a half keeps row t of [A; Co] in REGISTERS for the whole recursion (the generic path re-reads the
      // 32 KB matrix from LDS at each of the N + 1 steps: LDS-bandwidth bound), the current vectors are broadcast
      // 16-byte LDS reads.
      const int half = tid >> 7, t = tid & 127;
      const int nco = cx ? q : 0;
      const bool isA = t < L, isC = (t >= L) && (t < L + nco);
      T row[L_];
#pragma unroll
      for (int l = 0; l < L_; ++l) row[l] = isA ? sK[t * p + l] : (isC ? sC[(a.cy0 + t - L) * L + l] : T(0));
      const T bs = (a.du_mode && isA) ? sK[t * p + L] * (half ? up : T(1)) : T(0);
      if (!cx && half == 0 && isA) sG[t] = sV[t];  // y = lifted state: g_0 = B
      typedef T T2 __attribute__((ext_vector_type(2)));
      int cur = 0;
      for (int j = 0; j <= N; ++j) {
        const T2* v2 = reinterpret_cast<const T2*>(__builtin_assume_aligned((half ? sW : sV) + cur * L, 2 * sizeof(T)));
        T ac4[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
        for (int l = 0; l < L_ / 2; ++l) {
          const T2 x2 = v2[l];
          ac4[(2 * l) & 3] += row[2 * l] * x2.x;
          ac4[(2 * l + 1) & 3] += row[2 * l + 1] * x2.y;
        }
        const T acc = ((ac4[0] + ac4[1]) + (ac4[2] + ac4[3])) + (isA ? bs : T(0));
        if (half == 0) {
          if (isA && j < N) {
            sV[(cur ^ 1) * L + t] = acc;                      // v_{j+1}
            if (!cx && j + 1 < N) sG[(j + 1) * q + t] = acc;  // g_{j+1} = v_{j+1}
          }
          if (isC && j < N) sG[j * q + (t - L)] = acc;        // g_j = Co v_j
        } else {
          if (isA && j < N) {
            sW[(cur ^ 1) * L + t] = acc;                      // w_{j+1}
            if (!cx) sEr[j * q + t] += acc;                   // e_{j+1} = w_{j+1} - r_j
          }
          if (isC && j >= 1) sEr[(j - 1) * q + (t - L)] += acc;  // e_j = Co w_j - r_{j-1}
        }
        block_sync<TPB>();
        cur ^= 1;
      }
    } else {
    int cur = 0;
      const int ntask = 2 * L + 2 * q;
      for (int j = 0; j <= N; ++j) {
        const T* v = sV + cur * L;
        const T* w = sW + cur * L;
        T* vn = sV + (cur ^ 1) * L;
        T* wn = sW + (cur ^ 1) * L;
        for (int t = tid; t < ntask; t += TPB) {
          if (t < L) {
            if (j < N) {
              T acc = T(0);
#pragma unroll
              for (int l = 0; l < L; ++l) acc += sK[t * p + l] * v[l];
              vn[t] = a.du_mode ? acc + sK[t * p + L] : acc;  // delta-u: x+ = A x + B s, s = 1
            }
          } else if (t < 2 * L) {
            if (j < N) {
              const int r = t - L;
              T acc = T(0);
#pragma unroll
              for (int l = 0; l < L; ++l) acc += sK[r * p + l] * w[l];
              wn[r] = a.du_mode ? acc + sK[r * p + L] * a.u_prev[b] : acc;  // s = u_prev
            }
          } else if (t < 2 * L + q) {
            if (j < N) {
              const int r = t - 2 * L;
              T g;
              if (cx) {
                g = T(0);
#pragma unroll
                for (int l = 0; l < L; ++l) g += sC[(a.cy0 + r) * L + l] * v[l];
              } else {
                g = v[r];
              }
              sG[j * q + r] = g;
            }
          } else {
            if (j >= 1) {
              const int r = t - 2 * L - q;
              T y;
              if (cx) {
                y = T(0);
#pragma unroll
                for (int l = 0; l < L; ++l) y += sC[(a.cy0 + r) * L + l] * w[l];
              } else {
                y = w[r];
              }
              sEr[(j - 1) * q + r] += y;
            }
          }
        }
        block_sync<TPB>();
        cur ^= 1;
      }
    }
    // H[a][b] = Qw * S(b-a, N-1-b) (+Rw on the diagonal),  S(d,t) = sum_{s<=t} g_{s+d}.g_s
    KTRACE(6);
    if constexpr (N_ > 0 && N_ <= 40 && TPB == 64) {
      // the previous minimiser of this lane's variable (register solver: variable (lane >> 3) + 8 (lane & 7)) is
      // requested now -- asked for at the start of the solve, it was a memory round trip nobody could hide
      if ((sv.phases & PH_QP) && a.x_warm) {
        const int mv = (tid >> 3) + 8 * (tid & 7);
        xw_pre = a.x_warm[(size_t)(mv < N_ ? mv : 0) * a.B + b];
      }
    }
    if constexpr (N_ > 0 && N_ <= 64 && TPB == 256) {
      // (four-wave solver, round 4: likewise, and the header of the carried tableau -- two dependent round trips at its start)
      if (sv.phases & PH_QP) {
        const int mv = (tid >> 4) + 16 * (tid & 15);
        if (a.x_warm) xw_pre = a.x_warm[(size_t)(mv < N_ ? mv : 0) * a.B + b];
        if (a.qp_carry) {
          const int32_t* const cs = a.qp_carry_set + 4 * (size_t)b;
          cs_pre[0] = cs[0]; cs_pre[1] = cs[1]; cs_pre[2] = cs[2];
        }
      }
    }
    if constexpr (TPB == 64 && N_ > 0 && N_ <= 32 && Q_ > 0) {
      // One pass for both: lane d < N walks diagonal d of H, lane 32 + a accumulates f[a].  Both are sums of
      // g_t . w_{t+idx} with w = g (H) or e (f), so the two halves of the wave share one instruction stream.
      const int hf = tid >> 5, idx = tid & 31;
      const T* const wb = (hf ? sEr : sG) + idx * Q_;
      const bool on = idx < N_;
      const bool vec = ((Q_ & 1) == 0) && ((((int)(sG - sm)) & 1) == 0) && ((((int)(sEr - sm)) & 1) == 0);
      T acc = T(0);
      // H lanes store H(aa, bb) and H(bb, aa) with bb = N-1-t, aa = bb - idx: element offsets bb (N+1) - idx N and
      // bb (N+1) - idx; the f lanes take part in the same stores with a scratch slot as target (no second mask)
      T* const h1 = hf ? red + 14 : sH - idx * N_;
      T* const h2 = hf ? red + 15 : sH - idx;
      const int hstep = hf ? 0 : N_ + 1;
      const T rdiag = (idx == 0 && !hf) ? a.Rw : T(0);
      auto pass = [&](auto dotq) {
        auto term = [&](int t) {
          if (on && t + idx < N_) {
            acc += dotq(sG + t * Q_, wb + t * Q_);
            const T hv = a.Qw * acc + rdiag;
            h1[(N_ - 1 - t) * hstep] = hv;
            h2[(N_ - 1 - t) * hstep] = hv;
          }
        };
        if constexpr (Q_ * N_ > 128) {  // (y = psi with a long horizon: a loop -- unrolled, the N q loads of a lane were hoisted and spilled; L = 8, N = 30: 103 -> 55 registers, +7 ... 12 %)
#pragma unroll 1
          for (int t = 0; t < N_; ++t) term(t);
        } else {
#pragma unroll
          for (int t = 0; t < N_; ++t) term(t);
        }
      };
      if (vec) {  // 16-byte LDS reads (the layout is even for the shipped dimension sets)
        typedef T T2 __attribute__((ext_vector_type(2)));
        pass([](const T* x, const T* y) {
          const T2* x2 = reinterpret_cast<const T2*>(__builtin_assume_aligned(x, 2 * sizeof(T)));
          const T2* y2 = reinterpret_cast<const T2*>(__builtin_assume_aligned(y, 2 * sizeof(T)));
          T s0 = T(0);
#pragma unroll
          for (int r = 0; r < Q_ / 2; ++r) { const T2 u = x2[r], w = y2[r]; s0 += u.x * w.x; s0 += u.y * w.y; }
          return s0;
        });
      } else {
        pass([](const T* x, const T* y) {
          T s0 = T(0);
#pragma unroll
          for (int r = 0; r < Q_; ++r) s0 += x[r] * y[r];
          return s0;
        });
      }
      if (hf && on) sf[idx] = T(2) * a.Qw * acc;
    } else if constexpr (ONE_REGION && TPB == 256 && N_ > 0 && Q_ > 0 && (TPB / N_) >= 2 && 5 * L_ >= (TPB / N_) * N_) {
      // Round 4 -- four-wave trajectories: a diagonal of H is a prefix sum of N - d terms g_{t+d}.g_t and an element of f a sum
      // of N - a terms g_t.e_{t+a}; N of the 256 threads used to walk them one after the other, 2 N dependent steps (9.4 of the
      // 85 us of a cfg5 trajectory-step).  Every sum is now cut into NCH = TPB / N chunks of CL terms, one thread each: the chunk
      // totals meet in LDS (where the recursion's vectors were) and a thread starts its prefix at the sum of the chunks before it.
      constexpr int NCH = TPB / N_, CL = (N_ + NCH - 1) / NCH;
      T* const part = va;  // NCH x N   (sy | sV | sW: dead since the recursion's last barrier)
      const int c = tid / N_, d = tid - c * N_, t0 = c * CL;
      const bool on = c < NCH;
      T prod[CL];
      T tot = T(0), ftot = T(0);
#pragma unroll
      for (int i = 0; i < CL; ++i) {
        const int t = t0 + i;
        T s0 = T(0), s1 = T(0);
        if (on && t + d < N_) {
#pragma unroll
          for (int r = 0; r < Q_; ++r) {
            const T gt = sG[t * Q_ + r];
            s0 += sG[(t + d) * Q_ + r] * gt;
            s1 += sEr[(t + d) * Q_ + r] * gt;
          }
        }
        prod[i] = s0;
        tot += s0;
        ftot += s1;
      }
      if (on) part[c * N_ + d] = tot;
      block_sync_lds<TPB>();
      T acc = T(0);
#pragma unroll
      for (int cc = 0; cc < NCH - 1; ++cc) acc += cc < c ? part[cc * N_ + d] : T(0);
      const T rdiag = d == 0 ? a.Rw : T(0);
#pragma unroll
      for (int i = 0; i < CL; ++i) {
        const int t = t0 + i;
        if (on && t + d < N_) {
          acc += prod[i];
          const int bb = N_ - 1 - t, aa = bb - d;
          const T hv = a.Qw * acc + rdiag;
          sH[aa * N_ + bb] = hv;
          sH[bb * N_ + aa] = hv;
        }
      }
      block_sync_lds<TPB>();  // the chunk totals of H have been read
      if (on) part[c * N_ + d] = ftot;
      block_sync_lds<TPB>();
      if (tid < N_) {
        T fs = T(0);
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc) fs += part[cc * N_ + tid];
        sf[tid] = T(2) * a.Qw * fs;
      }
    } else {
    for (int d = tid; d < N; d += TPB) {
        T acc = T(0);
#pragma unroll
        for (int t = 0; t < N; ++t) {
          if (t + d < N) {
            T s = T(0);
#pragma unroll
            for (int r = 0; r < q; ++r) s += sG[(t + d) * q + r] * sG[t * q + r];
            acc += s;
            const int bb = N - 1 - t, aa = bb - d;
            const T hv = a.Qw * acc + (d == 0 ? a.Rw : T(0));
            sH[aa * N + bb] = hv;
            sH[bb * N + aa] = hv;
          }
        }
      }
      // f on lanes 32.. so that it overlaps the H diagonals of lanes 0..N-1 when the wave has room
      {
        const int f0 = (TPB == 64 && N <= 32) ? 32 : 0;
        for (int aa = tid - f0; aa < N; aa += TPB) {
          if (aa < 0) continue;
          T acc = T(0);
#pragma unroll
          for (int t = 0; t < N; ++t)
            if (t + aa < N) {
#pragma unroll
              for (int r = 0; r < q; ++r) acc += sG[t * q + r] * sEr[(t + aa) * q + r];
            }
          sf[aa] = T(2) * a.Qw * acc;
        }
      }
    }
    if (a.Wterm) {
      // terminal block of Q_bar is PN instead of Qw I (Koopman_update.m:381); Wterm = PN - Qw I:
      //   H[a][b] += g_{N-1-a}' sym(W) g_{N-1-b},   f[a] += 2 g_{N-1-a}' W e_N
      block_sync<TPB>();
      const T* const Wt = a.Wterm + (a.wterm_per_traj ? (size_t)b * q * q : (size_t)0);
      for (int e = tid; e < N * N; e += TPB) {
        const int aa = e / N, bb = e - aa * N;
        const T* ga = sG + (N - 1 - aa) * q;
        const T* gb = sG + (N - 1 - bb) * q;
        T acc = T(0);
#pragma unroll 1
        for (int r = 0; r < q; ++r)
#pragma unroll 2
          for (int s2 = 0; s2 < q; ++s2) acc += ga[r] * (T(0.5) * (Wt[r * q + s2] + Wt[s2 * q + r])) * gb[s2];
        sH[e] += acc;
      }
      for (int aa = tid; aa < N; aa += TPB) {
        const T* ga = sG + (N - 1 - aa) * q;
        const T* eN = sEr + (N - 1) * q;
        T acc = T(0);
#pragma unroll 1
        for (int r = 0; r < q; ++r)
#pragma unroll 2
          for (int s2 = 0; s2 < q; ++s2) acc += ga[r] * Wt[r * q + s2] * eN[s2];
        sf[aa] += T(2) * acc;
      }
    }
    block_sync_lds<TPB>();
    KTRACE(7);
    if (a.H_out) {
      T* Hg = a.H_out + (size_t)b * N * N;
      for (int e = tid; e < N * N; e += TPB) Hg[e] = sH[e];
    }
    if (a.f_out) {
      T* fg = a.f_out + (size_t)b * N;
      for (int e = tid; e < N; e += TPB) fg[e] = sf[e];
    }
    if (a.c_out) {
      // the constant of the cost: c = Qw sum_j |e_j|^2 (+ e_N' W e_N with a terminal block), e_j = Co A^j psi - r_{j-1}
      // (sEr), so that u'Hu + f'u + c is costFunction(u) (duffing.py:540-581) and c + the QP's optimal value result.fun
      T part = T(0);
      for (int e = tid; e < q * N; e += TPB) part += sEr[e] * sEr[e];
      part *= a.Qw;
      if (a.Wterm) {
        const T* const Wt = a.Wterm + (a.wterm_per_traj ? (size_t)b * q * q : (size_t)0);
        const T* eN = sEr + (N - 1) * q;
        for (int e = tid; e < q * q; e += TPB) part += eN[e / q] * Wt[e] * eN[e % q];
      }
      const T csum = block_sum<T, TPB>(part, red);
      if (tid == 0) a.c_out[b] = csum;
    }
  } else if (sv.phases & PH_QP) {
    const T* Hg = a.H_in + (a.h_shared ? (size_t)0 : (size_t)b * N * N);
    if (!sv.h_global) for (int e = tid; e < N * N; e += TPB) sH[e] = Hg[e];
    if (a.F_in) {
      // shared-model mode: f_b = F psi_b + f0 (the per-trajectory part of the condensed QP)
      for (int i = tid; i < L; i += TPB) sy[i] = sv.psi_now[i * a.pn_sl + b * a.pn_sb];
      block_sync<TPB>();
      // (delta-u form: F has one more column, for u_prev -- the augmented state is [psi; u_prev], Tank_System.m:290)
      const int Lf = L + (a.du_mode ? 1 : 0);
      for (int e = tid; e < N; e += TPB) {
        T acc = a.f0_in[e];
        for (int l = 0; l < L; ++l) acc += a.F_in[e * Lf + l] * sy[l];
        if (a.du_mode) acc += a.F_in[e * Lf + L] * a.u_prev[b];
        sf[e] = acc;
      }
    } else {
      const T* fg = a.f_in + (size_t)b * N;
      for (int e = tid; e < N; e += TPB) sf[e] = fg[e];
    }
    block_sync<TPB>();
  }

  // =====================================================================================
  // phase 3: box QP  min u'Hu + f'u, lb <= u <= ub  -- projected Newton (Bertsekas 1982) on a
  // SWEPT tableau: T = sweep_F(2H) is kept in LDS, T_FF = -(2 H_FF)^-1, so the Newton step on the
  // free set F is one masked mat-vec and a change of the active set costs one O(N^2) symmetric
  // sweep per variable that enters or leaves F (no refactorisation).  Projected Armijo arc on the
  // true cost; termination on the componentwise KKT test evaluated with H itself, so rounding
  // in T only costs an extra (cheap) iteration.  Cold start at clip(0) as the reference
  // (duffing.py:634-635); the minimiser is unique, so the start only affects the work.
  // =====================================================================================
  if (sv.phases & PH_QP) {
    if constexpr (N_ > 0 && N_ <= 40 && TPB == 64) {
      if (!(sv.phases & PH_CONDENSE) && a.x_warm) {  // QP-only call: nothing has requested the warm start yet
        const int mv = (tid >> 3) + 8 * (tid & 7);
        xw_pre = a.x_warm[(size_t)(mv < N_ ? mv : 0) * a.B + b];
      }
      // register tableau; its rare "crawling" cases are finished by the active-set loop of the LDS solver
      if (qp_regs<T, N_, LOWREG, ASREG>(sv.h_global ? a.H_in : sH, sf, a, sv, b, red, qx, up, xw_pre)) {
        block_sync<TPB>();
        if constexpr (step_tableau_in_lds<TPB, N_, L_>() && !ONE_REGION) {
          qp_lds<T, TPB>(sH, sf, sM, qx, qxa, qg, red, a, sv, b, N, true);
        } else {
          // no tableau region in LDS: H moves to this trajectory's global scratch block (read-only from here on,
          // coalesced reads) and the tableau takes H's place in LDS, where the sweeps' read-modify-writes belong
          T* const Hg = a.qp_scratch + (size_t)b * N * N;
          for (int e = tid; e < N * N; e += TPB) Hg[e] = sH[e];
          __threadfence_block();
          block_sync<TPB>();
          qp_lds<T, TPB>(Hg, sf, sX, qx, qxa, qg, red, a, sv, b, N, true);
        }
      }
    } else if constexpr (N_ > 0 && N_ <= 64 && TPB == 256) {
      // four-wave register tableau; the workspace is the vector area behind qx / qxa / qg (dead set A of the phase)
      static_assert(L_ == 0 || 2 * (L_ + 1) + 6 * L_ + 1 + 2 * N_ * Q_ >= 3 * N_ + 324, "qp_regs256 workspace");
      static_assert(!ONE_REGION || (5 * L_ + 1 + 2 * N_ * Q_ >= 3 * N_ + 324 && 2 * (L_ + 1) + L_ <= 4 * L_), "qp_regs256 workspace / update vectors (one region)");
      if (!(sv.phases & PH_CONDENSE)) {  // QP-only call: nothing has requested these yet
        const int mv = (tid >> 4) + 16 * (tid & 15);
        if (a.x_warm) xw_pre = a.x_warm[(size_t)(mv < N_ ? mv : 0) * a.B + b];
        if (a.qp_carry) {
          const int32_t* const cs = a.qp_carry_set + 4 * (size_t)b;
          cs_pre[0] = cs[0]; cs_pre[1] = cs[1]; cs_pre[2] = cs[2];
        }
      }
      if (qp_regs256<T, N_, !ONE_REGION>(sH, sf, a, sv, b, red, qg + N, qx, up, xw_pre, cs_pre)) {
        block_sync<TPB>();
        if constexpr (step_tableau_in_lds<TPB, N_, L_>() && !ONE_REGION) {
          qp_lds<T, TPB>(sH, sf, sM, qx, qxa, qg, red, a, sv, b, N, true);
        } else {
          // no tableau region in LDS: H moves to this trajectory's global scratch block (read-only from here on,
          // coalesced reads) and the tableau takes H's place in LDS, where the sweeps' read-modify-writes belong
          T* const Hg = a.qp_scratch + (size_t)b * N * N;
          for (int e = tid; e < N * N; e += TPB) Hg[e] = sH[e];
          __threadfence_block();
          block_sync<TPB>();
          qp_lds<T, TPB>(Hg, sf, sX, qx, qxa, qg, red, a, sv, b, N, true);
        }
      }
    } else {
      qp_lds<T, TPB>(sH, sf, sM, qx, qxa, qg, red, a, sv, b, N, false);
    }
  }
}

}  // namespace kmpc
