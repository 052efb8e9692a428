// step_kernel.hip -- the per-trajectory hot path on gfx950: RLS-EDMD update -> condensed QP
// build -> box-QP solve (step_body, step_body.h), as a per-step kernel (one 64-lane wave per trajectory by default,
// four for large dimensions); the same step_body runs inside the fused roll-out kernel (all steps of the closed loop
// in one launch, 16 trajectories per workgroup, encoder on MFMA: rollout_kernel.hip).  Covariance / model / KKT
// tiles live in LDS, the persistent state streams HBM -> LDS -> HBM exactly once per step.
//
// Reference arithmetic restated here (file:line under the reference root):
//   RLS of [A B]  duffing.py:900, 927-938, 965-967 (lambda form Koopman_update.m:258-278)
//   RLS of C      duffing.py:943-953
//   condensed QP  Koopman_update.m:455-471, :213 with the Python weights duffing.py:580
//   solve         replaces optimize.minimize(..., bounds) duffing.py:857-861 /
//                 quadprog(2H, f, ..., lb, ub) Koopman_update.m:214 by an exact method
//
// LDS map (elements of T):  X[r1] : P -> bar_Q -> H      Y[r2] : K, C -> elimination matrix
//                           V     : vectors (RLS/condense set aliased with the QP set)
#include "step_body.h"
#ifndef KMPC_QP_HG_WAVES
#define KMPC_QP_HG_WAVES 3
#endif

namespace kmpc {

size_t step_lds_bytes(int n, int L, int q, int N, size_t elem, int* r1, int* r2, bool lds_tableau) {
  if (r1) *r1 = step_region1(L, N);
  if (r2) *r2 = step_region2(n, L, N, lds_tableau);
  return step_lds_elems(n, L, q, N, lds_tableau) * elem;
}

// one workgroup of TPB threads per trajectory
template <typename T, int TPB, int L_, int N_, int Q_>
// (four-wave trajectories: two workgroups per CU fit the LDS, so at most 256 registers; FOUR with one LDS region and H re-read
//  from LDS in the solve: 128 registers, 38.5 KB)
__global__ __launch_bounds__(TPB, (TPB == 256 ? (step_one_region<TPB, L_, N_, Q_>() ? 4 : 2) : 1)) void step_kernel(const StepArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (a.qp_need && a.phases == PH_QP && a.qp_need[blockIdx.x] == 0) return;  // (finished by shared_fast_kernel)
  const StepVar<T> sv{a.phases, a.first_update, a.plant_switched, a.psi_prev, a.psi_now, a.U0, nullptr, 0, T(0), T(0)};
  step_body<T, TPB, L_, N_, Q_>(a, sv, (int)blockIdx.x, reinterpret_cast<T*>(smem_raw));
}

// The solve alone (kmpc_qp_solve, the shared-model step of cfg4: H, T0, F come from shared_model_kernel) as a kernel of its
// own for the long horizons: the step kernel's register allocation is that of its hungriest phase (256 registers at N = 40:
// 8 trajectories per CU), the solve with H re-read from LDS needs half -- residency is what these kernels are bound by (4.1b).
// HG: the shared-model step -- H is one matrix for the whole batch and is read from global memory (L1 / L2) in the mat-vecs
// instead of from a per-trajectory LDS copy: 1.4 KB of LDS per trajectory, residency set by the registers alone
template <typename T, int TPB, int L_, int N_, int Q_, bool HG>
__global__ __launch_bounds__(TPB, (HG ? KMPC_QP_HG_WAVES : 3)) void step_qp_kernel(const StepArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const StepVar<T> sv{PH_QP, 0, a.plant_switched, a.psi_prev, a.psi_now, a.U0, nullptr, 0, T(0), T(0), HG};
  // The trajectories shared_fast_kernel left over come as a list that a fixed grid walks (StepArgs::qp_list), or -- without a list --
  // one workgroup per trajectory reads its flag.  ONE call site of the step body serves both (two inlined copies of it cost the
  // 168-register kernel 40 more spilled registers).
  const bool list = a.qp_list != nullptr;
  const int cnt = list ? (a.qp_count[0] < a.B ? a.qp_count[0] : a.B) : (int)blockIdx.x + 1;
  for (int i = blockIdx.x; i < cnt; i += gridDim.x) {
    const int b = list ? a.qp_list[i] : i;
    if (!list && a.qp_need && a.qp_need[b] == 0) break;  // (finished by shared_fast_kernel)
    step_body<T, TPB, L_, N_, Q_, true>(a, sv, b, reinterpret_cast<T*>(smem_raw));
    if (!list) break;
    block_sync<TPB>();
  }
}

// ---------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------
template <typename T, int TPB, int L_, int N_, int Q_> static hipError_t launch_impl(const StepArgs<T>& a, hipStream_t s) {
  StepArgs<T> k = a;
  constexpr bool TAB = step_tableau_in_lds<TPB, N_, L_>() && !step_one_region<TPB, L_, N_, Q_>();  // (one region: the fall-back tableau works in global scratch)
  if (!TAB && !a.qp_scratch && (a.phases & PH_QP)) return hipErrorInvalidValue;
  size_t lds = step_lds_bytes(a.n, a.L, a.q, a.N, sizeof(T), &k.r1, &k.r2, TAB);
  if constexpr (step_one_region<TPB, L_, N_, Q_>()) {  // region 2 is C alone ([A B] follows inv_K_G in region 1)
    // (C could wait where g_0 .. g_{N-1} go when it fits there -- n = 1 --, but that place now takes the partial sums of the update)
    const int r2 = (a.n * a.L + 1) & ~1;
    k.r2 = r2;
    lds = ((size_t)k.r1 + k.r2 + vec_elems_one_region(a.n, a.L, a.q, a.N)) * sizeof(T);
  }
  // a solve-only launch of a register solver (kmpc_qp_solve, the shared-model step) touches neither the model block
  // nor a tableau region: without it more trajectories fit on a CU (cfg4 sizes: 25 -> 16 KB each)
  constexpr bool QPK = TPB == 64 && N_ > 24 && L_ > 0 && !TAB;  // (has a solve-only kernel)
  const bool qp_only = (a.phases & (PH_RLS | PH_CONDENSE)) == 0;
  if (!TAB && qp_only) {
    lds -= (size_t)k.r2 * sizeof(T);
    k.r2 = 0;
    if (QPK) {  // ... and of the vector area only the solve's part: red | f | max(x, x_a, g ; psi behind z, Pz)
      const int p = a.L + 1, setq = 3 * a.N > 2 * p + a.L ? 3 * a.N : 2 * p + a.L;
      lds -= (size_t)(vec_elems(a.n, a.L, a.q, a.N) - (16 + a.N + setq)) * sizeof(T);
    }
  }
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  static size_t configured_dev[3][16] = {};  // (function attributes are per device and per kernel symbol)
  auto allow_lds = [&](const void* fn, int which, size_t bytes) -> hipError_t {
    size_t& configured = configured_dev[which][device_slot()];
    if (bytes > 64 * 1024 && bytes > configured) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      if (e != hipSuccess) return e;
      configured = bytes;
    }
    return hipSuccess;
  };
  if constexpr (QPK) {
    if (qp_only && a.phases == PH_QP) {
      // (shared H and the register safeguard compiled in and not switched off: the solve never leaves for the LDS tableau)
      static const bool no_hg = dbg_env("KMPC_QP_NO_HGLOBAL") != nullptr;
      if (a.h_shared && a.H_in && a.T_in && (a.qp_predict & 2) == 0 && !no_hg) {
        const int p = a.L + 1, setq = 3 * a.N > 2 * p + a.L ? 3 * a.N : 2 * p + a.L;
        k.r1 = 0;
        const size_t lq = (size_t)(16 + a.N + setq) * sizeof(T);
        if (hipError_t e = allow_lds(reinterpret_cast<const void*>(&step_qp_kernel<T, TPB, L_, N_, Q_, true>), 1, lq); e != hipSuccess) return e;
        const int gridq = a.qp_list ? (a.B < 2048 ? a.B : 2048) : a.B;  // (list mode: a fixed grid walks the flagged trajectories)
        hipLaunchKernelGGL((step_qp_kernel<T, TPB, L_, N_, Q_, true>), dim3(gridq), dim3(TPB), lq, s, k);
        return hipGetLastError();
      }
      if (hipError_t e = allow_lds(reinterpret_cast<const void*>(&step_qp_kernel<T, TPB, L_, N_, Q_, false>), 2, lds); e != hipSuccess) return e;
      const int gridq = a.qp_list ? (a.B < 2048 ? a.B : 2048) : a.B;
      hipLaunchKernelGGL((step_qp_kernel<T, TPB, L_, N_, Q_, false>), dim3(gridq), dim3(TPB), lds, s, k);
      return hipGetLastError();
    }
  }
  if (hipError_t e = allow_lds(reinterpret_cast<const void*>(&step_kernel<T, TPB, L_, N_, Q_>), 0, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL((step_kernel<T, TPB, L_, N_, Q_>), dim3(a.B), dim3(TPB), lds, s, k);
  return hipGetLastError();
}

template <typename T> hipError_t launch_step(const StepArgs<T>& a, int threads, hipStream_t s) {
  if (a.B <= 0) return hipSuccess;
  if (threads == 256) {
    // BASELINE cfg5 sizes (L = 64, N = 50, y = Cx): four waves per trajectory, dimensions fixed at compile time
    if (a.L == 64 && a.N == 50 && a.q == 2 && a.out_kind == OUT_CX) return launch_impl<T, 256, 64, 50, 2>(a, s);
    return launch_impl<T, 256, 0, 0, 0>(a, s);
  }
  // (the static instantiations take the output kind from q: y = C x has q <= n <= 4 < 8 <= L rows, y = psi has q = L)
  if ((a.q == a.L) != (a.out_kind == OUT_LIFT)) return launch_impl<T, 64, 0, 0, 0>(a, s);
  // compile-time specialisations: BASELINE cfg1/cfg2 (L=20, N=20, y = Cx) and the reference's own
  // dimensions (L=8, N=10; y = Cx duffing.py, y = lifted state vanderpol.py)
  if (a.L == 20 && a.N == 20 && a.q == 2) return launch_impl<T, 64, 20, 20, 2>(a, s);
  if (a.L == 8 && a.N == 30 && a.q == 2) return launch_impl<T, 64, 8, 30, 2>(a, s);
#if !defined(KMPC_DEV_CFG2_ONLY) || defined(KMPC_DEV_LIFT)
  if (a.L == 8 && a.N == 10 && a.q == 8) return launch_impl<T, 64, 8, 10, 8>(a, s);
  if (a.L == 8 && a.N == 30 && a.q == 8) return launch_impl<T, 64, 8, 30, 8>(a, s);
#endif
#ifndef KMPC_DEV_CFG2_ONLY
  if (a.L == 8 && a.N == 10 && a.q == 2) return launch_impl<T, 64, 8, 10, 2>(a, s);
  // BASELINE cfg3: Van der Pol tracking, 8 RBF / MLP observables, N = 30, y = lifted state
  if (a.L == 10 && a.N == 20 && a.q == 1) return launch_impl<T, 64, 10, 20, 1>(a, s);  // Tank_System.m dimensions
  if (a.L == 32 && a.N == 40 && a.q == 2) return launch_impl<T, 64, 32, 40, 2>(a, s);  // BASELINE cfg4 sizes
  if (a.L == 32 && a.N == 40 && a.q == 1) return launch_impl<T, 64, 32, 40, 1>(a, s);
  if (a.L == 20 && a.N == 30 && a.q == 2) return launch_impl<T, 64, 20, 30, 2>(a, s);
#endif
  return launch_impl<T, 64, 0, 0, 0>(a, s);
}

template hipError_t launch_step<float>(const StepArgs<float>&, int, hipStream_t);
template hipError_t launch_step<double>(const StepArgs<double>&, int, hipStream_t);

}  // namespace kmpc

#ifdef KMPC_TRACE
// (measurement build: ONE translation unit, so that the phase stamps of both kernels land in the same buffer)
#include "rollout_kernel.hip"
#endif
