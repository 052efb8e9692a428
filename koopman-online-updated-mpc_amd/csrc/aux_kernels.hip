// aux_kernels.hip -- small device helpers around the hot path: state initialisation, model
// export, and the plant simulators used to close the loop on the device.
//   Duffing   x1' = x2, x2' = -0.5 x2 + x1 - x1^3 + u            duffing.py:255
//             after step 100: x2' = -5 x2 + 2 x1 - 0.5 x1^3 + u   duffing.py:991-992
//   Van der Pol  x1' = 2 x2, x2' = 2 x2 - 10 x1^2 x2 - 0.8 x1 + u vanderpol_RBF.py:113
//             after step 100: x1' = x2, x2' = -3 x2 - 10 x1^2 x2 - 3 x1 + u  vanderpol.py:923-931
//   RK4, h = 0.05                                                duffing.py:256-261
#include <cstdio>
#include "kernels.h"
#include <cstdlib>
#include "plant_device.h"

namespace kmpc {

const char* dbg_env(const char* name) {
  static const bool on = getenv("KMPC_DEBUG") != nullptr;
  if (!on) {
    // a measurement switch in the environment of a process that has not asked for them: say so once instead of silently running the
    // default path (a script that compares "with" and "without" would otherwise compare the default with itself)
    static bool warned = false;
    if (!warned && getenv(name)) {
      warned = true;
      fprintf(stderr, "libkoopmpc: %s is set but ignored -- the KMPC_* measurement switches are read only when KMPC_DEBUG is set\n", name);
    }
    return nullptr;
  }
  return getenv(name);
}


template <typename T> __global__ __launch_bounds__(256) void plant_kernel(const PlantArgs<T> a) {
  const int B = a.B;
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
    T x1 = a.X[b], x2 = a.X[(size_t)B + b];
    plant_apply<T>(a.plant, a.switched, a.h, x1, x2, a.U[b]);
    a.X[b] = x1;
    a.X[(size_t)B + b] = x2;
  }
}

template <typename T> hipError_t launch_plant(const PlantArgs<T>& a, hipStream_t s) {
  if (a.B <= 0) return hipSuccess;
  int grid = (a.B + 255) / 256;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL((plant_kernel<T>), dim3(grid), dim3(256), 0, s, a);
  return hipGetLastError();
}

// P = P0 I, bar_Q = Q0 I, K = 0, C = 0 for every trajectory   (duffing.py:927-930, 944-946)
template <typename T>
__global__ __launch_bounds__(256) void fill_state_kernel(T* P, long strideP, int p, T P0, T* Qb, long strideQ, int L,
                                                         T Q0, T* K, long strideK, T* C, long strideC, int n) {
  const int b = blockIdx.x;
  T* Pb = P + (size_t)b * strideP;
  for (int e = threadIdx.x; e < (int)strideP; e += blockDim.x) Pb[e] = (e < p * p && e / p == e % p) ? P0 : T(0);
  T* Qq = Qb + (size_t)b * strideQ;
  for (int e = threadIdx.x; e < (int)strideQ; e += blockDim.x) Qq[e] = (e < L * L && e / L == e % L) ? Q0 : T(0);
  if (K) {
    T* Kb = K + (size_t)b * strideK;
    for (int e = threadIdx.x; e < (int)strideK; e += blockDim.x) Kb[e] = T(0);
  }
  if (C) {
    T* Cb = C + (size_t)b * strideC;
    for (int e = threadIdx.x; e < (int)strideC; e += blockDim.x) Cb[e] = T(0);
  }
}

template <typename T>
hipError_t launch_fill_state(T* P, long strideP, int p, T P0, T* Qb, long strideQ, int L, T Q0, T* K, long strideK,
                             T* C, long strideC, int n, int B, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL((fill_state_kernel<T>), dim3(B), dim3(256), 0, s, P, strideP, p, P0, Qb, strideQ, L, Q0, K,
                     strideK, C, strideC, n);
  return hipGetLastError();
}

// split K = [A B] (duffing.py:965-967) into dense per-trajectory blocks
template <typename T>
__global__ __launch_bounds__(256) void export_model_kernel(const T* K, long strideK, const T* C, long strideC, int n,
                                                           int L, T* A_out, T* B_out, T* C_out) {
  const int b = blockIdx.x, p = L + 1;
  const T* Kb = K + (size_t)b * strideK;
  if (A_out)
    for (int e = threadIdx.x; e < L * L; e += blockDim.x) A_out[(size_t)b * L * L + e] = Kb[(e / L) * p + (e % L)];
  if (B_out)
    for (int e = threadIdx.x; e < L; e += blockDim.x) B_out[(size_t)b * L + e] = Kb[e * p + L];
  if (C_out && C) {
    const T* Cb = C + (size_t)b * strideC;
    for (int e = threadIdx.x; e < n * L; e += blockDim.x) C_out[(size_t)b * n * L + e] = Cb[e];
  }
}

template <typename T> __global__ __launch_bounds__(64) void qp_value_kernel(const T* H, const T* f, const T* c, const T* U, int N, int B, T* fun) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const T* Hb = H + (size_t)b * N * N;
  T acc = c[b];
  for (int i = 0; i < N; ++i) {
    const T ui = U[(size_t)i * B + b];
    T hu = f[(size_t)b * N + i];
    for (int j = 0; j < N; ++j) hu += Hb[i * N + j] * U[(size_t)j * B + b];
    acc += ui * hu;
  }
  fun[b] = acc;
}
template <typename T> hipError_t launch_qp_value(const T* H, const T* f, const T* c, const T* U, int N, int B, T* fun, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL((qp_value_kernel<T>), dim3((B + 63) / 64), dim3(64), 0, s, H, f, c, U, N, B, fun);
  return hipGetLastError();
}

// the inverse of export_model_kernel: dense A, B, C blocks -> state layout (kmpc_mpc_solve)
template <typename T>
__global__ __launch_bounds__(256) void import_model_kernel(const T* A, const T* Bv, const T* C, int shared, int n, int L, T* K,
                                                           long strideK, T* Cs, long strideC) {
  const int b = blockIdx.x, p = L + 1;
  const size_t m = shared ? 0 : (size_t)b;
  T* Kb = K + (size_t)b * strideK;
  for (int e = threadIdx.x; e < L * p; e += blockDim.x) {
    const int r = e / p, c = e - r * p;
    Kb[e] = c < L ? A[m * L * L + (size_t)r * L + c] : Bv[m * L + r];
  }
  if (C && Cs) {
    T* Cb = Cs + (size_t)b * strideC;
    for (int e = threadIdx.x; e < n * L; e += blockDim.x) Cb[e] = C[m * n * L + e];
  }
}
template <typename T>
hipError_t launch_import_model(const T* A, const T* Bv, const T* C, int shared, int n, int L, int B, T* K, long strideK,
                               T* Cs, long strideC, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL((import_model_kernel<T>), dim3(B), dim3(256), 0, s, A, Bv, C, shared, n, L, K, strideK, Cs, strideC);
  return hipGetLastError();
}

// data_generate.py:17-79 on the device: one thread per trajectory walks its n_steps plant steps (sequential in
// time, parallel over trajectories) and leaves the transition pairs in the panels the offline fit reads
template <typename T>
__global__ __launch_bounds__(256) void datagen_kernel(int plant, T h, const T* X0, const T* U, int n_traj, int n_steps, T* X, T* Y) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_traj) return;
  const size_t M = (size_t)n_traj * n_steps;
  T x1 = X0[t], x2 = X0[(size_t)n_traj + t];
  for (int i = 0; i < n_steps; ++i) {
    const size_t c = (size_t)i * n_traj + t;
    X[c] = x1; X[M + c] = x2;
    plant_apply<T>(plant, 0, h, x1, x2, U[c]);
    Y[c] = x1; Y[M + c] = x2;
  }
}
template <typename T>
hipError_t launch_datagen(int plant, T h, const T* X0, const T* U, int n_traj, int n_steps, T* X, T* Y, hipStream_t s) {
  if (n_traj <= 0 || n_steps <= 0) return hipSuccess;
  hipLaunchKernelGGL((datagen_kernel<T>), dim3((n_traj + 255) / 256), dim3(256), 0, s, plant, h, X0, U, n_traj, n_steps, X, Y);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_export_model(const T* K, long strideK, const T* C, long strideC, int n, int L, int B, T* A_out,
                               T* B_out, T* C_out, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL((export_model_kernel<T>), dim3(B), dim3(256), 0, s, K, strideK, C, strideC, n, L, A_out, B_out,
                     C_out);
  return hipGetLastError();
}

// dst[b*stride + e] = src[e]  (offline model handed to every trajectory, duffing.py:811-813)
template <typename T>
__global__ __launch_bounds__(256) void broadcast_kernel(T* dst, long stride, const T* src, int count) {
  T* d = dst + (size_t)blockIdx.x * stride;
  for (int e = threadIdx.x; e < count; e += blockDim.x) d[e] = src[e];
}
template <typename T> hipError_t launch_broadcast(T* dst, long stride, const T* src, int count, int B, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL((broadcast_kernel<T>), dim3(B), dim3(256), 0, s, dst, stride, src, count);
  return hipGetLastError();
}

// g <- a * g + d   (Gram accumulation with forgetting)
__global__ __launch_bounds__(256) void axpby_kernel(double* g, const double* d, double a, int count) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < count; e += gridDim.x * blockDim.x) g[e] = a * g[e] + d[e];
}
hipError_t launch_axpby(double* g, const double* d, double a, int count, hipStream_t s) {
  hipLaunchKernelGGL(axpby_kernel, dim3((count + 255) / 256), dim3(256), 0, s, g, d, a, count);
  return hipGetLastError();
}

#define INST(T)                                                                                                      \
  template hipError_t launch_plant<T>(const PlantArgs<T>&, hipStream_t);                                             \
  template hipError_t launch_fill_state<T>(T*, long, int, T, T*, long, int, T, T*, long, T*, long, int, int,        \
                                           hipStream_t);                                                             \
  template hipError_t launch_broadcast<T>(T*, long, const T*, int, int, hipStream_t);                                \
  template hipError_t launch_export_model<T>(const T*, long, const T*, long, int, int, int, T*, T*, T*, hipStream_t); \
  template hipError_t launch_import_model<T>(const T*, const T*, const T*, int, int, int, int, T*, long, T*, long,   \
                                             hipStream_t);                                                           \
  template hipError_t launch_datagen<T>(int, T, const T*, const T*, int, int, T*, T*, hipStream_t);                  \
  template hipError_t launch_qp_value<T>(const T*, const T*, const T*, const T*, int, int, T*, hipStream_t);
INST(float)
INST(double)

// ---------------------------------------------------------------------------------------
// dense state blocks <-> wave image of the register-state step (step_v2.h: V2Dims).  One workgroup per trajectory;
// the image is walked element by element (coalesced on the image side), padding elements are written as zero.
// ---------------------------------------------------------------------------------------
template <bool TO_IMAGE>
__global__ __launch_bounds__(256) void state_image_kernel(double* P, long sP, double* K, long sK, double* Q, long sQ, double* C, long sC,
                                                          int n, int L, double* img, long stride) {
  const V2Dims d(L, n);
  const int b = blockIdx.x, p = L + 1;
  double* const im = img + (size_t)b * stride;
  const long l2 = (long)d.cp * d.s2 * 2, tot = d.elems();
  for (long e = threadIdx.x; e < tot; e += blockDim.x) {
    const bool lay2 = e < l2;
    const long r = lay2 ? e : e - l2;
    const int S = lay2 ? d.s2 : d.s1;
    const int col = (int)(r / (2 * S)) * 2 + (int)(r & 1), slot = (int)((r >> 1) % S);
    double* src = nullptr;
    if (lay2) {
      if (slot < p) { if (col < p) src = P + (size_t)b * sP + (size_t)slot * p + col; }
      else if (col < L) src = Q + (size_t)b * sQ + (size_t)(slot - p) * L + col;
    } else {
      if (slot < L) { if (col < p) src = K + (size_t)b * sK + (size_t)slot * p + col; }
      else if (col < L && C) src = C + (size_t)b * sC + (size_t)(slot - L) * L + col;
    }
    if (TO_IMAGE) im[e] = src ? *src : 0.0;
    else if (src) *src = im[e];
  }
}
long state_image_elems(int L, int n) { return V2Dims(L, n).elems(); }
hipError_t launch_state_to_image(const double* P, long sP, const double* K, long sK, const double* Q, long sQ, const double* C, long sC,
                                 int n, int L, int B, double* img, long stride, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL((state_image_kernel<true>), dim3(B), dim3(256), 0, s, const_cast<double*>(P), sP, const_cast<double*>(K), sK,
                     const_cast<double*>(Q), sQ, const_cast<double*>(C), sC, n, L, img, stride);
  return hipGetLastError();
}
hipError_t launch_image_to_state(const double* img, long stride, int n, int L, int B, double* P, long sP, double* K, long sK,
                                 double* Q, long sQ, double* C, long sC, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL((state_image_kernel<false>), dim3(B), dim3(256), 0, s, P, sP, K, sK, Q, sQ, C, sC, n, L, const_cast<double*>(img), stride);
  return hipGetLastError();
}

template <typename S, typename D> __global__ __launch_bounds__(256) void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, size_t count) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) dst[i] = (D)src[i];
}
template <typename S, typename D> hipError_t launch_cast(const S* src, D* dst, size_t count, hipStream_t s) {
  if (count == 0) return hipSuccess;
  const size_t g = (count + 255) / 256;
  hipLaunchKernelGGL((cast_kernel<S, D>), dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, s, src, dst, count);
  return hipGetLastError();
}
template hipError_t launch_cast<float, double>(const float*, double*, size_t, hipStream_t);
template hipError_t launch_cast<double, float>(const double*, float*, size_t, hipStream_t);

// place_kernel: rank of every trajectory by last launch's solver work (descending, ties by index), then the card deal of RolloutArgs::perm.
// Round 6: a stable least-significant-digit radix sort by ONE workgroup, O(B), for batches beyond 16 384 -- rounds 4-5 compared every pair
// (O(B^2 / 16): 9 us at B = 4096, 62 us at 16 384) and had NO deal beyond; place_pairs_kernel below still serves B <= 16384.  The key is the work clipped to 12 bits and inverted (heavy first); three passes of four
// bits over ping-pong buffers of packed (key << 20 | index) words in global memory (L2-resident: B words); the elements start in index
// order and every pass is stable, so ties stay in index order: the result is a function of `work` alone (deterministic, as before).
// A pass: every thread counts the digits of its contiguous chunk (16 counters), the 16 x 1024 counts are scanned digit-major in LDS,
// every thread scatters its chunk in order.
constexpr int PLACE_T = 1024;
__global__ __launch_bounds__(PLACE_T) void place_kernel(const int32_t* __restrict__ work, int B, uint32_t* __restrict__ buf0, uint32_t* __restrict__ buf1,
                                                        int32_t* __restrict__ perm) {
  __shared__ int sCnt[16 * PLACE_T];   // [digit][thread]
  __shared__ int sPart[PLACE_T];
  const int t = threadIdx.x;
  const int C = (B + PLACE_T - 1) / PLACE_T, e0 = t * C, e1 = e0 + C < B ? e0 + C : B;
  for (int e = e0; e < e1; ++e) {
    int w = work[e];
    w = w < 0 ? 0 : (w > 4095 ? 4095 : w);
    buf0[e] = ((uint32_t)(4095 - w) << 20) | (uint32_t)e;
  }
  __syncthreads();
  uint32_t* src = buf0;
  uint32_t* dst = buf1;
  for (int pass = 0; pass < 3; ++pass) {
    const int sh = 20 + 4 * pass;
    int cnt[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) cnt[d] = 0;
    for (int e = e0; e < e1; ++e) {
      const int d = (src[e] >> sh) & 15;
#pragma unroll
      for (int dd = 0; dd < 16; ++dd) cnt[dd] += (dd == d) ? 1 : 0;  // (no dynamic register index)
    }
#pragma unroll
    for (int d = 0; d < 16; ++d) sCnt[d * PLACE_T + t] = cnt[d];
    __syncthreads();
    // exclusive scan of the 16 x 1024 counts in digit-major order: thread t owns entries 16 t .. 16 t + 15 of the flattened array
    int loc[16], sum = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { loc[i] = sum; sum += sCnt[16 * t + i]; }
    // (1024 partial sums: inclusive scan inside each wave by shuffles, the sixteen wave totals through LDS)
    int inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(inc, off, 64);
      if ((t & 63) >= off) inc += v;
    }
    if ((t & 63) == 63) sPart[t >> 6] = inc;
    __syncthreads();
    int wbase = 0;
    for (int wv2 = 0; wv2 < (t >> 6); ++wv2) wbase += sPart[wv2];
    const int base = wbase + inc - sum;  // exclusive
    __syncthreads();  // (sPart is rewritten by the next pass)
#pragma unroll
    for (int i = 0; i < 16; ++i) sCnt[16 * t + i] = base + loc[i];
    __syncthreads();
    int pos[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) pos[d] = sCnt[d * PLACE_T + t];
    for (int e = e0; e < e1; ++e) {
      const uint32_t v = src[e];
      const int d = (v >> sh) & 15;
      int p = 0;
#pragma unroll
      for (int dd = 0; dd < 16; ++dd) { if (dd == d) { p = pos[dd]; pos[dd] += 1; } }
      dst[p] = v;
    }
    __threadfence_block();
    __syncthreads();
    uint32_t* const tmp = src; src = dst; dst = tmp;
  }
  // rank r -> slot of the card deal: stratum p = r / G goes to wave p of the G workgroups, in snake order
  const int G = B >> 4;
  for (int r = t; r < B; r += PLACE_T) {
    const int e = (int)(src[r] & 0xfffffu);
    const int p = r / G;
    int g = r - p * G;
    if (p & 1) g = G - 1 - g;
    // (stratum p -> wave p: age order.  Measured and dropped: strata 4-7 / 12-15 in reverse order so that the four SIMDs carry equal
    //  sums -- it undoes the gain: heavy solves on the OLDEST waves is what pays, profiles/r4_cfg2_placement_ab.txt)
    perm[g * 16 + p] = e;
  }
}
// The same table by comparing every pair (rounds 4-5): sixteen lanes share the scan of one element, B / 16 workgroups of 256 threads with
// the work panel in LDS (B <= 16384: 64 KB).  O(B^2 / 16), but spread over the whole chip: 8.7 us at B = 4096 and 62 us at 16 384, where
// the one-workgroup sort takes 28 and 93 us (three passes of global-memory round trips, strided chunk reads, barriers of sixteen waves:
// ~6 us per 1024 trajectories) -- it serves the batches up to 16 384, the sort everything beyond (where there was no deal at all).
__global__ __launch_bounds__(256) void place_pairs_kernel(const int32_t* __restrict__ work, int B, int32_t* __restrict__ perm) {
  extern __shared__ int32_t sW[];  // the whole work panel (B <= 16384: 64 KB): the scan reads it 16 times per workgroup
  for (int c = threadIdx.x; c < B; c += 256) { const int w = work[c]; sW[c] = w < 0 ? 0 : (w > 4095 ? 4095 : w); }  // (the sort's key)
  __syncthreads();
  const int e = blockIdx.x * 16 + (threadIdx.x >> 4), part = threadIdx.x & 15;
  const int we = e < B ? sW[e] : 0;
  int r = 0;
#pragma unroll 8
  for (int c = part; c < B; c += 16) {
    const int wc = sW[c];
    r += (wc > we || (wc == we && c < e)) ? 1 : 0;
  }
  r += __shfl_xor(r, 8, 64); r += __shfl_xor(r, 4, 64); r += __shfl_xor(r, 2, 64); r += __shfl_xor(r, 1, 64);
  if (part == 0 && e < B) {
    const int G = B >> 4, p = r / G;
    int g = r - p * G;
    if (p & 1) g = G - 1 - g;
    perm[g * 16 + p] = e;
  }
}
// scratch: 2 B words (the sort's buffers).  B a multiple of 16, at most 2^20 trajectories.
hipError_t launch_place(const int32_t* work, int B, int32_t* perm, uint32_t* scratch, hipStream_t s) {
  if (B <= 0 || (B & 15) || B > (1 << 20) || !scratch) return hipErrorInvalidValue;
  if (B <= 16384) {
    static bool configured_dev[16] = {};  // (function attributes are per device)
    bool& configured = configured_dev[device_slot()];
    if (!configured) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&place_pairs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
      configured = true;
    }
    hipLaunchKernelGGL(place_pairs_kernel, dim3(B / 16), dim3(256), (size_t)B * sizeof(int32_t), s, work, B, perm);
  } else {
    hipLaunchKernelGGL(place_kernel, dim3(1), dim3(PLACE_T), 0, s, work, B, scratch, scratch + B, perm);
  }
  return hipGetLastError();
}

// Row-major zero-padded weights (Mp x Hp) -> MFMA A-fragments [tile][k-step][lane]: lane l of tile t, k-step ks
// holds W[16 t + (l & 15)][4 ks + (l >> 4)], so a wave reads one fragment as 64 consecutive elements.
template <typename T> __global__ void pack_afrag_kernel(const T* src, int Mp, int Hp, int KS, T* dst) {
  const int total = (Mp / 16) * KS * 64;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int l = e & 63, ks = (e >> 6) % KS, t = (e >> 6) / KS;
    const int col = 4 * ks + (l >> 4);
    dst[e] = col < Hp ? src[(size_t)(16 * t + (l & 15)) * Hp + col] : T(0);
  }
}
template <typename T> hipError_t launch_pack_afrag(const T* src, int Mp, int Hp, int KS, T* dst, hipStream_t s) {
  const int total = (Mp / 16) * KS * 64;
  hipLaunchKernelGGL((pack_afrag_kernel<T>), dim3((total + 255) / 256), dim3(256), 0, s, src, Mp, Hp, KS, dst);
  return hipGetLastError();
}
template hipError_t launch_pack_afrag<float>(const float*, int, int, int, float*, hipStream_t);
template hipError_t launch_pack_afrag<double>(const double*, int, int, int, double*, hipStream_t);

}  // namespace kmpc
