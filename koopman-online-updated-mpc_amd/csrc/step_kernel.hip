// step_kernel.hip -- the per-trajectory hot path on gfx950: RLS-EDMD update -> condensed QP
// build -> box-QP solve.  One workgroup (one 64-lane wave by default, four for large
// dimensions) owns one trajectory; its covariance / model / KKT tiles live in LDS, the
// persistent state streams HBM -> LDS -> HBM exactly once per step.
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
#include "kernels.h"

namespace kmpc {

// ---------------------------------------------------------------------------------------
// host-side LDS sizing (shared with the launcher)
// ---------------------------------------------------------------------------------------
static inline int imax(int a, int b) { return a > b ? a : b; }

static inline int vec_elems(int n, int L, int q, int N) {
  const int p = L + 1;
  const int setA = 2 * p + 6 * L + n + 2 * N * q;  // sz sPz | sy sE sV(2) sW(2) | sx | sG sEr
  const int setB = 8 * N;                          // qx qxa qg qp qHx qHxa qrhs flags
  return imax(setA, setB) + N /*sf*/ + 16 /*reduction scratch*/;
}

size_t step_lds_bytes(int n, int L, int q, int N, size_t elem, int* r1, int* r2) {
  const int p = L + 1;
  int a = imax(imax(p * p, L * L), N * N);
  int b = imax(L * p + n * L, N * N);
  a = (a + 1) & ~1;
  b = (b + 1) & ~1;
  if (r1) *r1 = a;
  if (r2) *r2 = b;
  return (size_t)(a + b + vec_elems(n, L, q, N)) * elem;
}

// ---------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------
template <typename T> struct Tol;
template <> struct Tol<double> {
  static __device__ __forceinline__ double kkt() { return 1e-9; }
  static __device__ __forceinline__ double act() { return 1e-8; }
  static __device__ __forceinline__ double slack() { return 1e-14; }
};
template <> struct Tol<float> {
  static __device__ __forceinline__ float kkt() { return 2e-5f; }
  static __device__ __forceinline__ float act() { return 1e-5f; }
  static __device__ __forceinline__ float slack() { return 1e-6f; }
};

template <typename T> __device__ __forceinline__ T tabs(T v) { return v < T(0) ? -v : v; }
template <typename T> __device__ __forceinline__ T tclip(T v, T lo, T hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over the whole block; every thread gets the result.  `red` holds >= 8 elements.
template <typename T, int TPB> __device__ __forceinline__ T block_sum(T v, T* red) {
  v = wave_sum(v);
  if (TPB == 64) return v;
  const int w = threadIdx.x >> 6;
  __syncthreads();  // protect red[] from the previous use
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  T s = T(0);
#pragma unroll
  for (int i = 0; i < TPB / 64; ++i) s += red[i];
  return s;
}

// ---------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------
template <typename T, int TPB>
__global__ __launch_bounds__(TPB) void step_kernel(const StepArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* const sm = reinterpret_cast<T*>(smem_raw);
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int n = a.n, L = a.L, p = a.L + 1, q = a.q, N = a.N, B = a.B;

  T* const sX = sm;            // P / bar_Q / H
  T* const sY = sX + a.r1;     // K, C / elimination matrix
  T* const sK = sY;
  T* const sC = sY + L * p;
  T* const sH = sX;
  T* const sM = sY;
  T* const vec = sY + a.r2;
  T* const sf = vec;           // N   (lives condense -> QP)
  T* const red = sf + N;       // 16
  T* const va = red + 16;      // aliased vector sets
  // set A (RLS + condense)
  T* const sz = va;            // p
  T* const sPz = sz + p;       // p
  T* const sy = sPz + p;       // L   psi_now
  T* const sE = sy + L;        // L
  T* const sV = sE + L;        // 2L
  T* const sW = sV + 2 * L;    // 2L
  T* const sx = sW + 2 * L;    // n
  T* const sG = sx + n;        // N*q
  T* const sEr = sG + N * q;   // N*q
  // set B (QP)
  T* const qx = va;
  T* const qxa = qx + N;
  T* const qg = qxa + N;
  T* const qp = qg + N;
  T* const qHx = qp + N;
  T* const qHxa = qHx + N;
  T* const qrhs = qHxa + N;
  int* const flg = reinterpret_cast<int*>(qrhs + N);  // N ints


  // =====================================================================================
  // phase 1: recursive least squares (gain form; algebraically K_A inv_K_G of the reference)
  // =====================================================================================
  if (a.phases & PH_RLS) {
    const T* Pg = a.P + (size_t)b * a.strideP;
    T* Kg = a.K + (size_t)b * a.strideK;
    for (int e = tid; e < p * p; e += TPB) sX[e] = Pg[e];
    if (a.first_update) {
      for (int e = tid; e < L * p; e += TPB) sK[e] = T(0);
    } else {
      for (int e = tid; e < L * p; e += TPB) sK[e] = Kg[e];
    }
    for (int i = tid; i < L; i += TPB) {
      sz[i] = a.psi_prev[i * a.pp_sl + b * a.pp_sb];
      sy[i] = a.psi_now[i * a.pn_sl + b * a.pn_sb];
    }
    if (tid == 0) sz[L] = a.u_prev[b];
    for (int i = tid; i < n; i += TPB) sx[i] = a.x_now[(size_t)i * B + b];
    __syncthreads();

    // Pz (P symmetric: column walk is conflict-free in LDS)
    for (int i = tid; i < p; i += TPB) {
      T acc = T(0);
      for (int j = 0; j < p; ++j) acc += sX[j * p + i] * sz[j];
      sPz[i] = acc;
    }
    __syncthreads();
    T part = T(0);
    for (int i = tid; i < p; i += TPB) part += sz[i] * sPz[i];
    const T d = a.lam + block_sum<T, TPB>(part, red);
    const T dinv = T(1) / d;
    const T linv = T(1) / a.lam;

    // P <- (P - Pz Pz' / d) / lam                                   duffing.py:931-932
    T* Pw = a.P + (size_t)b * a.strideP;
    for (int e = tid; e < p * p; e += TPB) {
      const int i = e / p, j = e - i * p;
      Pw[e] = (sX[e] - (sPz[i] * sPz[j]) * dinv) * linv;
    }
    // innovation  y - K z
    for (int r = tid; r < L; r += TPB) {
      T acc = sy[r];
      for (int j = 0; j < p; ++j) acc -= sK[r * p + j] * sz[j];
      sE[r] = acc;
    }
    __syncthreads();
    // K <- K + (y - K z) g',  g = Pz / d                            duffing.py:927-938
    for (int e = tid; e < L * p; e += TPB) {
      const int r = e / p, j = e - r * p;
      const T v = sK[e] + sE[r] * (sPz[j] * dinv);
      sK[e] = v;
      Kg[e] = v;
    }

    if (a.out_kind == OUT_CX) {
      // ---- C = bar_X bar_Q, target x_{k+1}, regressor psi(x_k)      duffing.py:943-953
      const T* Qg = a.Qb + (size_t)b * a.strideQ;
      T* Cg = a.C + (size_t)b * a.strideC;
      __syncthreads();  // everyone is done with P in sX and with sE / sPz
      for (int e = tid; e < L * L; e += TPB) sX[e] = Qg[e];
      if (a.first_update) {
        for (int e = tid; e < n * L; e += TPB) sC[e] = T(0);
      } else {
        for (int e = tid; e < n * L; e += TPB) sC[e] = Cg[e];
      }
      __syncthreads();
      for (int i = tid; i < L; i += TPB) {
        T acc = T(0);
        for (int j = 0; j < L; ++j) acc += sX[j * L + i] * sz[j];
        sPz[i] = acc;
      }
      for (int r = tid; r < n; r += TPB) {
        T acc = sx[r];
        for (int j = 0; j < L; ++j) acc -= sC[r * L + j] * sz[j];
        sE[r] = acc;
      }
      __syncthreads();
      T part2 = T(0);
      for (int i = tid; i < L; i += TPB) part2 += sz[i] * sPz[i];
      const T dc = a.lam + block_sum<T, TPB>(part2, red);
      const T dcinv = T(1) / dc;
      T* Qw = a.Qb + (size_t)b * a.strideQ;
      for (int e = tid; e < L * L; e += TPB) {
        const int i = e / L, j = e - i * L;
        Qw[e] = (sX[e] - (sPz[i] * sPz[j]) * dcinv) * linv;
      }
      for (int e = tid; e < n * L; e += TPB) {
        const int r = e / L, j = e - r * L;
        const T v = sC[e] + sE[r] * (sPz[j] * dcinv);
        sC[e] = v;
        Cg[e] = v;
      }
    }
    __syncthreads();
  } else if (a.phases & PH_CONDENSE) {
    const T* Kg = a.K + (size_t)b * a.strideK;
    for (int e = tid; e < L * p; e += TPB) sK[e] = Kg[e];
    if (a.out_kind == OUT_CX) {
      const T* Cg = a.C + (size_t)b * a.strideC;
      for (int e = tid; e < n * L; e += TPB) sC[e] = Cg[e];
    }
    for (int i = tid; i < L; i += TPB) sy[i] = a.psi_now[i * a.pn_sl + b * a.pn_sb];
    __syncthreads();
  }

  // =====================================================================================
  // phase 2: condensed QP  H = Qw Phi'Phi + Rw I,  f = 2 Qw Phi'(Gamma psi - r)
  // =====================================================================================
  if (a.phases & PH_CONDENSE) {
    const T* ref = a.ref + (a.ref_per_traj ? (size_t)b * q * N : 0);
    const bool cx = (a.out_kind == OUT_CX);
    for (int i = tid; i < L; i += TPB) {
      sV[i] = sK[i * p + L];  // v_0 = B
      sW[i] = sy[i];          // w_0 = psi(x_k)
    }
    __syncthreads();
    // v_{j+1} = A v_j, w_{j+1} = A w_j;  g_j = Co v_j;  e_j = Co w_j - r_{j-1}
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
            for (int l = 0; l < L; ++l) acc += sK[t * p + l] * v[l];
            vn[t] = acc;
          }
        } else if (t < 2 * L) {
          if (j < N) {
            const int r = t - L;
            T acc = T(0);
            for (int l = 0; l < L; ++l) acc += sK[r * p + l] * w[l];
            wn[r] = acc;
          }
        } else if (t < 2 * L + q) {
          if (j < N) {
            const int r = t - 2 * L;
            T g;
            if (cx) {
              g = T(0);
              for (int l = 0; l < L; ++l) g += sC[r * L + l] * v[l];
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
              for (int l = 0; l < L; ++l) y += sC[r * L + l] * w[l];
            } else {
              y = w[r];
            }
            sEr[(j - 1) * q + r] = y - ref[r * N + (j - 1)];
          }
        }
      }
      __syncthreads();
      cur ^= 1;
    }
    // H[a][b] = Qw * S(b-a, N-1-b) (+Rw on the diagonal),  S(d,t) = sum_{s<=t} g_{s+d}.g_s
    for (int d = tid; d < N; d += TPB) {
      T acc = T(0);
      for (int t = 0; t + d < N; ++t) {
        T s = T(0);
        for (int r = 0; r < q; ++r) s += sG[(t + d) * q + r] * sG[t * q + r];
        acc += s;
        const int bb = N - 1 - t, aa = bb - d;
        const T hv = a.Qw * acc + (d == 0 ? a.Rw : T(0));
        sH[aa * N + bb] = hv;
        sH[bb * N + aa] = hv;
      }
    }
    for (int aa = tid; aa < N; aa += TPB) {
      T acc = T(0);
      for (int t = 0; t + aa < N; ++t)
        for (int r = 0; r < q; ++r) acc += sG[t * q + r] * sEr[(t + aa) * q + r];
      sf[aa] = T(2) * a.Qw * acc;
    }
    __syncthreads();
    if (a.H_out) {
      T* Hg = a.H_out + (size_t)b * N * N;
      for (int e = tid; e < N * N; e += TPB) Hg[e] = sH[e];
    }
    if (a.f_out) {
      T* fg = a.f_out + (size_t)b * N;
      for (int e = tid; e < N; e += TPB) fg[e] = sf[e];
    }
  } else if (a.phases & PH_QP) {
    const T* Hg = a.H_in + (size_t)b * N * N;
    const T* fg = a.f_in + (size_t)b * N;
    for (int e = tid; e < N * N; e += TPB) sH[e] = Hg[e];
    for (int e = tid; e < N; e += TPB) sf[e] = fg[e];
    __syncthreads();
  }

  // =====================================================================================
  // phase 3: box QP  min u'Hu + f'u, lb <= u <= ub  -- projected Newton (Bertsekas 1982):
  // Newton step on the free set via a masked symmetric elimination, projected Armijo arc.
  // Terminates on the componentwise KKT test; the answer is the exact solve on the final
  // active set (cold start at clip(0) as the reference, duffing.py:634-635).
  // =====================================================================================
  if (a.phases & PH_QP) {
    const T lb = a.lb, ub = a.ub;
    const T tol = (T)Tol<T>::kkt();
    const T eact = (T)Tol<T>::act() * (ub - lb);
    for (int i = tid; i < N; i += TPB) {
      qx[i] = tclip(T(0), lb, ub);
    }
    __syncthreads();
    for (int i = tid; i < N; i += TPB) {
      T acc = T(0);
      for (int j = 0; j < N; ++j) acc += sH[j * N + i] * qx[j];
      qHx[i] = acc;
    }
    __syncthreads();

    int it = 0;
    int status = 1;
    const int ti = (TPB == 64) ? (tid >> 3) : (tid >> 4);
    const int tj = (TPB == 64) ? (tid & 7) : (tid & 15);
    const int tstep = (TPB == 64) ? 8 : 16;

    while (true) {
      // gradient, cost, componentwise KKT residual
      T pJ = T(0), pbad = T(0);
      for (int i = tid; i < N; i += TPB) {
        const T x = qx[i];
        const T g = T(2) * qHx[i] + sf[i];
        qg[i] = g;
        pJ += x * (qHx[i] + sf[i]);
        T gs = tabs(sf[i]);
        for (int j = 0; j < N; ++j) gs += T(2) * tabs(sH[j * N + i]) * tabs(qx[j]);
        const T res = tabs(x - tclip(x - g, lb, ub));
        const T xs = tabs(x) > T(1) ? tabs(x) : T(1);
        const bool ok = (res <= tol * gs) || (res <= tol * xs);
        pbad += ok ? T(0) : T(1);
      }
      const T J0 = block_sum<T, TPB>(pJ, red);
      const T nbad = block_sum<T, TPB>(pbad, red);
      if (!(J0 == J0) || tabs(J0) > (T)1e300) { status = 2; break; }
      if (nbad == T(0)) { status = 0; break; }
      if (it >= a.max_iter) { status = 1; break; }

      // bound set I and right-hand side
      for (int i = tid; i < N; i += TPB) {
        const T x = qx[i], g = qg[i];
        const int inI = ((x <= lb + eact) && (g > T(0))) || ((x >= ub - eact) && (g < T(0)));
        flg[i] = inI;
        qrhs[i] = inI ? T(0) : -g;
      }
      __syncthreads();
      // masked matrix (upper triangle incl. diagonal): 2H on free x free, identity elsewhere
      for (int i = ti; i < N; i += tstep)
        for (int j = tj; j < N; j += tstep)
          if (j >= i) {
            const bool fr = !flg[i] && !flg[j];
            sM[i * N + j] = fr ? T(2) * sH[i * N + j] : (i == j ? T(1) : T(0));
          }
      __syncthreads();
      // symmetric Gaussian elimination without pivoting (SPD), carrying the rhs
      for (int k = 0; k < N - 1; ++k) {
        const T inv = T(1) / sM[k * N + k];
        for (int i = k + 1 + ti; i < N; i += tstep) {
          const T mi = sM[k * N + i] * inv;
          for (int j = k + 1 + tj; j < N; j += tstep)
            if (j >= i) sM[i * N + j] -= mi * sM[k * N + j];
        }
        const T rk = qrhs[k];
        for (int i = k + 1 + tid; i < N; i += TPB) qrhs[i] -= sM[k * N + i] * inv * rk;
        __syncthreads();
      }
      // back substitution  U p = rhs
      for (int k = N - 1; k >= 0; --k) {
        const T pk = qrhs[k] / sM[k * N + k];
        for (int i = tid; i < k; i += TPB) qrhs[i] -= sM[i * N + k] * pk;
        if (tid == 0) qp[k] = pk;
        __syncthreads();
      }
      // bound-set variables go to the bound the gradient pushes them to
      for (int i = tid; i < N; i += TPB)
        if (flg[i]) qp[i] = (qg[i] > T(0) ? lb : ub) - qx[i];
      __syncthreads();

      // projected Armijo search
      T alpha = T(1);
      while (true) {
        for (int i = tid; i < N; i += TPB) qxa[i] = tclip(qx[i] + alpha * qp[i], lb, ub);
        __syncthreads();
        T pJa = T(0), pdec = T(0);
        for (int i = tid; i < N; i += TPB) {
          T acc = T(0);
          for (int j = 0; j < N; ++j) acc += sH[j * N + i] * qxa[j];
          qHxa[i] = acc;
          const T xa = qxa[i];
          pJa += xa * (acc + sf[i]);
          pdec += flg[i] ? qg[i] * (qx[i] - xa) : alpha * (-qg[i] * qp[i]);
        }
        const T Ja = block_sum<T, TPB>(pJa, red);
        const T dec = block_sum<T, TPB>(pdec, red);
        const T mag = tabs(J0) > tabs(Ja) ? tabs(J0) : tabs(Ja);
        if ((J0 - Ja >= T(1e-4) * dec - (T)Tol<T>::slack() * mag) || alpha < T(1e-10)) break;
        alpha *= T(0.25);
        __syncthreads();
      }
      __syncthreads();
      for (int i = tid; i < N; i += TPB) {
        qx[i] = qxa[i];
        qHx[i] = qHxa[i];
      }
      __syncthreads();
      ++it;
    }

    for (int i = tid; i < N; i += TPB) {
      if (a.Useq) a.Useq[(size_t)i * B + b] = qx[i];
    }
    if (tid == 0) {
      if (a.U0) a.U0[b] = qx[0];
      if (a.u_store) a.u_store[b] = qx[0];
      if (a.status) a.status[b] = a.accumulate ? (a.status[b] > status ? a.status[b] : status) : status;
      if (a.iters) a.iters[b] = a.accumulate ? a.iters[b] + it : it;
    }
  }
}

// ---------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------
template <typename T, int TPB> static hipError_t launch_impl(const StepArgs<T>& a, hipStream_t s) {
  StepArgs<T> k = a;
  const size_t lds = step_lds_bytes(a.n, a.L, a.q, a.N, sizeof(T), &k.r1, &k.r2);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  static size_t configured = 0;
  if (lds > 64 * 1024 && lds > configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&step_kernel<T, TPB>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    configured = lds;
  }
  hipLaunchKernelGGL((step_kernel<T, TPB>), dim3(a.B), dim3(TPB), lds, s, k);
  return hipGetLastError();
}

template <typename T> hipError_t launch_step(const StepArgs<T>& a, int threads, hipStream_t s) {
  if (a.B <= 0) return hipSuccess;
  if (threads == 256) return launch_impl<T, 256>(a, s);
  return launch_impl<T, 64>(a, s);
}

template hipError_t launch_step<float>(const StepArgs<float>&, int, hipStream_t);
template hipError_t launch_step<double>(const StepArgs<double>&, int, hipStream_t);

}  // namespace kmpc
