// api.hip -- the C ABI of libkoopmpc.so (include/koopmpc.h): handle, parameter upload,
// persistent state and stream-ordered launches.  No compute happens on the host.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/koopmpc.h"
#include "kernels.h"
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enumerators only (ncclDouble, ncclSum): the library resolves ncclAllReduce at run time

using namespace kmpc;

static thread_local std::string g_create_error;

struct kmpc_handle {
  kmpc_config cfg{};
  std::string err;
  virtual ~kmpc_handle() {
    for (auto& mk : marks) (void)hipEventDestroy(mk.e);
  }
  virtual int set_encoder_layer(int layer, const double* W, const double* b, int rows, int cols) = 0;
  virtual int set_centres(const double* cx, int L, int n) = 0;
  virtual int set_model(const double* A, const double* B, const double* C) = 0;
  virtual int set_terminal_weight(const double* PN) = 0;
  virtual int terminal_from_dare(const double* Qh, double R, int maxiter, double eps, int per_traj, double* PN_out,
                                 int32_t* iters_out, hipStream_t s) = 0;
  virtual int set_terminal_refresh(int every, const double* Qh, double R, int maxiter, double eps) = 0;
  virtual int rollout_is_fused() const = 0;
  virtual int rollout_plugin_status(std::string* text) const = 0;
  virtual int reset(hipStream_t s) = 0;
  virtual int state_init(double P0, double barQ0, hipStream_t s) = 0;
  virtual int state_init_from(const double* KA0, const double* P0m, const double* barX0, const double* barQ0m, hipStream_t s) = 0;
  virtual int lift(const void* X, void* Psi, int B, hipStream_t s) = 0;
  virtual int rls_update(const void* psi, const void* u, const void* psin, const void* xn, int B, hipStream_t s) = 0;
  virtual int get_model(void* A, void* B, void* C, hipStream_t s) = 0;
  virtual int condense(const void* psi, const void* ref, int rpt, void* H, void* f, void* c, int B, hipStream_t s) = 0;
  virtual int mpc_solve(const void* A, const void* Bv, const void* C, int shared, const void* psi, const void* ref, int rpt,
                        double lb, double ub, double Qw, double Rw, const double* PN, void* U, void* U0, void* fun,
                        int32_t* st, int32_t* it, int B, hipStream_t s) = 0;
  virtual int generate_and_fit(int plant, const void* X0, const void* U, int n_traj, int n_steps, double hs, double ridge,
                               int init_rls, void* A, void* B, void* C, void* Xo, void* Yo, hipStream_t s) = 0;
  int device = 0;
  // The few entry points that work on the null stream (kmpc_set_model, checkpoints, the first shared-model call) wait for the handle's OWN
  // outstanding work instead of the whole device (a device-wide synchronisation stalls the caller's unrelated streams).
  // What that work is: every stream-ordered entry point ends by recording a handle-owned event on its stream (one event per distinct
  // stream the caller has used with this handle); sync_own() waits for those events.  An event stays valid after the caller has
  // destroyed the stream, and a roll-out on stream A followed by a call on stream B leaves both recorded (the raw handle of
  // "the last stream" dangled and covered one stream only).  Calls made while their stream is being captured into a graph leave no mark.
  struct StreamMark { hipStream_t s; hipEvent_t e; unsigned long long age; };
  std::vector<StreamMark> marks;
  unsigned long long mark_clock = 0;
  void mark_stream(hipStream_t s) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); return; }
    if (cap != hipStreamCaptureStatusNone) return;
    StreamMark* m = nullptr;
    for (auto& mk : marks)
      if (mk.s == s) { m = &mk; break; }
    if (!m) {
      if (marks.size() >= 16) {  // (a caller that keeps making streams: the least recently used mark is waited for and reused)
        m = &marks[0];
        for (auto& mk : marks)
          if (mk.age < m->age) m = &mk;
        (void)hipEventSynchronize(m->e);
        m->s = s;
      } else {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return; }
        marks.push_back(StreamMark{s, e, 0});
        m = &marks.back();
      }
    }
    m->age = ++mark_clock;
    if (hipEventRecord(m->e, s) != hipSuccess) (void)hipGetLastError();
  }
  hipError_t sync_marks() {
    for (auto& mk : marks) {
      const hipError_t e = hipEventSynchronize(mk.e);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }
  virtual int qp_solve(const void* H, const void* f, void* U, int32_t* st, int32_t* it, int B, hipStream_t s) = 0;
  virtual int step(const void* X, const void* ref, int rpt, void* U0, void* Useq, int32_t* st, int32_t* it,
                   hipStream_t s) = 0;
  virtual int plant_step(int plant, void* X, const void* U, double h, int sw, int B, hipStream_t s) = 0;
  virtual int set_applied_input(const void* U, int B, hipStream_t s) = 0;
  virtual int set_online_update(int on) = 0;
  virtual int rollout(int plant, void* X, const void* ref, int rpt, int steps, int step0, int switch_step, double hs,
                      void* Ulog, void* Xlog, int32_t* st, int32_t* it, hipStream_t s) = 0;
  virtual int offline_fit(const void* X, const void* Y, const void* U, int M, double ridge, int init_rls, void* A,
                          void* B, void* C, hipStream_t s) = 0;
  virtual int64_t gram_elems() const = 0;
  virtual int shared_local_gram(const void* X, double* delta, hipStream_t s) = 0;
  virtual int shared_solve_plant(const double* delta, const void* ref, void* U0, void* Useq, int32_t* st, int32_t* it, int plant, void* X,
                                 int switched, double hstep, hipStream_t s) = 0;
  virtual int shared_solve(const double* delta, const void* ref, void* U0, void* Useq, int32_t* st, int32_t* it,
                           hipStream_t s) = 0;
  virtual int shared_get_model(void* A, void* B, void* C, hipStream_t s) = 0;
  virtual int shared_rollout(int plant, void* X, const void* ref, int steps, int step0, int switch_step, double hstep, void* comm,
                             void* Ulog, void* Xlog, void* U0out, void* Useq, int32_t* st, int32_t* it, hipStream_t s) = 0;
  virtual int64_t state_bytes() const = 0;
  virtual int state_export(void* blob, int64_t bytes) = 0;
  virtual int state_import(const void* blob, int64_t bytes) = 0;
  virtual int profile_enable(int on) = 0;
  virtual int profile_read(double* ms2, int64_t* count, int reset) = 0;
  virtual int64_t algorithmic_bytes() const = 0;
};

// ncclAllReduce of the RCCL that lives in this process (under PyTorch: torch's own copy), else of the system library; null: none.
// The library itself does not link against RCCL.
static void* resolve_nccl_allreduce() {
  static void* fn = nullptr;
  if (!fn) {
    fn = dlsym(RTLD_DEFAULT, "ncclAllReduce");
    if (!fn) {
      void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (lib) fn = dlsym(lib, "ncclAllReduce");
    }
  }
  return fn;
}

#define HIPCHK(expr)                                                                  \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) {                                                           \
      err = std::string(#expr) + ": " + hipGetErrorString(e_);                        \
      return -(int)(1000 + (int)e_);                                                  \
    }                                                                                 \
  } while (0)

#define FAIL(code, msg) \
  do {                  \
    err = (msg);        \
    return (code);      \
  } while (0)

static inline long even_up(long v) { return (v + 1) & ~1L; }
// KMPC_PLANT_* with the optional KMPC_PLANT_RK4_MATLAB bit (not for the tank map, which has no Runge-Kutta step)
static inline bool plant_id_ok(int plant) {
  const int base = plant & ~KMPC_PLANT_RK4_MATLAB;
  return base >= KMPC_PLANT_DUFFING && base <= KMPC_PLANT_TANK && !(base == KMPC_PLANT_TANK && (plant & KMPC_PLANT_RK4_MATLAB));
}

// a device temporary that is freed on every return path (the HIPCHK macro returns from the middle of a function)
struct DevTmp {
  void* p = nullptr;
  ~DevTmp() { if (p) (void)hipFree(p); }
  template <typename U> U* as() const { return static_cast<U*>(p); }
};

template <typename T>
struct Impl : kmpc_handle {
  int n, m, L, p, q, N, B, threads;
  int Hp = 0, Lp = 0;
  int hid = 0, Lenc = 0;  // effective hidden width / the encoder's own output dimension (lift_offset)
  std::vector<std::vector<double>> hostW, hostb;  // the layers as given (row-major), kept for finalize_encoder
  long sP, sK, sQ, sC;
  // persistent state
  T *dP = nullptr, *dK = nullptr, *dQ = nullptr, *dC = nullptr;
  T *dPsi[2] = {nullptr, nullptr};  // [B][L] trajectory-major: [cur], [prev]
  T* dUprev = nullptr;
  T* dQpScr = nullptr;  // [B][N*N] fall-back tableau of the register QP solvers (global scratch, rarely touched)
  int qp_scr_cap = 0;
  // Register-state step (step_v2.h): the fused roll-out of the small y = C x dimension sets streams the state as a "wave image"
  // ([B][sImg] doubles).  The dense blocks above stay the form every other entry point works on; whichever of the two was
  // written last is the valid one and the other is rebuilt on demand (state_to_image / image_to_state, one pass over the state).
  double* dImg = nullptr;
  long sImg = 0;
  bool use_img = false, img_valid = false, dense_valid = true;
  // KMPC_F32 handle of a register-state dimension set (row g2: BASELINE configs[1] names fp32; SURVEY G6: fp32 panels, float64
  // covariance): kmpc_rollout runs the float64 fused roll-out behind float32 panels.  `core` is a float64 handle of the same
  // configuration that owns the state while roll-outs run (its wave image; psi_{k-1}, u_{k-1}, the warm start in float64); the
  // float32 blocks of this handle are the form every other entry point works on.  Whichever side was written last is the valid one
  // (dense_valid / img_valid as for the wave image) and the other is rebuilt on demand by a cast over the blocks (core_push / core_pull).
  Impl<double>* core = nullptr;
  bool is_core = false;  // this handle IS the float64 core of a float32 handle
  // roll-out plug-in of this handle's dimension set (rollout_plugin.hip): 0 built-in set or no fused roll-out at all, 1 loaded, -1 could not be made
  int plugin_state = 0;
  RolloutPluginKey plugin_key{};
  std::string plugin_msg;
  // four-wave solver (threads = 256, float64): every trajectory's last tableau and its variable set, kept from step to step
  // (StepArgs::qp_carry); any change of the model from outside forgets them (the next solve starts from 2H)
  T* dQpCarry = nullptr;
  int32_t* dQpCarrySet = nullptr;
  T* dWarm = nullptr;  // [N][B] last minimiser = start of the next solve (the reference restarts at zeros, duffing.py:634-635)
  int cur = 0;
  bool have_prev = false;  // a previous (psi, u) exists -> next step runs the RLS update
  bool rls_fresh = true;   // next RLS update starts from K_A = 0, bar_X = 0
  // parameters
  T *dW1 = nullptr, *db1 = nullptr, *dWh[2] = {nullptr, nullptr}, *dbh[2] = {nullptr, nullptr}, *dWo = nullptr,
    *dbo = nullptr, *dcx = nullptr, *dTmp = nullptr;
  std::vector<bool> layer_set;
  bool centres_set = false;
  // profiling
  bool prof = false;
  std::vector<hipEvent_t> ev;  // triples
  size_t ev_used = 0;
  static constexpr size_t EV_CAP = 3 * 4096;

  int init(const kmpc_config& c) {
    cfg = c;
    n = c.n; m = c.m; L = c.L; N = c.N; B = c.batch; p = L + 1;
    q = (c.output_kind == KMPC_OUT_LIFT) ? L : (c.out_rows > 0 ? c.out_rows : n);
    if (c.output_kind != KMPC_OUT_LIFT && (c.out_row0 < 0 || c.out_row0 + q > n)) FAIL(-2, "out_row0/out_rows outside C");
    if (c.delta_u && !(c.umax > c.umin)) FAIL(-2, "delta_u needs umin < umax");
    if (m != 1) FAIL(-2, "m must be 1 (the reference takes B_hat = K[:, Nlift], duffing.py:170-171)");
    if (n < 1 || n > 4) FAIL(-2, "n must be in 1..4");
    if (L < 1 || L > 64) FAIL(-2, "L must be in 1..64");
    if (N < 1 || N > 64) FAIL(-2, "N must be in 1..64");
    if (B < 1) FAIL(-2, "batch must be >= 1");
    if (!(c.lambda > 0.0 && c.lambda <= 1.0)) FAIL(-2, "lambda must be in (0, 1]");
    if (!(c.ub > c.lb)) FAIL(-2, "need lb < ub");
    threads = c.threads;
    if (threads == 0) threads = (p * p > 2048 || N * N > 2048) ? 256 : 64;
    if (threads != 64 && threads != 256) FAIL(-2, "threads must be 0, 64 or 256");
    int r1, r2;
    if (step_lds_bytes(n, L, q, N, sizeof(T), &r1, &r2) > 160 * 1024) FAIL(-2, "configuration exceeds 160 KB of LDS");
    sP = even_up((long)p * p); sK = even_up((long)L * p); sQ = even_up((long)L * L); sC = even_up((long)n * L);
    HIPCHK(hipMalloc(&dP, sizeof(T) * sP * B));
    HIPCHK(hipMalloc(&dK, sizeof(T) * sK * B));
    HIPCHK(hipMalloc(&dQ, sizeof(T) * sQ * B));
    HIPCHK(hipMalloc(&dC, sizeof(T) * sC * B));
    HIPCHK(hipMalloc(&dPsi[0], sizeof(T) * (size_t)L * B));
    HIPCHK(hipMalloc(&dPsi[1], sizeof(T) * (size_t)L * B));
    HIPCHK(hipMalloc(&dUprev, sizeof(T) * (size_t)B));
    HIPCHK(hipMalloc(&dQpScr, sizeof(T) * (size_t)B * N * N));
    qp_scr_cap = B;
    HIPCHK(hipMalloc(&dWarm, sizeof(T) * (size_t)N * B));
    HIPCHK(hipMalloc(&dTmp, sizeof(T) * (size_t)(L * p + n * L + 64)));
    HIPCHK(hipMemset(dK, 0, sizeof(T) * sK * B));
    HIPCHK(hipMemset(dC, 0, sizeof(T) * sC * B));
    HIPCHK(hipMemset(dPsi[0], 0, sizeof(T) * (size_t)L * B));
    HIPCHK(hipMemset(dPsi[1], 0, sizeof(T) * (size_t)L * B));
    HIPCHK(hipMemset(dUprev, 0, sizeof(T) * (size_t)B));
    HIPCHK(hipMemset(dWarm, 0, sizeof(T) * (size_t)N * B));
    if (sizeof(T) == 8 && threads == 256 && !dbg_env("KMPC_QP_NO_CARRY")) {  // (the variable is a measurement aid)
      HIPCHK(hipMalloc(&dQpCarry, sizeof(T) * (size_t)B * N * N));
      HIPCHK(hipMalloc(&dQpCarrySet, sizeof(int32_t) * (size_t)B * 4));
      HIPCHK(hipMemset(dQpCarrySet, 0, sizeof(int32_t) * (size_t)B * 4));
    }
    // a dimension set without a built-in fused roll-out gets its kernel now (rollout_plugin.hip: kernel cache, else hipcc, 4-8 s);
    // a set that cannot be served stays on per-step launches and kmpc_rollout_plugin_status says why
    {
      const bool rbf = c.lift_kind != KMPC_LIFT_MLP;
      const int hid0 = c.hidden + (c.lift_offset == 2 ? 2 * n : 0), Hp0 = hid0 <= 112 ? 112 : 128, Lp0 = ((L + 15) / 16) * 16, KS0 = (hid0 + 3) / 4;
      const bool io32 = sizeof(T) == 4;
      const bool wanted = threads == 64 && n == 2 && !is_core && (c.lift_kind != KMPC_LIFT_MLP || (hid0 >= 1 && hid0 <= 128)) &&
                          (io32 ? (c.output_kind != KMPC_OUT_LIFT && !c.delta_u && rollout_io32_available(n, L, N, q, rbf) && !dbg_env("KMPC_NO_IO32_ROLLOUT"))
                                : rollout_fused_available<double>(n, L, N, q, threads, rbf));
      if (wanted && rollout_plugin_key(n, L, N, q, rbf, Lp0, KS0, Hp0, B, io32, &plugin_key)) {
        plugin_state = rollout_plugin_get(plugin_key, &plugin_msg) ? 1 : -1;
      }
    }
    if constexpr (sizeof(T) == 8) {
      use_img = threads == 64 && plugin_state >= 0 && (c.output_kind != KMPC_OUT_LIFT || q == L) && rollout_uses_image(n, L, N, q) &&
                rollout_fused_available<T>(n, L, N, q, threads, c.lift_kind != KMPC_LIFT_MLP);
      if (use_img) {
        sImg = state_image_elems(L, n);
        HIPCHK(hipMalloc(&dImg, sizeof(double) * (size_t)sImg * B));
      }
    }
    if constexpr (sizeof(T) == 4) {
      if (threads == 64 && plugin_state >= 0 && c.output_kind != KMPC_OUT_LIFT && !c.delta_u && rollout_io32_available(n, L, N, q, c.lift_kind != KMPC_LIFT_MLP) &&
          !dbg_env("KMPC_NO_IO32_ROLLOUT")) {  // (the variable is a measurement / test aid: float32 handles as per-step launches)
        kmpc_config c64 = c;
        c64.dtype = KMPC_F64;
        core = new Impl<double>();
        core->is_core = true;  // (its launches carry float32 panels: this handle's plug-in, loaded above, serves them)
        const int rc = core->init(c64);
        if (rc) { err = "float64 core of the float32 handle: " + core->err; delete core; core = nullptr; return rc; }
      }
    }
    if (c.lift_kind == KMPC_LIFT_MLP) {
      if (c.layers != 2 && c.layers != 3) FAIL(-2, "layers (hidden layers) must be 2 or 3");
      if (c.hidden < 1 || c.hidden > 128) FAIL(-2, "hidden must be in 1..128");
      // lift_offset (the MATLAB scripts' lifts): 1: psi(x) - psi(0) (Koopman_update_Tracking_Lift.m:65) -- an output bias;
      // 2: [x; psi(x)] - [0; psi(0)] (Koopman_update.m:67, L = n + the encoder's outputs) -- 2n more hidden units carry
      // relu(x), relu(-x) through the layers with identity weights and the first n output rows read x = relu(x) - relu(-x)
      // back (exact: one of the two is zero).  Either way the kernels see an ordinary encoder (finalize_encoder).
      if (c.lift_offset < 0 || c.lift_offset > 2) FAIL(-2, "lift_offset must be 0, 1 or 2");
      hid = c.hidden + (c.lift_offset == 2 ? 2 * n : 0);
      Lenc = L - (c.lift_offset == 2 ? n : 0);
      if (Lenc < 1) FAIL(-2, "lift_offset = 2 needs L > n (L = n + the encoder's output dimension)");
      if (hid > 128) FAIL(-2, "hidden + 2 n must not exceed 128 with lift_offset = 2");
      Hp = hid <= 112 ? 112 : 128;
      Lp = ((L + 15) / 16) * 16;
      HIPCHK(hipMalloc(&dW1, sizeof(T) * Hp * 4));
      HIPCHK(hipMalloc(&db1, sizeof(T) * Hp));
      for (int k = 0; k < c.layers - 1; ++k) {
        HIPCHK(hipMalloc(&dWh[k], sizeof(T) * Hp * Hp));
        HIPCHK(hipMalloc(&dbh[k], sizeof(T) * Hp));
      }
      HIPCHK(hipMalloc(&dWo, sizeof(T) * Lp * Hp));
      HIPCHK(hipMalloc(&dbo, sizeof(T) * Lp));
      layer_set.assign(c.layers + 1, false);
      hostW.assign(c.layers + 1, std::vector<double>());
      hostb.assign(c.layers + 1, std::vector<double>());
    } else {
      if (c.lift_offset != 0) FAIL(-2, "lift_offset applies to the MLP encoder only");
      HIPCHK(hipMalloc(&dcx, sizeof(T) * L * n));
    }
    return reset(nullptr);
  }

  ~Impl() override {
    delete core;
    for (void* ptr : {(void*)dP, (void*)dK, (void*)dQ, (void*)dC, (void*)dPsi[0], (void*)dPsi[1], (void*)dUprev, (void*)dWarm,
                      (void*)dW1, (void*)db1, (void*)dWh[0], (void*)dWh[1], (void*)dbh[0], (void*)dbh[1], (void*)dWo,
                      (void*)dbo, (void*)dcx, (void*)dTmp, (void*)dU0, (void*)dGram, (void*)dPartial, (void*)dKs, (void*)dCs,
                      (void*)dHs, (void*)dFs, (void*)df0s, (void*)dWt, (void*)dWhp[0], (void*)dWhp[1], (void*)dWop,
                      (void*)dDareP, (void*)dDareIt, (void*)dWtB, (void*)dQpScr, (void*)dMsK, (void*)dMsC, (void*)dMsH,
                      (void*)dMsf, (void*)dMsc, (void*)dMsW, (void*)dTs, (void*)dNeed, (void*)dImg, (void*)dQpCarry, (void*)dQpCarrySet, (void*)dDelta, (void*)dQpList, (void*)dWork, (void*)dPerm, (void*)dTermQ, (void*)dTermScr, (void*)dEyeL, (void*)dFastPack})
      if (ptr) (void)hipFree(ptr);
    for (auto e : ev) (void)hipEventDestroy(e);
    if (evPlace) (void)hipEventDestroy(evPlace);
  }

  // the handle's own outstanding work has finished (the events behind its stream-ordered calls, then the null stream the caller is
  // about to use)
  int sync_own() {
    HIPCHK(sync_marks());
    HIPCHK(hipStreamSynchronize(nullptr));
    return 0;
  }
  // upload a (rows x cols) host double matrix into a zero-padded (prow x pcol) device matrix of T
  int upload_padded(T* dst, const double* src, int rows, int cols, int prow, int pcol) {
    std::vector<T> tmp((size_t)prow * pcol, T(0));
    for (int r = 0; r < rows; ++r)
      for (int c2 = 0; c2 < cols; ++c2) tmp[(size_t)r * pcol + c2] = (T)src[(size_t)r * cols + c2];
    HIPCHK(hipMemcpy(dst, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
  }

  // the dense blocks are about to be read or modified / the wave image is about to be used by the roll-out
  int ensure_dense(hipStream_t s, bool will_modify) {
    if constexpr (sizeof(T) == 8) {
      if (use_img && !dense_valid) {
        HIPCHK(launch_image_to_state(dImg, sImg, n, L, B, (double*)dP, sP, (double*)dK, sK, (double*)dQ, sQ, (double*)dC, sC, s));
        dense_valid = true;
      }
    } else {
      if (core && !dense_valid) {
        const int rc = core_pull(s);
        if (rc) return rc;
        dense_valid = true;
      }
    }
    if (will_modify) img_valid = false;
    return 0;
  }
  // float32 handle <-> its float64 core: the state blocks (same strides on both sides), psi_{k-1}, u_{k-1}, the warm start, the flags
  int core_push(hipStream_t s) {
    if constexpr (sizeof(T) == 4) {
      int rc = core->ensure_dense(s, true);
      if (rc) { err = core->err; return rc; }
      HIPCHK((launch_cast<float, double>((const float*)dP, core->dP, (size_t)sP * B, s)));
      HIPCHK((launch_cast<float, double>((const float*)dK, core->dK, (size_t)sK * B, s)));
      HIPCHK((launch_cast<float, double>((const float*)dQ, core->dQ, (size_t)sQ * B, s)));
      HIPCHK((launch_cast<float, double>((const float*)dC, core->dC, (size_t)sC * B, s)));
      HIPCHK((launch_cast<float, double>((const float*)dPsi[0], core->dPsi[0], (size_t)L * B, s)));
      HIPCHK((launch_cast<float, double>((const float*)dPsi[1], core->dPsi[1], (size_t)L * B, s)));
      HIPCHK((launch_cast<float, double>((const float*)dUprev, core->dUprev, (size_t)B, s)));
      HIPCHK((launch_cast<float, double>((const float*)dWarm, core->dWarm, (size_t)N * B, s)));
      core->cur = cur; core->have_prev = have_prev; core->rls_fresh = rls_fresh; core->update_on = update_on;
      core->cfg.P0 = cfg.P0; core->cfg.barQ0 = cfg.barQ0;
      core->dense_valid = true; core->img_valid = false;
    }
    return 0;
  }
  int core_pull(hipStream_t s) {
    if constexpr (sizeof(T) == 4) {
      int rc = core->ensure_dense(s, false);
      if (rc) { err = core->err; return rc; }
      HIPCHK((launch_cast<double, float>(core->dP, (float*)dP, (size_t)sP * B, s)));
      HIPCHK((launch_cast<double, float>(core->dK, (float*)dK, (size_t)sK * B, s)));
      HIPCHK((launch_cast<double, float>(core->dQ, (float*)dQ, (size_t)sQ * B, s)));
      HIPCHK((launch_cast<double, float>(core->dC, (float*)dC, (size_t)sC * B, s)));
      HIPCHK((launch_cast<double, float>(core->dPsi[0], (float*)dPsi[0], (size_t)L * B, s)));
      HIPCHK((launch_cast<double, float>(core->dPsi[1], (float*)dPsi[1], (size_t)L * B, s)));
      HIPCHK((launch_cast<double, float>(core->dUprev, (float*)dUprev, (size_t)B, s)));
      HIPCHK((launch_cast<double, float>(core->dWarm, (float*)dWarm, (size_t)N * B, s)));
      cur = core->cur; have_prev = core->have_prev; rls_fresh = core->rls_fresh;
    }
    return 0;
  }
  // (called where the model or the state is replaced from outside: set_model, reset, offline fit, checkpoint import)
  int ensure_image(hipStream_t s) {
    if constexpr (sizeof(T) == 8) {
      if (!img_valid) {
        HIPCHK(launch_state_to_image((const double*)dP, sP, (const double*)dK, sK, (const double*)dQ, sQ, (const double*)dC, sC, n, L, B, dImg, sImg, s));
        img_valid = true;
      }
      dense_valid = false;  // (the roll-out writes the image)
    } else if (core) {
      if (!img_valid) {
        const int rc = core_push(s);
        if (rc) return rc;
        img_valid = true;
      }
      dense_valid = false;  // (the roll-out advances the core's state)
    }
    return 0;
  }

  int forget_tableaux(hipStream_t s) {
    if (dQpCarrySet) HIPCHK(hipMemsetAsync(dQpCarrySet, 0, sizeof(int32_t) * (size_t)B * 4, s));
    return 0;
  }

  int set_encoder_layer(int layer, const double* W, const double* b, int rows, int cols) override {
    if (cfg.lift_kind != KMPC_LIFT_MLP) FAIL(-3, "handle was not created with KMPC_LIFT_MLP");
    const int nl = cfg.layers + 1;  // linear layers
    if (layer < 0 || layer >= nl) FAIL(-3, "layer index out of range");
    const int in_dim = layer == 0 ? n : cfg.hidden;
    const int out_dim = layer == nl - 1 ? Lenc : cfg.hidden;
    if (rows != out_dim || cols != in_dim) FAIL(-3, "encoder layer shape does not match the configuration");
    hostW[layer].assign(W, W + (size_t)rows * cols);
    hostb[layer].assign(b, b + rows);
    layer_set[layer] = true;
    packed_ok = false;
    if (core) { const int rc = core->set_encoder_layer(layer, W, b, rows, cols); if (rc) { err = core->err; return rc; } }
    for (int k = 0; k < nl; ++k)
      if (!layer_set[k]) return 0;
    return finalize_encoder();
  }
  // every layer is known: upload the network the kernels evaluate (the given one, or the one that realises lift_offset)
  int finalize_encoder() {
    const int nl = cfg.layers + 1, H = cfg.hidden, off = cfg.lift_offset, ex = off == 2 ? 2 * n : 0;
    std::vector<double> psi0((size_t)Lenc, 0.0);
    if (off) {  // psi(0) = W_o relu(... relu(b_1)) + b_o, float64 on the host
      std::vector<double> h(hostb[0]), hn;
      for (double& v : h) v = v > 0.0 ? v : 0.0;
      for (int k = 1; k < nl; ++k) {
        const int rows = k == nl - 1 ? Lenc : H;
        hn.assign((size_t)rows, 0.0);
        for (int r = 0; r < rows; ++r) {
          double acc = hostb[k][r];
          for (int c2 = 0; c2 < H; ++c2) acc += hostW[k][(size_t)r * H + c2] * h[c2];
          hn[r] = (k == nl - 1) ? acc : (acc > 0.0 ? acc : 0.0);
        }
        h.swap(hn);
      }
      psi0 = h;
    }
    int rc;
    for (int k = 0; k < nl; ++k) {
      const bool first = k == 0, last = k == nl - 1;
      const int rin = first ? n : H, rout = last ? Lenc : H;          // the given layer
      const int ein = first ? n : hid, eout = last ? L : hid;         // the effective one
      std::vector<double> We((size_t)eout * ein, 0.0), be((size_t)eout, 0.0);
      const int r0 = (last && off == 2) ? n : 0;                       // (the first n output rows are x)
      for (int r = 0; r < rout; ++r) {
        for (int c2 = 0; c2 < rin; ++c2) We[(size_t)(r0 + r) * ein + c2] = hostW[k][(size_t)r * rin + c2];
        be[r0 + r] = hostb[k][r] - (last ? psi0[r] : 0.0);
      }
      if (ex) {
        if (first) {
          for (int i = 0; i < n; ++i) { We[(size_t)(H + i) * ein + i] = 1.0; We[(size_t)(H + n + i) * ein + i] = -1.0; }
        } else if (!last) {
          for (int i = 0; i < ex; ++i) We[(size_t)(H + i) * ein + (H + i)] = 1.0;
        } else {
          for (int i = 0; i < n; ++i) { We[(size_t)i * ein + (H + i)] = 1.0; We[(size_t)i * ein + (H + n + i)] = -1.0; }
        }
      }
      if (first) {
        if ((rc = upload_padded(dW1, We.data(), eout, ein, Hp, 4))) return rc;
        if ((rc = upload_padded(db1, be.data(), 1, eout, 1, Hp))) return rc;
      } else if (last) {
        if ((rc = upload_padded(dWo, We.data(), eout, ein, Lp, Hp))) return rc;
        if ((rc = upload_padded(dbo, be.data(), 1, eout, 1, Lp))) return rc;
      } else {
        if ((rc = upload_padded(dWh[k - 1], We.data(), eout, ein, Hp, Hp))) return rc;
        if ((rc = upload_padded(dbh[k - 1], be.data(), 1, eout, 1, Hp))) return rc;
      }
    }
    packed_ok = false;
    return 0;
  }

  int set_centres(const double* cx, int L_, int n_) override {
    if (cfg.lift_kind == KMPC_LIFT_MLP) FAIL(-3, "handle was created with KMPC_LIFT_MLP");
    if (L_ != L || n_ != n) FAIL(-3, "centre shape does not match the configuration");
    int rc = upload_padded(dcx, cx, L, n, L, n);
    if (rc) return rc;
    if (core && (rc = core->set_centres(cx, L_, n_))) { err = core->err; return rc; }
    centres_set = true;
    return 0;
  }

  int set_model(const double* A, const double* Bm, const double* C) override {
    std::vector<T> k((size_t)L * p);
    for (int r = 0; r < L; ++r) {
      for (int c2 = 0; c2 < L; ++c2) k[(size_t)r * p + c2] = (T)A[(size_t)r * L + c2];
      k[(size_t)r * p + L] = (T)Bm[r];
    }
    // (a roll-out enqueued on a non-blocking stream may still be writing the wave image: the null-stream conversion below does
    //  not wait for such a stream by itself)
    { int rc = sync_own(); if (rc) return rc; }
    { int rc = ensure_dense(nullptr, true); if (rc) return rc; }
    { int rc = forget_tableaux(nullptr); if (rc) return rc; }
    HIPCHK(hipMemcpy(dTmp, k.data(), k.size() * sizeof(T), hipMemcpyHostToDevice));
    HIPCHK(launch_broadcast<T>(dK, sK, dTmp, L * p, B, nullptr));
    if (cfg.output_kind == KMPC_OUT_CX) {
      if (!C) FAIL(-3, "C is required for KMPC_OUT_CX");
      std::vector<T> c((size_t)n * L);
      for (size_t i = 0; i < c.size(); ++i) c[i] = (T)C[i];
      HIPCHK(hipMemcpy(dTmp + L * p, c.data(), c.size() * sizeof(T), hipMemcpyHostToDevice));
      HIPCHK(launch_broadcast<T>(dC, sC, dTmp + L * p, n * L, B, nullptr));
    }
    if (dKs) {
      HIPCHK(hipMemcpy(dKs, dK, sizeof(T) * (size_t)L * p, hipMemcpyDeviceToDevice));
      HIPCHK(hipMemcpy(dCs, dC, sizeof(T) * (size_t)n * L, hipMemcpyDeviceToDevice));
    }
    HIPCHK(hipStreamSynchronize(nullptr));
    return 0;
  }

  // (a float32 handle's float64 core gets the block this handle itself works with -- PN - Qw I rounded to float32 -- so that the
  //  fused roll-outs, the per-step entry points and a checkpoint loaded into another handle all use the same weight: core_weight)
  int core_weight(const std::vector<T>* w) {
    if (!core) return 0;
    std::vector<double> pn;
    if (w) {
      pn.resize(w->size());
      for (int r = 0; r < q; ++r)
        for (int c2 = 0; c2 < q; ++c2) pn[(size_t)r * q + c2] = (double)(*w)[(size_t)r * q + c2] + (r == c2 ? cfg.Qw : 0.0);
    }
    const int rc = core->set_terminal_weight(w ? pn.data() : nullptr);
    if (rc) err = core->err;
    return rc;
  }
  int set_terminal_weight(const double* PN) override {
    wterm_from_dare = false;
    if (!PN) { have_wterm = false; hostPN.clear(); return core_weight(nullptr); }
    std::vector<T> w((size_t)q * q);
    for (int r = 0; r < q; ++r)
      for (int c2 = 0; c2 < q; ++c2) w[(size_t)r * q + c2] = (T)(PN[(size_t)r * q + c2] - (r == c2 ? cfg.Qw : 0.0));
    { const int rc = core_weight(&w); if (rc) return rc; }
    if (!dWt) HIPCHK(hipMalloc(&dWt, sizeof(T) * (size_t)q * q));
    HIPCHK(hipMemcpy(dWt, w.data(), w.size() * sizeof(T), hipMemcpyHostToDevice));
    hostPN.assign(PN, PN + (size_t)q * q);
    have_wterm = true;
    return 0;
  }

  // Terminal block from the Riccati iteration of the reference (solve_DARE, duffing.py:583-598) on this handle's
  // model(s): P = DARE(A, B, Q_lift, R), Q_bar(end) = Co P Co' (Koopman_update.m:381).  per_traj = 0: the model of
  // trajectory 0 (after kmpc_set_model / kmpc_offline_fit every trajectory holds that one) gives ONE block for the
  // batch; per_traj = 1: every trajectory's current [A B], C gives its own block (the MATLAB controller recomputes
  // its terminal ingredients with the updated model, Koopman_update.m:215, 276-281).
  int terminal_from_dare(const double* Qh, double R, int maxiter, double eps, int per_traj, double* PN_out,
                         int32_t* iters_out, hipStream_t s) override {
    if constexpr (sizeof(T) != 8) {
      FAIL(-2, "kmpc_terminal_from_dare: float64 handles only (the reference's Riccati iteration is float64)");
    } else {
      if (!Qh || maxiter < 1) FAIL(-3, "kmpc_terminal_from_dare: bad arguments");
      const int nb = per_traj ? B : 1;
      if (nb == 0) return 0;
      { int rc = ensure_dense(s, false); if (rc) return rc; }
      DevTmp tQl, tEye;
      HIPCHK(hipMalloc(&tQl.p, sizeof(double) * (size_t)L * L));
      double* const dQl = tQl.as<double>();
      HIPCHK(hipMemcpyAsync(dQl, Qh, sizeof(double) * (size_t)L * L, hipMemcpyHostToDevice, s));
      { int rc = dare_buffers(nb); if (rc) return rc; }
      { int rc = dare_blocks(nb, dQl, R, maxiter, eps, s); if (rc) return rc; }
      if (PN_out || iters_out) {
        HIPCHK(hipStreamSynchronize(s));
        if (PN_out) {
          HIPCHK(hipMemcpy(PN_out, dWtB, sizeof(double) * (size_t)nb * q * q, hipMemcpyDeviceToHost));
          for (int m = 0; m < nb; ++m)
            for (int r = 0; r < q; ++r) PN_out[((size_t)m * q + r) * q + r] += cfg.Qw;  // hand back P_N, not P_N - Qw I
        }
        if (iters_out) HIPCHK(hipMemcpy(iters_out, dDareIt, sizeof(int32_t) * (size_t)nb, hipMemcpyDeviceToHost));
      }
      HIPCHK(hipStreamSynchronize(s));
      wterm_per_traj = per_traj != 0;
      wterm_from_dare = true;
      have_wterm = true;
      return 0;
    }
  }

  // Riccati workspace and the terminal block(s) for nb models (kmpc_terminal_from_dare, kmpc_set_terminal_refresh)
  int dare_buffers(int nb) {
    if (dare_cap >= nb) return 0;
    double* nP = nullptr; int32_t* nIt = nullptr; double* nW = nullptr;
    HIPCHK(hipMalloc(&nP, sizeof(double) * (size_t)nb * L * L));
    HIPCHK(hipMalloc(&nIt, sizeof(int32_t) * (size_t)nb));
    HIPCHK(hipMalloc(&nW, sizeof(double) * (size_t)nb * q * q));
    if (dDareP) (void)hipFree(dDareP);
    if (dDareIt) (void)hipFree(dDareIt);
    if (dWtB) (void)hipFree(dWtB);
    dDareP = nP; dDareIt = nIt; dWtB = nW;
    dare_cap = nb;
    return 0;
  }
  double* dEyeL = nullptr;  // L x L identity: the output map of y = psi
  // P = DARE(A, B, Qdev, R) of the first nb trajectories' current [A B] and their blocks Co P Co' - Qw I -> dWtB   (dare_kernel)
  int dare_blocks(int nb, const double* Qdev, double R, int maxiter, double eps, hipStream_t s) {
    if constexpr (sizeof(T) == 8) {
      // y = psi (lifted output): Co = I, the block is P itself; y = C x: rows cy0 .. cy0+q-1 of C
      const bool lift_out = cfg.output_kind == KMPC_OUT_LIFT;
      if (lift_out && !dEyeL) {
        std::vector<double> eye((size_t)L * L, 0.0);
        for (int i = 0; i < L; ++i) eye[(size_t)i * L + i] = 1.0;
        HIPCHK(hipMalloc(&dEyeL, sizeof(double) * (size_t)L * L));
        HIPCHK(hipMemcpy(dEyeL, eye.data(), sizeof(double) * (size_t)L * L, hipMemcpyHostToDevice));
      }
      DareArgs d{};
      d.nb = nb; d.L = L; d.q = q; d.maxiter = maxiter; d.shared_model = 0;
      d.A = (const double*)dK; d.strideA = sK; d.ldA = p;
      d.B = (const double*)dK + L; d.strideB = sK; d.incB = p;
      d.Q = Qdev; d.R = R; d.eps = eps;
      d.Co = lift_out ? dEyeL : (const double*)dC + (size_t)cfg.out_row0 * L;
      d.strideC = lift_out ? 0 : sC;
      d.P = dDareP; d.K = nullptr; d.PN = dWtB; d.pn_sub_diag = cfg.Qw; d.iters = dDareIt;
      HIPCHK(launch_dare(d, s));
    }
    return 0;
  }

  // kmpc_set_terminal_refresh: the MATLAB controller recomputes its terminal ingredients with the updated model at EVERY iteration
  // (Koopman_update.m:215 dlqr, :381 Q_bar(end) = C P C').  every > 0: kmpc_step / kmpc_rollout run the reference's Riccati iteration
  // (duffing.py:583-598) on every trajectory's freshly updated [A B] each `every`-th step and rebuild its block Co P Co' before the
  // condensed QP is formed -- INSIDE the launch for the fused roll-out (rollout_kernel<.., TERM>, a plug-in made here), as RLS launch /
  // dare_kernel / QP launch on the per-step route.  0: off (the blocks stay what the last refresh left).
  int term_every = 0, term_maxiter = 500, term_plugin_state = 0;
  long term_count = 0;
  double term_R = 0.01, term_eps = 0.01;
  double *dTermQ = nullptr, *dTermScr = nullptr;
  RolloutPluginKey term_key{};
  std::string term_msg;
  int set_terminal_refresh(int every, const double* Qh, double R, int maxiter, double eps) override {
    if constexpr (sizeof(T) != 8) {
      FAIL(-2, "kmpc_set_terminal_refresh: float64 handles only (the reference's Riccati iteration is float64)");
    } else {
      if (every < 0 || (every > 0 && (!Qh || maxiter < 1))) FAIL(-3, "kmpc_set_terminal_refresh: bad arguments");
      if (every == 0) { term_every = 0; return 0; }
      { int rc = sync_own(); if (rc) return rc; }
      const bool had = have_wterm, had_per_traj = have_wterm && wterm_from_dare && wterm_per_traj;
      const int old_cap = dare_cap;
      std::vector<double> w0((size_t)q * q, 0.0);  // the block every trajectory holds until its first refresh: the current one, else Qw I
      if (had && !had_per_traj) {
        if (wterm_from_dare) HIPCHK(hipMemcpy(w0.data(), dWtB, sizeof(double) * w0.size(), hipMemcpyDeviceToHost));
        else HIPCHK(hipMemcpy(w0.data(), dWt, sizeof(double) * w0.size(), hipMemcpyDeviceToHost));
      }
      if (!(had_per_traj && old_cap >= B)) {
        { int rc = dare_buffers(B); if (rc) return rc; }
        std::vector<double> wb((size_t)B * q * q);
        for (int b = 0; b < B; ++b) std::copy(w0.begin(), w0.end(), wb.begin() + (size_t)b * q * q);
        HIPCHK(hipMemcpy(dWtB, wb.data(), sizeof(double) * wb.size(), hipMemcpyHostToDevice));
      }
      if (!dTermQ) HIPCHK(hipMalloc(&dTermQ, sizeof(double) * (size_t)L * L));
      HIPCHK(hipMemcpy(dTermQ, Qh, sizeof(double) * (size_t)L * L, hipMemcpyHostToDevice));
      have_wterm = true; wterm_from_dare = true; wterm_per_traj = true;
      term_R = R; term_maxiter = maxiter; term_eps = eps; term_count = 0;
      // the fused roll-out of this configuration with the refresh inside: a plug-in of its own (also for the built-in dimension sets)
      term_plugin_state = 0;
      const bool rbf = cfg.lift_kind != KMPC_LIFT_MLP;
      if (threads == 64 && n == 2 && plugin_state >= 0 && rollout_fused_available<double>(n, L, N, q, threads, rbf) &&
          rollout_plugin_key(n, L, N, q, rbf, Lp, (hid + 3) / 4, Hp, B, false, &term_key, true)) {
        term_plugin_state = rollout_plugin_get(term_key, &term_msg) ? 1 : -1;
        if (term_plugin_state > 0 && !dTermScr) HIPCHK(hipMalloc(&dTermScr, sizeof(double) * (size_t)B * term_scratch_elems(L)));
      }
      term_every = every;
      return 0;
    }
  }

  int reset(hipStream_t s) override {
    if (dGram) HIPCHK(hipMemsetAsync(dGram, 0, sizeof(double) * (size_t)gram_elems(), s));
    shared_has_samples = false;
    { int rc = ensure_dense(s, true); if (rc) return rc; }
    { int rc = forget_tableaux(s); if (rc) return rc; }
    HIPCHK(launch_fill_state<T>(dP, sP, p, (T)cfg.P0, dQ, sQ, L, (T)cfg.barQ0, nullptr, sK, nullptr, sC, n, B, s));
    HIPCHK(hipMemsetAsync(dWarm, 0, sizeof(T) * (size_t)N * B, s));  // first solve starts at clip(0) like the reference's (duffing.py:634-635)
    have_prev = false;
    rls_fresh = true;
    cur = 0;
    return 0;
  }

  // kmpc_state_init: the reference's start of the online update (duffing.py:927-930, 944-946) with the given scales
  int state_init(double P0, double barQ0, hipStream_t s) override {
    if (!(P0 > 0.0) || !(barQ0 > 0.0)) FAIL(-3, "kmpc_state_init: the scales must be positive");
    cfg.P0 = P0; cfg.barQ0 = barQ0;
    return reset(s);
  }
  // kmpc_state_init_from: every trajectory continues from given reference-form accumulators (the MATLAB start from the
  // offline Gram, Koopman_update.m:264-265): K_A0 (L x p), inv_K_G = P0 (p x p), bar_X0 (n x L), bar_Q0 (L x L), host,
  // row-major.  The handle keeps the gain form: K = K_A0 P0 (duffing.py:938), C = bar_X0 bar_Q0 (duffing.py:953).
  int state_init_from(const double* KA0, const double* P0m, const double* barX0, const double* barQ0m, hipStream_t s) override {
    if (!KA0 || !P0m || !barQ0m || (cfg.output_kind == KMPC_OUT_CX && !barX0)) FAIL(-3, "kmpc_state_init_from: bad arguments");
    int rc = reset(s);
    if (rc) return rc;
    std::vector<T> hP((size_t)p * p), hQ((size_t)L * L), hK((size_t)L * p), hC((size_t)n * L, T(0));
    for (size_t i = 0; i < hP.size(); ++i) hP[i] = (T)P0m[i];
    for (size_t i = 0; i < hQ.size(); ++i) hQ[i] = (T)barQ0m[i];
    for (int r = 0; r < L; ++r)
      for (int c2 = 0; c2 < p; ++c2) {
        double acc = 0.0;
        for (int k = 0; k < p; ++k) acc += KA0[(size_t)r * p + k] * P0m[(size_t)k * p + c2];
        hK[(size_t)r * p + c2] = (T)acc;
      }
    if (cfg.output_kind == KMPC_OUT_CX)
      for (int r = 0; r < n; ++r)
        for (int c2 = 0; c2 < L; ++c2) {
          double acc = 0.0;
          for (int k = 0; k < L; ++k) acc += barX0[(size_t)r * L + k] * barQ0m[(size_t)k * L + c2];
          hC[(size_t)r * L + c2] = (T)acc;
        }
    DevTmp ttmp;
    const size_t tot = hP.size() + hQ.size() + hK.size() + hC.size();
    HIPCHK(hipMalloc(&ttmp.p, sizeof(T) * tot));
    T* const tmp = ttmp.as<T>();
    T* tP = tmp; T* tQ = tP + hP.size(); T* tK = tQ + hQ.size(); T* tC = tK + hK.size();
    HIPCHK(hipMemcpyAsync(tP, hP.data(), sizeof(T) * hP.size(), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(tQ, hQ.data(), sizeof(T) * hQ.size(), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(tK, hK.data(), sizeof(T) * hK.size(), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(tC, hC.data(), sizeof(T) * hC.size(), hipMemcpyHostToDevice, s));
    HIPCHK(launch_broadcast<T>(dP, sP, tP, p * p, B, s));
    HIPCHK(launch_broadcast<T>(dQ, sQ, tQ, L * L, B, s));
    HIPCHK(launch_broadcast<T>(dK, sK, tK, L * p, B, s));
    if (cfg.output_kind == KMPC_OUT_CX) HIPCHK(launch_broadcast<T>(dC, sC, tC, n * L, B, s));
    if (dKs) HIPCHK(hipMemcpyAsync(dKs, tK, sizeof(T) * (size_t)L * p, hipMemcpyDeviceToDevice, s));
    if (dCs) HIPCHK(hipMemcpyAsync(dCs, tC, sizeof(T) * (size_t)n * L, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    rls_fresh = false;  // the first online update continues from these accumulators
    return 0;
  }

  int check_lift_ready() {
    if (cfg.lift_kind == KMPC_LIFT_MLP) {
      for (size_t i = 0; i < layer_set.size(); ++i)
        if (!layer_set[i]) FAIL(-4, "encoder layer " + std::to_string(i) + " has not been set");
    } else if (!centres_set) {
      FAIL(-4, "RBF centres have not been set");
    }
    return 0;
  }

  int lift_to(const T* X, T* Psi, long ps_l, long ps_b, int Bc, hipStream_t s) {
    int rc = check_lift_ready();
    if (rc) return rc;
    LiftArgs<T> a{};
    a.B = Bc; a.n = n; a.L = L; a.hidden = hid; a.nlayers = cfg.layers;
    a.X = X; a.Psi = Psi; a.ps_l = ps_l; a.ps_b = ps_b;
    if (cfg.lift_kind == KMPC_LIFT_MLP) {
      a.W1 = dW1; a.b1 = db1; a.Wh[0] = dWh[0]; a.Wh[1] = dWh[1]; a.bh[0] = dbh[0]; a.bh[1] = dbh[1];
      a.Wo = dWo; a.bo = dbo; a.Hp = Hp; a.Lp = Lp;
      if ((rc = pack_encoder(s))) return rc;
      a.Whp[0] = dWhp[0]; a.Whp[1] = dWhp[1] ? dWhp[1] : dWhp[0]; a.Wop = dWop; a.KSp = (hid + 3) / 4;
      HIPCHK(launch_lift_mlp<T>(a, s));
    } else {
      a.cx = dcx; a.eps = (T)cfg.rbf_eps; a.rbf_matlab = cfg.lift_kind == KMPC_LIFT_RBF_MATLAB;
      HIPCHK(launch_lift_rbf<T>(a, s));
    }
    return 0;
  }

  int lift(const void* X, void* Psi, int Bc, hipStream_t s) override {
    if (Bc == 0) return 0;  // empty batch: nothing to do
    if (Bc < 0 || !X || !Psi) FAIL(-3, "kmpc_lift: bad arguments");
    return lift_to((const T*)X, (T*)Psi, Bc, 1, Bc, s);
  }

  StepArgs<T> base_args(int Bc) {
    StepArgs<T> a{};
    a.B = Bc; a.n = n; a.L = L; a.q = q; a.N = N;
    a.out_kind = cfg.output_kind == KMPC_OUT_LIFT ? OUT_LIFT : OUT_CX;
    a.max_iter = cfg.qp_max_iter > 0 ? cfg.qp_max_iter : 8 * N + 40;
    a.P = dP; a.strideP = sP; a.K = dK; a.strideK = sK; a.Qb = dQ; a.strideQ = sQ; a.C = dC; a.strideC = sC;
    a.Wterm = have_wterm ? (wterm_from_dare ? (T*)dWtB : dWt) : nullptr;
    a.wterm_per_traj = (have_wterm && wterm_from_dare && wterm_per_traj) ? 1 : 0;
    a.lam = (T)cfg.lambda; a.Qw = (T)cfg.Qw; a.Rw = (T)cfg.Rw; a.lb = (T)cfg.lb; a.ub = (T)cfg.ub;
    a.du_mode = cfg.delta_u ? 1 : 0;
    a.cy0 = cfg.output_kind == KMPC_OUT_LIFT ? 0 : cfg.out_row0;
    a.c_skip_first = cfg.c_skip_first ? 1 : 0;
    a.umin = (T)cfg.umin; a.umax = (T)cfg.umax;
    a.u_prev = dUprev;  // delta-u reads the absolute previous input in every phase
    a.qp_scratch = dQpScr;
    static const int predict = dbg_env("KMPC_QP_PREDICT") ? atoi(dbg_env("KMPC_QP_PREDICT")) : 1;  // measurement aid: 0 = plain projected Newton
    a.qp_predict = predict;
    a.plant = -1;
    return a;
  }

  int rls_update(const void* psi, const void* u, const void* psin, const void* xn, int Bc, hipStream_t s) override {
    if (Bc != B) FAIL(-3, "kmpc_rls_update: B must equal the handle's batch (the state is per trajectory)");
    if (!psi || !u || !psin || !xn) FAIL(-3, "kmpc_rls_update: null pointer");
    { int rc = ensure_dense(s, true); if (rc) return rc; }
    StepArgs<T> a = base_args(Bc);
    a.phases = PH_RLS;
    a.first_update = rls_fresh ? 1 : 0;
    a.psi_prev = (const T*)psi; a.pp_sl = Bc; a.pp_sb = 1;
    a.psi_now = (const T*)psin; a.pn_sl = Bc; a.pn_sb = 1;
    a.u_prev = (const T*)u; a.x_now = (const T*)xn;
    HIPCHK(launch_step<T>(a, threads, s));
    rls_fresh = false;
    return 0;
  }

  int get_model(void* A, void* Bm, void* C, hipStream_t s) override {
    { int rc = ensure_dense(s, false); if (rc) return rc; }
    HIPCHK(launch_export_model<T>(dK, sK, cfg.output_kind == KMPC_OUT_CX ? dC : nullptr, sC, n, L, B, (T*)A, (T*)Bm,
                                  (T*)C, s));
    return 0;
  }

  int condense(const void* psi, const void* ref, int rpt, void* H, void* f, void* c, int Bc, hipStream_t s) override {
    if (Bc != B) FAIL(-3, "kmpc_condense: B must equal the handle's batch");
    if (!psi || !ref || !H || !f) FAIL(-3, "kmpc_condense: null pointer");
    { int rc = ensure_dense(s, false); if (rc) return rc; }
    StepArgs<T> a = base_args(Bc);
    a.phases = PH_CONDENSE;
    a.psi_now = (const T*)psi; a.pn_sl = Bc; a.pn_sb = 1;
    a.ref = (const T*)ref; a.ref_per_traj = rpt;
    a.H_out = (T*)H; a.f_out = (T*)f; a.c_out = (T*)c;
    HIPCHK(launch_step<T>(a, threads, s));
    return 0;
  }

  // kmpc_mpc_solve: the solve wrapper with the model as an argument; scratch model blocks, H, f, c are the handle's
  T *dMsK = nullptr, *dMsC = nullptr, *dMsH = nullptr, *dMsf = nullptr, *dMsc = nullptr, *dMsW = nullptr;
  int mpc_solve(const void* A, const void* Bv, const void* C, int shared, const void* psi, const void* ref, int rpt,
                double lb_, double ub_, double Qw_, double Rw_, const double* PN, void* U, void* U0, void* fun,
                int32_t* st, int32_t* it, int Bc, hipStream_t s) override {
    if (Bc != B) FAIL(-3, "kmpc_mpc_solve: B must equal the handle's batch");
    if (!A || !Bv || !psi || !ref || !U) FAIL(-3, "kmpc_mpc_solve: null pointer");
    if (cfg.output_kind == KMPC_OUT_CX && !C) FAIL(-3, "kmpc_mpc_solve: C is required for KMPC_OUT_CX");
    if (cfg.delta_u) FAIL(-3, "kmpc_mpc_solve: not available in the delta-u form (its state includes the handle's u_prev)");
    if (!(ub_ > lb_)) FAIL(-3, "kmpc_mpc_solve: need lb < ub");
    if (!dMsK) {
      HIPCHK(hipMalloc(&dMsK, sizeof(T) * (size_t)sK * B));
      HIPCHK(hipMalloc(&dMsC, sizeof(T) * (size_t)sC * B));
      HIPCHK(hipMalloc(&dMsH, sizeof(T) * (size_t)B * N * N));
      HIPCHK(hipMalloc(&dMsf, sizeof(T) * (size_t)B * N));
      HIPCHK(hipMalloc(&dMsc, sizeof(T) * (size_t)B));
      HIPCHK(hipMalloc(&dMsW, sizeof(T) * (size_t)q * q));
    }
    HIPCHK(launch_import_model<T>((const T*)A, (const T*)Bv, (const T*)C, shared, n, L, B, dMsK, sK,
                                  cfg.output_kind == KMPC_OUT_CX ? dMsC : nullptr, sC, s));
    if (PN) {
      std::vector<T> w((size_t)q * q);
      for (int r = 0; r < q; ++r)
        for (int c2 = 0; c2 < q; ++c2) w[(size_t)r * q + c2] = (T)(PN[(size_t)r * q + c2] - (r == c2 ? Qw_ : 0.0));
      HIPCHK(hipMemcpyAsync(dMsW, w.data(), w.size() * sizeof(T), hipMemcpyHostToDevice, s));
      HIPCHK(hipStreamSynchronize(s));  // (the host vector goes out of scope)
    }
    StepArgs<T> a = base_args(B);
    a.K = dMsK; a.C = dMsC;
    a.lb = (T)lb_; a.ub = (T)ub_; a.Qw = (T)Qw_; a.Rw = (T)Rw_;
    // the terminal block: P_N of this call, else the handle's (kmpc_set_terminal_weight / kmpc_terminal_from_dare), as
    // kmpc_condense and kmpc_step use it -- the same cost on every route through the handle
    if (PN) { a.Wterm = dMsW; a.wterm_per_traj = 0; }
    else if (have_wterm && Qw_ != cfg.Qw) {
      if (wterm_from_dare || hostPN.empty()) FAIL(-3, "kmpc_mpc_solve: the handle's terminal block was formed with the handle's Q; pass P_N with another Q");
      std::vector<T> w((size_t)q * q);
      for (int r = 0; r < q; ++r)
        for (int c2 = 0; c2 < q; ++c2) w[(size_t)r * q + c2] = (T)(hostPN[(size_t)r * q + c2] - (r == c2 ? Qw_ : 0.0));
      HIPCHK(hipMemcpyAsync(dMsW, w.data(), w.size() * sizeof(T), hipMemcpyHostToDevice, s));
      HIPCHK(hipStreamSynchronize(s));
      a.Wterm = dMsW; a.wterm_per_traj = 0;
    }
    a.phases = PH_CONDENSE | PH_QP;
    a.psi_now = (const T*)psi; a.pn_sl = B; a.pn_sb = 1;
    a.ref = (const T*)ref; a.ref_per_traj = rpt;
    a.H_out = fun ? dMsH : nullptr; a.f_out = fun ? dMsf : nullptr; a.c_out = fun ? dMsc : nullptr;
    a.Useq = (T*)U; a.U0 = (T*)U0; a.status = st; a.iters = it;
    HIPCHK(launch_step<T>(a, threads, s));
    if (fun) HIPCHK(launch_qp_value<T>(dMsH, dMsf, dMsc, (const T*)U, N, B, (T*)fun, s));
    return 0;
  }

  int generate_and_fit(int plant, const void* X0, const void* U, int n_traj, int n_steps, double hs, double ridge,
                       int init_rls, void* A, void* Bm, void* C, void* Xo, void* Yo, hipStream_t s) override {
    if (!X0 || !U || n_traj < 1 || n_steps < 1) FAIL(-3, "kmpc_generate_and_fit: bad arguments");
    if (n != 2) FAIL(-3, "plants are two-state systems");
    if (!plant_id_ok(plant)) FAIL(-3, "unknown plant");
    const size_t M = (size_t)n_traj * n_steps;
    T *gx = nullptr, *gy = nullptr;
    if (!Xo) HIPCHK(hipMalloc(&gx, sizeof(T) * 2 * M));
    if (!Yo) { hipError_t e = hipMalloc(&gy, sizeof(T) * 2 * M); if (e != hipSuccess) { if (gx) (void)hipFree(gx); HIPCHK(e); } }
    T* Xp = Xo ? (T*)Xo : gx; T* Yp = Yo ? (T*)Yo : gy;
    int rc = 0;
    hipError_t e = launch_datagen<T>(plant, (T)hs, (const T*)X0, (const T*)U, n_traj, n_steps, Xp, Yp, s);
    if (e != hipSuccess) { err = std::string("kmpc_generate_and_fit: ") + hipGetErrorString(e); rc = -(int)(1000 + (int)e); }
    if (!rc) rc = offline_fit(Xp, Yp, U, (int)M, ridge, init_rls, A, Bm, C, s);  // (synchronises the stream before it returns)
    if (gx) (void)hipFree(gx);
    if (gy) (void)hipFree(gy);
    return rc;
  }

  int qp_solve(const void* H, const void* f, void* U, int32_t* st, int32_t* it, int Bc, hipStream_t s) override {
    if (Bc == 0) return 0;  // empty batch: nothing to do
    if (Bc < 0 || !H || !f || !U) FAIL(-3, "kmpc_qp_solve: bad arguments");
    if (cfg.delta_u && Bc != B) FAIL(-3, "kmpc_qp_solve: with delta_u the batch must equal the handle's (per-trajectory u_prev)");
    if (Bc > qp_scr_cap) {  // (a stateless solve may bring a larger batch than the handle's)
      HIPCHK(hipStreamSynchronize(s));
      (void)hipFree(dQpScr);
      dQpScr = nullptr; qp_scr_cap = 0;
      HIPCHK(hipMalloc(&dQpScr, sizeof(T) * (size_t)Bc * N * N));
      qp_scr_cap = Bc;
    }
    StepArgs<T> a = base_args(Bc);
    a.phases = PH_QP;
    a.H_in = (const T*)H; a.f_in = (const T*)f;
    a.Useq = (T*)U; a.status = st; a.iters = it;
    HIPCHK(launch_step<T>(a, threads, s));
    return 0;
  }

  int step(const void* X, const void* ref, int rpt, void* U0, void* Useq, int32_t* st, int32_t* it,
           hipStream_t s) override {
    if (!X || !ref || !U0) FAIL(-3, "kmpc_step: null pointer");
    // Configurations with a fused roll-out instantiation and the MLP lift take ONE launch for the step as well: the
    // roll-out kernel with a single step and no plant (encoder inside on MFMA, no separate lift kernel, psi handed
    // over in registers).  Same results up to the summation order of the encoder.
    if (sizeof(T) == 8 && fuse_plant < 0 && !accumulate && cfg.lift_kind == KMPC_LIFT_MLP && fused_rollout_ok()) {
      static const bool two = dbg_env("KMPC_STEP_TWO_KERNELS") != nullptr;  // measurement aid
      if (!two)
        return rollout_fused(-1, const_cast<void*>(X), ref, rpt, 1, 0, -1, 0.05, nullptr, nullptr, st, it, s, U0, Useq);
    }
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    const bool rec = prof && ev_used + 3 <= EV_CAP;
    if (rec) {
      while (ev.size() < ev_used + 3) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        ev.push_back(e);
      }
      e0 = ev[ev_used]; e1 = ev[ev_used + 1]; e2 = ev[ev_used + 2];
      HIPCHK(hipEventRecord(e0, s));
    }
    int rc = ensure_dense(s, true);  // (first: on a float32 handle the ping-pong index follows the core's)
    if (rc) return rc;
    T* psi_now = dPsi[cur];
    T* psi_prev = dPsi[cur ^ 1];
    if ((rc = lift_to((const T*)X, psi_now, 1, L, B, s))) return rc;
    if (rec) HIPCHK(hipEventRecord(e1, s));
    StepArgs<T> a = base_args(B);
    a.phases = PH_CONDENSE | PH_QP | ((have_prev && update_on) ? PH_RLS : 0);
    a.first_update = rls_fresh ? 1 : 0;
    a.psi_prev = psi_prev; a.pp_sl = 1; a.pp_sb = L;
    a.psi_now = psi_now; a.pn_sl = 1; a.pn_sb = L;
    a.u_prev = dUprev; a.x_now = (const T*)X;
    a.ref = (const T*)ref; a.ref_per_traj = rpt;
    a.U0 = (T*)U0; a.Useq = (T*)Useq; a.u_store = dUprev; a.status = st; a.iters = it;
    a.x_warm = cfg.cold_start ? nullptr : dWarm;
    a.qp_carry = dQpCarry; a.qp_carry_set = dQpCarrySet;
    a.accumulate = accumulate ? 1 : 0;
    if (fuse_plant >= 0) { a.plant = fuse_plant; a.plant_switched = fuse_switched; a.plant_h = (T)fuse_h; a.X_rw = (T*)const_cast<void*>(X); }
    if (term_every > 0) {
      // per-step terminal refresh on the per-step route: the RLS update as a launch of its own, the Riccati iteration on the updated
      // models (dare_kernel, every term_every-th step), then condensed QP and solve with the refreshed blocks
      const int ph = a.phases;
      if (ph & PH_RLS) {
        StepArgs<T> a1 = a;
        a1.phases = PH_RLS; a1.plant = -1; a1.U0 = nullptr; a1.Useq = nullptr; a1.u_store = nullptr; a1.status = nullptr; a1.iters = nullptr;
        HIPCHK(launch_step<T>(a1, threads, s));
      }
      if (term_count % term_every == 0) { const int rc2 = dare_blocks(B, dTermQ, term_R, term_maxiter, term_eps, s); if (rc2) return rc2; }
      a.phases = ph & ~PH_RLS;
      a.Wterm = (T*)dWtB; a.wterm_per_traj = 1;
      term_count += 1;
    }
    HIPCHK(launch_step<T>(a, threads, s));
    if (rec) {
      HIPCHK(hipEventRecord(e2, s));
      ev_used += 3;
      prof_steps += 1;
    }
    if (have_prev && update_on) rls_fresh = false;
    have_prev = true;
    cur ^= 1;
    return 0;
  }

  // the input that was actually applied at the last step, when it is not the one kmpc_step returned (actuator limits, a
  // logged trajectory that is being followed): the next RLS update regresses on it (z = [psi; u], duffing.py:900)
  // the loop WITHOUT the online update (duffing.py:738-805, vanderpol.py:645-722): kmpc_step / kmpc_rollout skip the RLS phase and
  // solve with the model as it is; the lift / input history keeps running, so that switching the update back on continues
  bool update_on = true;
  int set_online_update(int on) override { update_on = on != 0; if (core) core->update_on = update_on; return 0; }
  int set_applied_input(const void* U, int Bc, hipStream_t s) override {
    if (!U || Bc != B) FAIL(-3, "kmpc_set_applied_input: U must hold one input per trajectory of the handle");
    if (core) { const int rc = ensure_dense(s, true); if (rc) return rc; }  // (u_{k-1} is part of what the core owns during roll-outs)
    HIPCHK(hipMemcpyAsync(dUprev, U, sizeof(T) * (size_t)B, hipMemcpyDeviceToDevice, s));
    return 0;
  }

  int plant_step(int plant, void* X, const void* U, double h, int sw, int Bc, hipStream_t s) override {
    if (n != 2) FAIL(-3, "plants are two-state systems");
    if (!plant_id_ok(plant)) FAIL(-3, "unknown plant");
    PlantArgs<T> a{};
    a.B = Bc; a.plant = plant; a.switched = sw; a.h = (T)h; a.X = (T*)X; a.U = (const T*)U;
    HIPCHK(launch_plant<T>(a, s));
    return 0;
  }

  // ---- fused roll-out (one launch for all the steps; rollout_kernel in step_kernel.hip) ----------------
  T *dWhp[2] = {nullptr, nullptr}, *dWop = nullptr;  // hidden / output weights as MFMA A-fragments
  bool packed_ok = false;
  int64_t prof_steps = 0;   // profiling: control steps covered by the recorded events
  int pack_encoder(hipStream_t s) {
    if (packed_ok) return 0;
    const int KS = (hid + 3) / 4;
    const int nhh = cfg.layers - 1;
    for (int k = 0; k < nhh; ++k) {
      if (!dWhp[k]) HIPCHK(hipMalloc(&dWhp[k], sizeof(T) * (size_t)(Hp / 16) * KS * 64));
      HIPCHK(launch_pack_afrag<T>(dWh[k], Hp, Hp, KS, dWhp[k], s));
    }
    if (!dWop) HIPCHK(hipMalloc(&dWop, sizeof(T) * (size_t)(Lp / 16) * KS * 64));
    HIPCHK(launch_pack_afrag<T>(dWo, Lp, Hp, KS, dWop, s));
    packed_ok = true;
    return 0;
  }
  int rollout_is_fused() const override { return fused_rollout_ok() ? 1 : 0; }
  bool fused_rollout_ok() const {
    static const bool off = dbg_env("KMPC_NO_FUSED_ROLLOUT") != nullptr;  // measurement aid: per-step launches
    if (core) return !off && core->fused_rollout_ok();  // (float32 panels around the float64 roll-out)
    if (term_every > 0 && term_plugin_state <= 0) return false;  // (the refresh needs the TERM plug-in; else the per-step route)
    return !off && n == 2 && plugin_state >= 0 && rollout_fused_available<T>(n, L, N, q, threads, cfg.lift_kind != KMPC_LIFT_MLP);
  }
  // 0: the library's own instantiation, 1: a plug-in (text: its file and whether it was compiled now or found in the kernel cache),
  // -1: the plug-in could not be made (text: why; the handle works with per-step launches), 2: this configuration has no fused roll-out
  int rollout_plugin_status(std::string* text) const override {
    if (term_every > 0 && term_plugin_state > 0) { if (text) *text = rollout_plugin_describe(term_key); return 1; }
    if (term_every > 0 && term_plugin_state < 0) { if (text) *text = term_msg; return -1; }
    if (plugin_state > 0) { if (text) *text = rollout_plugin_describe(plugin_key); return 1; }
    if (plugin_state < 0) { if (text) *text = plugin_msg; return -1; }
    if (!fused_rollout_ok()) { if (text) *text = "no fused roll-out for this configuration (per-step launches)"; return 2; }
    if (text) *text = "built-in instantiation of libkoopmpc.so";
    return 0;
  }
  // io32: the caller-owned panels (X, ref, Ulog, Xlog, U0out, Useqout) are float32 -- the call comes from a KMPC_F32 handle's core
  int rollout_fused(int plant, void* X, const void* ref, int rpt, int steps, int step0, int switch_step, double hs,
                    void* Ulog, void* Xlog, int32_t* st, int32_t* it, hipStream_t s, void* U0out = nullptr,
                    void* Useqout = nullptr, bool io32 = false) {
    int rc = check_lift_ready();
    if (rc) return rc;
    if constexpr (sizeof(T) == 4) {
      if (core) {
        if ((rc = ensure_image(s))) return rc;
        core->prof = prof;
        rc = core->rollout_fused(plant, X, ref, rpt, steps, step0, switch_step, hs, Ulog, Xlog, st, it, s, U0out, Useqout, true);
        if (rc) { err = core->err; return rc; }
        have_prev = core->have_prev; rls_fresh = core->rls_fresh; cur = core->cur;
        return 0;
      }
    }
    RolloutArgs<T> r{};
    r.s = base_args(B);
    r.s.u_prev = dUprev; r.s.x_now = (const T*)X;
    r.s.ref = (const T*)ref; r.s.ref_per_traj = rpt;
    r.s.U0 = U0out ? (T*)U0out : dU0; r.s.Useq = (T*)Useqout; r.s.u_store = dUprev; r.s.status = st; r.s.iters = it;
    r.s.x_warm = cfg.cold_start ? nullptr : dWarm;
    r.s.plant = plant; r.s.plant_h = (T)hs; r.s.X_rw = (T*)X;
    r.s.pp_sl = 1; r.s.pp_sb = L; r.s.pn_sl = 1; r.s.pn_sb = L;
    r.s.accumulate = 1;
    if (cfg.lift_kind == KMPC_LIFT_MLP) {
      if ((rc = pack_encoder(s))) return rc;
      r.lift_rbf = 0;
      r.W1 = dW1; r.b1 = db1; r.Whp[0] = dWhp[0]; r.Whp[1] = dWhp[1]; r.bh[0] = dbh[0]; r.bh[1] = dbh[1];
      r.Wop = dWop; r.bo = dbo; r.Hp = Hp; r.Lp = Lp; r.KS = (hid + 3) / 4; r.nhh = cfg.layers - 1;
    } else {
      r.lift_rbf = 1;
      r.cx = dcx; r.eps = (T)cfg.rbf_eps; r.rbf_matlab = cfg.lift_kind == KMPC_LIFT_RBF_MATLAB ? 1 : 0;
    }
    r.psi[0] = dPsi[0]; r.psi[1] = dPsi[1]; r.cur = cur;
    r.steps = steps; r.step0 = step0; r.switch_step = switch_step;
    r.have_prev = have_prev ? 1 : 0; r.rls_fresh = rls_fresh ? 1 : 0;
    r.no_update = update_on ? 0 : 1;
    r.U_log = (T*)Ulog; r.X_log = (T*)Xlog;
    r.io_f32 = io32 ? 1 : 0;
    if (io32 && !r.s.U0) r.s.U0 = dU0;
    if constexpr (sizeof(T) == 8) {
      if (term_every > 0) {
        if (io32) FAIL(-3, "the per-step terminal refresh is a float64 feature");
        r.term_every = term_every; r.term_count0 = (int)(term_count % term_every); r.term_maxiter = term_maxiter;
        r.term_R = term_R; r.term_eps = term_eps; r.term_Q = dTermQ; r.term_W = dWtB;
        r.term_scratch = dTermScr; r.term_scratch_stride = term_scratch_elems(L); r.term_iters = dDareIt;
        r.s.Wterm = dWtB; r.s.wterm_per_traj = 1;
      }
    }
    if (!dbg_env("KMPC_ROLLOUT_NO_PLACE")) {  // (the trajectories' solver work of the previous launch: RolloutArgs::work)
      if (!dWork) {
        HIPCHK(hipMalloc(&dWork, sizeof(int32_t) * (size_t)B));
        HIPCHK(hipMemsetAsync(dWork, 0, sizeof(int32_t) * (size_t)B, s));
      }
      r.work = dWork;
      static const int tail = dbg_env("KMPC_PLACE_TAIL") ? atoi(dbg_env("KMPC_PLACE_TAIL")) : 5;  // (measurement aid; 0 = the whole launch)
      r.work_tail = tail;
      // ... across the workgroups too where the batch allows the card deal (whole workgroups; the rank kernel is O(B^2 / lanes))
      if (place_valid && cfg.lift_kind == KMPC_LIFT_MLP && (B & 15) == 0 && B <= (1 << 20) && !dbg_env("KMPC_ROLLOUT_NO_GLOBAL_PLACE")) r.perm = dPerm;
      // (work and perm were written on the stream of the previous call: a call on another stream waits for them -- a torn perm would
      //  hand one trajectory to two waves and skip another: ADVICE r4)
      if (evPlace && place_stream != s) HIPCHK(hipStreamWaitEvent(s, evPlace, 0));
    }
    if constexpr (sizeof(T) == 8) {
      if (use_img) {
        if ((rc = ensure_image(s))) return rc;
        r.img = dImg; r.img_stride = sImg;
      } else if ((rc = ensure_dense(s, true))) return rc;
    }
    const bool rec = prof && ev_used + 3 <= EV_CAP;
    if (rec) {
      while (ev.size() < ev_used + 3) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        ev.push_back(e);
      }
      HIPCHK(hipEventRecord(ev[ev_used], s));
      HIPCHK(hipEventRecord(ev[ev_used + 1], s));  // (no separate lift kernel)
    }
    HIPCHK(launch_rollout_fused<T>(r, s));
    // (profiling: the closing event sits right behind the roll-out kernel -- until round 5 it sat behind the rank pass below, whose
    //  9 us at B = 4096 were counted into "the dominant kernel's duration": rocprofv3 0.630 ms against 0.647 ms from these events)
    if (rec) {
      HIPCHK(hipEventRecord(ev[ev_used + 2], s));
      ev_used += 3;
      prof_steps += steps;
    }
    if (r.work && steps >= 4 && cfg.lift_kind == KMPC_LIFT_MLP && (B & 15) == 0 && B <= (1 << 20)) {
      if (!dPerm) HIPCHK(hipMalloc(&dPerm, sizeof(int32_t) * (size_t)B * 3));  // (perm, then the sort's two ping-pong buffers)
      HIPCHK(launch_place(dWork, B, dPerm, reinterpret_cast<uint32_t*>(dPerm + B), s));  // (the next launch's deal: RolloutArgs::perm)
      place_valid = true;
    } else {
      place_valid = false;
    }
    if (r.work) {
      if (!evPlace) HIPCHK(hipEventCreateWithFlags(&evPlace, hipEventDisableTiming));
      HIPCHK(hipEventRecord(evPlace, s));
      place_stream = s;
    }
    if (term_every > 0) term_count += steps;
    // the host-side flags follow the kernel's own bookkeeping
    if (update_on && (steps >= 2 || (steps == 1 && have_prev))) rls_fresh = false;
    have_prev = true;
    cur ^= (steps & 1);
    return 0;
  }

  T* dU0 = nullptr;  // rollout scratch [B]
  bool accumulate = false;
  int fuse_plant = -1, fuse_switched = 0;  // rollouts: the step kernel advances the plant itself
  double fuse_h = 0.05;
  void* fuse_X = nullptr;  // (shared-model solve with the plant inside)
  int rollout(int plant, void* X, const void* ref, int rpt, int steps, int step0, int switch_step, double hs,
              void* Ulog, void* Xlog, int32_t* st, int32_t* it, hipStream_t s) override {
    if (!X || !ref || steps < 0) FAIL(-3, "kmpc_rollout: bad arguments");
    if (!dU0) HIPCHK(hipMalloc(&dU0, sizeof(T) * (size_t)B));
    if (n != 2) FAIL(-3, "plants are two-state systems");
    if (!plant_id_ok(plant)) FAIL(-3, "unknown plant");
    // (the fused roll-out with the MLP lift accumulates status / iterations in LDS and WRITES them when it ends: no zeroing
    //  launches in front of it -- two of the ~70 us a 20-step call spends outside its kernel)
    const bool kernel_writes_totals = steps > 0 && fused_rollout_ok() && cfg.lift_kind == KMPC_LIFT_MLP;
    if (!kernel_writes_totals) {
      if (st) HIPCHK(hipMemsetAsync(st, 0, sizeof(int32_t) * (size_t)B, s));
      if (it) HIPCHK(hipMemsetAsync(it, 0, sizeof(int32_t) * (size_t)B, s));
    }
    if (steps > 0 && fused_rollout_ok())
      return rollout_fused(plant, X, ref, rpt, steps, step0, switch_step, hs, Ulog, Xlog, st, it, s);
    if (steps == 0 && fused_rollout_ok()) {
      // a call without steps brings the handle into the form the fused roll-out works on (the wave image of a state that was
      // set, restored or stepped through the dense blocks): set-up that the next kmpc_rollout would otherwise pay
      if constexpr (sizeof(T) == 8) {
        if (use_img) return ensure_image(s);
      } else if (core) {
        int rc = ensure_image(s);
        if (!rc && core->use_img && (rc = core->ensure_image(s))) err = core->err;
        return rc;
      }
      return 0;
    }
    for (int i = 0; i < steps; ++i) {
      const int gi = step0 + i;
      T* u = Ulog ? (T*)Ulog + (size_t)i * B : dU0;
      if (n != 2) FAIL(-3, "plants are two-state systems");
      if (!plant_id_ok(plant)) FAIL(-3, "unknown plant");
      accumulate = true;
      fuse_plant = plant; fuse_switched = (switch_step >= 0 && gi >= switch_step) ? 1 : 0; fuse_h = hs;
      int rc = step(X, ref, rpt, u, nullptr, st, it, s);  // lift kernel + fused RLS/condense/QP/plant kernel
      accumulate = false;
      fuse_plant = -1;
      if (rc) return rc;
      if (Xlog) HIPCHK(hipMemcpyAsync((T*)Xlog + (size_t)i * n * B, X, sizeof(T) * (size_t)n * B,
                                      hipMemcpyDeviceToDevice, s));
    }
    return 0;
  }

  // ---- shared-model mode -----------------------------------------------------------------------------
  double *dGram = nullptr, *dPartial = nullptr;
  T *dKs = nullptr, *dCs = nullptr, *dHs = nullptr, *dFs = nullptr, *df0s = nullptr, *dTs = nullptr;
  double* dFastPack = nullptr;   // F, T0, H of the shared model as shared_fast_kernel's operand fragments (SharedModel2Args::pack)
  bool fast_pack_valid = false;  // ... written by the last model launch
  int32_t* dNeed = nullptr;  // [B] flags of shared_fast_kernel
  int32_t* dWork = nullptr;  // [B] solver work of every trajectory in the last fused launch (RolloutArgs::work)
  int32_t* dPerm = nullptr;  // [B] slot -> trajectory for the next fused launch (RolloutArgs::perm); valid after a launch that wrote dWork
  bool place_valid = false;
  hipEvent_t evPlace = nullptr;      // recorded behind the launch (and the rank pass) that wrote dWork / dPerm
  hipStream_t place_stream = nullptr;
  int32_t* dQpList = nullptr;  // two alternating counters, then the list of the trajectories shared_fast_kernel left to the solve-only kernel
  int qp_list_parity = 0;
  T* dWt = nullptr;  // PN - Qw I (terminal block of Q_bar)
  std::vector<double> hostPN;  // P_N as given to kmpc_set_terminal_weight
  bool have_wterm = false;
  // kmpc_terminal_from_dare: Riccati workspace and the block(s) it produced ([1] or [B][q*q], PN - Qw I)
  double* dDareP = nullptr; int32_t* dDareIt = nullptr; double* dWtB = nullptr;
  int dare_cap = 0;
  bool wterm_from_dare = false, wterm_per_traj = false;
  bool shared_has_samples = false;
  static constexpr int GRAM_BLOCKS = 256;  // partial blocks of the Gram kernels (512 measured slower: lift + Gram 21.8 vs 17.2 us, reduce 7.6 vs 4.8 us)
  int64_t gram_elems() const override { return (int64_t)(p + L + n) * p; }
  int shared_alloc() {
    if (dGram) return 0;
    const int R = p + L + n, MT = (R + 15) / 16, NT = (p + 15) / 16;
    HIPCHK(hipMalloc(&dGram, sizeof(double) * (size_t)R * p));
    HIPCHK(hipMemset(dGram, 0, sizeof(double) * (size_t)R * p));
    HIPCHK(hipMalloc(&dPartial, sizeof(double) * (size_t)GRAM_BLOCKS * MT * 16 * NT * 16));
    HIPCHK(hipMalloc(&dKs, sizeof(T) * (size_t)L * p));
    HIPCHK(hipMalloc(&dCs, sizeof(T) * (size_t)n * L));
    HIPCHK(hipMalloc(&dHs, sizeof(T) * (size_t)N * N));
    HIPCHK(hipMalloc(&dFs, sizeof(T) * (size_t)N * (L + 1)));  // (one more column in the delta-u form)
    HIPCHK(hipMalloc(&df0s, sizeof(T) * (size_t)N));
    // until samples exist the shared model is the offline one (trajectory 0's copy, duffing.py:811-813)
    { int rc = sync_own(); if (rc) return rc; }  // (before the conversion: non-blocking streams do not order with the null stream)
    { int rc = ensure_dense(nullptr, false); if (rc) return rc; HIPCHK(hipStreamSynchronize(nullptr)); }
    HIPCHK(hipMemcpy(dKs, dK, sizeof(T) * (size_t)L * p, hipMemcpyDeviceToDevice));
    HIPCHK(hipMemcpy(dCs, dC, sizeof(T) * (size_t)n * L, hipMemcpyDeviceToDevice));
    return 0;
  }
  int shared_local_gram(const void* X, double* delta, hipStream_t s) override {
    if (!X || !delta) FAIL(-3, "kmpc_shared_local_gram: null pointer");
    int rc = shared_alloc();
    if (rc) return rc;
    if (core && (rc = ensure_dense(s, true))) return rc;  // (psi, u_{k-1} and the ping-pong index are this side's from here on)
    T* psi_now = dPsi[cur];
    T* psi_prev = dPsi[cur ^ 1];
    GramArgs<T> g{};
    g.B = B; g.n = n; g.L = L; g.max_blocks = GRAM_BLOCKS;
    g.psi_prev = psi_prev; g.pp_sl = 1; g.pp_sb = L;
    g.psi_now = psi_now; g.pn_sl = 1; g.pn_sb = L;
    g.u_prev = dUprev; g.x_now = (const T*)X; g.partial = dPartial;
    if constexpr (sizeof(T) == 8) {
      // float64 MLP lift: the lift and the Gram sums of the transitions in ONE launch (lift_coop_kernel<.., GRAM>), then the reduce
      const bool two = dbg_env("KMPC_SHARED_LIFT_GRAM_2") != nullptr;  // measurement / test aid (read per call): the two round-3 launches
      if (have_prev && !two && cfg.lift_kind == KMPC_LIFT_MLP) {
        if ((rc = check_lift_ready())) return rc;
        LiftArgs<T> a{};
        a.B = B; a.n = n; a.L = L; a.hidden = hid; a.nlayers = cfg.layers;
        a.X = (const T*)X; a.Psi = psi_now; a.ps_l = 1; a.ps_b = L;
        a.W1 = dW1; a.b1 = db1; a.Wh[0] = dWh[0]; a.Wh[1] = dWh[1]; a.bh[0] = dbh[0]; a.bh[1] = dbh[1];
        a.Wo = dWo; a.bo = dbo; a.Hp = Hp; a.Lp = Lp;
        if ((rc = pack_encoder(s))) return rc;
        a.Whp[0] = dWhp[0]; a.Whp[1] = dWhp[1] ? dWhp[1] : dWhp[0]; a.Wop = dWop; a.KSp = (hid + 3) / 4;
        if (lift_gram_available(a)) {
          int nb = 0;
          HIPCHK(launch_lift_gram(a, g, &nb, s));
          HIPCHK(launch_gram_reduce(dPartial, nb, L, n, 0.0, delta, s));  // (forget = 0 overwrites delta)
          return 0;
        }
      }
    }
    rc = lift_to((const T*)X, psi_now, 1, L, B, s);
    if (rc) return rc;
    if (!have_prev) {
      HIPCHK(hipMemsetAsync(delta, 0, sizeof(double) * (size_t)gram_elems(), s));
      return 0;
    }
    HIPCHK(launch_gram<T>(g, 0.0, delta, s));  // (forget = 0 overwrites delta)
    return 0;
  }
  int shared_solve_plant(const double* delta, const void* ref, void* U0, void* Useq, int32_t* st, int32_t* it, int plant, void* X,
                         int switched, double hstep, hipStream_t s) override {
    if (!X) FAIL(-3, "kmpc_shared_solve_plant: null pointer");
    if (n != 2) FAIL(-3, "plants are two-state systems");
    if (!plant_id_ok(plant)) FAIL(-3, "unknown plant");
    fuse_plant = plant; fuse_switched = switched ? 1 : 0; fuse_h = hstep; fuse_X = X;
    const int rc = shared_solve(delta, ref, U0, Useq, st, it, s);
    fuse_plant = -1; fuse_X = nullptr;
    return rc;
  }
  // kmpc_shared_rollout: the stages of the shared-model step, `steps` times, on one stream, the collective between them
  double* dDelta = nullptr;  // this rank's Gram sums of a step (all-reduced in place)
  int shared_rollout(int plant, void* X, const void* ref, int steps, int step0, int switch_step, double hstep, void* comm, void* Ulog,
                     void* Xlog, void* U0out, void* Useq, int32_t* st, int32_t* it, hipStream_t s) override {
    if (!X || !ref || steps < 0) FAIL(-3, "kmpc_shared_rollout: bad arguments");
    if (n != 2) FAIL(-3, "plants are two-state systems");
    if (!plant_id_ok(plant)) FAIL(-3, "unknown plant");
    int rc = shared_alloc();
    if (rc) return rc;
    if (!dDelta) HIPCHK(hipMalloc(&dDelta, sizeof(double) * (size_t)gram_elems()));
    if (!dU0) HIPCHK(hipMalloc(&dU0, sizeof(T) * (size_t)B));
    typedef ncclResult_t (*allreduce_fn)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    allreduce_fn ar = nullptr;
    if (comm) {
      ar = (allreduce_fn)resolve_nccl_allreduce();
      if (!ar) FAIL(-4, "kmpc_shared_rollout: no RCCL in this process (ncclAllReduce not found)");
    }
    if (st) HIPCHK(hipMemsetAsync(st, 0, sizeof(int32_t) * (size_t)B, s));
    if (it) HIPCHK(hipMemsetAsync(it, 0, sizeof(int32_t) * (size_t)B, s));
    for (int i = 0; i < steps; ++i) {
      const int gi = step0 + i;
      if ((rc = shared_local_gram(X, dDelta, s))) return rc;
      if (ar && have_prev) {  // (before the first transition every rank's block is zero: nothing to sum)
        const ncclResult_t nr = ar(dDelta, dDelta, (size_t)gram_elems(), ncclDouble, ncclSum, (ncclComm_t)comm, s);
        if (nr != ncclSuccess) FAIL(-(2000 + (int)nr), "kmpc_shared_rollout: ncclAllReduce failed");
      }
      T* const u = Ulog ? (T*)Ulog + (size_t)i * B : (U0out ? (T*)U0out : dU0);
      accumulate = true;
      rc = shared_solve_plant(dDelta, ref, u, Useq, st, it, plant, X, (switch_step >= 0 && gi >= switch_step) ? 1 : 0, hstep, s);
      accumulate = false;
      if (rc) return rc;
      if (Xlog) HIPCHK(hipMemcpyAsync((T*)Xlog + (size_t)i * n * B, X, sizeof(T) * (size_t)n * B, hipMemcpyDeviceToDevice, s));
    }
    if (Ulog && U0out && steps > 0)
      HIPCHK(hipMemcpyAsync(U0out, (T*)Ulog + (size_t)(steps - 1) * B, sizeof(T) * (size_t)B, hipMemcpyDeviceToDevice, s));
    return 0;
  }
  int shared_solve(const double* delta, const void* ref, void* U0, void* Useq, int32_t* st, int32_t* it,
                   hipStream_t s) override {
    if (!delta || !ref || !U0) FAIL(-3, "kmpc_shared_solve: null pointer");
    if (have_wterm && wterm_from_dare && wterm_per_traj)
      FAIL(-3, "kmpc_shared_solve: per-trajectory terminal blocks do not apply to the shared model (kmpc_terminal_from_dare with per_trajectory = 0)");
    int rc = shared_alloc();
    if (rc) return rc;
    if (core && (rc = ensure_dense(s, true))) return rc;
    const T* wt = !have_wterm ? nullptr : (wterm_from_dare ? (const T*)dWtB : dWt);
    const int okind = cfg.output_kind == KMPC_OUT_LIFT ? OUT_LIFT : OUT_CX;
    const int cy0 = cfg.output_kind == KMPC_OUT_LIFT ? 0 : cfg.out_row0;
    static const bool two_kernels = dbg_env("KMPC_SHARED_TWO_KERNELS") != nullptr;  // measurement aid: the round-1 launches
    const bool one_launch = !two_kernels && cfg.output_kind == KMPC_OUT_CX && shared_model_available(L, n, q, N, cfg.delta_u ? 1 : 0);
    if (have_prev && !one_launch) HIPCHK(launch_axpby(dGram, delta, cfg.lambda, (int)gram_elems(), s));
    if (one_launch) {
      // model solve + condensed QP + the tableau the box QPs start from, one launch (shared_model_kernel)
      if (!dTs) HIPCHK(hipMalloc(&dTs, sizeof(T) * (size_t)N * N));
      const bool old_kernel = dbg_env("KMPC_SHARED_MODEL_R3") != nullptr;  // measurement / test aid (read per call): the round-3 model kernel
      bool on16 = false;
      if constexpr (sizeof(T) == 8) {
        if (!old_kernel && shared_model2_available(L, n, q, N, cfg.delta_u ? 1 : 0)) {
          SharedModel2Args m{};
          m.gram = dGram; m.delta = have_prev ? delta : nullptr; m.forget = cfg.lambda; m.ref = (const double*)ref;
          m.Lm = L; m.n = n; m.q = q; m.N = N; m.dP = 1.0 / cfg.P0; m.dQ = 1.0 / cfg.barQ0; m.have_samples = have_prev ? 1 : 0;
          m.Qw = cfg.Qw; m.Rw = cfg.Rw;
          m.Kio = (double*)dKs; m.Cio = (double*)dCs; m.Hout = (double*)dHs; m.Fout = (double*)dFs; m.f0out = (double*)df0s; m.Tout = (double*)dTs;
          m.Wt = (const double*)wt; m.du_mode = cfg.delta_u ? 1 : 0; m.cy0 = cy0;
          // (F, T0, H once more as the operand fragments of shared_fast_kernel: written by the model kernel beside the dense blocks)
          int fmt = 0, fk1 = 0, fk2 = 0;
          if (N <= 64 && shared_fast_shape(N, L + (cfg.delta_u ? 1 : 0), &fmt, &fk1, &fk2)) {
            if (!dFastPack) {
              const size_t bytes = sizeof(double) * shared_fast_pack_elems(fmt, fk1, fk2);
              HIPCHK(hipMalloc(&dFastPack, bytes));
              HIPCHK(hipMemsetAsync(dFastPack, 0, bytes, s));  // (the padding of the fragments stays zero: only elements of the matrices are written)
            }
            m.pack = dFastPack; m.pack_mt = fmt; m.pack_ks1 = fk1; m.pack_ks2 = fk2;
          }
          HIPCHK(launch_shared_model2(m, s));
          on16 = true;
          fast_pack_valid = m.pack != nullptr;
        }
      }
      if (!on16) {
        fast_pack_valid = false;
        HIPCHK(launch_shared_model<T>(dGram, have_prev ? delta : nullptr, cfg.lambda, (const T*)ref, L, n, q, N, 1.0 / cfg.P0, 1.0 / cfg.barQ0, 1, have_prev ? 1 : 0, cfg.Qw,
                                      cfg.Rw, dKs, dCs, dHs, dFs, df0s, dTs, wt, cfg.delta_u ? 1 : 0, cy0, s));
      }
      if (have_prev) shared_has_samples = true;
    } else {
      if (have_prev) {
        HIPCHK(launch_shared_solve<T>(dGram, L, n, 1.0 / cfg.P0, 1.0 / cfg.barQ0,
                                      cfg.output_kind == KMPC_OUT_CX ? 1 : 0, dKs, dCs, s));
        shared_has_samples = true;
      }
      HIPCHK(launch_shared_condense<T>(dKs, dCs, (const T*)ref, L, n, q, N, okind, cfg.Qw, cfg.Rw, dHs, dFs, df0s, s, wt,
                                       cfg.delta_u ? 1 : 0, cy0));
    }
    StepArgs<T> a = base_args(B);
    a.phases = PH_QP;
    a.H_in = dHs; a.h_shared = 1; a.F_in = dFs; a.f0_in = df0s;
    a.T_in = one_launch ? dTs : nullptr;
    a.fast_pack = (one_launch && fast_pack_valid) ? dFastPack : nullptr;
    a.psi_now = dPsi[cur]; a.pn_sl = 1; a.pn_sb = L;
    a.U0 = (T*)U0; a.Useq = (T*)Useq; a.u_store = dUprev; a.status = st; a.iters = it;
    a.x_warm = cfg.cold_start ? nullptr : dWarm;
    a.accumulate = accumulate ? 1 : 0;  // (kmpc_shared_rollout: worst status / total Newton solves over the steps)
    // (kmpc_shared_solve_plant: the solve also advances every trajectory's plant with its u_k, as the roll-outs do)
    if (fuse_plant >= 0) { a.plant = fuse_plant; a.plant_switched = fuse_switched; a.plant_h = (T)fuse_h; a.X_rw = (T*)fuse_X; }
    const bool rec = prof && ev_used + 3 <= EV_CAP;  // (profiling: the QP launch of the shared-model step)
    if (rec) {
      while (ev.size() < ev_used + 3) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        ev.push_back(e);
      }
      HIPCHK(hipEventRecord(ev[ev_used], s));
      HIPCHK(hipEventRecord(ev[ev_used + 1], s));
    }
    // the trajectories whose unconstrained minimiser lies inside the box are finished 16 per wave on the matrix cores
    // (shared_fast_kernel); the solve-only kernel then only runs for the flagged rest
    const bool no_fast = dbg_env("KMPC_SHARED_NO_FAST") != nullptr;  // measurement / test aid (read per call): every trajectory through the QP kernel
    if (one_launch && sizeof(T) == 8 && N <= 64 && !no_fast) {
      if (!dNeed) HIPCHK(hipMalloc(&dNeed, sizeof(int32_t) * (size_t)B));
      // (the flagged trajectories also go into a list that a fixed grid of the solve-only kernel walks; two counters alternate)
      if (!dQpList) {
        HIPCHK(hipMalloc(&dQpList, sizeof(int32_t) * (size_t)(B + 2)));
        HIPCHK(hipMemsetAsync(dQpList, 0, sizeof(int32_t) * (size_t)(B + 2), s));
      }
      if (!dbg_env("KMPC_SHARED_NO_LIST")) {  // (measurement / test aid, read per call: one workgroup per trajectory, flags only)
        a.qp_list = dQpList + 2;
        a.qp_count = dQpList + qp_list_parity;
        a.qp_count_next = dQpList + (qp_list_parity ^ 1);
        qp_list_parity ^= 1;
      }
      HIPCHK(launch_shared_fast<T>(a, dNeed, s));
      a.qp_need = dNeed;
    }
    HIPCHK(launch_step<T>(a, threads, s));
    if (rec) {
      HIPCHK(hipEventRecord(ev[ev_used + 2], s));
      ev_used += 3;
      prof_steps += 1;
    }
    have_prev = true;
    cur ^= 1;
    return 0;
  }
  int offline_fit(const void* X, const void* Y, const void* U, int M, double ridge, int init_rls, void* A, void* Bm,
                  void* C, hipStream_t s) override {
    if (!X || !Y || !U || M < 1) FAIL(-3, "kmpc_offline_fit: bad arguments");
    int rc = shared_alloc();
    if (rc) return rc;
    if ((rc = ensure_dense(s, true))) return rc;
    if ((rc = forget_tableaux(s))) return rc;
    DevTmp tpx, tpy, tpinv, tg;
    if (init_rls) HIPCHK(hipMalloc(&tpinv.p, sizeof(T) * (size_t)(p * p + L * L)));
    HIPCHK(hipMalloc(&tpx.p, sizeof(T) * (size_t)L * M));
    HIPCHK(hipMalloc(&tpy.p, sizeof(T) * (size_t)L * M));
    HIPCHK(hipMalloc(&tg.p, sizeof(double) * (size_t)gram_elems()));
    T* const px = tpx.as<T>(); T* const py = tpy.as<T>(); T* const pinv = tpinv.as<T>();
    double* const g = tg.as<double>();
    rc = lift_to((const T*)X, px, M, 1, M, s);           // panels (L x M)
    if (!rc) rc = lift_to((const T*)Y, py, M, 1, M, s);
    if (!rc) {
      GramArgs<T> ga{};
      ga.B = M; ga.n = n; ga.L = L; ga.max_blocks = GRAM_BLOCKS;
      ga.psi_prev = px; ga.pp_sl = M; ga.pp_sb = 1;
      ga.psi_now = py; ga.pn_sl = M; ga.pn_sb = 1;
      ga.u_prev = (const T*)U; ga.x_now = (const T*)X;  // C pairs X with PHIX (duffing.py:177)
      ga.partial = dPartial;
      hipError_t e = hipMemsetAsync(g, 0, sizeof(double) * (size_t)gram_elems(), s);
      if (e == hipSuccess) e = launch_gram<T>(ga, 0.0, g, s);
      if (e == hipSuccess)
        e = launch_shared_solve<T>(g, L, n, ridge, ridge, 1, dTmp, dTmp + L * p, s, pinv, pinv ? pinv + p * p : nullptr);
      if (e == hipSuccess && init_rls) {
        // the online RLS continues from the offline data: K_A = Ylift V', inv_K_G = pinv(V V')
        // (Koopman_update.m:258-278) -- in gain form: K = offline [A B], P = (V V')^-1
        e = launch_broadcast<T>(dP, sP, pinv, p * p, B, s);
        if (e == hipSuccess) e = launch_broadcast<T>(dQ, sQ, pinv + p * p, L * L, B, s);
      }
      if (e == hipSuccess) e = launch_broadcast<T>(dK, sK, dTmp, L * p, B, s);
      if (e == hipSuccess && cfg.output_kind == KMPC_OUT_CX) e = launch_broadcast<T>(dC, sC, dTmp + L * p, n * L, B, s);
      if (e == hipSuccess) e = hipMemcpyAsync(dKs, dTmp, sizeof(T) * (size_t)L * p, hipMemcpyDeviceToDevice, s);
      if (e == hipSuccess) e = hipMemcpyAsync(dCs, dTmp + L * p, sizeof(T) * (size_t)n * L, hipMemcpyDeviceToDevice, s);
      if (e == hipSuccess && (A || Bm || C))
        e = launch_export_model<T>(dTmp, (long)L * p, dTmp + L * p, (long)n * L, n, L, 1, (T*)A, (T*)Bm, (T*)C, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      if (e != hipSuccess) { err = std::string("kmpc_offline_fit: ") + hipGetErrorString(e); rc = -(int)(1000 + (int)e); }
    }
    if (!rc && init_rls) rls_fresh = false;  // the first online update refines this model instead of restarting from K_A = 0
    return rc;
  }

  int shared_get_model(void* A, void* Bm, void* C, hipStream_t s) override {
    int rc = shared_alloc();
    if (rc) return rc;
    HIPCHK(launch_export_model<T>(dKs, (long)L * p, cfg.output_kind == KMPC_OUT_CX ? dCs : nullptr, (long)n * L, n, L,
                                  1, (T*)A, (T*)Bm, (T*)C, s));
    return 0;
  }

  // ---- state blob (version 2): header | P | K | Q | C | psi_prev | last minimiser | u_prev
  //      | shared-model section (Gram sums, shared [A B], C) when the handle has one
  //      | terminal-weight section (PN - Qw I: one block, or the Riccati blocks of kmpc_terminal_from_dare)
  struct BlobHeader {
    int32_t magic, version, dtype, n, L, N, B, have_prev, rls_fresh;
    int32_t has_shared, shared_has_samples, have_wterm, wterm_from_dare, wterm_per_traj, wterm_blocks, reserved;
    double P0, barQ0;
  };
  int64_t base_elems() const { return (sP + sK + sQ + sC) * (int64_t)B + (int64_t)L * B + B + (int64_t)N * B; }
  int64_t shared_bytes() const { return dGram ? (int64_t)sizeof(double) * gram_elems() + (int64_t)sizeof(T) * ((int64_t)L * p + (int64_t)n * L) : 0; }
  int64_t wterm_blocks() const { return !have_wterm ? 0 : (wterm_from_dare ? (wterm_per_traj ? B : 1) : 1); }
  int64_t wterm_bytes() const { return !have_wterm ? 0 : (wterm_from_dare ? (int64_t)sizeof(double) : (int64_t)sizeof(T)) * wterm_blocks() * q * q; }
  int64_t state_bytes() const override {
    return (int64_t)sizeof(BlobHeader) + (int64_t)sizeof(T) * base_elems() + shared_bytes() + wterm_bytes();
  }
  int state_export(void* blob, int64_t bytes) override {
    if (bytes < state_bytes() || !blob) FAIL(-3, "kmpc_state_export: buffer too small");
    { int rc = sync_own(); if (rc) return rc; }  // (a roll-out on a non-blocking stream may still be writing the wave image)
    { int rc = ensure_dense(nullptr, false); if (rc) return rc; }
    { int rc = forget_tableaux(nullptr); if (rc) return rc; }  // (a checkpoint is a synchronisation point: exporter and importer continue alike)
    if (core) img_valid = false;  // (... a float32 handle from its float32 blocks, as the importer will, not from its core's unrounded state)
    HIPCHK(hipStreamSynchronize(nullptr));
    BlobHeader hd{};
    hd.magic = 0x4b4d5043; hd.version = 2; hd.dtype = cfg.dtype; hd.n = n; hd.L = L; hd.N = N; hd.B = B;
    hd.have_prev = have_prev ? 1 : 0; hd.rls_fresh = rls_fresh ? 1 : 0;
    hd.has_shared = dGram ? 1 : 0; hd.shared_has_samples = shared_has_samples ? 1 : 0;
    hd.have_wterm = have_wterm ? 1 : 0; hd.wterm_from_dare = wterm_from_dare ? 1 : 0; hd.wterm_per_traj = wterm_per_traj ? 1 : 0;
    hd.wterm_blocks = (int32_t)wterm_blocks();
    hd.P0 = cfg.P0; hd.barQ0 = cfg.barQ0;
    char* o = (char*)blob;  // host or device memory (unified addressing)
    HIPCHK(hipMemcpy(o, &hd, sizeof(hd), hipMemcpyDefault)); o += sizeof(hd);
    struct { const void* ptr; size_t bytes; } parts[] = {
        {dP, sizeof(T) * (size_t)sP * B}, {dK, sizeof(T) * (size_t)sK * B}, {dQ, sizeof(T) * (size_t)sQ * B},
        {dC, sizeof(T) * (size_t)sC * B}, {dPsi[cur ^ 1], sizeof(T) * (size_t)L * B}, {dWarm, sizeof(T) * (size_t)N * B},
        {dUprev, sizeof(T) * (size_t)B},
        {dGram, dGram ? sizeof(double) * (size_t)gram_elems() : 0}, {dKs, dGram ? sizeof(T) * (size_t)L * p : 0},
        {dCs, dGram ? sizeof(T) * (size_t)n * L : 0},
        {have_wterm ? (wterm_from_dare ? (const void*)dWtB : (const void*)dWt) : nullptr, (size_t)wterm_bytes()}};
    for (auto& pt : parts) {
      if (!pt.bytes) continue;
      HIPCHK(hipMemcpy(o, pt.ptr, pt.bytes, hipMemcpyDefault));
      o += pt.bytes;
    }
    return 0;
  }
  int state_import(const void* blob, int64_t bytes) override {
    if (bytes < (int64_t)sizeof(BlobHeader) || !blob) FAIL(-3, "kmpc_state_import: buffer too small");
    BlobHeader hd;
    const char* o = (const char*)blob;
    HIPCHK(hipMemcpy(&hd, o, sizeof(hd), hipMemcpyDefault)); o += sizeof(hd);
    if (hd.magic != 0x4b4d5043 || hd.version != 2 || hd.dtype != cfg.dtype || hd.n != n || hd.L != L || hd.N != N || hd.B != B)
      FAIL(-3, "kmpc_state_import: blob does not match this handle");
    // a consistent header: flags are 0 / 1, the number of terminal blocks is what the flags imply, Riccati blocks only where
    // kmpc_terminal_from_dare would have produced them (float64)  (round-1 blobs, version 1, are not accepted: they carry no
    // shared-model / terminal sections and were written with another state order)
    auto flag01 = [](int32_t v) { return v == 0 || v == 1; };
    if (!flag01(hd.have_prev) || !flag01(hd.rls_fresh) || !flag01(hd.has_shared) || !flag01(hd.shared_has_samples) || !flag01(hd.have_wterm) ||
        !flag01(hd.wterm_from_dare) || !flag01(hd.wterm_per_traj))
      FAIL(-3, "kmpc_state_import: inconsistent header (flags)");
    const int32_t want_blocks = !hd.have_wterm ? 0 : ((hd.wterm_from_dare && hd.wterm_per_traj) ? B : 1);
    if (hd.wterm_blocks != want_blocks || (!hd.have_wterm && (hd.wterm_from_dare || hd.wterm_per_traj)) || (hd.wterm_per_traj && !hd.wterm_from_dare))
      FAIL(-3, "kmpc_state_import: inconsistent header (terminal blocks)");
    if (hd.wterm_from_dare && sizeof(T) != 8) FAIL(-3, "kmpc_state_import: Riccati terminal blocks need a float64 handle");
    const int64_t wb = !hd.have_wterm ? 0 : (hd.wterm_from_dare ? (int64_t)sizeof(double) : (int64_t)sizeof(T)) * hd.wterm_blocks * q * q;
    const int64_t sb = hd.has_shared ? (int64_t)sizeof(double) * gram_elems() + (int64_t)sizeof(T) * ((int64_t)L * p + (int64_t)n * L) : 0;
    if (bytes < (int64_t)sizeof(BlobHeader) + (int64_t)sizeof(T) * base_elems() + sb + wb) FAIL(-3, "kmpc_state_import: buffer too small");
    { int rc = sync_own(); if (rc) return rc; }
    dense_valid = true; img_valid = false;  // (the blob overwrites every dense block)
    { int rc = forget_tableaux(nullptr); if (rc) return rc; }
    if (hd.has_shared) { int rc = shared_alloc(); if (rc) return rc; }
    if (hd.have_wterm) {
      if (hd.wterm_from_dare) {
        if (dare_cap < hd.wterm_blocks) {
          if (dDareP) (void)hipFree(dDareP);
          if (dDareIt) (void)hipFree(dDareIt);
          if (dWtB) (void)hipFree(dWtB);
          dDareP = nullptr; dDareIt = nullptr; dWtB = nullptr; dare_cap = 0;
          HIPCHK(hipMalloc(&dDareP, sizeof(double) * (size_t)hd.wterm_blocks * L * L));
          HIPCHK(hipMalloc(&dDareIt, sizeof(int32_t) * (size_t)hd.wterm_blocks));
          HIPCHK(hipMalloc(&dWtB, sizeof(double) * (size_t)hd.wterm_blocks * q * q));
          dare_cap = hd.wterm_blocks;
        }
      } else if (!dWt) {
        HIPCHK(hipMalloc(&dWt, sizeof(T) * (size_t)q * q));
      }
    }
    struct { void* ptr; size_t bytes; } parts[] = {
        {dP, sizeof(T) * (size_t)sP * B}, {dK, sizeof(T) * (size_t)sK * B}, {dQ, sizeof(T) * (size_t)sQ * B},
        {dC, sizeof(T) * (size_t)sC * B}, {dPsi[cur ^ 1], sizeof(T) * (size_t)L * B}, {dWarm, sizeof(T) * (size_t)N * B},
        {dUprev, sizeof(T) * (size_t)B},
        {dGram, hd.has_shared ? sizeof(double) * (size_t)gram_elems() : 0}, {dKs, hd.has_shared ? sizeof(T) * (size_t)L * p : 0},
        {dCs, hd.has_shared ? sizeof(T) * (size_t)n * L : 0},
        {hd.have_wterm ? (hd.wterm_from_dare ? (void*)dWtB : (void*)dWt) : nullptr, (size_t)wb}};
    for (auto& pt : parts) {
      if (!pt.bytes) continue;
      HIPCHK(hipMemcpy(pt.ptr, o, pt.bytes, hipMemcpyDefault));
      o += pt.bytes;
    }
    have_prev = hd.have_prev != 0;
    rls_fresh = hd.rls_fresh != 0;
    if (hd.has_shared) shared_has_samples = hd.shared_has_samples != 0;
    else if (dGram) {  // the blob comes from a handle that never ran a shared-model step
      HIPCHK(hipMemset(dGram, 0, sizeof(double) * (size_t)gram_elems()));
      shared_has_samples = false;
    }
    have_wterm = hd.have_wterm != 0; wterm_from_dare = hd.wterm_from_dare != 0; wterm_per_traj = hd.wterm_per_traj != 0;
    cfg.P0 = hd.P0; cfg.barQ0 = hd.barQ0;
    // the host copy of a given P_N (kmpc_mpc_solve with another Q re-forms the block from it) and -- float32 handle with a float64
    // core -- the core's terminal weight follow the blob: the core only learns about a weight through set_terminal_weight, so a
    // checkpoint loaded into a fresh handle would otherwise run its fused roll-outs WITHOUT the terminal block, and one without a block
    // loaded over a handle that had one would keep using it
    hostPN.clear();
    if (have_wterm && !wterm_from_dare) {
      std::vector<T> w((size_t)q * q);
      HIPCHK(hipMemcpy(w.data(), dWt, sizeof(T) * w.size(), hipMemcpyDeviceToHost));
      hostPN.resize(w.size());
      for (int r = 0; r < q; ++r)
        for (int c2 = 0; c2 < q; ++c2) hostPN[(size_t)r * q + c2] = (double)w[(size_t)r * q + c2] + (r == c2 ? cfg.Qw : 0.0);
      return core_weight(&w);
    }
    return core_weight(nullptr);
  }

  int profile_enable(int on) override {
    prof = on != 0;
    ev_used = 0;
    prof_steps = 0;
    if (core) core->profile_enable(on);
    return 0;
  }
  int profile_read(double* ms2, int64_t* count, int reset_) override {
    double a0 = 0, a1 = 0;
    int64_t core_steps = 0;
    if (core) {  // (the fused launches of a float32 handle are recorded by its core; both halves keep their totals on a non-resetting read)
      double c2[2] = {0, 0}; int64_t cc = 0;
      const int rc = core->profile_read(c2, &cc, reset_);
      if (rc) { err = core->err; return rc; }
      a0 += c2[0]; a1 += c2[1]; core_steps = cc;
    }
    for (size_t i = 0; i + 2 < ev_used + 0 && i + 2 < ev.size(); i += 3) {
      HIPCHK(hipEventSynchronize(ev[i + 2]));
      float t0 = 0, t1 = 0;
      HIPCHK(hipEventElapsedTime(&t0, ev[i], ev[i + 1]));
      HIPCHK(hipEventElapsedTime(&t1, ev[i + 1], ev[i + 2]));
      a0 += t0; a1 += t1;
    }
    if (ms2) { ms2[0] = a0; ms2[1] = a1; }
    if (count) *count = prof_steps + core_steps;  // control steps covered by the recorded launches
    if (reset_) { ev_used = 0; prof_steps = 0; }
    return 0;
  }

  int64_t algorithmic_bytes() const override {
    // s [ 2 (p^2 + L p + L^2 + n L) + (L + L p + n L + N m + n) + (n + m + q N) ]
    const int64_t s = sizeof(T);
    return s * (2 * ((int64_t)p * p + L * p + L * L + n * L) + (L + L * p + n * L + N * m + n) + (n + m + q * N));
  }
};

// every entry point runs on the device the handle was created on (lazy allocations and launches included), whatever
// the caller's current device is, and leaves the caller's device as it found it
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
// ---------------------------------------------------------------------------------------
extern "C" {

int kmpc_version(void) { return 100; }

int kmpc_create(const kmpc_config* cfg, kmpc_handle** out) {
  if (!cfg || !out) { g_create_error = "kmpc_create: null argument"; return -1; }
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    g_create_error = "kmpc_create: no HIP device visible (libkoopmpc has no CPU path)";
    return -5;
  }
  kmpc_handle* h = nullptr;
  int rc;
  if (cfg->dtype == KMPC_F64) { auto* i = new Impl<double>(); rc = i->init(*cfg); h = i; }
  else if (cfg->dtype == KMPC_F32) { auto* i = new Impl<float>(); rc = i->init(*cfg); h = i; }
  else { g_create_error = "kmpc_create: dtype must be KMPC_F32 or KMPC_F64"; return -2; }
  if (rc) { g_create_error = h->err; delete h; return rc; }
  if (hipGetDevice(&h->device) != hipSuccess) h->device = 0;
  *out = h;
  return 0;
}

int kmpc_destroy(kmpc_handle* h) {
  if (!h) return 0;
  DeviceGuard g(h->device);
  delete h;
  return 0;
}
const char* kmpc_last_error(const kmpc_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

#define NN(h) if (!(h)) return -1; DeviceGuard dev_guard_((h)->device)
// (stream-ordered entry points: when the call returns, the handle's event for this stream is recorded behind what it enqueued)
struct StreamMarkAtExit {
  kmpc_handle* h; hipStream_t s;
  ~StreamMarkAtExit() { h->mark_stream(s); }
};
#define NNS(h, s) NN(h); StreamMarkAtExit mark_at_exit_{(h), (hipStream_t)(s)}
int kmpc_set_encoder(kmpc_handle* h, int layer, const double* W, const double* b, int rows, int cols) { NN(h); return h->set_encoder_layer(layer, W, b, rows, cols); }
int kmpc_set_encoder_layer(kmpc_handle* h, int layer, const double* W, const double* b, int rows, int cols) { return kmpc_set_encoder(h, layer, W, b, rows, cols); }
int kmpc_set_centres(kmpc_handle* h, const double* cx, int L, int n) { NN(h); return h->set_centres(cx, L, n); }
int kmpc_set_model(kmpc_handle* h, const double* A, const double* B, const double* C) { NN(h); return h->set_model(A, B, C); }
int kmpc_set_terminal_weight(kmpc_handle* h, const double* PN) { NN(h); return h->set_terminal_weight(PN); }
int kmpc_terminal_from_dare(kmpc_handle* h, const double* Q, double R, int maxiter, double eps, int per_trajectory,
                            double* PN_out, int32_t* iters_out, void* s) {
  NNS(h, s);
  return h->terminal_from_dare(Q, R, maxiter, eps, per_trajectory, PN_out, iters_out, (hipStream_t)s);
}
int kmpc_solve_dare(const void* A, const void* B, const double* Q, double R, int maxiter, double eps, int nb, int L,
                    void* P, void* K, int32_t* iters, void* s) {
  if (nb == 0) return 0;
  if (!A || !B || !Q || !P || nb < 0 || L < 1 || L > 64 || maxiter < 1) return -3;
  double* dQl = nullptr;
  if (hipMalloc(&dQl, sizeof(double) * (size_t)L * L) != hipSuccess) return -1001;
  int rc = 0;
  if (hipMemcpyAsync(dQl, Q, sizeof(double) * (size_t)L * L, hipMemcpyHostToDevice, (hipStream_t)s) != hipSuccess) rc = -1002;
  DareArgs d{};
  d.nb = nb; d.L = L; d.q = 0; d.maxiter = maxiter; d.shared_model = 0;
  d.A = (const double*)A; d.strideA = (long)L * L; d.ldA = L;
  d.B = (const double*)B; d.strideB = L; d.incB = 1;
  d.Q = dQl; d.R = R; d.eps = eps; d.Co = nullptr; d.strideC = 0;
  d.P = (double*)P; d.K = (double*)K; d.PN = nullptr; d.pn_sub_diag = 0.0; d.iters = iters;
  if (!rc && launch_dare(d, (hipStream_t)s) != hipSuccess) rc = -1003;
  if (hipStreamSynchronize((hipStream_t)s) != hipSuccess && !rc) rc = -1004;
  (void)hipFree(dQl);
  return rc;
}
int kmpc_set_terminal_refresh(kmpc_handle* h, int every, const double* Q, double R, int maxiter, double eps) { NN(h); return h->set_terminal_refresh(every, Q, R, maxiter, eps); }
int kmpc_rollout_is_fused(const kmpc_handle* h) { NN(h); return h->rollout_is_fused(); }
// Makes (or finds) the plug-in a handle of this configuration would load -- no device needed: hipcc cross-compiles.  For builds that
// want the kernel cache filled before the first kmpc_create (__graft_entry__.build(), an installer, the first rank of a node).
int kmpc_rollout_plugin_prebuild(int n, int L, int N, int out_rows, int lift_kind, int hidden_eff, int batch, int dtype, char* text, int text_bytes) {
  auto say = [&](const std::string& t) { if (text && text_bytes > 0) snprintf(text, (size_t)text_bytes, "%s", t.c_str()); };
  const bool rbf = lift_kind != KMPC_LIFT_MLP;
  const int q = out_rows > 0 ? out_rows : n;
  if (n != 2 || L < 1 || N < 1 || batch < 1 || (!rbf && (hidden_eff < 1 || hidden_eff > 128)) || (dtype != KMPC_F32 && dtype != KMPC_F64)) { say("bad arguments"); return -3; }
  const int Hp = hidden_eff <= 112 ? 112 : 128, Lp = ((L + 15) / 16) * 16, KS = (hidden_eff + 3) / 4;
  const bool io32 = dtype == KMPC_F32;
  if (!(io32 ? rollout_io32_available(n, L, N, q, rbf) : rollout_fused_available<double>(n, L, N, q, 64, rbf))) { say("no fused roll-out for this configuration"); return 2; }
  RolloutPluginKey k{};
  if (!rollout_plugin_key(n, L, N, q, rbf, Lp, KS, Hp, batch, io32, &k)) { say("built-in instantiation of libkoopmpc.so"); return 0; }
  std::string err;
  if (!rollout_plugin_get(k, &err)) { say(err); return -1; }
  say(rollout_plugin_describe(k));
  return 1;
}
int kmpc_rollout_plugin_status(const kmpc_handle* h, char* text, int text_bytes) {
  if (!h) return -100;
  std::string t;
  const int rc = h->rollout_plugin_status(&t);
  if (text && text_bytes > 0) { snprintf(text, (size_t)text_bytes, "%s", t.c_str()); }
  return rc;
}
// The placement pass of the fused roll-out as a function of its own (tests, tools): perm_dev receives, in its first B entries, the
// slot -> trajectory table of the card deal for the per-trajectory work counters work_dev[B]; it must hold 3 B int32 (the sort's buffers).
int kmpc_rank_by_work(const int32_t* work_dev, int B, int32_t* perm_dev, void* s) {
  if (!work_dev || !perm_dev) return -3;
  const hipError_t e = kmpc::launch_place(work_dev, B, perm_dev, reinterpret_cast<uint32_t*>(perm_dev + B), (hipStream_t)s);
  return e == hipSuccess ? 0 : -(1000 + (int)e);
}
int kmpc_set_rollout_workgroup(int trajectories) {
  if (trajectories != 0 && trajectories != 4 && trajectories != 8 && trajectories != 16) return -1;
  kmpc::set_rollout_workgroup(trajectories);
  return 0;
}
int kmpc_reset(kmpc_handle* h, void* s) { NNS(h, s); return h->reset((hipStream_t)s); }
int kmpc_state_init(kmpc_handle* h, double P0_scale, double barQ0_scale, void* s) { NNS(h, s); return h->state_init(P0_scale, barQ0_scale, (hipStream_t)s); }
int kmpc_state_init_from(kmpc_handle* h, const double* K_A0, const double* P0, const double* barX0, const double* barQ0, void* s) {
  NNS(h, s);
  return h->state_init_from(K_A0, P0, barX0, barQ0, (hipStream_t)s);
}
int kmpc_lift(kmpc_handle* h, const void* X, void* Psi, int B, void* s) { NNS(h, s); return h->lift(X, Psi, B, (hipStream_t)s); }
int kmpc_rls_update(kmpc_handle* h, const void* psi, const void* u, const void* psin, const void* xn, int B, void* s) { NNS(h, s); return h->rls_update(psi, u, psin, xn, B, (hipStream_t)s); }
int kmpc_get_model(kmpc_handle* h, void* A, void* B, void* C, void* s) { NNS(h, s); return h->get_model(A, B, C, (hipStream_t)s); }
int kmpc_condense(kmpc_handle* h, const void* psi, const void* ref, int rpt, void* H, void* f, int B, void* s) { NNS(h, s); return h->condense(psi, ref, rpt, H, f, nullptr, B, (hipStream_t)s); }
int kmpc_condense_cost(kmpc_handle* h, const void* psi, const void* ref, int rpt, void* H, void* f, void* c, int B, void* s) { NNS(h, s); return h->condense(psi, ref, rpt, H, f, c, B, (hipStream_t)s); }
int kmpc_mpc_solve(kmpc_handle* h, const void* A, const void* Bv, const void* C, int shared, const void* psi, const void* ref, int rpt,
                   double lb, double ub, double Qw, double Rw, const double* PN, void* U, void* U0, void* fun, int32_t* st, int32_t* it,
                   int B, void* s) {
  NNS(h, s);
  return h->mpc_solve(A, Bv, C, shared, psi, ref, rpt, lb, ub, Qw, Rw, PN, U, U0, fun, st, it, B, (hipStream_t)s);
}
int kmpc_generate_and_fit(kmpc_handle* h, int plant, const void* X0, const void* U, int n_traj, int n_steps, double hs, double ridge,
                          int init_rls, void* A, void* B, void* C, void* Xo, void* Yo, void* s) {
  NNS(h, s);
  return h->generate_and_fit(plant, X0, U, n_traj, n_steps, hs, ridge, init_rls, A, B, C, Xo, Yo, (hipStream_t)s);
}
int kmpc_qp_solve(kmpc_handle* h, const void* H, const void* f, void* U, int32_t* st, int32_t* it, int B, void* s) { NNS(h, s); return h->qp_solve(H, f, U, st, it, B, (hipStream_t)s); }
int kmpc_step(kmpc_handle* h, const void* X, const void* ref, int rpt, void* U0, void* Useq, int32_t* st, int32_t* it, void* s) { NNS(h, s); return h->step(X, ref, rpt, U0, Useq, st, it, (hipStream_t)s); }
int kmpc_set_applied_input(kmpc_handle* h, const void* U, int B, void* s) { NNS(h, s); return h->set_applied_input(U, B, (hipStream_t)s); }
int kmpc_set_online_update(kmpc_handle* h, int on) { NN(h); return h->set_online_update(on); }
int kmpc_plant_step(kmpc_handle* h, int plant, void* X, const void* U, double hs, int sw, int B, void* s) { NNS(h, s); return h->plant_step(plant, X, U, hs, sw, B, (hipStream_t)s); }
int kmpc_rollout(kmpc_handle* h, int plant, void* X, const void* ref, int rpt, int steps, int step0, int sw, double hs, void* Ulog, void* Xlog, int32_t* st, int32_t* it, void* s) { NNS(h, s); return h->rollout(plant, X, ref, rpt, steps, step0, sw, hs, Ulog, Xlog, st, it, (hipStream_t)s); }
int kmpc_offline_fit(kmpc_handle* h, const void* X, const void* Y, const void* U, int M, double ridge, int init_rls, void* A, void* B, void* C, void* s) { NNS(h, s); return h->offline_fit(X, Y, U, M, ridge, init_rls, A, B, C, (hipStream_t)s); }
int64_t kmpc_gram_elems(const kmpc_handle* h) { return h ? h->gram_elems() : -1; }
// The one collective of the path for native callers: sum of the per-rank Gram blocks over an RCCL communicator.
// RCCL is resolved at run time from the process (the caller created the communicator with the RCCL it loaded; under
// PyTorch that is torch's own copy and KoopmanMPC.shared_step goes through torch.distributed instead) -- the library
// itself does not link against it.
int kmpc_allreduce_gram(kmpc_handle* h, double* delta, void* nccl_comm, void* s) {
  NNS(h, s);
  if (!delta || !nccl_comm) return -3;
  typedef ncclResult_t (*allreduce_fn)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  const allreduce_fn fn = (allreduce_fn)resolve_nccl_allreduce();
  if (!fn) return -4;  // no RCCL in this process
  const ncclResult_t rc = fn(delta, delta, (size_t)h->gram_elems(), ncclDouble, ncclSum, (ncclComm_t)nccl_comm, (hipStream_t)s);
  return rc == ncclSuccess ? 0 : -(2000 + (int)rc);
}
int kmpc_shared_local_gram(kmpc_handle* h, const void* X, double* delta, void* s) { NNS(h, s); return h->shared_local_gram(X, delta, (hipStream_t)s); }
int kmpc_gram_accumulate(kmpc_handle* h, const void* X, double* delta, void* s) { return kmpc_shared_local_gram(h, X, delta, s); }
int kmpc_shared_solve(kmpc_handle* h, const double* delta, const void* ref, void* U0, void* Useq, int32_t* st, int32_t* it, void* s) { NNS(h, s); return h->shared_solve(delta, ref, U0, Useq, st, it, (hipStream_t)s); }
int kmpc_shared_solve_plant(kmpc_handle* h, const double* delta, const void* ref, void* U0, void* Useq, int32_t* st, int32_t* it,
                            int plant, void* X, int switched, double hstep, void* s) {
  NNS(h, s);
  return h->shared_solve_plant(delta, ref, U0, Useq, st, it, plant, X, switched, hstep, (hipStream_t)s);
}
int kmpc_shared_get_model(kmpc_handle* h, void* A, void* B, void* C, void* s) { NNS(h, s); return h->shared_get_model(A, B, C, (hipStream_t)s); }
int kmpc_shared_rollout(kmpc_handle* h, int plant, void* X, const void* ref, int steps, int step0, int switch_step, double hstep, void* comm,
                        void* Ulog, void* Xlog, void* U0out, void* Useq, int32_t* st, int32_t* it, void* s) {
  NNS(h, s);
  return h->shared_rollout(plant, X, ref, steps, step0, switch_step, hstep, comm, Ulog, Xlog, U0out, Useq, st, it, (hipStream_t)s);
}
int64_t kmpc_state_bytes(const kmpc_handle* h) { return h ? h->state_bytes() : -1; }
int kmpc_state_export(kmpc_handle* h, void* blob, int64_t bytes) { NN(h); return h->state_export(blob, bytes); }
int kmpc_state_import(kmpc_handle* h, const void* blob, int64_t bytes) { NN(h); return h->state_import(blob, bytes); }
int kmpc_profile_enable(kmpc_handle* h, int on) { NN(h); return h->profile_enable(on); }
int kmpc_profile_read(kmpc_handle* h, double* ms2, int64_t* count, int reset) { NN(h); return h->profile_read(ms2, count, reset); }
int64_t kmpc_algorithmic_bytes_per_step(const kmpc_handle* h) { return h ? h->algorithmic_bytes() : -1; }

}  // extern "C"
