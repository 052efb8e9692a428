/*
 * koopmpc.h -- C ABI of libkoopmpc.so: the batched Koopman online-updated MPC step on MI355X.
 *
 * The reference (MichaelMillerCSU/Koopman-online-updated-MPC) has no FFI: its hot path is a
 * set of statements inlined in script loops.  Each entry point below replaces one of those
 * call sites, batched over B independent trajectories; the citation is the reference
 * statement it replaces (file:line under the reference root).
 *
 * Conventions
 *   - plain C, no torch / STL types; every function returns 0 on success, <0 on error
 *     (kmpc_last_error() gives the text).  Kernels never abort: per-trajectory QP status
 *     is returned in status[B] (0 optimal, 1 iteration cap, 2 non-finite data).
 *   - "dev" pointers are device (HBM) pointers of the handle's dtype (float or double);
 *     "host" pointers are host doubles.  `stream` is a hipStream_t passed as void*.
 *   - batched vectors are batch-contiguous panels: element (row r, trajectory b) of an
 *     (R x B) panel is at [r*B + b].  Per-trajectory matrices exported by kmpc_get_model
 *     are trajectory-major blocks ([b][row][col]).
 *   - the caller owns every I/O buffer; the handle owns only the encoder weights, the
 *     persistent RLS state (P, [A B], bar_Q, C), the previous lifted state / input and
 *     scratch.  No hidden host<->device copies on the step path.
 *   - one handle per GPU / stream; a handle is not thread-safe.
 */
#ifndef KOOPMPC_H
#define KOOPMPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kmpc_handle kmpc_handle;

enum { KMPC_F32 = 0, KMPC_F64 = 1 };
enum {
  KMPC_LIFT_MLP = 0,        /* AutoEncoder.Encoder, duffing.py:17-29                      */
  KMPC_LIFT_RBF_PY = 1,     /* d^2 log(d + eps), vanderpol_RBF.py:20-23                   */
  KMPC_LIFT_RBF_MATLAB = 2  /* r2 log(sqrt(r2)), NaN -> 0, rbf.m:24-29                    */
};
enum {
  KMPC_OUT_CX = 0,   /* y = C x, C adapted by RLS (duffing.py:553, 943-953); q = n        */
  KMPC_OUT_LIFT = 1  /* y = lifted state (vanderpol.py:456-459); q = L, C not used        */
};
enum { KMPC_PLANT_DUFFING = 0, KMPC_PLANT_VDP = 1, KMPC_PLANT_TANK = 2 /* Tank_System.m:9-10, 194-195, 211 */ };
/* OR-ed to KMPC_PLANT_DUFFING / KMPC_PLANT_VDP: the Runge-Kutta step as the MATLAB scripts write it, k4 = f(x + h k1) instead
 * of f(x + h k3) (Koopman_update.m:24, Koopman_update_Tracking_Lift.m)                          */
enum { KMPC_PLANT_RK4_MATLAB = 16 };
/* kmpc_config.lift_offset: the lifts of the MATLAB scripts on top of the encoder (KMPC_LIFT_MLP only)
 *   KMPC_LIFT_OFFSET_PSI0    psi(x) - psi(0)               Koopman_update_Tracking_Lift.m:65   (L = encoder outputs)
 *   KMPC_LIFT_OFFSET_X_PSI0  [x; psi(x)] - [0; psi(0)]     Koopman_update.m:67, 70             (L = n + encoder outputs) */
enum { KMPC_LIFT_OFFSET_NONE = 0, KMPC_LIFT_OFFSET_PSI0 = 1, KMPC_LIFT_OFFSET_X_PSI0 = 2 };

/* kmpc_step / kmpc_run phases (bit mask) */
enum { KMPC_PH_RLS = 1, KMPC_PH_CONDENSE = 2, KMPC_PH_QP = 4 };

typedef struct kmpc_config {
  int32_t n;           /* plant state dimension (2)                                        */
  int32_t m;           /* inputs; 1 as in the reference (duffing.py:170-171)               */
  int32_t L;           /* lifted dimension Nlift (duffing.py:66)                           */
  int32_t N;           /* horizon MPCHorizon = ControlHorizon (duffing.py:632-633)         */
  int32_t hidden;      /* MLP hidden width (100)                                           */
  int32_t layers;      /* MLP hidden layers (3; Encoder_Tank.m has 2)                      */
  int32_t lift_kind;   /* KMPC_LIFT_*                                                      */
  int32_t output_kind; /* KMPC_OUT_*                                                       */
  int32_t dtype;       /* KMPC_F32 / KMPC_F64                                              */
  int32_t batch;       /* B: trajectories resident on this GPU                             */
  int32_t qp_max_iter; /* Newton-solve cap per QP (0 -> 8N + 40)                           */
  int32_t threads;     /* threads per trajectory block: 0 auto, 64 or 256                  */
  int32_t delta_u;     /* 1: decision variable is the input increment, MPC state [psi; u_prev],
                          A~ = [A B; 0 I], B~ = [B; I], C~ = [C 0] (Tank_System.m:110-113, 265-268, 290);
                          the step returns u_k = u_{k-1} + dU*(1) (Tank_System.m:192)          */
  int32_t out_row0;    /* KMPC_OUT_CX: outputs are rows out_row0 .. out_row0+out_rows-1 of C x   */
  int32_t out_rows;    /*   (Cy = [0 1] -> out_row0 = 1, out_rows = 1, Tank_System.m:113); 0 -> all n rows */
  int32_t c_skip_first;/* 1: the first C update only downdates bar_Q (Tank_System.m:252-254)     */
  int32_t cold_start;  /* 0: kmpc_step / kmpc_rollout start each solve at the previous minimiser; 1: always at
                          clip(0), the reference's start (its pastRes_loc stays zeros, duffing.py:634-635, 859).
                          The minimiser is unique, so this only changes the work, not the answer.  No primal
                          start is kept; inside a fused roll-out the solver still keeps its factorisation from
                          step to step and with it the face of the box the last minimiser lay on: clip(0) is
                          moved onto that face before the first Newton point (qp_rl.h)                        */
  int32_t lift_offset; /* KMPC_LIFT_OFFSET_* (0: the raw encoder, as the Python scripts use it)                 */
  double lambda;       /* RLS forgetting factor (1.0; Koopman_update.m:258)                */
  double P0;           /* inv_K_G init scale (1e4 duffing.py:929-930; 1e5 vanderpol.py:874)*/
  double barQ0;        /* bar_Q init scale (100 duffing.py:946)                            */
  double Qw, Rw;       /* stage weights (100, 1e-4: duffing.py:580)                        */
  double lb, ub;       /* input box (+-2 duffing.py:636; +-6 vanderpol.py:542-544)         */
  double rbf_eps;      /* 1e-4 (vanderpol_RBF.py:22)                                       */
  double umin, umax;   /* delta_u: absolute input range folded into the first increment's box
                          (A_cons, b_cons of Tank_System.m:182-188)                           */
} kmpc_config;

/* ---- life cycle -------------------------------------------------------------------- */
int kmpc_create(const kmpc_config* cfg, kmpc_handle** out);
int kmpc_destroy(kmpc_handle* h);
const char* kmpc_last_error(const kmpc_handle* h); /* h may be NULL: last create() error */
int kmpc_version(void);

/* ---- parameters (host pointers, float64, row-major; one-off uploads) ---------------- */
/* net.Encoder layer `layer` (0-based): y = W x + b, W rows x cols   duffing.py:21-28,
 * Encoder_Duffing.m:2-6 */
int kmpc_set_encoder(kmpc_handle* h, int layer, const double* W_host, const double* b_host,
                     int rows, int cols);
/* (the same call under its round-1 name) */
int kmpc_set_encoder_layer(kmpc_handle* h, int layer, const double* W_host, const double* b_host,
                           int rows, int cols);
/* RBF centres cx (L x n)                                             vanderpol_RBF.py:44-46 */
int kmpc_set_centres(kmpc_handle* h, const double* cx_host, int L, int n);
/* model used until the first online update exists: Aloc_d, Bloc_d, Cloc_d = offline A, B, C
 * (duffing.py:811-813).  A (L x L), B (L), C (n x L; ignored for KMPC_OUT_LIFT); broadcast
 * to every trajectory.                                                                      */
int kmpc_set_model(kmpc_handle* h, const double* A_host, const double* B_host, const double* C_host);

/* Optional terminal weight: the last q x q block of Q_bar becomes PN instead of Qw*I
 * (Q_bar(end-n+1:end, end-n+1:end) = C*P*C', Koopman_update.m:381; the LMI that produces P there is host
 * work and out of scope).  PN: host, q x q row-major (q = output rows); shared by all trajectories; NULL
 * restores Qw*I.  Applies to kmpc_condense / kmpc_step / kmpc_rollout and the shared-model mode.          */
int kmpc_set_terminal_weight(kmpc_handle* h, const double* PN);
/* re-initialise the online state: K_A = 0, inv_K_G = P0 I, bar_X = 0, bar_Q = barQ0 I and
 * forget the previous transition (duffing.py:927-930, 944-946)                              */
int kmpc_reset(kmpc_handle* h, void* stream);
/* the same start with other scales than the configuration's (1e4 / 100 duffing.py:929-930, 946; 1e5 / 1e5
 * vanderpol.py:875, 888; 1e4 / 1e4 Tank_System.m:240, 255)                                      */
int kmpc_state_init(kmpc_handle* h, double P0_scale, double barQ0_scale, void* stream);
/* every trajectory continues from given reference-form accumulators (the MATLAB start from the offline
 * Gram, Koopman_update.m:264-265): K_A0 (L x p), inv_K_G = P0 (p x p), bar_X0 (n x L, NULL for y = psi),
 * bar_Q0 (L x L); host, row-major.  The handle keeps the algebraically identical gain form: [A B] =
 * K_A0 P0 (duffing.py:938), C = bar_X0 bar_Q0 (duffing.py:953); the first online update refines them.  */
int kmpc_state_init_from(kmpc_handle* h, const double* K_A0, const double* P0, const double* barX0,
                         const double* barQ0, void* stream);

/* ---- the hot path, one call per reference statement --------------------------------- */
/* Psi = lift(X): net.Encoder(x) duffing.py:847 / rbf(x, cx) vanderpol_RBF.py:372.
 * X_dev (n x B) -> Psi_dev (L x B).                                                         */
int kmpc_lift(kmpc_handle* h, const void* X_dev, void* Psi_dev, int B, void* stream);

/* The "Koopman update" block duffing.py:900, 927-953, 965-967 for every trajectory:
 * z = [psi; u]; RLS update of [A B] with target psi_next, and of C with target x_next.
 * psi (L x B), u (B), psi_next (L x B), x_next (n x B).                                     */
int kmpc_rls_update(kmpc_handle* h, const void* psi_dev, const void* u_dev, const void* psi_next_dev,
                    const void* x_next_dev, int B, void* stream);

/* current per-trajectory model: A_dev [B][L][L], B_dev [B][L], C_dev [B][n][L] (any may be
 * NULL)                                                       duffing.py:965-967, 978-984 */
int kmpc_get_model(kmpc_handle* h, void* A_dev, void* B_dev, void* C_dev, void* stream);

/* Condensed QP of the current model (Koopman_update.m:455-471, :213; weights duffing.py:580):
 * H_dev [B][N][N], f_dev [B][N] such that J(u) = u'Hu + f'u + const is costFunction
 * (duffing.py:540-581).  psi (L x B); ref (q x N) shared by all trajectories when
 * ref_per_traj == 0, else [B][q][N].                                                        */
int kmpc_condense(kmpc_handle* h, const void* psi_dev, const void* ref_dev, int ref_per_traj,
                  void* H_dev, void* f_dev, int B, void* stream);
/* ... with the constant as well: c_dev [B] such that u'Hu + f'u + c IS costFunction(u) (duffing.py:540-581);
 * c + the optimal value of the QP is the `result.fun` of duffing.py:859.  c_dev may be NULL.                 */
int kmpc_condense_cost(kmpc_handle* h, const void* psi_dev, const void* ref_dev, int ref_per_traj,
                       void* H_dev, void* f_dev, void* c_dev, int B, void* stream);

/* The MPC solve wrapper as a stateless call (duffing.py:857-861 with the model as an argument; SURVEY 8b
 * mpc_solve(A, B, C, xlift, r, lb, ub, Q, R[, P_N])): condensed QP of the GIVEN models and exact box-QP, nothing of
 * the handle's estimator state is read or written.  A_dev [B][L][L], B_dev [B][L], C_dev [B][n][L] (NULL for
 * KMPC_OUT_LIFT), or ONE model for the batch when model_shared = 1; psi_dev (L x B); ref as in kmpc_condense;
 * lb, ub, Qw, Rw override the handle's for this call; PN_host (q x q) the terminal block of this call, NULL: the handle's
 * (kmpc_set_terminal_weight / kmpc_terminal_from_dare; none if the handle has none) -- the cost kmpc_condense and kmpc_step use.
 * Outputs: U_dev (N x B), U0_dev [B] (may be NULL), fun_dev [B] = J at the minimiser (may be NULL), status / iters.
 * Every solve starts at clip(0) like the reference.  B must equal the handle's batch.                          */
int kmpc_mpc_solve(kmpc_handle* h, const void* A_dev, const void* B_dev, const void* C_dev, int model_shared,
                   const void* psi_dev, const void* ref_dev, int ref_per_traj, double lb, double ub, double Qw,
                   double Rw, const double* PN_host, void* U_dev, void* U0_dev, void* fun_dev, int32_t* status_dev,
                   int32_t* iters_dev, int B, void* stream);

/* Box QP  min u'Hu + f'u, lb <= u <= ub  -- replaces optimize.minimize(..., bounds=...)
 * duffing.py:857-861 and quadprog(2H, f, ...) Koopman_update.m:214.  H_dev [B][N][N],
 * f_dev [B][N] -> U_dev (N x B), status_dev[B], iters_dev[B] (may be NULL).  Stateless: always
 * starts at clip(0) (duffing.py:634-635); the warm start of the loop lives in kmpc_step.      */
int kmpc_qp_solve(kmpc_handle* h, const void* H_dev, const void* f_dev, void* U_dev,
                  int32_t* status_dev, int32_t* iters_dev, int B, void* stream);

/* One closed-loop control step for all trajectories, in the reference's order
 * (duffing.py:847-984): lift x_k; if a previous transition exists, RLS-update the model with
 * (psi_{k-1}, u_{k-1}, psi_k, x_k); condense; solve; u_k.  X_dev (n x B), ref as above,
 * U0_dev (B), Useq_dev (N x B, may be NULL), status_dev / iters_dev [B] (may be NULL).
 * The handle keeps psi_k and u_k for the next call.  One kernel launch where the fused roll-out has
 * an instantiation (float64, MLP lift, static dimension set: the roll-out kernel with one step and no
 * plant), else the lift kernel followed by the step kernel.                                   */
int kmpc_step(kmpc_handle* h, const void* X_dev, const void* ref_dev, int ref_per_traj,
              void* U0_dev, void* Useq_dev, int32_t* status_dev, int32_t* iters_dev, void* stream);

/* ---- shared-model mode (BASELINE cfg4): ONE [A B], C for all trajectories ---------------
 * The batch's transitions are pooled in Gram form, V = [Xlift; U], W = [Ylift; X]: G = V V', W V'
 * (Koopman_update.m:94-98), [A B] = (Ylift V')(G + I/P0)^-1, C = (X Xlift')(G_LL + I/barQ0)^-1 -- B rank-one
 * RLS updates of a shared inv_K_G (duffing.py:927-953) are exactly G <- G + sum_b z_b z_b'.  With the batch
 * sharded over GPUs the only exchange of the path is the sum of `delta_gram` over ranks between the two
 * stages (ncclAllReduce / torch.distributed.all_reduce on the caller's side; (p+L+n)*p float64 values).   */
int64_t kmpc_gram_elems(const kmpc_handle* h);
/* the exchange itself for native callers: delta_gram_dev <- sum over the ranks of `nccl_comm` (an
 * ncclComm_t of the RCCL the caller loaded; ncclAllReduce, float64, sum) on `stream`.  RCCL is looked up in
 * the process at run time (-4: none found); Python callers use torch.distributed.all_reduce instead
 * (KoopmanMPC.shared_step).                                                                     */
int kmpc_allreduce_gram(kmpc_handle* h, double* delta_gram_dev, void* nccl_comm, void* stream);
/* stage 1: lift x_k; delta_gram_dev (float64, overwritten) = Gram sums of THIS rank's transitions
 * (psi_{k-1}, u_{k-1}) -> (psi_k, x_k), rows [Z Z' (p x p); Ylift Z' (L x p); X Z' (n x p)]; zeros on the
 * first call (no transition yet).  The contraction runs on v_mfma_f64_16x16x4_f64.                        */
int kmpc_shared_local_gram(kmpc_handle* h, const void* X_dev, double* delta_gram_dev, void* stream);
/* the same entry point under the name SURVEY.md 8b lists for it */
int kmpc_gram_accumulate(kmpc_handle* h, const void* X_dev, double* delta_gram_dev, void* stream);
/* stage 2: G <- lambda G + delta_gram (already summed over ranks); model from G; condensed QP of the shared
 * model (H, F, f0 with f_b = F psi_b + f0); box QP of every trajectory.  ref_dev (q x N) is shared.        */
int kmpc_shared_solve(kmpc_handle* h, const double* delta_gram_dev, const void* ref_dev, void* U0_dev,
                      void* Useq_dev, int32_t* status_dev, int32_t* iters_dev, void* stream);
/* ... and the plant inside the solve (the shared-model counterpart of kmpc_rollout's fused plant): after
 * u_k every trajectory's state advances in place, X_dev (n x B) <- f(X_dev, u_k); plant / switched / hstep as kmpc_plant_step
 * (Tank_System.m:193-196, 211 for the tanks).  One launch less per step of the shared-model loop.                           */
int kmpc_shared_solve_plant(kmpc_handle* h, const double* delta_gram_dev, const void* ref_dev, void* U0_dev, void* Useq_dev,
                            int32_t* status_dev, int32_t* iters_dev, int plant, void* X_dev, int switched, double hstep,
                            void* stream);
/* The shared-model loop itself (Tank_System.m:170-291 with one model for every trajectory of every rank; the loop bench.py
 * times for BASELINE cfg4), `steps` iterations enqueued from C++ on ONE stream with the collective inside:
 *   lift + local Gram sums (one launch) -> ncclAllReduce(sum, float64) of the (2L + n + 1)(L + 1)-element block over
 *   nccl_comm (RCCL; resolved at run time as in kmpc_allreduce_gram; NULL: single rank, no collective) -> model, condensed QP,
 *   T0 -> interior trajectories on the matrix cores / box QPs, plant inside (X_dev <- f(X_dev, u_k), parameters switched from
 *   global iteration switch_step on, -1: never).
 * No host round trip and no Python between the stages.  U_log_dev (steps x B) / X_log_dev (steps x n x B) optional;
 * status_dev / iters_dev receive every trajectory's worst QP status and total Newton solves; U0_dev (B, optional) the last
 * step's inputs, Useq_dev (N x B, optional) its sequences.  Same results as the loop of kmpc_shared_local_gram -> [all-reduce] -> kmpc_shared_solve_plant.      */
int kmpc_shared_rollout(kmpc_handle* h, int plant, void* X_dev, const void* ref_dev, int steps, int step0, int switch_step, double hstep,
                        void* nccl_comm, void* U_log_dev, void* X_log_dev, void* U0_dev, void* Useq_dev, int32_t* status_dev,
                        int32_t* iters_dev, void* stream);
/* the shared model: A_dev (L x L), B_dev (L), C_dev (n x L) in the handle dtype                           */
int kmpc_shared_get_model(kmpc_handle* h, void* A_dev, void* B_dev, void* C_dev, void* stream);

/* Offline EDMD fit on the device: K_hat = PHIY pinv([PHIX; U]), C = X pinv(PHIX)
 * (duffing.py:152-177) evaluated in Gram form M = (W V')(V V')^-1 (Koopman_update.m:94-101): lift X and Y,
 * Gram sums on the MFMA kernel, p x p solve with the ridge `ridge` (0 reproduces the pseudo-inverse when
 * V V' has full rank).  X_dev, Y_dev (n x M), U_dev (M).  The fitted model becomes every trajectory's
 * model (as kmpc_set_model) and is returned in A_dev (L x L), B_dev (L), C_dev (n x L) when non-NULL.
 * init_rls = 1 additionally starts every trajectory's online RLS FROM the offline data, as the MATLAB twin
 * does (K_A = Ylift V', inv_K_G = pinv(V V'), Koopman_update.m:258-278): P = (V V')^-1, bar_Q =
 * (Xlift Xlift')^-1, and the first online update refines the fitted model instead of restarting from
 * K_A = 0 (the Python scripts' behaviour, duffing.py:927-930, which init_rls = 0 keeps).                    */
int kmpc_offline_fit(kmpc_handle* h, const void* X_dev, const void* Y_dev, const void* U_dev, int M,
                     double ridge, int init_rls, void* A_dev, void* B_dev, void* C_dev, void* stream);
/* Data generation + offline fit as ONE device pipeline (data_generate.py:17-79 -> duffing.py:152-177):
 * n_traj trajectories start at X0_dev (n x n_traj) and take n_steps plant steps (RK4 Duffing / Van der Pol, tank map;
 * nominal parameters) with the inputs U_dev (n_steps x n_traj) -- the caller draws x0 and u (the reference uses
 * np.random with seed 101) --, the M = n_traj * n_steps transitions are lifted, their Gram sums formed on MFMA and the
 * model solved exactly as kmpc_offline_fit does; nothing returns to the host in between.  X_out_dev / Y_out_dev
 * (n x M, sample i * n_traj + t; may be NULL) receive the generated samples.                                       */
int kmpc_generate_and_fit(kmpc_handle* h, int plant, const void* X0_dev, const void* U_dev, int n_traj, int n_steps,
                          double hstep, double ridge, int init_rls, void* A_dev, void* B_dev, void* C_dev,
                          void* X_out_dev, void* Y_out_dev, void* stream);

/* The input that was APPLIED at the last step, when it is not the u_k kmpc_step returned (actuator limits, a logged
 * trajectory that is being followed): U_dev [B]; the next RLS update regresses on it (z = [psi; u], duffing.py:900). */
int kmpc_set_applied_input(kmpc_handle* h, const void* U_dev, int B, void* stream);

/* ---- adjacent to the path: the plant on the device ---------------- */
/* X <- RK4(f, X, U, h) in place: duffing.py:250-261 / vanderpol_RBF.py:113; `switched`
 * selects the parameters after step 100 (duffing.py:991-992, vanderpol.py:923-931).         */
int kmpc_plant_step(kmpc_handle* h, int plant, void* X_dev, const void* U_dev, double hstep,
                    int switched, int B, void* stream);

/* `steps` iterations of the reference loop body (duffing.py:823-1012) without returning to the host:
 * for i = step0 .. step0+steps-1: kmpc_step(X) -> u_i; X <- plant(X, u_i) with the switched
 * parameters once i >= switch_step (the reference switches at the end of iteration 101, i.e.
 * switch_step = 102; pass a negative value for "never").  Optional outputs: U_log_dev (steps x B),
 * X_log_dev (steps x n x B, the state AFTER each plant step), status_dev[B] (worst QP status of
 * each trajectory over the call), iters_dev[B] (its total number of Newton solves).
 * For single-wave configurations (kmpc_rollout_is_fused: n = 2, y = C x with q <= 2 or y = psi, (L+1)^2 and N^2 <= 2048, as far as
 * the LDS holds a workgroup) the whole
 * call is ONE kernel launch: 16 trajectories per workgroup, the encoder evaluated inside on MFMA, no
 * synchronisation between workgroups from the first step to the last.  Same results as the per-step
 * launches up to the summation order of the encoder (1e-12 on the controls).
 * steps = 0 changes nothing and brings the handle into the form the fused roll-out works on (set-up
 * that the next call would otherwise pay after a state import or per-step calls).                  */
int kmpc_rollout(kmpc_handle* h, int plant, void* X_dev, const void* ref_dev, int ref_per_traj,
                 int steps, int step0, int switch_step, double hstep, void* U_log_dev, void* X_log_dev,
                 int32_t* status_dev, int32_t* iters_dev, void* stream);

/* ---- terminal ingredients ------------------------------------------- */
/* The reference's Riccati iteration and LQR gain, batched over nb models on the device, float64:
 *   solve_DARE(A, B, Q, R)  duffing.py:583-598:  X0 = Q, X <- A'XA - A'XB (R + B'XB)^+ B'XA + Q until
 *                           max|X_new - X| < eps (the reference uses eps = 0.01) or maxiter (500)
 *   dlqr(A, B, Q, R)        duffing.py:600-613:  K = (B'XB + R)^+ B'XA
 * A [nb][L][L], B [nb][L] (one input), P [nb][L][L], K [nb][L] or null, iters [nb] or null are device
 * pointers; Q (L x L) is a host pointer.  L <= 64.  Synchronises the stream.                          */
int kmpc_solve_dare(const void* A, const void* B, const double* Q, double R, int maxiter, double eps, int nb,
                    int L, void* P, void* K, int32_t* iters, void* stream);
/* Terminal block of Q_bar from that iteration on the handle's own model(s) (replaces the MATLAB
 * controller's LMI terminal cost by its LQR stand-in): P = solve_DARE(A, B, Q, R), P_N = Co P Co'
 * (Koopman_update.m:381; Co = the output rows of C, or I for y = psi).  per_trajectory = 0: the model of
 * trajectory 0 (every trajectory holds it after kmpc_set_model / kmpc_offline_fit) gives one block for
 * the batch; 1: every trajectory's current [A B], C gives its own block (call it as often as the
 * terminal ingredients should follow the online model; Koopman_update.m:215 does so every step).
 * PN_out (host, [1 or B][q*q]) and iters_out (host, [1 or B]) may be null.  float64 handles only.
 * kmpc_set_terminal_weight(h, PN or NULL) replaces / removes the block again.                    */
int kmpc_terminal_from_dare(kmpc_handle* h, const double* Q, double R, int maxiter, double eps,
                            int per_trajectory, double* PN_out, int32_t* iters_out, void* stream);

/* The MATLAB controller recomputes its terminal ingredients with the UPDATED model at every iteration (Koopman_update.m:215
 * K = -dlqr(A, B, ...), :381 Q_bar(end-n+1:end, ...) = C*P*C').  every > 0: kmpc_step / kmpc_rollout run the reference's Riccati iteration
 * (solve_DARE, duffing.py:583-598; Q (L x L, host, row-major), R, maxiter, eps as there) on every trajectory's freshly updated [A B] at
 * each `every`-th step -- after the RLS update, before the condensed QP is formed -- and rebuild its block Co P Co'.  For the fused
 * roll-out this happens INSIDE the launch (a plug-in variant of the kernel, made by this call: kmpc_rollout_plugin_status); on the
 * per-step route as RLS launch, Riccati kernel, QP launch.  Until a trajectory's first refresh its block is the handle's current one
 * (kmpc_set_terminal_weight / kmpc_terminal_from_dare), else Qw I.  every = 0: off, the blocks stay what the last refresh left.
 * float64 handles, per-trajectory models (not the shared-model mode).                                                            */
int kmpc_set_terminal_refresh(kmpc_handle* h, int every, const double* Q_host, double R, int maxiter, double eps);

/* ---- state hand-over / checkpoint ------------------------------------------------------ */
/* bytes of the persistent state blob (P, K, bar_Q, C, psi_prev, u_prev, flags)              */
int64_t kmpc_state_bytes(const kmpc_handle* h);
int kmpc_state_export(kmpc_handle* h, void* host_blob, int64_t bytes);
int kmpc_state_import(kmpc_handle* h, const void* host_blob, int64_t bytes);

/* ---- measurement ------------------------------------------------------------------------ */
/* when enabled, kmpc_step brackets its kernels with HIP events on `stream`                  */
int kmpc_profile_enable(kmpc_handle* h, int on);
/* accumulated milliseconds since enable/reset: [0] lift kernel, [1] step kernel (fused roll-outs:
 * [0] = 0, [1] = the roll-out kernel); count = number of control steps the recorded launches
 * covered.  Synchronises the recorded events.                                               */
int kmpc_profile_read(kmpc_handle* h, double* ms2, int64_t* count, int reset);
/* 1 if kmpc_rollout runs as one fused kernel for this handle's configuration, else 0         */
int kmpc_rollout_is_fused(const kmpc_handle* h);
/* Where that kernel comes from.  The reference's dimensions are constants edited in its scripts (duffing.py:66 Nlift, :632-633
 * MPCHorizon; Koopman_update.m:67, 70, 113: L = 10, N = 10), so the fused roll-out is not tied to a list: libkoopmpc.so holds the
 * instantiations of the BASELINE / reference dimension sets, and kmpc_create makes the kernel of any other set as a PLUG-IN --
 * looked up in the kernel cache ($KMPC_KERNEL_CACHE, <library dir>/kernel_cache, ~/.cache/koopmpc), else compiled from the sources
 * that ship next to the library (csrc/rollout_jit.hip) with hipcc ($KMPC_HIPCC, /opt/rocm/bin/hipcc): 4-8 s, once per set and machine.
 * Returns 0: built-in instantiation, 1: plug-in, -1: the plug-in could not be made (the handle works with per-step launches),
 * 2: the configuration has no fused roll-out; `text` (optional, text_bytes long) receives the file / the reason.                */
int kmpc_rollout_plugin_status(const kmpc_handle* h, char* text, int text_bytes);
/* Makes (or finds in the kernel cache) the plug-in that kmpc_create would load for a configuration with these dimensions: n, L, N,
 * out_rows (0 = n; L for y = psi), lift_kind, the encoder's effective hidden width (kmpc_config.hidden, + 2 n with lift_offset 2),
 * batch and dtype as in kmpc_config.  Needs no device (hipcc cross-compiles): for installers and for the first rank of a node.
 * Returns 0 / 1 / -1 / 2 as kmpc_rollout_plugin_status, -3 for bad arguments.                                                   */
int kmpc_rollout_plugin_prebuild(int n, int L, int N, int out_rows, int lift_kind, int hidden_eff, int batch, int dtype, char* text,
                                 int text_bytes);
/* The placement pass kmpc_rollout runs behind a fused launch, as a function of its own (tests, tools): from every trajectory's solver-work
 * counter work_dev[B] (B a multiple of 16, at most 2^20) the slot -> trajectory table of the next launch -- trajectories ranked by work
 * (descending, ties by index: every pair compared up to 16384 trajectories, a stable radix sort by one workgroup, O(B), beyond), rank r dealt to wave r / (B/16) of workgroup r mod (B/16) in
 * snake order.  perm_dev must hold 3 B int32: the table in its first B entries, the sort's buffers behind it.                        */
int kmpc_rank_by_work(const int32_t* work_dev, int B, int32_t* perm_dev, void* stream);
/* Trajectories per workgroup of the fused roll-out kernel (MLP lift): 4, 8 or 16; 0 = automatic (most
 * trajectories per CU, ties to the larger workgroup).  Process-wide tuning / test knob: every
 * trajectory's arithmetic is the same for every choice, only the scheduling differs.
 * Returns -1 for any other value.                                                            */
int kmpc_set_rollout_workgroup(int trajectories);
/* algorithmic bytes of one trajectory-step (SURVEY.md 8d formula) for this configuration   */
int64_t kmpc_algorithmic_bytes_per_step(const kmpc_handle* h);
/* on = 0: kmpc_step / kmpc_rollout run the loop WITHOUT the online update (the reference's comparison loop,
 * duffing.py:738-805, vanderpol.py:645-722): no RLS phase, the model stays as kmpc_set_model / kmpc_offline_fit or the
 * updates so far left it; on = 1 (default) resumes the update with the next transition                               */
int kmpc_set_online_update(kmpc_handle* h, int on);

#ifdef __cplusplus
}
#endif
#endif /* KOOPMPC_H */
