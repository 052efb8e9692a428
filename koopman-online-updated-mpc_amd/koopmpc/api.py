"""Host-side mirror of the reference's call surface for the hot path, batched on a trailing axis.

The reference inlines the path in script loops (duffing.py:823-1012, vanderpol.py:738-951,
vanderpol_RBF.py:348-526); the names below are the ones those loops use:

    net.Encoder(x)            duffing.py:847      ->  KoopmanMPC.Encoder(x)
    rbf(X, cx)                vanderpol_RBF.py:20 ->  KoopmanMPC.rbf(X)
    "Update LTV-Model" block  duffing.py:900-967  ->  KoopmanMPC.Koopman_update(xlift, u, ylift, x_next)
    costFunction + minimize   duffing.py:857-861  ->  KoopmanMPC.mpc_solve(psi, r)  (condense + exact box-QP)
    one loop iteration        duffing.py:847-984  ->  KoopmanMPC.step(x, r)
    solve_DARE / dlqr         duffing.py:583-613  ->  solve_DARE(A, B, Q, R), dlqr(A, B, Q, R) (batched, device)

Everything runs in hand-written HIP kernels through libkoopmpc.so (ctypes); torch only owns
the device buffers and the stream.  batch = 1 with NumPy inputs reproduces the reference's
shapes ((L,1) lifted states, (1,1) inputs).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _ffi
from .sharding import force_collectives


def _dptr(a):
    return np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.POINTER(C.c_double))


def solve_DARE(A, B, Q, R, maxiter=500, eps=0.01, device="cuda:0", return_iters=False):
    """The reference's Riccati iteration (duffing.py:583-598; its constants maxiter = 500, eps = 0.01 are the
    defaults) on the device, batched: A (L, L) or (nb, L, L), B (L,), (L, 1) or (nb, L[, 1]), Q (L, L), R scalar.
    Returns P as a float64 device tensor of A's shape (and the iteration counts when asked)."""
    P, K, it = _dare(A, B, Q, R, maxiter, eps, device, want_gain=False)
    return (P, it) if return_iters else P


def dlqr(A, B, Q, R, maxiter=500, eps=0.01, device="cuda:0"):
    """K = (B'XB + R)^+ B'XA with X = solve_DARE(A, B, Q, R)  (duffing.py:600-613): (1, L) or (nb, 1, L)."""
    P, K, it = _dare(A, B, Q, R, maxiter, eps, device, want_gain=True)
    return K


def _dare(A, B, Q, R, maxiter, eps, device, want_gain):
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("solve_DARE runs on the HIP device only")
    At = torch.as_tensor(np.asarray(A, dtype=np.float64) if not torch.is_tensor(A) else A).to(device=dev, dtype=torch.float64)
    single = At.dim() == 2
    At = At.reshape(-1, At.shape[-1], At.shape[-1]).contiguous()
    nb, L = At.shape[0], At.shape[-1]
    Bt = torch.as_tensor(np.asarray(B, dtype=np.float64) if not torch.is_tensor(B) else B).to(device=dev, dtype=torch.float64)
    Bt = Bt.reshape(nb, L).contiguous()
    Qh = np.ascontiguousarray(Q, dtype=np.float64)
    if Qh.shape != (L, L):
        raise ValueError("Q must be (%d, %d)" % (L, L))
    P = torch.empty(nb, L, L, dtype=torch.float64, device=dev)
    K = torch.empty(nb, L, dtype=torch.float64, device=dev) if want_gain else None
    it = torch.zeros(nb, dtype=torch.int32, device=dev)
    lib = _ffi.load()
    with torch.cuda.device(dev):
        rc = lib.kmpc_solve_dare(C.c_void_p(At.data_ptr()), C.c_void_p(Bt.data_ptr()), _dptr(Qh), float(np.ravel(R)[0]),
                                 int(maxiter), float(eps), nb, L, C.c_void_p(P.data_ptr()),
                                 C.c_void_p(K.data_ptr()) if want_gain else None, C.c_void_p(it.data_ptr()),
                                 C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError("kmpc_solve_dare failed with code %d" % rc)
    if single:
        return P[0], (K[0].reshape(1, L) if want_gain else None), it[0]
    return P, (K.reshape(nb, 1, L) if want_gain else None), it


class KoopmanMPC:
    """Batched Koopman online-updated MPC controller on one MI355X.

    Parameters follow the reference's constants: n=2 states, m=1 input, L = Nlift, N = MPCHorizon
    (= ControlHorizon), bounds (lb, ub), weights Qw=100, Rw=1e-4 (duffing.py:580), RLS inits
    P0=1e4 (duffing.py:929-930), barQ0=100 (duffing.py:946).
    ``output``: "Cx" -> y = C x with C adapted online (duffing.py); "lift" -> y = lifted state
    (vanderpol.py:456-459).
    ``lift_offset``: None (the raw encoder, as the Python scripts lift), "psi0" -> psi(x) - psi(0)
    (Koopman_update_Tracking_Lift.m:65), "x_psi0" -> [x; psi(x)] - [0; psi(0)] with L = n + the encoder's outputs
    (Koopman_update.m:67).
    """

    def __init__(self, n=2, L=8, N=10, batch=1, lift="mlp", weights=None, centres=None, hidden=100, layers=3,
                 output="Cx", dtype=torch.float64, lam=1.0, P0=1e4, barQ0=100.0, Qw=100.0, Rw=1e-4, lb=-2.0,
                 ub=2.0, rbf_eps=1e-4, qp_max_iter=0, threads=0, device=None, delta_u=False, out_row0=0, out_rows=0,
                 c_skip_first=False, umin=-8.0, umax=8.0, cold_start=False, lift_offset=None):
        if not torch.cuda.is_available():
            raise RuntimeError("koopmpc needs a HIP device (MI355X); there is no CPU path")
        self.lib = _ffi.load()
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        torch.cuda.set_device(self.device)
        if dtype not in (torch.float64, torch.float32):
            raise ValueError("dtype must be torch.float64 or torch.float32")
        self.dtype = dtype
        self.n, self.L, self.N, self.B = int(n), int(L), int(N), int(batch)
        self.q = self.L if output == "lift" else (int(out_rows) if out_rows else self.n)
        self.output = output
        lift_kind = {"mlp": _ffi.KMPC_LIFT_MLP, "rbf": _ffi.KMPC_LIFT_RBF_PY, "rbf_matlab": _ffi.KMPC_LIFT_RBF_MATLAB}[lift]
        cfg = _ffi.KmpcConfig(
            n=n, m=1, L=L, N=N, hidden=hidden, layers=layers, lift_kind=lift_kind,
            output_kind=_ffi.KMPC_OUT_LIFT if output == "lift" else _ffi.KMPC_OUT_CX,
            dtype=_ffi.KMPC_F64 if dtype == torch.float64 else _ffi.KMPC_F32, batch=batch,
            qp_max_iter=qp_max_iter, threads=threads, lam=lam, P0=P0, barQ0=barQ0, Qw=Qw, Rw=Rw, lb=lb, ub=ub,
            rbf_eps=rbf_eps, delta_u=int(bool(delta_u)), out_row0=int(out_row0), out_rows=int(out_rows),
            c_skip_first=int(bool(c_skip_first)), umin=umin, umax=umax, cold_start=int(bool(cold_start)),
            lift_offset={None: 0, "none": 0, "psi0": _ffi.KMPC_LIFT_OFFSET_PSI0, "x_psi0": _ffi.KMPC_LIFT_OFFSET_X_PSI0}[lift_offset])
        self.cfg = cfg
        h = C.c_void_p()
        rc = self.lib.kmpc_create(C.byref(cfg), C.byref(h))
        _ffi.check(self.lib, None, rc, "kmpc_create")
        self.h = h
        if weights is not None:
            self.set_encoder(weights)
        if centres is not None:
            self.set_centres(centres)
        B = self.B
        self.U0 = torch.zeros(B, dtype=dtype, device=self.device)
        self.Useq = torch.zeros(N, B, dtype=dtype, device=self.device)
        self.status = torch.zeros(B, dtype=torch.int32, device=self.device)
        self.iters = torch.zeros(B, dtype=torch.int32, device=self.device)

    # ------------------------------------------------------------------ plumbing
    def __del__(self):
        h = getattr(self, "h", None)
        if h:
            self.lib.kmpc_destroy(h)
            self.h = None

    def _chk(self, rc, what):
        _ffi.check(self.lib, self.h, rc, what)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, a, shape=None):
        """-> contiguous device tensor of the handle dtype (accepts numpy / torch / scalars)."""
        if not torch.is_tensor(a):
            a = torch.as_tensor(np.asarray(a, dtype=np.float64))
        a = a.to(device=self.device, dtype=self.dtype)
        if shape is not None:
            a = a.reshape(shape)
        return a.contiguous()

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr())

    # ------------------------------------------------------------------ parameters
    def set_encoder(self, weights):
        """weights = [(W1, b1), ..., (Wd+1, bd+1)], W (out x in)   (duffing.py:21-28)."""
        for k, (W, b) in enumerate(weights):
            W = np.ascontiguousarray(W, dtype=np.float64)
            b = np.ascontiguousarray(np.reshape(b, -1), dtype=np.float64)
            self._chk(self.lib.kmpc_set_encoder(self.h, k, _dptr(W), _dptr(b), W.shape[0], W.shape[1]), "kmpc_set_encoder")

    def set_centres(self, cx):
        cx = np.ascontiguousarray(cx, dtype=np.float64)
        self._chk(self.lib.kmpc_set_centres(self.h, _dptr(cx), cx.shape[0], cx.shape[1]), "kmpc_set_centres")

    def set_model(self, A, B, Cm=None):
        """Aloc_d, Bloc_d, Cloc_d = A, B, C  (duffing.py:811-813): the model used until the first update."""
        A = np.ascontiguousarray(A, dtype=np.float64)
        Bv = np.ascontiguousarray(np.reshape(B, -1), dtype=np.float64)
        cp = _dptr(Cm) if Cm is not None else None
        self._chk(self.lib.kmpc_set_model(self.h, _dptr(A), _dptr(Bv), cp), "kmpc_set_model")

    def set_terminal_weight(self, PN=None):
        """Q_bar(end-n+1:end, end-n+1:end) = C*P*C'  (Koopman_update.m:381): PN (q, q) replaces the last
        Qw*I block of the output weight; None restores it."""
        if PN is None:
            self._chk(self.lib.kmpc_set_terminal_weight(self.h, None), "kmpc_set_terminal_weight")
            return
        PN = np.ascontiguousarray(PN, dtype=np.float64)
        if PN.shape != (self.q, self.q):
            raise ValueError("PN must be (%d, %d)" % (self.q, self.q))
        self._chk(self.lib.kmpc_set_terminal_weight(self.h, _dptr(PN)), "kmpc_set_terminal_weight")

    def terminal_from_dare(self, Q=None, R=0.01, per_trajectory=False, maxiter=500, eps=0.01):
        """Terminal block from the reference's Riccati iteration on this controller's own model(s): P =
        solve_DARE(A, B, Q, R) (duffing.py:583-598; Q defaults to the reference's 10 I, R to its 0.01,
        duffing.py:667-668), Q_bar(end) = Co P Co' (Koopman_update.m:381).  per_trajectory=False: one block from
        trajectory 0's model (the shared one after set_model / offline_fit); True: every trajectory's current model
        gives its own block.  Returns (PN, iterations) as NumPy arrays ((q, q) or (B, q, q))."""
        Qh = np.ascontiguousarray(10.0 * np.eye(self.L) if Q is None else Q, dtype=np.float64)
        if Qh.shape != (self.L, self.L):
            raise ValueError("Q must be (%d, %d)" % (self.L, self.L))
        nb = self.B if per_trajectory else 1
        PN = np.zeros((nb, self.q, self.q))
        it = np.zeros(nb, dtype=np.int32)
        self._chk(self.lib.kmpc_terminal_from_dare(self.h, _dptr(Qh), float(R), int(maxiter), float(eps), 1 if per_trajectory else 0,
                                                   PN.ctypes.data_as(C.POINTER(C.c_double)), it.ctypes.data_as(C.POINTER(C.c_int32)),
                                                   self._stream()), "kmpc_terminal_from_dare")
        return (PN, it) if per_trajectory else (PN[0], int(it[0]))

    def set_terminal_refresh(self, every=1, Q=None, R=0.01, maxiter=500, eps=0.01):
        """The MATLAB controller's per-iteration terminal ingredients (Koopman_update.m:215 dlqr with the updated model, :381
        Q_bar(end) = C*P*C'): every `every`-th step each trajectory runs the reference's Riccati iteration (solve_DARE,
        duffing.py:583-598; Q defaults to its 10 I, R to its 0.01) on its freshly updated [A B] and rebuilds its terminal block before
        the QP is formed -- inside the launch for fused roll-outs.  every=0 switches it off."""
        Qh = np.ascontiguousarray(10.0 * np.eye(self.L) if Q is None else Q, dtype=np.float64)
        if Qh.shape != (self.L, self.L):
            raise ValueError("Q must be (%d, %d)" % (self.L, self.L))
        self._chk(self.lib.kmpc_set_terminal_refresh(self.h, int(every), _dptr(Qh), float(R), int(maxiter), float(eps)), "kmpc_set_terminal_refresh")

    def offline_fit(self, X, Y, U, ridge=0.0, init_rls=False):
        """K_hat = PHIY pinv([PHIX; U]), C = X pinv(PHIX) (duffing.py:152-177) on the device in Gram form
        (Koopman_update.m:94-101): lift, MFMA Gram sums, p x p solve.  X, Y (n, M), U (M,).  The result
        becomes every trajectory's model; returns (A, B, C) as device tensors.  init_rls=True also starts
        the online RLS from the offline data as Koopman_update.m:258-278 does (inv_K_G = pinv(V V'))."""
        Xd = self._dev(X, (self.n, -1))
        M = Xd.shape[1]
        Yd = self._dev(Y, (self.n, M))
        Ud = self._dev(U, (M,))
        A = torch.empty(self.L, self.L, dtype=self.dtype, device=self.device)
        Bm = torch.empty(self.L, 1, dtype=self.dtype, device=self.device)
        Cm = torch.empty(self.n, self.L, dtype=self.dtype, device=self.device)
        self._chk(self.lib.kmpc_offline_fit(self.h, self._p(Xd), self._p(Yd), self._p(Ud), M, float(ridge), int(bool(init_rls)), self._p(A),
                                            self._p(Bm), self._p(Cm), self._stream()), "kmpc_offline_fit")
        return A, Bm, Cm

    def generate_and_fit(self, kind, x0, U, ridge=0.0, init_rls=False, h=0.05, return_data=False):
        """data_generate.py:17-79 -> duffing.py:152-177 as one device pipeline: from the initial states x0 (n, n_traj) and the
        inputs U (n_steps, n_traj) (drawn by the caller, the reference uses np.random with seed 101) the plant roll-out,
        the lift of the n_traj * n_steps transitions, their Gram sums (MFMA) and the model solve run without returning to the
        host.  Returns (A, B, C) [+ (X, Y) panels (n, M) with return_data=True]; the model becomes every trajectory's."""
        plant = {"duffing": _ffi.KMPC_PLANT_DUFFING, "vdp": _ffi.KMPC_PLANT_VDP, "tank": _ffi.KMPC_PLANT_TANK}[kind]
        Ud = self._dev(U)
        n_steps, n_traj = Ud.shape
        X0 = self._dev(x0, (self.n, n_traj))
        M = n_steps * n_traj
        A = torch.empty(self.L, self.L, dtype=self.dtype, device=self.device)
        Bm = torch.empty(self.L, 1, dtype=self.dtype, device=self.device)
        Cm = torch.empty(self.n, self.L, dtype=self.dtype, device=self.device)
        Xo = torch.empty(self.n, M, dtype=self.dtype, device=self.device) if return_data else None
        Yo = torch.empty(self.n, M, dtype=self.dtype, device=self.device) if return_data else None
        self._chk(self.lib.kmpc_generate_and_fit(self.h, plant, self._p(X0), self._p(Ud), n_traj, n_steps, float(h), float(ridge),
                                                 int(bool(init_rls)), self._p(A), self._p(Bm), self._p(Cm),
                                                 self._p(Xo) if return_data else None, self._p(Yo) if return_data else None,
                                                 self._stream()), "kmpc_generate_and_fit")
        return (A, Bm, Cm, Xo, Yo) if return_data else (A, Bm, Cm)

    def state_init(self, P0=None, barQ0=None, K_A=None, inv_K_G=None, bar_X=None, bar_Q=None):
        """Start of the online update.  Scales only: K_A = 0, inv_K_G = P0 I, bar_X = 0, bar_Q = barQ0 I
        (duffing.py:927-930, 944-946).  With the reference's accumulators (K_A (L,p), inv_K_G (p,p), bar_X (n,L),
        bar_Q (L,L)): every trajectory continues from them (Koopman_update.m:264-265)."""
        if K_A is None:
            self._chk(self.lib.kmpc_state_init(self.h, float(self.cfg.P0 if P0 is None else P0),
                                               float(self.cfg.barQ0 if barQ0 is None else barQ0), self._stream()), "kmpc_state_init")
            return
        self._chk(self.lib.kmpc_state_init_from(self.h, _dptr(K_A), _dptr(inv_K_G), _dptr(bar_X) if bar_X is not None else None,
                                                _dptr(bar_Q), self._stream()), "kmpc_state_init_from")

    def reset(self):
        self._chk(self.lib.kmpc_reset(self.h, self._stream()), "kmpc_reset")

    # ------------------------------------------------------------------ a1/a2 lift
    def Encoder(self, x):
        """net.Encoder(x) (duffing.py:847).  x: (n,) | (n,1) | (n,B) -> (L,) | (L,1) | (L,B)."""
        was_np = not torch.is_tensor(x)
        shp = tuple(np.shape(x)) if was_np else tuple(x.shape)
        X = self._dev(x, (self.n, -1))
        Bc = X.shape[1]
        Psi = torch.empty(self.L, Bc, dtype=self.dtype, device=self.device)
        self._chk(self.lib.kmpc_lift(self.h, self._p(X), self._p(Psi), Bc, self._stream()), "kmpc_lift")
        if len(shp) == 1:
            Psi = Psi.reshape(self.L)
        return Psi.cpu().numpy() if was_np else Psi

    def rbf(self, X, cx=None):
        """rbf(X, cx) (vanderpol_RBF.py:20-23); returns (L,1) for a single state like the reference."""
        if cx is not None:
            self.set_centres(cx)
        was_np = not torch.is_tensor(X)
        out = self.Encoder(X)
        if was_np and out.ndim == 1:
            out = out.reshape(self.L, 1)
        return out

    # ------------------------------------------------------------------ a3/a4 RLS
    def Koopman_update(self, xlift, u, ylift, x_next):
        """The "Update LTV-Model" block (duffing.py:900, 927-953, 965-967) for every trajectory.
        xlift, ylift: (L,B); u: (B,) or (1,B); x_next: (n,B).  Returns (A, B, C) as
        ([B,L,L], [B,L,1], [B,n,L]) tensors (C is None for output="lift")."""
        xl = self._dev(xlift, (self.L, self.B))
        yl = self._dev(ylift, (self.L, self.B))
        uu = self._dev(u, (self.B,))
        xn = self._dev(x_next, (self.n, self.B))
        self._chk(self.lib.kmpc_rls_update(self.h, self._p(xl), self._p(uu), self._p(yl), self._p(xn), self.B,
                                           self._stream()), "kmpc_rls_update")
        return self.get_model()

    def get_model(self):
        A = torch.empty(self.B, self.L, self.L, dtype=self.dtype, device=self.device)
        Bm = torch.empty(self.B, self.L, 1, dtype=self.dtype, device=self.device)
        Cm = torch.empty(self.B, self.n, self.L, dtype=self.dtype, device=self.device) if self.output == "Cx" else None
        self._chk(self.lib.kmpc_get_model(self.h, self._p(A), self._p(Bm), self._p(Cm) if Cm is not None else None,
                                          self._stream()), "kmpc_get_model")
        return A, Bm, Cm

    # ------------------------------------------------------------------ a5-a8 MPC
    def _ref(self, r):
        # (fast path: a reference that already is a contiguous device tensor of the handle's dtype and shape -- the loop of a running
        #  controller hands the same tensor over at every call; the conversions below cost ~10 us of host time per call otherwise)
        if torch.is_tensor(r) and r.dtype == self.dtype and r.device == self.device and r.is_contiguous():
            if r.dim() == 2 and tuple(r.shape) == (self.q, self.N):
                return r, 0
            if r.dim() == 3 and tuple(r.shape) == (self.B, self.q, self.N):
                return r, 1
        r = self._dev(r)
        if r.dim() == 2:
            return r.reshape(self.q, self.N).contiguous(), 0
        return r.reshape(self.B, self.q, self.N).contiguous(), 1

    def condense(self, psi, r, return_const=False):
        """H [B,N,N], f [B,N] of the current model: J(u) = u'Hu + f'u + c is the reference's
        costFunction (duffing.py:540-581; Koopman_update.m:455-471).  r: (q,N) or (B,q,N).
        return_const=True also returns c [B] (so that c + the QP's optimal value is `result.fun`, duffing.py:859)."""
        ps = self._dev(psi, (self.L, self.B))
        rr, per = self._ref(r)
        H = torch.empty(self.B, self.N, self.N, dtype=self.dtype, device=self.device)
        f = torch.empty(self.B, self.N, dtype=self.dtype, device=self.device)
        c = torch.empty(self.B, dtype=self.dtype, device=self.device) if return_const else None
        self._chk(self.lib.kmpc_condense_cost(self.h, self._p(ps), self._p(rr), per, self._p(H), self._p(f),
                                              self._p(c) if return_const else None, self.B, self._stream()), "kmpc_condense_cost")
        return (H, f, c) if return_const else (H, f)

    def qp_solve(self, H, f):
        """Exact box-QP (replaces optimize.minimize duffing.py:857-861 / quadprog Koopman_update.m:214).
        H (Bq,N,N), f (Bq,N) -> U (N,Bq), status (Bq,), iters (Bq,)."""
        H = self._dev(H, (-1, self.N, self.N))
        Bq = H.shape[0]
        f = self._dev(f, (Bq, self.N))
        U = torch.empty(self.N, Bq, dtype=self.dtype, device=self.device)
        st = torch.empty(Bq, dtype=torch.int32, device=self.device)
        it = torch.empty(Bq, dtype=torch.int32, device=self.device)
        self._chk(self.lib.kmpc_qp_solve(self.h, self._p(H), self._p(f), self._p(U), self._p(st), self._p(it), Bq,
                                         self._stream()), "kmpc_qp_solve")
        return U, st, it

    def mpc_solve(self, *args, lb=None, ub=None, Q=None, R=None, P_N=None):
        """The MPC solve wrapper (duffing.py:857-861).

        mpc_solve(psi, r): `result.x` for the handle's current model -> (U (N,B), status, iters).
        mpc_solve(A, B, C, xlift, r[, lb, ub, Q, R, P_N]): stateless, the model is an argument -- A (L,L) or
        (B,L,L), B (L,) / (L,1) / (B,L[,1]), C (n,L) / (B,n,L) or None for output="lift"; xlift (L,B); scalar weights
        Q, R and bounds default to the handle's; P_N (q,q) replaces the last block of the output weight
        (Koopman_update.m:381) -> (U (N,B), u0 (B,), status (B,), fun (B,) = J at the minimiser)."""
        if len(args) == 2:
            H, f = self.condense(args[0], args[1])
            return self.qp_solve(H, f)
        if len(args) < 5:
            raise TypeError("mpc_solve(psi, r) or mpc_solve(A, B, C, xlift, r[, lb, ub, Q, R, P_N])")
        A, Bm, Cm, xlift, r = args[:5]
        rest = list(args[5:]) + [None] * 5
        lb = rest[0] if lb is None else lb; ub = rest[1] if ub is None else ub
        Q = rest[2] if Q is None else Q; R = rest[3] if R is None else R; P_N = rest[4] if P_N is None else P_N
        A = self._dev(A); shared = A.dim() == 2
        A = A.reshape(-1, self.L, self.L).contiguous()
        nb = A.shape[0]
        if nb not in (1, self.B):
            raise ValueError("A must be (L,L) or (B,L,L)")
        Bv = self._dev(Bm).reshape(nb, self.L).contiguous()
        Cd = self._dev(Cm).reshape(nb, self.n, self.L).contiguous() if Cm is not None else None
        ps = self._dev(xlift, (self.L, self.B))
        rr, per = self._ref(r)
        U = torch.empty(self.N, self.B, dtype=self.dtype, device=self.device)
        u0 = torch.empty(self.B, dtype=self.dtype, device=self.device)
        fun = torch.empty(self.B, dtype=self.dtype, device=self.device)
        st = torch.empty(self.B, dtype=torch.int32, device=self.device)
        it = torch.empty(self.B, dtype=torch.int32, device=self.device)
        pn = _dptr(np.reshape(P_N, (self.q, self.q))) if P_N is not None else None
        self._chk(self.lib.kmpc_mpc_solve(self.h, self._p(A), self._p(Bv), self._p(Cd) if Cd is not None else None, 1 if (shared or nb == 1) else 0,
                                          self._p(ps), self._p(rr), per, float(self.cfg.lb if lb is None else lb),
                                          float(self.cfg.ub if ub is None else ub), float(self.cfg.Qw if Q is None else np.ravel(Q)[0]),
                                          float(self.cfg.Rw if R is None else np.ravel(R)[0]), pn, self._p(U), self._p(u0), self._p(fun),
                                          self._p(st), self._p(it), self.B, self._stream()), "kmpc_mpc_solve")
        return U, u0, st, fun

    def step(self, x, r):
        """One loop iteration (duffing.py:847-984) for all trajectories: returns u_k (B,) (a view of
        self.U0; self.Useq, self.status, self.iters hold the rest).  x: (n,B); r: (q,N) | (B,q,N)."""
        X = self._dev(x, (self.n, self.B))
        rr, per = self._ref(r)
        self._chk(self.lib.kmpc_step(self.h, self._p(X), self._p(rr), per, self._p(self.U0), self._p(self.Useq),
                                     self._p(self.status), self._p(self.iters), self._stream()), "kmpc_step")
        return self.U0

    def set_applied_input(self, u):
        """The input that was actually applied at the last step when it is not the returned u_k (actuator limits, a logged
        trajectory that is being followed): the next update regresses on it (z = [psi; u], duffing.py:900).  u: (B,) or scalar."""
        uu = self._dev(np.broadcast_to(np.asarray(u, dtype=np.float64), (self.B,)) if not torch.is_tensor(u) else u, (self.B,))
        self._chk(self.lib.kmpc_set_applied_input(self.h, self._p(uu), self.B, self._stream()), "kmpc_set_applied_input")

    # ------------------------------------------------------------------ shared-model mode
    def shared_local_gram(self, x):
        """Stage 1 of a shared-model step: lift x_k and return this rank's Gram sums of the transitions
        (psi_{k-1}, u_{k-1}) -> (psi_k, x_k): float64 tensor ((p+L+n), p) = [Z Z'; Ylift Z'; X Z']
        (Koopman_update.m:94-98).  Zeros on the first call."""
        X = self._dev(x, (self.n, self.B))
        p = self.L + 1
        if getattr(self, "_delta", None) is None:
            self._delta = torch.zeros(p + self.L + self.n, p, dtype=torch.float64, device=self.device)
        self._chk(self.lib.kmpc_shared_local_gram(self.h, self._p(X), self._p(self._delta), self._stream()),
                  "kmpc_shared_local_gram")
        return self._delta

    def shared_solve(self, delta_gram, r, plant=None, X=None, switched=False, h=0.05):
        """Stage 2: G <- lambda G + delta_gram (summed over ranks by the caller), shared [A B], C from G,
        condensed QP of the shared model, box QP of every trajectory.  Returns u_k (B,).
        plant = "duffing" | "vdp" | "tank" with X (n, B): the solve also advances the plant, X <- f(X, u_k) in place
        (kmpc_shared_solve_plant; one launch less per step of a closed loop)."""
        rr, per = self._ref(r)
        if per:
            raise ValueError("shared-model mode takes one reference (q, N) for the whole batch")
        d = delta_gram.to(device=self.device, dtype=torch.float64).contiguous()
        if plant is None:
            self._chk(self.lib.kmpc_shared_solve(self.h, self._p(d), self._p(rr), self._p(self.U0), self._p(self.Useq),
                                                 self._p(self.status), self._p(self.iters), self._stream()),
                      "kmpc_shared_solve")
        else:
            pl = {"duffing": _ffi.KMPC_PLANT_DUFFING, "vdp": _ffi.KMPC_PLANT_VDP, "tank": _ffi.KMPC_PLANT_TANK}[plant]
            assert X is not None and X.is_cuda and X.dtype == self.dtype and X.is_contiguous() and tuple(X.shape) == (self.n, self.B)
            self._chk(self.lib.kmpc_shared_solve_plant(self.h, self._p(d), self._p(rr), self._p(self.U0), self._p(self.Useq),
                                                       self._p(self.status), self._p(self.iters), pl, self._p(X),
                                                       int(bool(switched)), float(h), self._stream()),
                      "kmpc_shared_solve_plant")
        return self.U0

    def shared_step(self, x, r, plant=None, switched=False, h=0.05):
        """One control step with ONE model for all trajectories of all ranks: local Gram sums (MFMA) ->
        all-reduce over the process group (RCCL on GPUs; the only collective of the path) -> model, QP.
        With `plant`, x (a contiguous (n, B) device tensor) is advanced in place by the solve kernel."""
        import torch.distributed as dist

        delta = self.shared_local_gram(x)
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_collectives()):
            if dist.get_backend() == "gloo":  # (rehearsals of the multi-rank path on CPU-side collectives: gloo reduces host tensors)
                hd = delta.cpu()
                dist.all_reduce(hd, op=dist.ReduceOp.SUM)
                delta.copy_(hd)
            else:
                dist.all_reduce(delta, op=dist.ReduceOp.SUM)  # RCCL over xGMI: the one collective of the path
        return self.shared_solve(delta, r, plant=plant, X=x if plant is not None else None, switched=switched, h=h)

    def shared_rollout(self, kind, X, r, steps, step0=0, switch_step=101, h=0.05, comm=None, log=False):
        """`steps` iterations of the shared-model loop (Tank_System.m:170-291 with one model for all trajectories of all ranks)
        enqueued from C++ on the current stream: lift + local Gram sums -> ncclAllReduce over `comm` (an RCCL communicator handle,
        koopmpc.sharding.NcclCommunicator; None: single rank) -> model, condensed QP -> box QPs with the plant inside
        (kmpc_shared_rollout).  X (n, B) is advanced in place; self.status / self.iters receive the worst status / total Newton
        solves, self.U0 / self.Useq the last step's inputs / sequences.  Returns (U_log, X_log) with log=True."""
        plant = self._plant_id(kind)
        assert X.is_cuda and X.dtype == self.dtype and X.is_contiguous() and tuple(X.shape) == (self.n, self.B)
        rr, per = self._ref(r)
        if per:
            raise ValueError("shared-model mode takes one reference (q, N) for the whole batch")
        Ul = torch.empty(steps, self.B, dtype=self.dtype, device=self.device) if log else None
        Xl = torch.empty(steps, self.n, self.B, dtype=self.dtype, device=self.device) if log else None
        cptr = None if comm is None else (comm.handle if hasattr(comm, "handle") else comm)
        self._chk(self.lib.kmpc_shared_rollout(self.h, plant, self._p(X), self._p(rr), int(steps), int(step0), int(switch_step), float(h),
                                               cptr, self._p(Ul) if log else None, self._p(Xl) if log else None, self._p(self.U0), self._p(self.Useq),
                                               self._p(self.status), self._p(self.iters), self._stream()), "kmpc_shared_rollout")
        self._keep = (rr,)
        return (Ul, Xl) if log else None

    def shared_model(self):
        A = torch.empty(self.L, self.L, dtype=self.dtype, device=self.device)
        Bm = torch.empty(self.L, 1, dtype=self.dtype, device=self.device)
        Cm = torch.empty(self.n, self.L, dtype=self.dtype, device=self.device) if self.output == "Cx" else None
        self._chk(self.lib.kmpc_shared_get_model(self.h, self._p(A), self._p(Bm), self._p(Cm) if Cm is not None else None,
                                                 self._stream()), "kmpc_shared_get_model")
        return A, Bm, Cm

    # ------------------------------------------------------------------ plant (adjacent)
    @staticmethod
    def _plant_id(kind):
        """"duffing" | "vdp" | "tank"; a "_matlab" suffix selects the Runge-Kutta step as the MATLAB scripts write it
        (k4 evaluated at x + h k1, Koopman_update.m:24)."""
        base = {"duffing": _ffi.KMPC_PLANT_DUFFING, "vdp": _ffi.KMPC_PLANT_VDP, "tank": _ffi.KMPC_PLANT_TANK}
        if kind.endswith("_matlab"):
            return base[kind[:-7]] | _ffi.KMPC_PLANT_RK4_MATLAB
        return base[kind]

    def set_online_update(self, on=True):
        """on=False: step / rollout run the reference's loop WITHOUT the online update (duffing.py:738-805): the model stays."""
        self._chk(self.lib.kmpc_set_online_update(self.h, int(bool(on))), "kmpc_set_online_update")

    def plant_step(self, kind, X, U, h=0.05, switched=False):
        """X <- f_update(0, X, U) in place on the device (duffing.py:256-261)."""
        plant = self._plant_id(kind)
        assert X.is_cuda and X.dtype == self.dtype and X.is_contiguous()
        Uu = self._dev(U, (X.shape[1],))
        self._chk(self.lib.kmpc_plant_step(self.h, plant, self._p(X), self._p(Uu), float(h), int(bool(switched)),
                                           X.shape[1], self._stream()), "kmpc_plant_step")
        return X

    def rollout(self, kind, X, r, steps, step0=0, switch_step=102, h=0.05, log=False):
        """`steps` iterations of the reference loop body (duffing.py:823-1012) enqueued from C++:
        u_i = step(X); X <- plant(X, u_i), switched parameters from iteration `switch_step` on
        (the reference flips them at the end of iteration 101).  X (n,B) device tensor, updated in place.
        self.status / self.iters receive each trajectory's worst QP status and total Newton solves.
        Returns (U_log (steps,B), X_log (steps,n,B)) when log=True."""
        plant = self._plant_id(kind)
        assert X.is_cuda and X.dtype == self.dtype and X.is_contiguous() and tuple(X.shape) == (self.n, self.B)
        rr, per = self._ref(r)
        Ul = torch.empty(steps, self.B, dtype=self.dtype, device=self.device) if log else None
        Xl = torch.empty(steps, self.n, self.B, dtype=self.dtype, device=self.device) if log else None
        self._chk(self.lib.kmpc_rollout(self.h, plant, self._p(X), self._p(rr), per, int(steps), int(step0),
                                        int(switch_step), float(h), self._p(Ul) if log else None,
                                        self._p(Xl) if log else None, self._p(self.status), self._p(self.iters),
                                        self._stream()), "kmpc_rollout")
        self._keep = (rr,)  # keep the reference tensor alive until the stream has consumed it
        return (Ul, Xl) if log else None

    # ------------------------------------------------------------------ checkpoint
    def state_dict(self):
        nb = self.lib.kmpc_state_bytes(self.h)
        buf = np.empty(nb, dtype=np.uint8)
        self._chk(self.lib.kmpc_state_export(self.h, C.c_void_p(buf.ctypes.data), nb), "kmpc_state_export")
        return {"blob": buf}

    def state_to(self, blob=None):
        """Snapshot into a uint8 tensor (host or device; allocated on the device when None) -- the blob of
        state_dict() without the host round trip."""
        nb = self.lib.kmpc_state_bytes(self.h)
        if blob is None:
            blob = torch.empty(nb, dtype=torch.uint8, device=self.device)
        assert blob.dtype == torch.uint8 and blob.numel() >= nb and blob.is_contiguous()
        self._chk(self.lib.kmpc_state_export(self.h, C.c_void_p(blob.data_ptr()), nb), "kmpc_state_export")
        return blob

    def state_from(self, blob):
        self._chk(self.lib.kmpc_state_import(self.h, C.c_void_p(blob.data_ptr()), blob.numel()), "kmpc_state_import")

    def load_state_dict(self, sd):
        buf = np.ascontiguousarray(sd["blob"], dtype=np.uint8)
        self._chk(self.lib.kmpc_state_import(self.h, C.c_void_p(buf.ctypes.data), buf.size), "kmpc_state_import")

    # ------------------------------------------------------------------ measurement
    def profile(self, on=True):
        self._chk(self.lib.kmpc_profile_enable(self.h, int(on)), "kmpc_profile_enable")

    def profile_read(self, reset=True):
        ms = (C.c_double * 2)()
        cnt = C.c_int64()
        self._chk(self.lib.kmpc_profile_read(self.h, ms, C.byref(cnt), int(reset)), "kmpc_profile_read")
        return {"lift_ms": ms[0], "step_ms": ms[1], "count": cnt.value}

    def rollout_is_fused(self):
        """True if rollout() runs as one fused kernel launch for this configuration."""
        return bool(self.lib.kmpc_rollout_is_fused(self.h))

    def rollout_plugin_status(self):
        """(code, text): where the fused roll-out kernel of this controller comes from -- 0 the library's own instantiation, 1 a
        plug-in made for this dimension set when the controller was created (text: its file, found in the kernel cache or compiled
        with hipcc in so many seconds), -1 the plug-in could not be made (text: why; per-step launches), 2 no fused roll-out."""
        buf = C.create_string_buffer(1024)
        code = int(self.lib.kmpc_rollout_plugin_status(self.h, buf, len(buf)))
        return code, buf.value.decode("utf-8", "replace")

    def algorithmic_bytes_per_step(self):
        return int(self.lib.kmpc_algorithmic_bytes_per_step(self.h))


# ----------------------------------------------------------------------------------------------------------------------
# The reference's own names at module level: net.Encoder(x), rbf(X, cx), costFunction(uSequence, r, AB, C, x0, ...),
# Koopman_update, mpc_solve, mpc_step.  Thin: each one is a call of a (cached) KoopmanMPC handle, i.e. of the HIP kernels.
# ----------------------------------------------------------------------------------------------------------------------
_handles = {}


def _handle(key, make):
    h = _handles.get(key)
    if h is None:
        h = _handles[key] = make()
    return h


class AutoEncoder:
    """`net = AutoEncoder(...)`; `net.Encoder(x)` (duffing.py:17-29, 847): the encoder half only (the decoder is not on the path).
    weights = [(W1, b1), ..., (Wd+1, bd+1)] with W (out x in), as `scipy.io.loadmat` of the reference's weight files gives them
    (koopmpc.io.load_encoder_mat).  Encoder(x): (n,) | (n, 1) | (n, B) -> (L,) | (L, 1) | (L, B), on the device (kmpc_lift)."""

    def __init__(self, weights, device=None, lift_offset=None):
        self.weights = [(np.asarray(W, dtype=np.float64), np.asarray(b, dtype=np.float64).reshape(-1)) for W, b in weights]
        n, L = self.weights[0][0].shape[1], self.weights[-1][0].shape[0]
        if lift_offset == "x_psi0":
            L += n
        self._mpc = KoopmanMPC(n=n, L=L, N=2, batch=1, weights=self.weights, hidden=self.weights[0][0].shape[0],
                               layers=len(self.weights) - 1, device=device, lift_offset=lift_offset)

    def Encoder(self, x):
        return self._mpc.Encoder(x)

    __call__ = Encoder


def rbf(X, cx, eps=1e-4, form="python", device=None):
    """rbf(X, cx) of vanderpol_RBF.py:20-23 / duffing_RBF.py:20-23 (form="matlab": rbf.m:24-29): psi_j = d_j^2 log(d_j + eps).
    X (n,) | (n, 1) | (n, B), cx (L, n) -> (L, 1) for a single state like the reference, else (L, B)."""
    cx = np.ascontiguousarray(cx, dtype=np.float64)
    L, n = cx.shape
    m = _handle(("rbf", L, n, form, float(eps), str(device)),
                lambda: KoopmanMPC(n=n, L=L, N=2, batch=1, lift="rbf" if form == "python" else "rbf_matlab", centres=cx, rbf_eps=eps, device=device))
    return m.rbf(X, cx)


def costFunction(uSequence, r, AB, C, x0, pastu=None, Np=None, Nc=None, d=None, ek=None, Qw=100.0, Rw=1e-4, device=None):
    """costFunction(uSequence, r, AB, C, x0, pastu, Np, Nc, d, ek) of duffing.py:540-581 / vanderpol.py:445-487 for caller-supplied
    input sequences, evaluated on the device: J(u) = Qw sum_k |C x_k - r_{:,k-1}|^2 + Rw sum u^2, x_k = AB [x_{k-1}; u_{k-1}]
    (the condensed form u'Hu + f'u + c of kmpc_condense_cost, which equals the reference's value to 1e-12).
    uSequence (N,) or (N, S) for S sequences at once; r (q, N); AB (L, L+1); C (q, L) or None for y = the lifted state
    (vanderpol.py:456-459); x0 (L,) | (L, 1).  pastu, ek and d are dead in the reference (:576-580; d is zeros, :776) -- accepted and
    ignored, a non-zero d is refused; Np must equal Nc (= N, the reference's setting).  Returns a float (one sequence) or (S,)."""
    U = np.asarray(uSequence.detach().cpu() if torch.is_tensor(uSequence) else uSequence, dtype=np.float64)
    single = U.ndim == 1
    U = U.reshape(U.shape[0], -1)
    N, S = U.shape
    if (Np is not None and Np != N) or (Nc is not None and Nc != N):
        raise ValueError("costFunction: Np = Nc = len(uSequence) is the only setting the reference uses (duffing.py:632-633)")
    if d is not None and np.any(np.asarray(d) != 0.0):
        raise ValueError("costFunction: the offset d is zeros in the reference (duffing.py:776); a non-zero d is not supported")
    AB = np.asarray(AB, dtype=np.float64)
    L = AB.shape[0]
    out = "lift" if C is None else "Cx"
    q = L if C is None else np.asarray(C).shape[0]
    m = _handle(("cost", L, N, S, out, q, float(Qw), float(Rw), str(device)),
                lambda: KoopmanMPC(n=q if C is not None else 2, L=L, N=N, batch=S, lift="rbf", centres=np.zeros((L, q if C is not None else 2)),
                                   output=out, Qw=Qw, Rw=Rw, device=device))
    m.set_model(AB[:, :L], AB[:, L], np.asarray(C, dtype=np.float64) if C is not None else None)
    psi = np.tile(np.asarray(x0, dtype=np.float64).reshape(L, 1), (1, S))
    H, f, c = m.condense(psi, np.asarray(r, dtype=np.float64).reshape(q, N), return_const=True)
    Ud = torch.as_tensor(U.T.copy(), device=m.device)  # (S, N)
    J = torch.einsum("si,sij,sj->s", Ud, H, Ud) + (f * Ud).sum(1) + c
    return float(J[0]) if single else J.cpu().numpy()


def Koopman_update(state, xlift, u, ylift, x_next, lam=None):
    """The "Update LTV-Model" block (duffing.py:900-967; Koopman_update.m:258-278) on `state`, a KoopmanMPC handle that owns the
    accumulators (K_A inv_K_G, bar_X bar_Q) of its trajectories: returns (A, B, C).  lam must be the handle's (it is a launch constant)."""
    if lam is not None and abs(float(lam) - float(state.cfg.lam)) > 0:
        raise ValueError("Koopman_update: lam is fixed when the handle is created (KoopmanMPC(lam=...))")
    return state.Koopman_update(xlift, u, ylift, x_next)


koopman_update = Koopman_update  # (SURVEY 8b spells it in lower case)


def mpc_solve(A, B, C, xlift, r, lb=-2.0, ub=2.0, Q=100.0, R=1e-4, P_N=None, device=None):
    """optimize.minimize(lamdaCostFunction(AB, C, x0, r), zeros(N), bounds=...) of duffing.py:857-861 with the model as an argument
    (SURVEY 8b): exact minimiser instead of L-BFGS-B.  A (L, L), B (L,) | (L, 1), C (q, L) | None, xlift (L,) | (L, B), r (q, N).
    Returns (U (N, B), u0 (B,), status (B,)) as NumPy arrays."""
    A = np.asarray(A, dtype=np.float64)
    L = A.shape[0]
    xl = np.asarray(xlift.detach().cpu() if torch.is_tensor(xlift) else xlift, dtype=np.float64).reshape(L, -1)
    Bn = xl.shape[1]
    rr = np.asarray(r, dtype=np.float64)
    q, N = rr.shape
    out = "lift" if C is None else "Cx"
    m = _handle(("solve", L, N, Bn, out, q, str(device)),
                lambda: KoopmanMPC(n=q if C is not None else 2, L=L, N=N, batch=Bn, lift="rbf", centres=np.zeros((L, q if C is not None else 2)),
                                   output=out, device=device))
    U, u0, st, fun = m.mpc_solve(A, B, C, xl, rr, lb, ub, Q, R, P_N)
    return U.cpu().numpy(), u0.cpu().numpy(), st.cpu().numpy()


def mpc_step(state, x, r):
    """One iteration of the reference loop (duffing.py:847-984) on the handle `state`: lift, RLS update with the previous transition,
    condensed QP, box QP -> u_k (B,)."""
    return state.step(x, r)
