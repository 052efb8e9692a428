"""CPU oracle for the Koopman online-updated MPC hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain NumPy (float64) restatement of the reference's per-step arithmetic
(MichaelMillerCSU/Koopman-online-updated-MPC): lift -> RLS-EDMD update of [A B] and C ->
condensed QP build -> box-QP solve.  Every function cites the reference lines it follows.
It is the CHECKER for the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product (``koopmpc`` +
``libkoopmpc.so``) never does and fails loudly without its HIP library.

Parity status ("pinned" = checked against values computed by the reference itself, see
``tests/golden/make_golden.py`` and ``tests/test_oracle_golden.py``):

  * mlp_lift, rbf_lift (Python eps form)  pinned   (reference encoder / rbf outputs, 64 states each)
  * rls_update_reference                   pinned   (130 consecutive steps of duffing.py / vanderpol.py)
  * cost_function                          pinned   (reference costFunction values)
  * condense (Gamma, Phi, H, f)            pinned through cost_function:  u'Hu + f'u + c == J(u)
                                           (the MATLAB construction Koopman_update.m:455-471 itself
                                           cannot run here: no MATLAB/Octave -> "parity unpinned"
                                           for the MATLAB-only weights/terminal block)
  * qp_exact                               the exact minimiser of the reference's own cost; the
                                           reference's solver (SciPy L-BFGS-B with finite-difference
                                           gradients, duffing.py:857-861) only approximates it
                                           (|u0 - u0*| up to 5e-4), so its stored outputs are a
                                           loose cross-check, and solve_lbfgsb() reproduces it
  * rbf_lift (MATLAB form), tank Delta-u augmentation, forgetting factor:  parity unpinned
    (MATLAB-only, restated from the .m text)

Third-party arithmetic on the path: SciPy ``optimize.minimize`` (L-BFGS-B; 1.15.3 in this
image, the reference pins no version) and LAPACK ``pinv``; both are outside /root/reference.
"""
from __future__ import annotations

import numpy as np

# ----------------------------------------------------------------------------------------
# a1 / a2  lifting
# ----------------------------------------------------------------------------------------


def mlp_lift(weights, X):
    """psi = W_d relu(... relu(W_1 x + b_1) ...) + b_d  for every column of X (n x B).

    duffing.py:17-29 (nn.Sequential Linear/ReLU stack), Revise_2/Encoder_Duffing.m:3-6,
    Encoder_Tank.m:3-5 (two hidden layers).  ``weights`` = [(W1, b1), ..., (Wd, bd)] with
    W (out x in), b (out,).
    """
    H = np.asarray(X, dtype=np.float64)
    if H.ndim == 1:
        H = H[:, None]
    for k, (W, b) in enumerate(weights):
        H = W @ H + np.reshape(b, (-1, 1))
        if k + 1 < len(weights):
            H = np.maximum(H, 0.0)
    return H


def mlp_lift_offset(weights, X, form):
    """The lifts of the MATLAB scripts on top of the encoder:
    form "psi0":   liftFun = @(x) [Encoder_VDP(x)] - [Encoder_VDP(zeros(2, 1))]                     VDP_Revise_2/Koopman_update_Tracking_Lift.m:65
    form "x_psi0": liftFun = @(x) [x; Encoder_Duffing(x)] - [zeros(2, 1); Encoder_Duffing(zeros(2, 1))]   Revise_2/Koopman_update.m:67 (Nlift = n + 8, :70)"""
    X = np.asarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X[:, None]
    E = mlp_lift(weights, X) - mlp_lift(weights, np.zeros((X.shape[0], 1)))
    if form == "psi0":
        return E
    if form == "x_psi0":
        return np.concatenate([X, E], axis=0)
    raise ValueError(form)


def load_mlp_weights(npz):
    """[(W1,b1),...] from a tests/golden/weights_*.npz (b stored 1 x k as in the .mat)."""
    out = []
    k = 1
    while "W%d" % k in npz:
        out.append((np.array(npz["W%d" % k], float), np.array(npz["b%d" % k], float).reshape(-1)))
        k += 1
    return out


def rbf_lift(X, cx, eps=1e-4, form="python"):
    """Thin-plate RBF dictionary, one row per centre.

    form="python": psi_j = d_j^2 * log(d_j + 1e-4), d_j = ||x - c_j||  (vanderpol_RBF.py:20-23,
                   with sklearn's  d = sqrt(max(|x|^2 - 2 x.c + |c|^2, 0)) ).
    form="matlab": psi_j = r2 * log(sqrt(r2)), NaN -> 0               (rbf.m:24-29).
    X: n x B, cx: L x n  ->  L x B.
    """
    X = np.asarray(X, float)
    if X.ndim == 1:
        X = X[:, None]
    cx = np.asarray(cx, float)
    if form == "python":
        xx = np.sum(X * X, axis=0)[None, :]
        cc = np.sum(cx * cx, axis=1)[:, None]
        d2 = np.maximum(xx - 2.0 * (cx @ X) + cc, 0.0)
        d = np.sqrt(d2)
        return d * d * np.log(d + eps)
    r2 = np.sum((X[None, :, :] - cx[:, :, None]) ** 2, axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        y = r2 * np.log(np.sqrt(r2))
    y[np.isnan(y)] = 0.0
    return y


# ----------------------------------------------------------------------------------------
# a3 / a4  recursive least squares ("Koopman update")
# ----------------------------------------------------------------------------------------


class RlsStateRef:
    """The reference's module-level globals K_A, inv_K_G, bar_X, bar_Q (duffing.py:927-953)."""

    def __init__(self, L, m, n, P0=1e4, barQ0=100.0):
        p = L + m
        self.K_A = np.zeros((L, p))
        self.P = P0 * np.eye(p)  # inv_K_G = pinv(1e-4 I)            duffing.py:929-930
        self.bar_X = np.zeros((n, L))
        self.bar_Q = barQ0 * np.eye(L)  # duffing.py:946
        self.K = np.zeros((L, p))
        self.C = np.zeros((n, L))


def rls_update_reference(st, xlift, u, ylift, x_next, lam=1.0):
    """One online update exactly as written in the reference (lambda = 1 is the only value it uses).

    z = [xlift; u]                                              duffing.py:900
    K_A += ylift z' ; P -= (P z z' P) / (1 + z' P z) ; K = K_A P duffing.py:927-938
    bar_X += x_{k+1} xlift' ; bar_Q -= ... ; C = bar_X bar_Q    duffing.py:943-953
    (the pairing of x_{k+1} with psi(x_k) is the reference's, reproduced as written)
    lam != 1 follows Koopman_update.m:270-274 (P only is discounted).
    """
    xlift = np.reshape(xlift, (-1, 1))
    ylift = np.reshape(ylift, (-1, 1))
    z = np.concatenate([xlift, np.reshape(u, (-1, 1))], axis=0)
    st.K_A = st.K_A + ylift @ z.T
    P = st.P
    st.P = P / lam - (1.0 / lam) * (P @ z @ z.T @ P) / (lam + z.T @ P @ z)
    st.K = st.K_A @ st.P
    x_next = np.reshape(x_next, (-1, 1))
    st.bar_X = st.bar_X + x_next @ xlift.T
    Qb = st.bar_Q
    st.bar_Q = Qb - (Qb @ xlift @ xlift.T @ Qb) / (1.0 + xlift.T @ Qb @ xlift)
    st.C = st.bar_X @ st.bar_Q
    L = xlift.shape[0]
    return st.K[:, :L].copy(), st.K[:, L:].copy(), st.C.copy()  # duffing.py:965-967


def rls_update_gain(K, P, z, y, lam=1.0):
    """Gain form of rls_update_reference, algebraically identical for every lam: with K = K_A P, P symmetric,
    g = P z / (lam + z' P z) one has P_new z = g, so K_A_new P_new = (K - K z g') / lam + y g'
    (K_A is NOT discounted, Koopman_update.m:270-274); lam = 1: K += (y - K z) g'.
    P <- (P - g (P z)') / lam.  This is the form the HIP kernel evaluates; the tests check it against
    rls_update_reference (also at lam < 1)."""
    z = np.reshape(z, (-1, 1))
    y = np.reshape(y, (-1, 1))
    Pz = P @ z
    d = lam + float((z.T @ Pz)[0, 0])
    g = Pz / d
    Kn = (K - (K @ z) @ g.T) / lam + y @ g.T if lam != 1.0 else K + (y - K @ z) @ g.T
    Pn = (P - Pz @ Pz.T / d) / lam
    return Kn, Pn


class SharedEdmd:
    """Shared-model EDMD in Gram form (Koopman_update.m:94-101): V = [Xlift; U], W = [Ylift; X],
    G = V V', [A B] = (Ylift V')(G + I/P0)^-1, C = (X Xlift')(G_LL + I/barQ0)^-1.  The ridge terms are
    the reference's RLS initialisations (duffing.py:929-930, 946): for one trajectory this IS the
    recursion of rls_update_reference (matrix inversion lemma), for B trajectories it pools them."""

    def __init__(self, L, n, P0=1e4, barQ0=100.0, lam=1.0):
        p = L + 1
        self.L, self.n, self.p, self.P0, self.barQ0, self.lam = L, n, p, P0, barQ0, lam
        self.G = np.zeros((p, p)); self.YZ = np.zeros((L, p)); self.XZ = np.zeros((n, p))

    @staticmethod
    def gram(Psi, U, PsiN, XN):
        """Gram sums of a batch of transitions: Psi, PsiN (L,B), U (B,), XN (n,B) -> (G, YZ, XZ)."""
        Z = np.concatenate([Psi, np.reshape(U, (1, -1))], axis=0)
        return Z @ Z.T, PsiN @ Z.T, XN @ Z.T

    def add(self, G, YZ, XZ):
        self.G = self.lam * self.G + G
        self.YZ = self.lam * self.YZ + YZ
        self.XZ = self.lam * self.XZ + XZ

    def model(self):
        L, p = self.L, self.p
        K = np.linalg.solve(self.G + np.eye(p) / self.P0, self.YZ.T).T
        C = np.linalg.solve(self.G[:L, :L] + np.eye(L) / self.barQ0, self.XZ[:, :L].T).T
        return K[:, :L].copy(), K[:, L:].copy(), C


# ----------------------------------------------------------------------------------------
# a5  shooting cost == a7 condensed QP
# ----------------------------------------------------------------------------------------


def cost_function(u, r, AB, C, x0, Qw=100.0, Rw=1e-4):
    """J(u) = Qw * sum_k |C x_k - r_{k-1}|^2 + Rw * sum u^2,  x_k = AB [x_{k-1}; u_{k-1}].

    duffing.py:540-581 (d = 0, Np = Nc); vanderpol.py:445-487 uses y = x (pass C = None).
    """
    x = np.reshape(np.asarray(x0, float), (-1, 1))
    J = 0.0
    for k, uk in enumerate(np.asarray(u, float).reshape(-1)):
        x = AB @ np.concatenate([x, [[uk]]], axis=0)
        y = x if C is None else C @ x
        e = y - np.reshape(r[:, k], (-1, 1))
        J += Qw * float(np.sum(e * e))
    return J + Rw * float(np.sum(np.square(u)))


def condense(A, B, Co, psi, r, N, Qw=100.0, Rw=1e-4, PN=None):
    """Condensed (dense) QP of the linear MPC:  minimise  u'Hu + f'u  (+ const).

    Gamma_k = Co A^k (k = 1..N), g_j = Co A^j B, Phi[i,j] = g_{i-j} (j <= i)
        Revise_2/Koopman_update.m:455-467 (Compact_Form1 / Compact_Form2, Co = Cy*C)
    H = Phi' Qbar Phi + Rbar, symmetrised;  f = 2 (Gamma psi - Yr)' Qbar Phi
        Koopman_update.m:469-471, :213   (quadprog(2H, f) minimises u'Hu + f'u)
    Qbar = kron(I_N, Qw I_q), optional terminal block PN (q x q) replaces the last Qw I_q
        (Koopman_update.m:381).  Python weights Qw = 100, Rw = 1e-4 (duffing.py:580).
    Co: q x L (None -> identity: output = lifted state, vanderpol.py:456-459); r: q x N.
    Returns Gamma (qN x L), Phi (qN x N), H (N x N), f (N,), const (so that J = u'Hu+f'u+const).
    """
    A = np.asarray(A, float)
    B = np.reshape(np.asarray(B, float), (A.shape[0], -1))
    assert B.shape[1] == 1, "m = 1 as in the reference (duffing.py:170-171)"
    L = A.shape[0]
    Co = np.eye(L) if Co is None else np.asarray(Co, float)
    q = Co.shape[0]
    Gam = np.zeros((q * N, L))
    Phi = np.zeros((q * N, N))
    M = Co.copy()  # Co A^j
    g = []
    for j in range(N):
        g.append(M @ B)  # g_j
        M = M @ A
        Gam[q * j : q * (j + 1), :] = M  # Co A^{j+1}
    for i in range(N):
        for j in range(i + 1):
            Phi[q * i : q * (i + 1), j : j + 1] = g[i - j]
    Qbar = Qw * np.eye(q * N)
    if PN is not None:
        Qbar[-q:, -q:] = PN
    H = Phi.T @ Qbar @ Phi + Rw * np.eye(N)
    H = 0.5 * (H + H.T)
    e = Gam @ np.reshape(psi, (-1, 1)) - np.reshape(np.asarray(r, float).T, (-1, 1))  # stacked (k, q)
    f = 2.0 * (Phi.T @ Qbar @ e).reshape(-1)
    const = float((e.T @ Qbar @ e)[0, 0])
    return Gam, Phi, H, f, const


# ----------------------------------------------------------------------------------------
# a6 / a8  the solve
# ----------------------------------------------------------------------------------------


def solve_dare(A, B, Q, R, maxiter=500, eps=0.01):
    """duffing.py:583-598: X = Q; X <- A'XA - A'XB pinv(R + B'XB) B'XA + Q, stop after the first iterate with
    max|X_new - X| < eps (returned) or after maxiter.  Also returns the number of iterations done."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64).reshape(A.shape[0], -1)
    X = np.asarray(Q, dtype=np.float64)
    R = np.atleast_2d(np.asarray(R, dtype=np.float64))
    Xn, it = X, 0
    for it in range(1, maxiter + 1):
        Xn = A.T @ X @ A - A.T @ X @ B @ np.linalg.pinv(R + B.T @ X @ B) @ B.T @ X @ A + Q
        if np.abs(Xn - X).max() < eps:
            break
        X = Xn
    return Xn, it


def dlqr(A, B, Q, R, maxiter=500, eps=0.01):
    """duffing.py:600-613: K = pinv(B'XB + R) (B'XA), X = solve_DARE(A, B, Q, R)."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64).reshape(A.shape[0], -1)
    X, _ = solve_dare(A, B, Q, R, maxiter, eps)
    R = np.atleast_2d(np.asarray(R, dtype=np.float64))
    return np.linalg.pinv(B.T @ X @ B + R) @ (B.T @ X @ A)


def terminal_block(Co, P):
    """Q_bar(end-n+1:end, end-n+1:end) = C*P*C'   (Koopman_update.m:381)."""
    Co = np.asarray(Co, dtype=np.float64)
    return Co @ np.asarray(P, dtype=np.float64) @ Co.T


def qp_exact(H, f, lb, ub, tol=1e-11, max_iter=None):
    """Exact minimiser of u'Hu + f'u over the box lb <= u <= ub (H SPD): primal active set.

    This is the parity TARGET for the MPC solve: the unique minimiser of the reference's own
    cost (duffing.py:540-581 == condense()).  quadprog(2H, f, ..., lb, ub) of
    Koopman_update.m:214 returns the same point.  Finite termination (strictly convex).
    """
    H = np.asarray(H, float)
    f = np.asarray(f, float).reshape(-1)
    n = f.size
    lb = np.broadcast_to(np.asarray(lb, float), (n,)).copy()
    ub = np.broadcast_to(np.asarray(ub, float), (n,)).copy()
    x = np.clip(np.zeros(n), lb, ub)
    act = np.zeros(n, dtype=int)  # 0 free, -1 at lb, +1 at ub
    act[x <= lb] = -1
    act[x >= ub] = 1
    max_iter = max_iter or 20 * n + 20
    gscale = np.abs(f) + 2.0 * (np.abs(H) @ np.maximum(np.abs(lb), np.abs(ub)))  # magnitude of the gradient's terms
    at_min = False  # x is the minimiser on the current working set (a full step was just taken)
    for it in range(max_iter):
        F = act == 0
        grad = 2.0 * (H @ x) + f
        p = np.zeros(n)
        if F.any() and not at_min:
            p[F] = np.linalg.solve(2.0 * H[np.ix_(F, F)], -grad[F])
        if at_min or np.max(np.abs(p)) <= 1e-9 * max(1.0, np.max(np.abs(x))):
            # stationary on the working set: check multipliers
            viol = np.where(act == -1, -grad, np.where(act == 1, grad, 0.0)) / np.maximum(gscale, 1e-300)
            j = int(np.argmax(viol))  # > 0 means wrong sign
            if viol[j] <= tol:
                return x, it
            act[j] = 0
            at_min = False
            continue
        alpha, blk = 1.0, -1
        for i in np.nonzero(F)[0]:
            if p[i] < 0 and x[i] + p[i] < lb[i]:
                a = (lb[i] - x[i]) / p[i]
                if a < alpha:
                    alpha, blk = a, i
            elif p[i] > 0 and x[i] + p[i] > ub[i]:
                a = (ub[i] - x[i]) / p[i]
                if a < alpha:
                    alpha, blk = a, i
        x = x + alpha * p
        at_min = blk < 0
        if blk >= 0:
            if p[blk] < 0:
                x[blk] = lb[blk]
                act[blk] = -1
            else:
                x[blk] = ub[blk]
                act[blk] = 1
    raise RuntimeError("qp_exact: iteration cap")


def kkt_residual(H, f, lb, ub, x):
    """max-norm of the projected gradient  x - clip(x - grad, lb, ub)  (0 at the minimiser)."""
    g = 2.0 * (H @ x) + f
    return float(np.max(np.abs(x - np.clip(x - g, lb, ub))))


def solve_lbfgsb(AB, C, x0, r, N, lb, ub, Qw=100.0, Rw=1e-4):
    """The reference's actual solver call: optimize.minimize(cost, zeros(N), bounds=...) ->
    L-BFGS-B with 2-point finite-difference gradient, cold start (duffing.py:634-636, 857-861)."""
    from scipy import optimize

    fun = lambda u: cost_function(u, r, AB, C, x0, Qw, Rw)
    res = optimize.minimize(fun, np.zeros(N), bounds=[(lb, ub)] * N)
    return res.x, res


# ----------------------------------------------------------------------------------------
# plants (adjacent to the hot path; used to build closed-loop test inputs)
# ----------------------------------------------------------------------------------------


def _rk4(f, x, u, h, matlab=False):
    """duffing.py:256-261.  matlab=True: Revise_2/Koopman_update.m:21-25 -- its k4 is evaluated at x + k1 deltaT (not k3)."""
    k1 = f(x, u)
    k2 = f(x + 0.5 * h * k1, u)
    k3 = f(x + 0.5 * h * k2, u)
    k4 = f(x + h * (k1 if matlab else k3), u)
    return x + (h / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)


def duffing_f(x, u, switched=False):
    """duffing.py:255 (nominal) and :991-992 (after step 100)."""
    if switched:
        return np.array([x[1], -10.0 * 0.5 * x[1] + 2.0 * x[0] - 0.5 * x[0] ** 3.0 + u])
    return np.array([x[1], -0.5 * x[1] + x[0] - x[0] ** 3.0 + u])


def vdp_f(x, u, switched=False):
    """vanderpol.py:250 region (fv) and :923-931 (after step 100)."""
    if switched:
        return np.array([x[1], -3.0 * x[1] - 10.0 * x[0] ** 2.0 * x[1] - 3.0 * x[0] + u])
    return np.array([2.0 * x[1], 2.0 * x[1] - 10.0 * x[0] ** 2.0 * x[1] - 0.8 * x[0] + u])


def plant_step(kind, x, u, h=0.05, switched=False):
    """One RK4 step (duffing.py:256-261) on x (2,) or (2,B) with input u scalar or (B,); kind "duffing" | "vdp", with the
    suffix "_matlab" the step as Koopman_update.m:21-25 writes it."""
    matlab = kind.endswith("_matlab")
    f = {"duffing": duffing_f, "vdp": vdp_f}[kind[:-7] if matlab else kind]
    return _rk4(lambda xx, uu: f(xx, uu, switched), np.asarray(x, float), np.asarray(u, float), h, matlab)


def tank_step(x, u, switched=False):
    """Tank_System.m:9-10 (nominal), :194-195 (after step 100), clip at 0 (:211)."""
    x = np.asarray(x, float)
    if switched:
        xn = np.array([x[0] - 0.53 * np.sqrt(x[0]) + 0.3 * u, x[1] + 0.1 * np.sqrt(x[0]) - 0.35 * np.sqrt(x[1])])
    else:
        xn = np.array([x[0] - 0.5 * np.sqrt(x[0]) + 0.4 * u, x[1] + 0.2 * np.sqrt(x[0]) - 0.3 * np.sqrt(x[1])])
    return np.maximum(xn, 0.0)


# ----------------------------------------------------------------------------------------
# the step, in the reference's order (duffing.py:847-984), for one trajectory
# ----------------------------------------------------------------------------------------


class OracleController:
    """lift x_k -> [RLS with (psi_{k-1}, u_{k-1}, psi_k, x_k)] -> condense -> exact QP -> u_k.

    Same dependency order as the reference loop: the model used for the solve at step k is the
    one updated with the transition that ended in x_k (duffing.py:856-861, 927-984); until the
    first transition exists the offline model (A0, B0, C0) is used (duffing.py:811-813).
    ``output`` = "Cx" (duffing: y = C x, C adapted by RLS) or "lift" (vanderpol.py: y = lifted state).
    """

    def __init__(self, lift, L, n, N, lb, ub, A0, B0, C0, P0=1e4, barQ0=100.0, Qw=100.0, Rw=1e-4,
                 output="Cx", solver="exact", rls="reference", update=True):
        self.lift, self.L, self.n, self.N = lift, L, n, N
        self.update = update  # False: the loop without the online update (duffing.py:738-805, vanderpol.py:645-722)
        self.lb, self.ub, self.Qw, self.Rw = lb, ub, Qw, Rw
        self.A, self.B, self.C = np.array(A0, float), np.array(B0, float).reshape(L, 1), np.array(C0, float)
        self.rls = RlsStateRef(L, 1, n, P0, barQ0)
        self.output, self.solver, self.rls_form = output, solver, rls
        # gain-form state (same estimator, evaluated without the K_A * inv_K_G product)
        self.gK, self.gP = np.zeros((L, L + 1)), P0 * np.eye(L + 1)
        self.gC, self.gQ = np.zeros((n, L)), barQ0 * np.eye(L)
        self.prev = None

    def step(self, x, r):
        psi = self.lift(np.reshape(x, (-1, 1))).reshape(-1)
        if self.prev is not None and self.update:
            ppsi, pu = self.prev
            if self.rls_form == "reference":
                self.A, self.B, self.C = rls_update_reference(self.rls, ppsi, pu, psi, x)
            else:
                self.gK, self.gP = rls_update_gain(self.gK, self.gP, np.concatenate([ppsi, [pu]]), psi)
                self.gC, self.gQ = rls_update_gain(self.gC, self.gQ, ppsi, np.reshape(x, -1))
                self.A, self.B, self.C = self.gK[:, :-1].copy(), self.gK[:, -1:].copy(), self.gC.copy()
        Co = None if self.output == "lift" else self.C
        if self.solver == "exact":
            _, _, H, f, _ = condense(self.A, self.B, Co, psi, r, self.N, self.Qw, self.Rw)
            U, _ = qp_exact(H, f, self.lb, self.ub)
        else:
            U, _ = solve_lbfgsb(np.concatenate([self.A, self.B], axis=1), Co, psi.reshape(-1, 1), r,
                                self.N, self.lb, self.ub, self.Qw, self.Rw)
        self.prev = (psi, float(U[0]))
        return float(U[0]), U, psi


class OracleDeltaUController:
    """Tank_System.m loop (:170-291), restated from the MATLAB text (no Octave here: parity unpinned).

    RLS of [A B] / C on z = [psi; u] as usual (:234-263; first C update only downdates bar_Q, :252-254);
    the MPC runs on the increment form  A~ = [A B; 0 I], B~ = [B; I], C~ = [C 0], Cy = rows
    cy0..cy0+q-1 (:110-113, 265-268), state Lift_xu = [psi(x); U0] (:290); every increment in
    [lb, ub] = +-0.5 and the first one also inside [umin - U0, umax - U0] (:182-188);
    U0 <- U0 + dU*(1) (:192).  Weights Q = 10, R = 1e-3 (:117-118).  Gain-form RLS (see rls_update_gain).
    """

    def __init__(self, lift, L, n, N, A0, B0, C0, cy0=1, q=1, lb=-0.5, ub=0.5, umin=-8.0, umax=8.0, P0=1e4, barQ0=1e4,
                 Qw=10.0, Rw=1e-3, c_skip_first=True):
        self.lift, self.L, self.n, self.N = lift, L, n, N
        self.cy0, self.q, self.lb, self.ub, self.umin, self.umax = cy0, q, lb, ub, umin, umax
        self.Qw, self.Rw, self.c_skip_first = Qw, Rw, c_skip_first
        self.A, self.B, self.C = np.array(A0, float), np.array(B0, float).reshape(L, 1), np.array(C0, float)
        self.gK, self.gP = np.zeros((L, L + 1)), P0 * np.eye(L + 1)
        self.gC, self.gQ = np.zeros((n, L)), barQ0 * np.eye(L)
        self.prev, self.u, self.nupd = None, 0.0, 0

    def qp(self, psi):
        L = self.L
        At = np.block([[self.A, self.B], [np.zeros((1, L)), np.eye(1)]])
        Bt = np.concatenate([self.B, [[1.0]]], axis=0)
        Ct = np.concatenate([self.C, np.zeros((self.n, 1))], axis=1)
        Co = Ct[self.cy0: self.cy0 + self.q]
        xt = np.concatenate([psi, [self.u]])
        return At, Bt, Co, xt

    def step(self, x, r):
        psi = self.lift(np.reshape(x, (-1, 1))).reshape(-1)
        if self.prev is not None:
            ppsi, pu = self.prev
            self.gK, self.gP = rls_update_gain(self.gK, self.gP, np.concatenate([ppsi, [pu]]), psi)
            Cn, self.gQ = rls_update_gain(self.gC, self.gQ, ppsi, np.reshape(x, -1))
            self.gC = np.zeros_like(self.gC) if (self.c_skip_first and self.nupd == 0) else Cn
            self.nupd += 1
            self.A, self.B, self.C = self.gK[:, :-1].copy(), self.gK[:, -1:].copy(), self.gC.copy()
        At, Bt, Co, xt = self.qp(psi)
        _, _, H, f, _ = condense(At, Bt, Co, xt, r, self.N, self.Qw, self.Rw)
        lbv = np.full(self.N, self.lb); ubv = np.full(self.N, self.ub)
        lbv[0] = max(self.lb, self.umin - self.u)
        ubv[0] = min(self.ub, self.umax - self.u)
        dU, _ = qp_exact(H, f, lbv, ubv)
        self.u = self.u + float(dU[0])
        self.prev = (psi, self.u)
        return self.u, dU, psi

