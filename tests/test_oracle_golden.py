"""Pins the CPU oracle (oracle/koopman_oracle.py) to values computed by the reference itself
(tests/golden/*.npz, made by tests/golden/make_golden.py from duffing.py / vanderpol.py /
vanderpol_RBF.py executed in the build container).  CPU only."""
import os

import numpy as np
import pytest

from oracle import koopman_oracle as ko

G = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return np.load(os.path.join(G, name))


@pytest.fixture(scope="module")
def duff():
    return _load("duffing_loop.npz")


@pytest.fixture(scope="module")
def vdp():
    return _load("vanderpol_loop.npz")


@pytest.fixture(scope="module")
def vrbf():
    return _load("vanderpol_rbf_loop.npz")


# ------------------------------------------------------------------ lift
@pytest.mark.parametrize("which,wfile", [("duffing_loop.npz", "weights_duffing.npz"), ("vanderpol_loop.npz", "weights_vdp.npz")])
def test_mlp_lift_matches_reference_encoder(which, wfile):
    g = _load(which)
    w = ko.load_mlp_weights(_load(wfile))
    psi = ko.mlp_lift(w, g["lift_X"].T).T
    assert np.abs(psi - g["lift_Psi"]).max() < 1e-13


def test_known_answer_lifts_from_survey():
    # psi(0,0) and psi(-2,-2) of the shipped duffing / vdp encoders
    w = ko.load_mlp_weights(_load("weights_duffing.npz"))
    p0 = ko.mlp_lift(w, np.array([[0.0], [0.0]])).ravel()
    p2 = ko.mlp_lift(w, np.array([[-2.0], [-2.0]])).ravel()
    assert np.allclose(p0, [-0.13070173, 0.01165256, 0.15532107, 0.30123124, 0.21592677, 0.29037692, 0.14559477, -0.31636275], atol=1e-8)
    assert np.allclose(p2, [-2.75444189, 1.12538874, -2.53037454, 0.81205281, 0.92008022, -1.16281013, 0.26535957, -1.45176999], atol=1e-8)
    w = ko.load_mlp_weights(_load("weights_vdp.npz"))
    p0 = ko.mlp_lift(w, np.array([[0.0], [0.0]])).ravel()
    assert np.allclose(p0, [-0.22947356, -0.39910619, 0.25949734, -0.29341398, -0.0271047, 0.21488658, -0.15604981, -0.1823129], atol=1e-8)


def test_tank_encoder_has_two_hidden_layers():
    w = ko.load_mlp_weights(_load("weights_tank.npz"))
    assert len(w) == 3 and w[-1][0].shape == (10, 100)
    assert ko.mlp_lift(w, np.zeros((2, 5))).shape == (10, 5)


def test_rbf_lift_matches_reference_rbf(vrbf):
    psi = ko.rbf_lift(vrbf["lift_X"].T, vrbf["cx"]).T
    assert np.abs(psi - vrbf["lift_Psi"]).max() < 1e-12
    # the loop's own lifted states
    psi = ko.rbf_lift(vrbf["logXloc"][:, :-1], vrbf["cx"])
    assert np.abs(psi - vrbf["logXLOClift"][:, 1:]).max() < 1e-12


def test_rbf_matlab_form_nan_to_zero():
    cx = np.array([[0.0, 0.0], [1.0, 0.0]])
    y = ko.rbf_lift(np.array([[0.0], [0.0]]), cx, form="matlab")
    assert y[0, 0] == 0.0 and abs(y[1, 0]) < 1e-15  # r2=1 -> 1*log(1) = 0


# ------------------------------------------------------------------ RLS
@pytest.mark.parametrize("name,P0,Q0", [("duffing_loop.npz", 1e4, 100.0), ("vanderpol_loop.npz", 1e5, 1e5)])
def test_rls_replay_is_bit_exact_with_reference(name, P0, Q0):
    g = _load(name)
    st = ko.RlsStateRef(8, 1, 2, P0, Q0)
    for k in range(len(g["loop_i"])):
        ko.rls_update_reference(st, g["loop_xlift"][k], g["loop_u_loc"][k], g["loop_ylift"][k], g["loop_x_loc"][k])
        assert np.array_equal(st.K, g["loop_K_ext"][k])
        assert np.array_equal(st.P, g["loop_inv_K_G"][k])
        assert np.array_equal(st.K_A, g["loop_K_A"][k])
        assert np.array_equal(st.C, g["loop_C_prev"][k])
        assert np.array_equal(st.bar_Q, g["loop_bar_Q"][k])
        assert np.array_equal(st.bar_X, g["loop_bar_X"][k])


def test_gain_form_equals_reference_form_up_to_its_own_rounding_floor(duff):
    """K += (y-Kz)g' is algebraically K_A inv_K_G.  The reference form multiplies the accumulated
    K_A by a P that carries 1e4-scale cancellation, so ANY re-association of it moves K by ~5e-8
    relative (measured: reversing the summation order of K_A @ P alone gives 5.0e-8); the gain
    form sits inside that floor."""
    g = duff
    st = ko.RlsStateRef(8, 1, 2)
    K, P = np.zeros((8, 9)), 1e4 * np.eye(9)
    worst = 0.0
    for k in range(130):
        ko.rls_update_reference(st, g["loop_xlift"][k], g["loop_u_loc"][k], g["loop_ylift"][k], g["loop_x_loc"][k])
        z = np.concatenate([g["loop_xlift"][k].ravel(), g["loop_u_loc"][k].ravel()])
        K, P = ko.rls_update_gain(K, P, z, g["loop_ylift"][k].ravel())
        worst = max(worst, np.abs(K - st.K).max() / np.abs(st.K).max())
        assert np.abs(P - st.P).max() <= 1e-9 * np.abs(st.P).max()
    assert worst < 2e-7


@pytest.mark.parametrize("name,P0,lifted,bnd", [("duffing_loop.npz", 1e4, False, 2.0), ("vanderpol_loop.npz", 1e5, True, 6.0)])
def test_reference_form_reassociation_floor(name, P0, lifted, bnd):
    """How reproducible is the reference's OWN arithmetic?  Evaluate the same K_A / inv_K_G recursion with
    the rank-one term associated as (Pz)(Pz)'/d instead of (P z z' P)/d (duffing.py:931-932; identical in
    exact arithmetic): [A B] moves by ~5e-8 (inv_K_G0 = 1e4 I) / ~4e-5 (1e5 I) relative and the exact QP
    minimiser by ~1e-6 / ~1e-3 (printed).  No implementation other than a bit-identical one can be closer to the
    reference than this; the HIP path (gain form) is checked against these floors."""
    g = _load(name)
    st = ko.RlsStateRef(8, 1, 2, P0, P0)
    KA, P = np.zeros((8, 9)), P0 * np.eye(9)
    dK, dU = 0.0, 0.0
    for k in range(len(g["loop_i"]) - 1):
        A, B, Cm = ko.rls_update_reference(st, g["loop_xlift"][k], g["loop_u_loc"][k], g["loop_ylift"][k], g["loop_x_loc"][k])
        z = np.concatenate([g["loop_xlift"][k].ravel(), g["loop_u_loc"][k].ravel()])
        KA = KA + np.outer(g["loop_ylift"][k].ravel(), z)
        Pz = P @ z
        P = P - np.outer(Pz, Pz) / (1.0 + z @ Pz)
        K2 = KA @ P
        dK = max(dK, np.abs(K2 - st.K).max() / np.abs(st.K).max())
        psi, r = g["loop_ylift"][k], g["loop_r"][k + 1]
        Co = None if lifted else g["loop_C_prev"][k]
        _, _, H, f, _ = ko.condense(A, B, Co, psi, r, 10)
        _, _, H2, f2, _ = ko.condense(K2[:, :8], K2[:, 8:], Co, psi, r, 10)
        dU = max(dU, np.abs(ko.qp_exact(H, f, -bnd, bnd)[0] - ko.qp_exact(H2, f2, -bnd, bnd)[0]).max())
    print("%s: re-association moves K by %.1e (rel), U* by %.1e" % (name, dK, dU))
    assert 1e-9 < dK < 1e-9 * P0   # measured 5.0e-8 (1e4) / 4.2e-5 (1e5)
    assert dU < 1e-7 * P0


def test_shared_edmd_is_the_reference_rls_for_one_trajectory(duff):
    """Gram-form shared model (Koopman_update.m:94-101 with the RLS initialisations as ridge) pooled over ONE
    trajectory reproduces the reference's recursive K_ext / C_prev (inside the K_A*inv_K_G floor)."""
    g = duff
    sh = ko.SharedEdmd(8, 2)
    for k in range(130):
        sh.add(*ko.SharedEdmd.gram(g["loop_xlift"][k], g["loop_u_loc"][k].ravel(), g["loop_ylift"][k], g["loop_x_loc"][k]))
        A, B, C = sh.model()
        K = np.concatenate([A, B], axis=1)
        assert np.abs(K - g["loop_K_ext"][k]).max() <= 2e-7 * np.abs(K).max()
        assert np.abs(C - g["loop_C_prev"][k]).max() <= 1e-9 * max(1e-3, np.abs(C).max())


# ------------------------------------------------------------------ cost / condensed QP
def _model(g, k):
    return g["loop_Ap"][k], g["loop_Bp"][k], g["loop_Cp"][k], g["loop_xlift"][k], g["loop_r"][k]


def test_cost_function_matches_reference_costFunction(duff, vdp, vrbf):
    for g, lifted in ((duff, False), (vdp, True), (vrbf, False)):
        for row, J in zip(g["cost_in"], g["cost_out"]):
            k = int(row[0])
            A, B, Cm, psi, r = _model(g, k)
            Jm = ko.cost_function(row[1:], r, np.concatenate([A, B], axis=1), None if lifted else Cm, psi)
            assert abs(Jm - J) <= 1e-12 * abs(J)


def test_condensed_qp_is_the_reference_cost(duff, vdp):
    for g, lifted in ((duff, False), (vdp, True)):
        for row, J in zip(g["cost_in"], g["cost_out"]):
            k = int(row[0])
            A, B, Cm, psi, r = _model(g, k)
            _, _, H, f, c = ko.condense(A, B, None if lifted else Cm, psi, r, 10)
            u = row[1:]
            assert abs(u @ H @ u + f @ u + c - J) <= 1e-11 * abs(J)
            assert np.array_equal(H, H.T)


def test_condense_structure_matches_matlab_spec():
    rng = np.random.RandomState(3)
    L, N, q = 6, 5, 2
    A, B, Cm = rng.randn(L, L) * 0.4, rng.randn(L, 1), rng.randn(q, L)
    Gam, Phi, H, f, _ = ko.condense(A, B, Cm, rng.randn(L), rng.randn(q, N), N, Qw=10.0, Rw=0.01)
    # Compact_Form1 rows: Cy*C*A^k ; Compact_Form2: lower block Toeplitz of C A^(i-j) B  (Koopman_update.m:455-467)
    for k in range(1, N + 1):
        assert np.allclose(Gam[q * (k - 1): q * k], Cm @ np.linalg.matrix_power(A, k))
    for i in range(N):
        for j in range(N):
            blk = Phi[q * i: q * (i + 1), j]
            want = (Cm @ np.linalg.matrix_power(A, i - j) @ B).ravel() if j <= i else np.zeros(q)
            assert np.allclose(blk, want)
    assert np.allclose(H, Phi.T @ (10.0 * np.eye(q * N)) @ Phi + 0.01 * np.eye(N))


# ------------------------------------------------------------------ QP
def test_qp_exact_beats_or_ties_reference_solver(duff, vdp):
    from scipy.optimize import lsq_linear

    for g, lifted, bnd in ((duff, False, 2.0), (vdp, True, 6.0)):
        assert np.allclose(g["bounds"], [[-bnd, bnd]] * 10)
        worst_u0 = 0.0
        for k in range(0, len(g["loop_i"]), 3):
            A, B, Cm, psi, r = _model(g, k)
            _, _, H, f, c = ko.condense(A, B, None if lifted else Cm, psi, r, 10)
            U, _ = ko.qp_exact(H, f, -bnd, bnd)
            assert ko.kkt_residual(H, f, -bnd, bnd, U) <= 1e-9 * max(1.0, np.abs(f).max())
            Jx = U @ H @ U + f @ U + c
            # never worse than what the reference's L-BFGS-B found; its answer is only approximate
            assert Jx <= g["loop_J"][k] * (1 + 1e-12) + 1e-12
            worst_u0 = max(worst_u0, abs(U[0] - g["loop_Useq"][k][0]))
            Lc = np.linalg.cholesky(H).T
            bv = lsq_linear(Lc, -np.linalg.solve(Lc.T, f / 2), bounds=(-bnd, bnd), method="bvls", tol=1e-14).x
            assert np.abs(bv - U).max() < 1e-8
        assert worst_u0 < 5e-3


def test_qp_exact_edge_cases():
    H = np.array([[2.0, 0.5], [0.5, 1.0]])
    for f, want in ((np.array([-100.0, -100.0]), [2, 2]), (np.array([100.0, 100.0]), [-2, -2]), (np.zeros(2), [0, 0])):
        U, _ = ko.qp_exact(H, f, -2, 2)
        assert np.allclose(U, want)
    U, _ = ko.qp_exact(np.array([[3.0]]), np.array([-6.0]), -2, 2)  # N = 1, interior: u = 1
    assert np.allclose(U, [1.0])


def test_solve_lbfgsb_reproduces_the_reference_call(duff):
    for k in (0, 5, 60):
        A, B, Cm, psi, r = _model(duff, k)
        U, res = ko.solve_lbfgsb(np.concatenate([A, B], axis=1), Cm, psi, r, 10, -2, 2)
        # finite-difference L-BFGS-B amplifies last-bit differences of the cost evaluation into
        # ~3e-4 in U (flat valley); the attained cost agrees to 1e-7
        assert np.abs(U - duff["loop_Useq"][k]).max() < 2e-3
        assert abs(res.fun - duff["loop_J"][k]) <= 1e-7 * abs(duff["loop_J"][k])


# ------------------------------------------------------------------ plant + closed loop
def test_plants_reproduce_reference_trajectories(duff, vdp):
    for g, kind in ((duff, "duffing"), (vdp, "vdp")):
        x = np.array([-2.0, -2.0])
        for k in range(len(g["loop_i"])):
            x = ko.plant_step(kind, x, float(g["logUloc"][0, k]), float(g["h"]), switched=(k > 101))
            assert np.abs(x - g["logXloc"][:, k]).max() < 1e-12, k


def test_closed_loop_with_reference_style_solver_tracks_reference_log(duff):
    """Oracle controller on the reference's states: u within the reference solver's own error
    (|u0 - u0*| up to ~1e-3) of the logged u, for both solver variants."""
    w = ko.load_mlp_weights(_load("weights_duffing.npz"))
    lift = lambda x: ko.mlp_lift(w, x)
    r = duff["loop_r"][0]
    for solver, tol in (("lbfgsb", 5e-3), ("exact", 5e-3)):
        ctl = ko.OracleController(lift, 8, 2, 10, -2.0, 2.0, duff["A0"], duff["B0"], duff["C0"], solver=solver)
        x = np.array([-2.0, -2.0])
        for k in range(40):
            u, U, psi = ctl.step(x, r)
            assert abs(u - duff["logUloc"][0, k]) < tol, (solver, k)
            ctl.prev = (psi, float(duff["logUloc"][0, k]))  # follow the logged trajectory
            x = duff["logXloc"][:, k]


def test_nn_encoder_mat_prefix_is_a_vdp_trajectory():
    """NN_Encoder.mat (vanderpol.py:1112): X_Collection_NO = logX and U_Collection = logU are the loop WITHOUT update,
    X_Collection = logXloc the loop with it.  logX[:, k] is the state after input logU[k]: x_k = RK4(x_{k-1}, u_k) with
    x_{-1} = (-2, -2) -- exact for the oracle's plant while the nominal parameters are active."""
    d = _load("vdp_nn_encoder_first200.npz")
    XN, U = d["X_Collection_NO"], d["U_Collection"]
    x = np.array([-2.0, -2.0])
    for k in range(100):
        xn = ko.plant_step("vdp", x, float(U[0, k]), 0.05)
        assert np.abs(xn - XN[:, k]).max() < 1e-12, k
        x = XN[:, k]
    assert np.all(np.abs(U) <= 6.0 + 1e-12)


def test_free_running_oracle_reproduces_nn_encoder_mat():
    """The with-update loop of vanderpol.py restated (exact QP, free-running from (-2, -2) with the reference's offline model)
    lands within 1e-4 of X_Collection[:, :120] of the shipped NN_Encoder.mat (3.5e-5: the reference's L-BFGS-B leaves ~1e-3
    in u, which moves x by that much) -- the same bar the GPU script is held to in tests/test_gpu_round2.py."""
    d = _load("vdp_nn_encoder_first200.npz")
    g = _load("vanderpol_loop.npz")
    w = ko.load_mlp_weights(_load("weights_vdp.npz"))
    lift = lambda x: ko.mlp_lift(w, x)
    r = np.tile(lift(np.array([[1.0], [0.0]])).reshape(8, 1), (1, 10))
    ctl = ko.OracleController(lift, 8, 2, 10, -6.0, 6.0, g["A0"], g["B0"], g["C0"], P0=1e5, barQ0=1e5, output="lift", solver="exact")
    x = np.array([-2.0, -2.0])
    worst = 0.0
    for k in range(120):
        u, _, _ = ctl.step(x, r)
        x = ko.plant_step("vdp", x, u, 0.05, switched=(k >= 102))
        worst = max(worst, np.abs(x - d["X_Collection"][:, k]).max())
    assert worst < 1e-4, worst


def test_storage_update_is_the_recursion_from_the_offline_gram():
    """vanderpol_RBF.py:434-438 ("storage" update: refit from all stored samples) against the recursive estimator continued
    from the Gram sums the script held before its first iteration: the reference's own logged K_ext / C_prev, 40 steps."""
    g = _load("vanderpol_rbf_loop.npz")
    L = 8
    z = lambda k: np.concatenate([g["loop_xlift"][k, :, 0], g["loop_u_loc"][k, :, 0]])
    st = ko.RlsStateRef(L, 1, 2)
    st.P = np.linalg.inv(g["loop_stor_KG"][0] - np.outer(z(0), z(0)))
    st.K_A = g["loop_K_A"][0] - np.outer(g["loop_ylift"][0, :, 0], z(0))
    st.bar_Q = np.linalg.inv(g["loop_stor_GX"][0] - np.outer(z(0)[:L], z(0)[:L]))
    st.bar_X = g["loop_stor_XXE"][0] - np.outer(g["loop_x_loc"][0, :, 0], z(0)[:L])
    for k in range(g["loop_i"].shape[0]):
        A, B, C = ko.rls_update_reference(st, g["loop_xlift"][k], g["loop_u_loc"][k], g["loop_ylift"][k], g["loop_x_loc"][k])
        K = np.concatenate([A, B], axis=1)
        assert np.abs(K - g["loop_K_ext"][k]).max() < 1e-8 * np.abs(g["loop_K_ext"][k]).max(), k
        assert np.abs(C - g["loop_C_prev"][k]).max() < 1e-8 * np.abs(g["loop_C_prev"][k]).max(), k


@pytest.mark.parametrize("lam", [1.0, 0.98, 0.9])
def test_gain_form_equals_reference_form_with_forgetting(lam):
    """rls_update_gain is rls_update_reference for every lambda (Koopman_update.m:270-274 discounts inv_K_G only): 30 steps,
    [A B] within the K_A inv_K_G product's own rounding."""
    rng = np.random.RandomState(0)
    L, n = 8, 2
    st = ko.RlsStateRef(L, 1, n)
    K, P = np.zeros((L, L + 1)), 1e4 * np.eye(L + 1)
    for k in range(30):
        psi, psin, u, xn = rng.randn(L), rng.randn(L), rng.randn(), rng.randn(n)
        A, B, _ = ko.rls_update_reference(st, psi, u, psin, xn, lam=lam)
        K, P = ko.rls_update_gain(K, P, np.concatenate([psi, [u]]), psin, lam=lam)
        assert np.abs(K - np.concatenate([A, B], axis=1)).max() < 1e-5 * max(1.0, np.abs(K).max()), k


def test_dare_and_dlqr_are_the_reference_functions():
    """solve_DARE / dlqr (duffing.py:583-613) evaluated by the reference's own functions on its own call (offline
    model, Q = 10 I, R = 0.01, duffing.py:667-671), on online-updated models of the golden loop and on seeded random
    models (tests/golden/make_golden_dare.py): the restatement reproduces P and K bit for bit, including the two
    cases the reference leaves unconverged after its 500 iterations."""
    g = _load("dare.npz")
    for k in range(int(g["n"])):
        A, B, Q, R = g["A%d" % k], g["B%d" % k], g["Q%d" % k], float(g["R%d" % k])
        P, it = ko.solve_dare(A, B, Q, R)
        assert 1 <= it <= 500
        assert np.array_equal(P, g["P%d" % k]), k
        assert np.array_equal(ko.dlqr(A, B, Q, R), g["K%d" % k]), k
    # the terminal block is what the MATLAB controller writes into Q_bar (Koopman_update.m:381)
    Co = np.array([[1.0, 2.0, 0.0], [0.0, 1.0, -1.0]])
    P = np.diag([1.0, 2.0, 3.0])
    assert np.allclose(ko.terminal_block(Co, P), Co @ P @ Co.T)


# ------------------------------------------------------------------ round 3: reference variants
def test_fixed_model_loop_of_the_oracle(duff):
    """duffing.py:738-805: the loop without the online update (logX, logU).  The oracle with update=False, exact QP, free-running
    from (-2, -2) with the reference's offline model lands on the reference's own log (its L-BFGS-B leaves ~1e-3 in u)."""
    w = ko.load_mlp_weights(_load("weights_duffing.npz"))
    ctl = ko.OracleController(lambda x: ko.mlp_lift(w, x), 8, 2, 10, -2.0, 2.0, duff["A0"], duff["B0"], duff["C0"], update=False)
    r = duff["loop_r"][0]
    x = np.array([-2.0, -2.0])
    ex = eu = 0.0
    for k in range(130):
        u, _, _ = ctl.step(x, r)
        x = ko.plant_step("duffing", x, u, switched=(k > 101))
        ex = max(ex, np.abs(x - duff["logX"][:, k]).max())
        eu = max(eu, abs(u - duff["logU"][0, k]))
    assert np.array_equal(ctl.A, duff["A0"])  # the model never moved
    assert ex < 1e-4 and eu < 3e-3, (ex, eu)


def test_matlab_lift_forms_and_rk4():
    """Koopman_update.m:67 / Koopman_update_Tracking_Lift.m:65: both lifts vanish at the origin, the long form starts with x;
    Koopman_update.m:21-25: the MATLAB Runge-Kutta step differs from the Python scripts' at O(h^4) only."""
    w = ko.load_mlp_weights(_load("weights_duffing.npz"))
    X = np.array([[0.0, 0.3, -1.2], [0.0, -0.7, 0.4]])
    a = ko.mlp_lift_offset(w, X, "psi0")
    b = ko.mlp_lift_offset(w, X, "x_psi0")
    assert a.shape == (8, 3) and b.shape == (10, 3)
    assert np.abs(a[:, 0]).max() < 1e-15 and np.abs(b[:, 0]).max() < 1e-15  # (BLAS sums a 3-column product in another order)
    assert np.array_equal(b[:2], X) and np.array_equal(b[2:], a)
    assert np.allclose(a, ko.mlp_lift(w, X) - ko.mlp_lift(w, np.zeros((2, 1))), rtol=0, atol=1e-15)
    x = np.array([0.4, -1.1])
    xp, xm = ko.plant_step("duffing", x, 0.7), ko.plant_step("duffing_matlab", x, 0.7)
    assert 1e-9 < np.abs(xp - xm).max() < 1e-4


def test_delta_u_controller_against_a_literal_transcription_of_tank_system_m():
    """cfg4's checker, OracleDeltaUController, restates Tank_System.m from its text (no MATLAB / Octave here and the reference holds no
    fixture for it: VERDICT r4 weak 1).  A second, independent transcription pins it: (1) the condensed QP built the way the .m file
    writes it -- A~ = [A B; 0 I], B~ = [B; I], C~ = C [I 0], Cy = [0 1] (:110-113, 265-268), Compact_Form1 rows Cy C~ A~^k by matrix
    POWERS, Compact_Form2 by the Shift_Matrix construction (:272-288), H = CF2' Q_bar CF2 + R_bar, f = 2 (CF1 Lift_xu)' Q_bar CF2 -
    2 Yr' Q_bar CF2 (:186, 287-289) -- equals the oracle's recursion-built H, f to rounding; (2) the cost the increments produce when
    the ORIGINAL model x+ = A x + B u is simulated with absolute inputs u_k = U0 + sum dU (no augmentation at all) equals
    dU' H dU + f' dU + const; (3) quadprog's constraint set -- A_cons / b_cons on the first increment (:182-185) inside the box
    +-0.5 (:188) -- is the box the oracle folds it into, and the oracle's minimiser passes the KKT test of that constraint set."""
    rng = np.random.RandomState(4)
    L, n, N, Qw, Rw = 6, 2, 9, 10.0, 1e-3
    A = 0.9 * np.linalg.qr(rng.randn(L, L))[0] + 0.05 * rng.randn(L, L)
    B = rng.randn(L, 1)
    C = rng.randn(n, L)
    psi = rng.randn(L)
    U0, umin, umax = 7.8, -8.0, 8.0
    Yr = np.ones((N, 1))
    ctl = ko.OracleDeltaUController(lambda x: x, L, n, N, A, B, C, cy0=1, q=1, Qw=Qw, Rw=Rw)
    ctl.u = U0
    At, Bt, Co, xt = ctl.qp(psi)
    _, _, H, f, const = ko.condense(At, Bt, Co, xt, np.ones((1, N)), N, Qw, Rw)
    # ---- (1) the .m file's own construction
    nu = 1
    Am = np.block([[A, B], [np.zeros((nu, L)), np.eye(nu)]])
    Bm = np.concatenate([B, np.eye(nu)], axis=0)
    Cm = C @ np.concatenate([np.eye(L), np.zeros((L, nu))], axis=1)
    Cy = np.array([[0.0, 1.0]])
    Shift = np.kron(np.concatenate([np.zeros((1, N)), np.concatenate([np.eye(N - 1), np.zeros((N - 1, 1))], axis=1)], axis=0), np.eye(nu))
    CF1 = np.concatenate([Cy @ Cm @ np.linalg.matrix_power(Am, k) for k in range(1, N + 1)], axis=0)
    CF2 = np.zeros((0, N))
    for k in range(1, N + 1):
        vt = np.zeros((1, 0))
        for j in range(1, N + 1):
            vt = np.concatenate([Cy @ Cm @ np.linalg.matrix_power(Am, j - 1) @ Bm, vt], axis=1)
        CF2 = np.concatenate([vt @ np.linalg.matrix_power(Shift, k - 1), CF2], axis=0)
    Qbar, Rbar = np.kron(np.eye(N), Qw * np.eye(1)), np.kron(np.eye(N), Rw * np.eye(1))
    Hm = CF2.T @ Qbar @ CF2 + Rbar
    Hm = (Hm + Hm.T) / 2
    Lift_xu = np.concatenate([psi, [U0]]).reshape(-1, 1)
    fm = (2 * (CF1 @ Lift_xu).T @ Qbar @ CF2 - 2 * Yr.T @ Qbar @ CF2).ravel()
    assert np.abs(H - Hm).max() <= 1e-11 * np.abs(Hm).max() and np.abs(f - fm).max() <= 1e-11 * np.abs(fm).max()
    # ---- (2) the original model simulated with absolute inputs
    for _ in range(5):
        dU = rng.uniform(-0.5, 0.5, N)
        x, u, J = psi.copy(), U0, 0.0
        for k in range(N):
            u = u + dU[k]
            x = A @ x + B[:, 0] * u
            J += Qw * float((C @ x)[1] - 1.0) ** 2 + Rw * dU[k] ** 2
        assert abs(J - (dU @ H @ dU + f @ dU + const)) <= 1e-10 * max(1.0, abs(J))
    # ---- (3) the constraint set: first increment inside [umin - U0, umax - U0] and every increment inside +-0.5
    u_new, dUo, _ = ctl.step(psi, np.ones((1, N)))  # (lift = identity: x is psi here)
    lbv = np.full(N, -0.5); ubv = np.full(N, 0.5)
    ubv[0] = min(0.5, umax - U0); lbv[0] = max(-0.5, umin - U0)
    assert ubv[0] == pytest.approx(0.2) and abs(u_new - (U0 + dUo[0])) < 1e-15 and u_new <= umax + 1e-12
    g = 2 * Hm @ dUo + fm
    for i in range(N):  # KKT of min dU' H dU + f' dU over that set
        if dUo[i] <= lbv[i] + 1e-9:
            assert g[i] >= -1e-7 * max(1.0, np.abs(fm).max())
        elif dUo[i] >= ubv[i] - 1e-9:
            assert g[i] <= 1e-7 * max(1.0, np.abs(fm).max())
        else:
            assert abs(g[i]) <= 1e-7 * max(1.0, np.abs(fm).max())
