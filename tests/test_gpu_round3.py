"""Round-3 parity tests of the HIP path (through the C ABI): the reference variants the round-2 review listed.

  the lifts of the MATLAB scripts, psi(x) - psi(0) and [x; psi(x)] - [0; psi(0)]      Koopman_update_Tracking_Lift.m:65, Koopman_update.m:67
  the loop WITHOUT the online update against the reference's own logs                 duffing.py:738-805, vanderpol.py:645-722
  the Runge-Kutta step as the MATLAB scripts write it (k4 at x + h k1)                 Koopman_update.m:21-25
  the stateless solve keeps the handle's terminal block; the wave image of the state survives every entry point

Runs on the MI355X box:  python -m pytest tests -m gpu
"""
import os
import sys

import numpy as np
import pytest

from oracle import koopman_oracle as ko

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def _load(name):
    return np.load(os.path.join(G, name))


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device; there is no CPU fallback")
    return torch


@pytest.fixture(scope="module")
def KM(torch_mod):
    from koopmpc import KoopmanMPC

    return KoopmanMPC


def _t(torch, a, dtype=None):
    return torch.tensor(np.asarray(a), dtype=dtype or torch.float64, device="cuda:0")


def _rand_model(rng, L, q, rho=0.95):
    A = rng.randn(L, L)
    A *= rho / np.abs(np.linalg.eigvals(A)).max()
    return A, rng.randn(L, 1) * 0.1, rng.randn(q, L) * 0.5


# ------------------------------------------------------------------ lift offsets of the MATLAB scripts
@pytest.mark.parametrize("form,wfile", [("psi0", "weights_vdp.npz"), ("x_psi0", "weights_duffing.npz")])
def test_lift_offsets_of_the_matlab_scripts(torch_mod, KM, form, wfile):
    """liftFun = Enc(x) - Enc(0) (Koopman_update_Tracking_Lift.m:65) and [x; Enc(x)] - [0; Enc(0)] (Koopman_update.m:67,
    Nlift = n + 8) with the reference's own encoder weights: the lift of 300 states within 1e-12 (relative to the largest
    observable) of the oracle's restatement, psi(0) = 0 to rounding, and for the second form the first n observables ARE x,
    bit for bit.  (MATLAB-only forms: pinned to the text of the .m files, no MATLAB output exists.)"""
    torch = torch_mod
    w = ko.load_mlp_weights(_load(wfile))
    Lenc = w[-1][0].shape[0]
    L = Lenc + (2 if form == "x_psi0" else 0)
    B = 300
    rng = np.random.RandomState(5)
    X = 4 * rng.rand(2, B) - 2
    X[:, 0] = 0.0
    m = KM(n=2, L=L, N=10, batch=B, weights=w, lift_offset=form)
    Psi = m.Encoder(_t(torch, X)).cpu().numpy()
    ref = ko.mlp_lift_offset(w, X, form)
    scale = np.abs(ref).max()
    assert Psi.shape == ref.shape == (L, B)
    assert np.abs(Psi - ref).max() <= 1e-12 * scale
    assert np.abs(Psi[:, 0]).max() <= 1e-13 * scale  # psi(0) = 0
    if form == "x_psi0":
        assert np.array_equal(Psi[:2], X)
    # the fused roll-out lifts inside the kernel with the same effective encoder: one closed-loop step from the same state agrees
    A, Bm, Cm = _rand_model(rng, L, 2)
    m.set_model(A, Bm, Cm)
    r = np.tile(np.array([[1.0], [0.0]]), (1, 10))
    u = m.step(_t(torch, X), r).cpu().numpy()
    ctl = ko.OracleController(lambda x: ko.mlp_lift_offset(w, x, form), L, 2, 10, -2.0, 2.0, A, Bm, Cm)
    for b in range(0, B, 37):
        uo, _, _ = ctl.step(X[:, b], r)
        ctl.prev = None
        assert abs(u[b] - uo) < 1e-7, b
    assert int(m.status.max().item()) == 0


def test_lift_offset_closed_loop_with_update(torch_mod, KM):
    """Five closed-loop steps with the [x; psi(x)] - [0; psi(0)] lift (L = 10) and the online update, plant on the device:
    inputs within 1e-6 of per-trajectory oracle controllers that lift the same way."""
    torch = torch_mod
    w = ko.load_mlp_weights(_load("weights_duffing.npz"))
    L, N, B = 10, 10, 7
    rng = np.random.RandomState(11)
    A, Bm, Cm = _rand_model(rng, L, 2)
    m = KM(n=2, L=L, N=N, batch=B, weights=w, lift_offset="x_psi0")
    m.set_model(A, Bm, Cm)
    lift = lambda x: ko.mlp_lift_offset(w, x, "x_psi0")
    ctl = [ko.OracleController(lift, L, 2, N, -2.0, 2.0, A, Bm, Cm, rls="gain") for _ in range(B)]
    X0 = 4 * rng.rand(2, B) - 2
    X = _t(torch, X0)
    xs = X0.copy()
    r = np.tile(np.array([[1.0], [0.0]]), (1, N))
    for k in range(5):
        u = m.step(X, r).clone()
        for b in range(B):
            uo, _, _ = ctl[b].step(xs[:, b], r)
            assert abs(float(u[b]) - uo) < 1e-6, (k, b)
            xs[:, b] = ko.plant_step("duffing", xs[:, b], uo)
        X = m.plant_step("duffing", X, u)
    assert int(m.status.max().item()) == 0


# ------------------------------------------------------------------ the loop without the online update
def _run_script(name, args, monkeypatch, tmp_path):
    import importlib

    out = str(tmp_path / ("%s.npz" % name))
    mod = importlib.import_module("koopmpc.scripts.%s" % name)
    monkeypatch.setattr(sys, "argv", ["x"] + args + ["--out", out])
    mod.main()
    return np.load(out)


def test_fixed_model_loop_duffing_against_reference_log(torch_mod, monkeypatch, tmp_path):
    """duffing.py:738-805, the comparison loop WITHOUT the online update (logX, logU of the executed reference, 130 steps across
    the parameter switch): scripts/duffing.py --no-update, B = 1, free-running from (-2, -2) with the reference's offline
    model.  Measured: x 2.6e-5, u 6.6e-4 (the reference's L-BFGS-B leaves ~1e-3 in u); asserted at x 1e-4, u 3e-3, the
    tolerances of the with-update loop."""
    g = _load("duffing_loop.npz")
    res = _run_script("duffing", ["--weights", os.path.join(G, "weights_duffing.npz"), "--model", os.path.join(G, "duffing_loop.npz"),
                                  "--steps", "130", "--no-update"], monkeypatch, tmp_path)
    X = res["logXloc"][:, :, 0].T
    U = res["logUloc"][:, 0]
    e_x = np.abs(X - g["logX"][:, :130]).max()
    e_u = np.abs(U - g["logU"][0, :130]).max()
    print("fixed-model loop (duffing) vs the executed reference: x %.2e u %.2e" % (e_x, e_u))
    assert e_x < 1e-4 and e_u < 3e-3
    # and it IS the loop without update: the same run with the update differs from logX by much more than from logXloc
    assert np.abs(X - g["logXloc"][:, :130]).max() > 10 * e_x


def test_fixed_model_loop_vanderpol_against_nn_encoder_mat(torch_mod, monkeypatch, tmp_path):
    """vanderpol.py:645-722: X_Collection_NO / U_Collection of the shipped NN_Encoder.mat are the loop without the update.
    scripts/vanderpol.py --no-update, B = 1, 100 steps (nominal plant parameters) against the file and against the executed
    reference's logX / logU (measured 1.5e-5 / 5.8e-4; asserted at 1e-4 / 3e-3).  The shipped file itself is 7.1e-2 away from
    the reference's own re-run in this loop (its L-BFGS-B residue is not corrected by any update; SURVEY 8c: "diverges at the
    1e-2 level without-update"), so against the file only that same distance can be asserted."""
    d = _load("vdp_nn_encoder_first200.npz")
    g = _load("vanderpol_loop.npz")
    res = _run_script("vanderpol", ["--weights", os.path.join(G, "weights_vdp.npz"), "--model", os.path.join(G, "vanderpol_loop.npz"),
                                    "--steps", "100", "--no-update"], monkeypatch, tmp_path)
    X = res["logXloc"][:, :, 0].T
    U = res["logUloc"][:, 0]
    e_file = np.abs(X - d["X_Collection_NO"][:, :100]).max()
    e_run = np.abs(X - g["logX"][:, :100]).max()
    e_u = np.abs(U - g["logU"][0, :100]).max()
    print("fixed-model loop (vanderpol) vs NN_Encoder.mat %.2e, vs the executed reference %.2e (u %.2e)" % (e_file, e_run, e_u))
    ref_gap = np.abs(g["logX"][:, :100] - d["X_Collection_NO"][:, :100]).max()  # the reference against its own file
    assert e_run < 1e-4 and e_u < 3e-3 and e_file < ref_gap + 1e-3


def test_online_update_switch(torch_mod, KM):
    """kmpc_set_online_update(0): the model does not move, through kmpc_step and through the fused roll-out; switched back on
    it continues with the next transition exactly like a controller that never stopped would from the same state."""
    torch = torch_mod
    from koopmpc.synth import random_mlp_weights

    rng = np.random.RandomState(2)
    L = N = 20
    B = 33
    w = random_mlp_weights(2, 100, 3, L, seed=3)
    A, Bm, Cm = _rand_model(rng, L, 2)
    r = np.tile(np.array([[1.0], [0.0]]), (1, N))
    m = KM(n=2, L=L, N=N, batch=B, weights=w)
    m.set_model(A, Bm, Cm)
    m.set_online_update(False)
    X = _t(torch, 4 * rng.rand(2, B) - 2)
    m.rollout("duffing", X, r, 6)
    u = m.step(X, r).clone()
    A1, B1, C1 = m.get_model()
    assert np.array_equal(A1[0].cpu().numpy(), A) and np.array_equal(C1[3].cpu().numpy(), Cm)
    # the inputs are those of the fixed model: the stateless solve with (A, B, C) at the same lift gives the same u
    psi = m.Encoder(X)
    U2, u2, _, st = m.mpc_solve(A, Bm, Cm, psi, r)
    assert float((u - u2).abs().max()) < 1e-9
    m.set_online_update(True)
    X = m.plant_step("duffing", X, u)
    m.step(X, r)
    A2, _, _ = m.get_model()
    assert float((A2[0] - A1[0]).abs().max()) > 1e-8  # the update is running again


# ------------------------------------------------------------------ MATLAB Runge-Kutta variant
@pytest.mark.parametrize("kind", ["duffing", "vdp"])
def test_matlab_rk4_variant(torch_mod, KM, kind):
    """f_ud of Koopman_update.m:21-25: k4 = f(x + k1 deltaT).  plant_step and the plant inside the fused roll-out against the
    oracle's restatement (2e-16 relative), and it is a different map than the Python scripts' RK4."""
    torch = torch_mod
    from koopmpc.synth import random_mlp_weights

    rng = np.random.RandomState(8)
    B = 200
    m = KM(n=2, L=8, N=10, batch=B, lift="rbf", centres=rng.rand(8, 2))
    X0 = 4 * rng.rand(2, B) - 2
    U = 4 * rng.rand(B) - 2
    Xm = m.plant_step(kind + "_matlab", _t(torch, X0), U).cpu().numpy()
    Xp = m.plant_step(kind, _t(torch, X0), U).cpu().numpy()
    ref = ko.plant_step(kind + "_matlab", X0, U)
    assert np.abs(Xm - ref).max() <= 4e-16 * max(1.0, np.abs(ref).max())
    assert np.abs(Xm - Xp).max() > 1e-6
    # inside the fused roll-out (cfg2 dimensions): states of a 3-step roll-out = step + plant_step(matlab) loop
    L = N = 20
    w = random_mlp_weights(2, 100, 3, L, seed=3)
    A, Bm, Cm = _rand_model(rng, L, 2)
    r = np.tile(np.array([[1.0], [0.0]]), (1, N))
    m1 = KM(n=2, L=L, N=N, batch=16, weights=w)
    m2 = KM(n=2, L=L, N=N, batch=16, weights=w)
    m1.set_model(A, Bm, Cm); m2.set_model(A, Bm, Cm)
    X1, X2 = _t(torch, X0[:, :16]), _t(torch, X0[:, :16])
    m1.rollout(kind + "_matlab", X1, r, 3)
    for k in range(3):
        X2 = m2.plant_step(kind + "_matlab", X2, m2.step(X2, r).clone())
    assert float((X1 - X2).abs().max()) < 1e-9


# ------------------------------------------------------------------ the wave image of the state
def test_wave_image_round_trips_through_every_entry_point(torch_mod, KM):
    """The fused roll-out of the small y = Cx sets keeps the state as a wave image; every other entry point works on the dense
    blocks.  Mixed use: roll-out, get_model, kmpc_rls_update, roll-out, checkpoint, kmpc_step -- against a handle that only
    ever takes single steps (dense blocks <-> image conversions on one side, none of them on the other)."""
    torch = torch_mod
    from koopmpc.synth import random_mlp_weights

    rng = np.random.RandomState(21)
    L = N = 20
    B = 40
    w = random_mlp_weights(2, 100, 3, L, seed=3)
    A, Bm, Cm = _rand_model(rng, L, 2)
    r = np.tile(np.array([[1.0], [0.0]]), (1, N))
    m1 = KM(n=2, L=L, N=N, batch=B, weights=w)
    m2 = KM(n=2, L=L, N=N, batch=B, weights=w)
    m1.set_model(A, Bm, Cm); m2.set_model(A, Bm, Cm)
    X0 = 4 * rng.rand(2, B) - 2
    X1, X2 = _t(torch, X0), _t(torch, X0)
    m1.rollout("duffing", X1, r, 3)
    for k in range(3):
        X2 = m2.plant_step("duffing", X2, m2.step(X2, r).clone())
    Aa, _, Ca = m1.get_model()          # image -> dense
    Ab, _, Cb = m2.get_model()
    assert float((Aa - Ab).abs().max()) < 1e-9 and float((Ca - Cb).abs().max()) < 1e-9
    m1.rollout("duffing", X1, r, 2)     # dense -> image (nothing changed in between: bitwise the image of before)
    for k in range(2):
        X2 = m2.plant_step("duffing", X2, m2.step(X2, r).clone())
    assert float((X1 - X2).abs().max()) < 1e-9
    sd = m1.state_dict()                # checkpoint reads the dense blocks
    m3 = KM(n=2, L=L, N=N, batch=B, weights=w)
    m3.load_state_dict(sd)
    X3 = X1.clone()
    m1.rollout("duffing", X1, r, 2)
    m3.rollout("duffing", X3, r, 2)
    assert torch.equal(X1, X3)
    assert int(m1.status.max().item()) == 0


def test_mpc_solve_keeps_the_handles_terminal_block(torch_mod, KM):
    """kmpc_mpc_solve without P_N uses the handle's terminal block (kmpc_set_terminal_weight), as its docstring says: the same
    minimiser as condense() + qp_solve() on that handle, and `fun` includes the terminal term (advisor finding, round 2)."""
    torch = torch_mod
    rng = np.random.RandomState(4)
    L, N, B = 8, 10, 6
    m = KM(n=2, L=L, N=N, batch=B, lift="rbf", centres=rng.rand(L, 2))
    A, Bm, Cm = _rand_model(rng, L, 2)
    m.set_model(A, Bm, Cm)
    PN = np.array([[700.0, 90.0], [90.0, 300.0]])
    m.set_terminal_weight(PN)
    psi = _t(torch, rng.randn(L, B))
    r = np.tile(np.array([[1.0], [0.0]]), (1, N))
    H, f, c = m.condense(psi, r, return_const=True)
    Uq, st, _ = m.qp_solve(H, f)
    U, u0, st2, fun = m.mpc_solve(A, Bm, Cm, psi, r)
    assert float((U - Uq).abs().max()) < 1e-8
    for b in range(B):
        _, _, Ho, fo, co = ko.condense(A, Bm, Cm, psi[:, b].cpu().numpy(), r, N, 100.0, 1e-4, PN=PN)
        Ub = U[:, b].cpu().numpy()
        assert abs(float(fun[b]) - (Ub @ Ho @ Ub + fo @ Ub + co)) <= 1e-9 * max(1.0, abs(co))


# ------------------------------------------------------------------ multi-rank readiness that a one-GPU box can prove
def _bench_child(extra, timeout=900):
    import json
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # rank 0 prints ONE line
    return json.loads(lines[0])


@pytest.mark.parametrize("B", [37, 256])
def test_shared_step_interior_kernel_equals_the_per_trajectory_solve(torch_mod, KM, monkeypatch, B):
    """The shared-model step finishes the trajectories whose unconstrained minimiser lies inside the box 16 per wave on the
    matrix cores (shared_fast_kernel) and leaves the rest to the per-trajectory solve.  The same closed loop with that kernel
    switched off (KMPC_SHARED_NO_FAST, read per call): same inputs, sequences, states and statuses; the loop starts with
    saturated inputs (both kernels at work in one step) and settles into the interior (cfg4 sizes, delta-u tank form)."""
    from koopmpc.synth import random_mlp_weights

    L, N = 32, 40
    w = random_mlp_weights(2, 100, 2, L, seed=9)
    rng = np.random.RandomState(5)
    Ub = 10 * rng.rand(60, 100) - 5
    xc = np.maximum(4 * rng.rand(2, 100) - 2, 0.0)
    Xs, Ys, Us = [], [], []
    for i in range(60):
        xn = ko.tank_step(xc, Ub[i])
        Xs.append(xc); Ys.append(xn); Us.append(Ub[i][None, :]); xc = xn
    Xd, Yd, Ud = np.concatenate(Xs, 1), np.concatenate(Ys, 1), np.concatenate(Us, 1)
    lift_fn = lambda x: ko.mlp_lift(w, x)
    V = np.concatenate([lift_fn(Xd), Ud], 0)
    M = np.concatenate([lift_fn(Yd), Xd], 0) @ V.T @ np.linalg.pinv(V @ V.T)
    X0 = np.abs(rng.rand(2, B)) * np.linspace(0.05, 2.0, B)[None, :]
    r = np.ones((1, N))

    def make():
        mpc = KM(n=2, L=L, N=N, batch=B, weights=w, layers=2, lb=-0.5, ub=0.5, umin=-8.0, umax=8.0, Qw=10.0, Rw=1e-3, P0=1e4,
                 barQ0=1e4, delta_u=True, out_row0=1, out_rows=1)
        mpc.set_model(M[:L, :L], M[:L, L:], M[L:, :L])
        return mpc

    # two handles in lockstep on the SAME state and input history (the second one is told the first one's inputs), so that
    # every step compares the two kernels on one problem instead of two closed loops that drift apart by rounding
    fast, slow = make(), make()
    X = X0.copy()
    worst, interior, bound = 0.0, 0, 0
    for k in range(40):
        monkeypatch.delenv("KMPC_SHARED_NO_FAST", raising=False)
        ua = fast.shared_step(X, r).cpu().numpy()
        monkeypatch.setenv("KMPC_SHARED_NO_FAST", "1")
        ub_ = slow.shared_step(X, r).cpu().numpy()
        slow.set_applied_input(ua)
        Ua, Ub_ = fast.Useq.cpu().numpy(), slow.Useq.cpu().numpy()
        sa, sb = fast.status.cpu().numpy(), slow.status.cpu().numpy()
        assert (sa == sb).all(), k
        ok = sa == 0
        worst = max(worst, np.abs(ua - ub_)[ok].max(), np.abs(Ua - Ub_)[:, ok].max())
        at_bound = (np.abs(Ub_) >= 0.5 - 1e-9).any(axis=0)
        interior += int((~at_bound).sum()); bound += int(at_bound.sum())
        X = ko.tank_step(X, ua)
    monkeypatch.delenv("KMPC_SHARED_NO_FAST", raising=False)
    print("interior kernel: %d interior / %d bound solves, worst difference %.2e" % (interior, bound, worst))
    assert interior > 0 and bound > 0
    # (both points pass the same gradient certificate, 1e-12 of the gradient scale; H of this form has a condition number near
    #  1e8 -- Rw = 1e-3 against Qw = 10 -- so the points themselves agree to about 1e-8, as two solver paths of one kernel do)
    assert worst < 1e-7

@pytest.mark.parametrize("L,N,lift", [(20, 20, "mlp"), (8, 30, "rbf")])
def test_cold_start_inside_a_fused_rollout_is_the_same_closed_loop(torch_mod, KM, L, N, lift):
    """cold_start = 1 inside a fused roll-out: no primal start is kept, but the solver carries its tableau from step to step and starts
    each solve at clip(0) moved onto the face of that tableau (qp_rl.h) -- with saturated sequences in the first steps and after the
    parameter switch.  The minimiser is unique: the cold and the warm loop must log the same inputs and states (to the certificate
    of the solve, amplified by 60 closed-loop steps), every status 0, and the cold one needs at least as many Newton solves."""
    torch = torch_mod
    from koopmpc.synth import initial_states, offline_data, random_mlp_weights

    B = 256
    rng = np.random.RandomState(3)
    kw = dict(weights=random_mlp_weights(2, 100, 3, L, seed=4)) if lift == "mlp" else dict(lift="rbf", centres=4 * rng.rand(L, 2) - 2)
    r = np.tile(np.array([[1.0], [0.0]]), (1, N))
    logs, iters = [], []
    for cold in (False, True):
        m = KM(n=2, L=L, N=N, batch=B, cold_start=cold, **kw)
        assert m.rollout_is_fused()
        m.offline_fit(*offline_data(), ridge=1e-8, init_rls=(lift == "rbf"))  # (RBF set: the RLS continues from the offline Gram, as vanderpol_RBF.py's storage update)
        X = torch.tensor(initial_states(B, seed=9), dtype=torch.float64, device="cuda:0").contiguous()
        U, Xl, it = [], [], 0.0
        for i in range(3):  # three launches: a launch starts without a carried tableau, the later steps of it with one
            u, x = m.rollout("duffing", X, r, 20, step0=20 * i, switch_step=30, log=True)
            assert int(m.status.max().item()) == 0, (cold, i)
            U.append(u.cpu().numpy()); Xl.append(x.cpu().numpy()); it += float(m.iters.double().mean())
        logs.append((np.concatenate(U), np.concatenate(Xl))); iters.append(it / 60)
    du = np.abs(logs[0][0] - logs[1][0]).max()
    dx = np.abs(logs[0][1] - logs[1][1]).max()
    dU = np.abs(logs[0][0] - logs[1][0])
    big = int((dU > 1e-6).sum())
    sat = float((np.abs(logs[1][0]) >= 2.0 - 1e-9).mean())
    print("cold vs warm fused loop L=%d N=%d: max |dU| %.2e, max |dX| %.2e, saturated first moves %.3f, Newton solves per step warm %.2f cold %.2f"
          % (L, N, du, dx, sat, iters[0], iters[1]))
    print("   inputs that differ by more than 1e-6: %d of %d (median difference %.1e)" % (big, dU.size, np.median(dU)))
    assert sat > 0.01            # (the loop does meet its bounds)
    # Two certified minimisers of the same QP differ by (KKT tolerance) / (smallest eigenvalue of H on the face): right after the RLS
    # reset and after the parameter switch single trajectories have nearly singular H (one-sample models, App. B of the survey), and a
    # difference made there decays over the next steps.  So: all but a handful of the 15 360 inputs agree to 1e-6, none differs by
    # more than 1e-4, the typical difference is at rounding level, and the states agree to 1e-5.
    assert big <= 0.002 * dU.size and du < 1e-4 and np.median(dU) < 1e-9 and dx < 1e-5


@pytest.mark.parametrize("L,N,lift", [(20, 20, "mlp"), (8, 30, "rbf")])
def test_fused_rollout_reports_non_finite_data_per_trajectory(torch_mod, KM, L, N, lift):
    """A trajectory whose state is not finite: the register-state solver (qp_rl.h) reports status 2 for it -- the test is on the
    gradient, the cost sums are only formed when a line search needs them -- and hands back the feasible point clip(0), never a NaN;
    its neighbours in the same workgroup are not touched (same inputs as a batch without the bad trajectory, bit for bit)."""
    torch = torch_mod
    from koopmpc.synth import initial_states, offline_data, random_mlp_weights

    B, bad = 32, 5
    rng = np.random.RandomState(3)
    kw = dict(weights=random_mlp_weights(2, 100, 3, L, seed=4)) if lift == "mlp" else dict(lift="rbf", centres=4 * rng.rand(L, 2) - 2)
    r = np.tile(np.array([[1.0], [0.0]]), (1, N))
    x0 = initial_states(B, seed=9)
    out = []
    for poison in (False, True):
        m = KM(n=2, L=L, N=N, batch=B, **kw)
        assert m.rollout_is_fused()
        m.offline_fit(*offline_data(), ridge=1e-8)
        xs = x0.copy()
        if poison:
            xs[0, bad] = np.nan
        X = torch.tensor(xs, dtype=torch.float64, device="cuda:0").contiguous()
        U, _ = m.rollout("duffing", X, r, 4, log=True)
        out.append((U.cpu().numpy(), m.status.cpu().numpy().copy()))
    (Uok, sok), (Ubad, sbad) = out
    others = np.arange(B) != bad
    assert (sok == 0).all() and (sbad[others] == 0).all() and sbad[bad] == 2
    assert np.array_equal(Uok[:, others], Ubad[:, others])
    # (the encoder's ReLU / the RBF's r > 0 test map the NaN state to a finite lift, so the first step is an ordinary solve; from the
    #  second step on the output-map update has seen the state itself: status 2, the input is clip(0) of the box [-2, 2])
    assert np.isfinite(Ubad[:, bad]).all() and np.abs(Ubad[0, bad]) <= 2.0 and np.abs(Ubad[1:, bad]).max() == 0.0


@pytest.mark.parametrize("cfg", ["cfg2", "cfg4"])
def test_bench_two_ranks_as_a_child_process(torch_mod, cfg):
    """`python bench.py --gpus 2 --backend gloo --same-device` as the driver would start it for N > 1, minus the second GPU:
    bench.py spawns its own two ranks (torch.distributed.run as a child, before any GPU call), they rendezvous on 127.0.0.1,
    shard the trajectories, run the timed region between barriers and reduce the MAX; cfg4 also all-reduces the Gram block
    between the two stages of every step.  The JSON line must say what the process group reported."""
    d = _bench_child(["--gpus", "2", "--backend", "gloo", "--same-device", "--config", cfg, "--steps", "4", "--warmup", "2",
                      "--cpu-seconds", "0", "--spin-seconds", "0", "--no-extras", "--settle", "6", "--batch", "512"])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2
    pg = d["config"]["process_group"]
    assert pg["world_size"] == 2 and pg["backend"] == "gloo" and pg["ranks_share_device"] is True
    assert d["config"]["global_batch"] == 2 * 512
    assert d["config"]["worst_qp_status"] == 0 and d["config"]["finite"] is True
    assert d["value"] > 0 and abs(d["value"] - 2 * 512 * 4 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]
    assert "rehearsal" in d["config"]  # (never to be read as a scaling figure)


@pytest.mark.parametrize("cfg", ["cfg2", "cfg4"])
def test_bench_on_a_one_rank_rccl_process_group(torch_mod, cfg):
    """What a one-GPU box can prove about the `nccl` (= RCCL) branch the driver's multi-GPU runs take: `bench.py --gpus 1
    --force-process-group` initialises a ONE-rank process group on RCCL and executes the path's collectives on it -- the barriers
    around the timed region, the MAX of the elapsed time, and for cfg4 the all-reduce of the Gram block between the two stages of
    every step (device tensors, torch's current stream, float64 SUM).  One rank: the sums are the identity, so the result must equal
    the run without a process group up to the clock."""
    args = ["--gpus", "1", "--config", cfg, "--steps", "4", "--warmup", "2", "--cpu-seconds", "0", "--spin-seconds", "0", "--no-extras",
            "--settle", "6", "--batch", "512"]
    d = _bench_child(args + ["--force-process-group"])
    pg = d["config"]["process_group"]
    assert pg["world_size"] == 1 and pg["backend"] == "nccl"
    assert d["n_gpus"] == 1 and d["config"]["worst_qp_status"] == 0 and d["config"]["finite"] is True and d["value"] > 0
    plain = _bench_child(args)
    assert plain["config"]["process_group"]["backend"].startswith("none")
    assert plain["config"]["qp"] == d["config"]["qp"]  # (same Newton-solve statistics: the forced collectives changed no number)


_RCCL_CHILD = r"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(%(root)r, "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import KoopmanMPC, _ffi
# the RCCL of this process: torch's own copy when it exports the symbols, else the system library
proc = C.CDLL(None)
try:
    proc.ncclCommInitRank
    rccl = proc
except AttributeError:
    rccl = C.CDLL("librccl.so", mode=C.RTLD_GLOBAL)
class UID(C.Structure):
    _fields_ = [("b", C.c_char * 128)]
uid = UID()
rc = rccl.ncclGetUniqueId(C.byref(uid))
comm = C.c_void_p()
if rc == 0:
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UID, C.c_int]
    rc = rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0)
if rc != 0:
    print("NO_COMM", rc); sys.exit(0)
m = KoopmanMPC(n=2, L=8, N=10, batch=64, lift="rbf", centres=np.random.RandomState(0).rand(8, 2))
lib = _ffi.load()
ne = int(lib.kmpc_gram_elems(m.h))
g = torch.arange(ne, dtype=torch.float64, device="cuda:0") * 0.5 + 1.0
ref = g.clone()
rc = lib.kmpc_allreduce_gram(m.h, C.c_void_p(g.data_ptr()), comm, C.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
print("RC", rc, "EQUAL", bool(torch.equal(g, ref)), "ELEMS", ne)
rccl.ncclCommDestroy.argtypes = [C.c_void_p]
rccl.ncclCommDestroy(comm)
"""


def test_allreduce_gram_through_rccl(torch_mod):
    """kmpc_allreduce_gram on a real RCCL communicator (one rank: the sum over ranks is the block itself): the library resolves
    ncclAllReduce from the process, passes the handle's element count as float64 / sum on the caller's stream and maps the
    result code.  Run in a child process; when the box cannot create a communicator at all the test says so and skips."""
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, "-c", _RCCL_CHILD % {"root": ROOT}], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    out = p.stdout.strip().splitlines()
    if p.returncode == 0 and out and out[-1].startswith("NO_COMM"):
        pytest.skip("RCCL could not create a one-rank communicator on this box: %s" % out[-1])
    assert p.returncode == 0, p.stderr[-3000:]
    assert out and out[-1].startswith("RC 0 EQUAL True"), (p.stdout[-500:], p.stderr[-1500:])
