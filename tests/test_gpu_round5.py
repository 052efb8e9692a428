"""Round-5 parity tests of the HIP path (through the C ABI).

  the fused roll-out behind float32 panels (BASELINE configs[1] "fp32", SURVEY G6: fp32 I/O, float64 state)    duffing.py:823-1012, 927-953
  module-level Koopman_update / mpc_solve / mpc_step (the reference's call surface)                    duffing.py:857-861, 900-967
  bench.py as four ranks on one device (rehearsal of the driver's 8-GPU launch: sharding, MAX, cfg4's sum)       SURVEY 8e
  the work-ranked placement of a fused launch never changes a trajectory's arithmetic                             (kernels.h RolloutArgs::perm)

Runs on the MI355X box:  python -m pytest tests -m gpu
"""
import json
import os
import subprocess
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


def _f32_exact(a):
    """the float64 array whose values are exactly representable in float32"""
    return np.asarray(a, dtype=np.float32).astype(np.float64)


# ------------------------------------------------------------------ row g2: float32 panels around the float64 roll-out
def _cfg2_pair(KM, torch, B, seed=101):
    """A KMPC_F32 and a KMPC_F64 controller of BASELINE cfg2's dimensions with the SAME (float32-representable) offline model,
    initial states and reference: what the float32 boundary hands over is then exact on both sides."""
    from koopmpc.synth import initial_states, offline_data, random_mlp_weights

    L, N = 20, 20
    w = random_mlp_weights(2, 100, 3, L, seed=2024)
    Xo, Yo, Uo = offline_data()
    PX, PY = ko.mlp_lift(w, Xo), ko.mlp_lift(w, Yo)
    Z = np.concatenate([PX, Uo[None, :]], 0)
    K0 = PY @ np.linalg.pinv(Z)
    A0, B0, C0 = _f32_exact(K0[:, :L]), _f32_exact(K0[:, L:]), _f32_exact(Xo @ np.linalg.pinv(PX))
    X0 = _f32_exact(initial_states(B, seed=seed))
    r = np.tile(np.array([[1.0], [0.0]]), (1, N))
    m32 = KM(n=2, L=L, N=N, batch=B, weights=w, dtype=torch.float32, device="cuda:0")
    m64 = KM(n=2, L=L, N=N, batch=B, weights=w, dtype=torch.float64, device="cuda:0")
    for m in (m32, m64):
        m.set_model(A0, B0, C0)
    return w, (A0, B0, C0), X0, r, m32, m64


def test_f32_io_fused_rollout_is_the_f64_rollout_behind_float32_panels(torch_mod, KM):
    """BASELINE configs[1] as it is named: Duffing, 20-dim lift, N = 20, 4096 trajectories, float32 panels.  kmpc_rollout on a
    KMPC_F32 handle is ONE launch of the float64 register-state roll-out (rollout_kernel<.., float>): state (wave image), psi, u_{k-1},
    warm start and every operation in float64, x_{k+1} carried in LDS in float64, only X / ref / U_log / X_log rounded at the boundary.
    From float32-representable inputs the launch therefore computes bit for bit what the float64 handle's launch computes, and its
    outputs are the float32 roundings of that launch's outputs: inputs and states within one float32 ulp (relative 1.2e-7 < the
    north star's 1e-6), the models handed over (kmpc_get_model: [A B], C) within 6e-8 relative, the same worst status.  A second launch
    continues from the ROUNDED states (the float64 twin is given the same ones): same statement again.  Against the oracle
    (duffing.py:847-984, gain-form RLS, exact QP) 24 trajectories of both launches: |u - u_oracle| <= 1e-6."""
    torch = torch_mod
    B, K = 4096, 20
    w, (A0, B0, C0), X0, r, m32, m64 = _cfg2_pair(KM, torch, B)
    assert m32.rollout_is_fused() and m64.rollout_is_fused()
    X32 = torch.tensor(X0, dtype=torch.float32, device="cuda:0").contiguous()
    X64 = torch.tensor(X0, dtype=torch.float64, device="cuda:0").contiguous()
    Ulog, Xlog = [], []
    for launch in range(2):
        U32, L32 = m32.rollout("duffing", X32, r, K, step0=launch * K, log=True)
        U64, L64 = m64.rollout("duffing", X64, r, K, step0=launch * K, log=True)
        assert U32.dtype == torch.float32 and L32.dtype == torch.float32
        assert int(m32.status.max().item()) == 0 and int(m64.status.max().item()) == 0
        assert torch.equal(m32.iters, m64.iters)  # (the same solves, solve for solve)
        du = float((U32.double() - U64).abs().max())
        dxl = float((L32.double() - L64).abs().max())
        # one rounding to float32: half an ulp of values up to 2 (inputs) / ~3 (states)
        assert torch.equal(U32, U64.float()) and torch.equal(L32, L64.float()), (launch, du, dxl)
        assert torch.equal(X32, X64.float())
        Ulog.append(U64.cpu().numpy()); Xlog.append(L64.cpu().numpy())
        A32, B32, C32 = m32.get_model()
        A64, B64, C64 = m64.get_model()
        assert A32.dtype == torch.float32
        for a32, a64 in ((A32, A64), (B32, B64), (C32, C64)):
            rel = float(((a32.double() - a64).abs() / a64.abs().clamp(min=1.0)).max())
            assert rel <= 6e-8, (launch, rel)  # [A B], C: the float64 model, rounded once
        print("launch %d: float32-I/O roll-out vs float64 roll-out: max |du| %.2e |dx| %.2e (one float32 rounding), model rel %.1e" % (launch, du, dxl, rel))
        # (get_model only READS the core's state through the float32 blocks: the handle's state stays float64 from launch to launch;
        #  what the boundary rounds between two launches is X -- the twin continues from the same rounded states)
        X64.copy_(X32.double())
    # ---- the oracle on 24 trajectories, following the device's (float64 twin's) states launch by launch
    L, N = 20, 20
    lift_fn = lambda x: ko.mlp_lift(w, x)
    Uall, Xall = np.concatenate(Ulog), np.concatenate(Xlog)
    rng = np.random.RandomState(11)
    worst = 0.0
    for b in rng.choice(B, 24, replace=False):
        ctl = ko.OracleController(lift_fn, L, 2, N, -2.0, 2.0, A0, B0, C0, rls="gain")
        x = X0[:, b].copy()
        for k in range(K):  # (the first launch: from the reset on, the twin's own float64 states)
            uo, _, _ = ctl.step(x, r)
            worst = max(worst, abs(float(np.float32(Uall[k, b])) - uo))
            ctl.prev = (ctl.prev[0], float(Uall[k, b]))
            x = Xall[k, :, b].copy()
    print("   24 trajectories of the float32-I/O launch vs per-trajectory oracles: max |u - u_oracle| %.2e" % worst)
    assert worst < 1e-6


def test_f32_handle_mixes_rollouts_and_per_step_calls(torch_mod, KM):
    """A KMPC_F32 handle whose roll-outs run on its float64 core stays ONE controller: kmpc_step (float32 kernels) after a roll-out
    continues from the core's state (pulled into the float32 blocks), a roll-out after kmpc_step from the blocks (pushed to the core);
    kmpc_set_applied_input, kmpc_reset and checkpoints see the same state.  Checked against a float64 handle driven the same way
    (loosely: the float32 per-step arithmetic right after the reset parts from float64 at the 1e-2 level) and for the hand-over itself:
    a checkpoint taken between two launches and loaded into a fresh float32 handle gives the same next launch bit for bit."""
    torch = torch_mod
    B = 256
    w, _, X0, r, m32, m64 = _cfg2_pair(KM, torch, B, seed=5)
    X32 = torch.tensor(X0, dtype=torch.float32, device="cuda:0").contiguous()
    X64 = torch.tensor(X0, dtype=torch.float64, device="cuda:0").contiguous()
    m32.rollout("duffing", X32, r, 12)
    m64.rollout("duffing", X64, r, 12)
    assert torch.equal(X32, X64.float())
    # per-step calls on both (float32 arithmetic on the float32 handle)
    for k in range(3):
        u32 = m32.step(X32, r).clone()
        u64 = m64.step(X64, r).clone()
        assert int(m32.status.max().item()) == 0
        # (float32 RLS / condense / QP a dozen steps after the 1e4 I reset, free-running against float64: the two closed loops part
        #  at the 1e-2 level here -- SURVEY G6 is why the roll-outs keep the state in float64; this test is about the hand-over)
        assert float((u32.double() - u64).abs().max()) < 0.25, k
        X32 = m32.plant_step("duffing", X32, u32)
        X64 = m64.plant_step("duffing", X64, u64)
    # back to roll-outs: the core continues from the blocks the steps left
    U32, _ = m32.rollout("duffing", X32, r, 8, step0=15, log=True)
    U64, _ = m64.rollout("duffing", X64, r, 8, step0=15, log=True)
    assert int(m32.status.max().item()) == 0 and bool(torch.isfinite(X32).all())
    assert float((U32.double() - U64).abs().median()) < 1e-2
    # hand-over is loss-free beyond the one rounding: checkpoint -> a fresh float32 handle -> the same next launch
    sd = m32.state_dict()
    m32b = KM(n=2, L=20, N=20, batch=B, weights=w, dtype=torch.float32, device="cuda:0")
    m32b.load_state_dict(sd)
    Xa, Xb = X32.clone(), X32.clone()
    Ua, _ = m32.rollout("duffing", Xa, r, 6, step0=23, log=True)
    Ub, _ = m32b.rollout("duffing", Xb, r, 6, step0=23, log=True)
    assert torch.equal(Ua, Ub) and torch.equal(Xa, Xb)
    m32.reset()
    m32.rollout("duffing", Xa, r, 4)
    assert int(m32.status.max().item()) == 0 and bool(torch.isfinite(Xa).all())


def test_f32_io_rbf_rollout(torch_mod, KM):
    """The register-state RBF set (cfg3's dimensions: 8 thin-plate RBFs, N = 30, y = C x, the estimator continued from the offline
    Gram as vanderpol_RBF.py:434-438) behind float32 panels.  Two things are rounded that the MLP sets do not round: the state the
    float32 handle is INITIALISED with (kmpc_state_init_from fills its float32 blocks: inv_K_G, cond ~ 1e10, to 6e-8 relative) and x
    once per step (every wave re-reads its state from the X panel; no cooperative lift keeps it in LDS).  Stated tolerance against the
    float64 handle over 15 closed-loop steps from the same start: inputs within 1e-2 (0.25 % of the box), states within 1e-3; every
    solve optimal (measured 2.3e-3 / 1.3e-4)."""
    torch = torch_mod
    from koopmpc.synth import initial_states, offline_data, vdp_rk4

    L, N, B = 8, 30, 512
    Xo, Yo, Uo = [_f32_exact(a) for a in offline_data(plant=vdp_rk4)]
    cx = Xo[:, np.random.RandomState(0).choice(Xo.shape[1], L, replace=False)].T.copy()
    PX, PY = ko.rbf_lift(Xo, cx), ko.rbf_lift(Yo, cx)
    Z = np.concatenate([PX, Uo[None, :]], 0)
    P = np.linalg.inv(Z @ Z.T + 1e-9 * np.eye(L + 1)); KA = PY @ Z.T
    bQ = np.linalg.inv(PX @ PX.T + 1e-9 * np.eye(L)); bX = Xo @ PX.T
    ms = []
    for dt in (torch.float32, torch.float64):
        m = KM(n=2, L=L, N=N, batch=B, lift="rbf", centres=cx, P0=1e5, barQ0=1e5, dtype=dt, device="cuda:0")
        m.set_model((KA @ P)[:, :L], (KA @ P)[:, L], bX @ bQ)
        m.state_init(K_A=KA, inv_K_G=P, bar_X=bX, bar_Q=bQ)
        ms.append(m)
    m32, m64 = ms
    assert m32.rollout_is_fused()
    X0 = _f32_exact(initial_states(B, seed=3))
    r = np.tile(np.array([[1.0], [0.0]]), (1, N))
    X32 = torch.tensor(X0, dtype=torch.float32, device="cuda:0").contiguous()
    X64 = torch.tensor(X0, dtype=torch.float64, device="cuda:0").contiguous()
    U32, _ = m32.rollout("vdp", X32, r, 15, log=True)
    U64, _ = m64.rollout("vdp", X64, r, 15, log=True)
    assert int(m32.status.max().item()) == 0 and int(m64.status.max().item()) == 0 and bool(torch.isfinite(X32).all())
    du, dx = float((U32.double() - U64).abs().max()), float((X32.double() - X64).abs().max())
    print("RBF set behind float32 panels, 15 steps: max |du| %.2e |dx| %.2e" % (du, dx))
    assert du < 1e-2 and dx < 1e-3


# ------------------------------------------------------------------ the reference's names at module level
def test_module_level_Koopman_update_mpc_solve_mpc_step(torch_mod, KM):
    """koopmpc.Koopman_update(state, xlift, u, ylift, x_next), koopmpc.mpc_solve(A, B, C, xlift, r, lb, ub, Q, R) and
    koopmpc.mpc_step(state, x, r) -- the module-level spellings of duffing.py:900-967, :857-861 and :847-984 -- on the reference's own
    loop (fixture duffing_loop.npz: 130 steps of duffing.py executed here): the update reproduces the reference's [A B], C from its
    own (xlift, u, ylift, x_loc) sequence (oracle gain form, 1e-7: the reference form's re-association floor), mpc_solve returns the
    exact minimiser of the reference's costFunction for the reference's models (oracle qp_exact, 1e-8; cost never above the
    reference's L-BFGS-B result), for one psi and for a batch of psi columns, with C = None for y = psi (vanderpol.py:456-459), and
    mpc_step strings them together like KoopmanMPC.step."""
    torch = torch_mod
    import koopmpc

    g = _load("duffing_loop.npz")
    w = ko.load_mlp_weights(_load("weights_duffing.npz"))
    L, n, N = 8, 2, 10
    # ---- Koopman_update on the reference's own sequence
    state = KM(n=n, L=L, N=N, batch=1, weights=w, device="cuda:0")
    A = Bm = C = None
    for k in range(40):
        A, Bm, C = koopmpc.Koopman_update(state, g["loop_xlift"][k], g["loop_u_loc"][k].reshape(1), g["loop_ylift"][k], g["loop_x_loc"][k])
    A, Bm, C = A[0].cpu().numpy(), Bm[0].cpu().numpy(), C[0].cpu().numpy()
    Kref, Cref = g["loop_K_ext"][39], g["loop_C_prev"][39]  # (the reference's own K_ext / C after its 40th update)
    assert np.abs(np.concatenate([A, Bm.reshape(L, 1)], axis=1) - Kref).max() <= 1e-7 * np.abs(Kref).max()
    assert np.abs(C - Cref).max() <= 1e-9 * max(1e-3, np.abs(Cref).max())
    with pytest.raises(ValueError):
        koopmpc.Koopman_update(state, g["loop_xlift"][0], g["loop_u_loc"][0].reshape(1), g["loop_ylift"][0], g["loop_x_loc"][0], lam=0.9)
    assert koopmpc.koopman_update is koopmpc.Koopman_update
    # ---- mpc_solve: the reference's models, one psi per call
    worst = 0.0
    for k in (3, 20, 60, 110):
        Ak, Bk, Ck, psi, r = g["loop_Ap"][k], g["loop_Bp"][k], g["loop_Cp"][k], g["loop_xlift"][k], g["loop_r"][k]
        U, u0, st = koopmpc.mpc_solve(Ak, Bk, Ck, psi.reshape(-1), r, -2.0, 2.0, 100.0, 1e-4)
        assert U.shape == (N, 1) and u0.shape == (1,) and int(st[0]) == 0
        _, _, H, f, c = ko.condense(Ak, Bk, Ck, psi.reshape(-1), r, N, 100.0, 1e-4)
        Uo, _ = ko.qp_exact(H, f, -2.0, 2.0)
        worst = max(worst, float(np.abs(U[:, 0] - Uo).max()))
        J = lambda u: float(u @ H @ u + f @ u + c)
        assert J(U[:, 0]) <= g["loop_J"][k] * (1 + 1e-9) + 1e-12  # never worse than the reference's own L-BFGS-B result
    assert worst <= 1e-8, worst
    # ---- a batch of psi columns against one model; C = None (y = the lifted state)
    k = 60
    Ak, Bk = g["loop_Ap"][k], g["loop_Bp"][k]
    Psi = np.stack([g["loop_xlift"][j].reshape(-1) for j in (10, 30, 60, 90, 120)], axis=1)  # (L, 5)
    rl = np.tile(ko.mlp_lift(w, np.array([1.0, 0.0])).reshape(L, 1), (1, N))
    U, u0, st = koopmpc.mpc_solve(Ak, Bk, None, Psi, rl, -6.0, 6.0)
    assert U.shape == (N, 5) and (st == 0).all()
    for j in range(5):
        _, _, H, f, _ = ko.condense(Ak, Bk, np.eye(L), Psi[:, j], rl, N, 100.0, 1e-4)
        Uo, _ = ko.qp_exact(H, f, -6.0, 6.0)
        # y = psi makes H ill-conditioned (cond ~ 1e8): the minimiser is pinned through its cost
        Jd, Jo = float(U[:, j] @ H @ U[:, j] + f @ U[:, j]), float(Uo @ H @ Uo + f @ Uo)
        assert Jd <= Jo + 1e-9 * max(1.0, abs(Jo)), (j, Jd, Jo)
        assert abs(u0[j] - U[0, j]) == 0.0
    # ---- mpc_step == KoopmanMPC.step
    s1 = KM(n=n, L=L, N=N, batch=3, weights=w, device="cuda:0")
    s2 = KM(n=n, L=L, N=N, batch=3, weights=w, device="cuda:0")
    for s in (s1, s2):
        s.set_model(g["A0"], g["B0"], g["C0"])
    X = _t(torch, np.array([[-2.0, 0.5, 1.0], [-2.0, 0.3, -1.0]]))
    for k in range(4):
        ua = koopmpc.mpc_step(s1, X, g["loop_r"][0]).clone()
        ub = s2.step(X, g["loop_r"][0]).clone()
        assert torch.equal(ua, ub)
        X = s1.plant_step("duffing", X.clone(), ua)


# ------------------------------------------------------------------ placement never changes a trajectory's arithmetic
def test_placement_by_work_leaves_results_bit_for_bit(torch_mod, KM, monkeypatch):
    """The fused roll-out deals its trajectories to the waves by the solver work of the last five steps of the previous launch
    (RolloutArgs::perm / work_tail).  A trajectory's arithmetic is its own: the same three chained launches with the placement
    switched off (KMPC_ROLLOUT_NO_PLACE, read per call) give the same inputs and states bit for bit."""
    torch = torch_mod
    B = 1024
    outs = []
    for off in (False, True):
        w, _, X0, r, _, m = _cfg2_pair(KM, torch, B, seed=9)
        X = torch.tensor(X0, dtype=torch.float64, device="cuda:0").contiguous()
        if off:
            monkeypatch.setenv("KMPC_ROLLOUT_NO_PLACE", "1")
        us = [m.rollout("duffing", X, r, k, step0=s0, log=True)[0].clone() for s0, k in ((0, 12), (12, 20), (32, 7))]
        monkeypatch.delenv("KMPC_ROLLOUT_NO_PLACE", raising=False)
        outs.append((torch.cat(us), X.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


# ------------------------------------------------------------------ the driver's 8-GPU launch, rehearsed on one device
def _bench_child(extra, timeout=1100):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # rank 0 prints ONE line
    return json.loads(lines[0])


@pytest.mark.parametrize("cfg", ["cfg2", "cfg4"])
def test_bench_four_ranks_rehearsal(torch_mod, cfg):
    """`python bench.py --gpus N --backend gloo --same-device` with as many ranks as a one-GPU box lets a job put on its card beside
    the test process itself (four: the box kills a job with more than six processes on the GPU; the driver's run has eight, one GPU
    each, on RCCL -- the rank arithmetic is the same code with another N): bench.py spawns the
    ranks before any GPU call, they rendezvous on 127.0.0.1, every rank takes its own trajectories (seed 101 + rank), the timed
    region sits between barriers, the MAX over the ranks is the time, rank 0 prints ONE line.  The line must say what the process
    group reported (world size, backend), the global batch, a finite closed loop -- and, for the shared model of cfg4, that the four
    shards solved with ONE model: the line's parity probe compares rank 0's inputs with the oracle given the pooled model the
    device exported, which only holds if the Gram blocks of all four ranks were summed."""
    N = 4
    d = _bench_child(["--gpus", str(N), "--backend", "gloo", "--same-device", "--config", cfg, "--steps", "4", "--warmup", "2",
                      "--cpu-seconds", "0", "--spin-seconds", "0", "--no-extras", "--settle", "6", "--batch", "256"])
    assert d["n_gpus"] == N and d["steps"] == 4 and d["warmup"] == 2
    pg = d["config"]["process_group"]
    assert pg["world_size"] == N and pg["backend"] == "gloo" and pg["ranks_share_device"] is True
    assert pg["rccl_ranks_seen"] is None  # (a gloo rehearsal: RCCL saw nobody, and the line says so)
    assert d["config"]["global_batch"] == N * 256
    assert d["config"]["worst_qp_status"] == 0 and d["config"]["finite"] is True
    assert d["value"] > 0 and abs(d["value"] - N * 256 * 4 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]
    assert "rehearsal" in d["config"]  # (never to be read as a scaling figure)
    err = d["config"]["parity_probe"]["max_abs_u_err"]
    assert err is not None and err < 1e-6, d["config"]["parity_probe"]


def test_bench_one_rank_rccl_reports_ranks_seen(torch_mod):
    """`bench.py --gpus 1 --force-process-group --config cfg4`: the shared-model loop all-reduces on THE process's RCCL communicator
    (koopmpc.sharding.process_communicator: one per process, destroyed at the end) and the line carries what RCCL itself counts
    (ncclCommCount) as `rccl_ranks_seen` -- 1 here, 8 on the driver's node."""
    d = _bench_child(["--gpus", "1", "--config", "cfg4", "--steps", "4", "--warmup", "2", "--cpu-seconds", "0", "--spin-seconds", "0",
                      "--no-extras", "--settle", "6", "--batch", "512", "--force-process-group"])
    pg = d["config"]["process_group"]
    assert pg["world_size"] == 1 and pg["backend"] == "nccl" and pg["rccl_ranks_seen"] == 1
    assert d["config"]["worst_qp_status"] == 0 and d["config"]["finite"] is True


def test_two_rccl_ranks_on_one_gpu_are_refused_cleanly(torch_mod):
    """The world > 1 branch of koopmpc.sharding.NcclCommunicator on the only hardware a pool box has: two processes, both on cuda:0.
    The unique id travels from rank 0 over the (gloo) process group, both ranks pass the hash check, both call
    ncclCommInitRank(world = 2) -- and RCCL refuses two ranks on one device (ncclInvalidUsage, code 5: duplicate GPU).  What the test
    pins: that branch runs to the RCCL call on both ranks, and a failure there is an exception and a non-zero exit on EVERY rank within
    seconds -- not a hang, not a re-exec.  (The same code with one GPU per rank is what the driver's 8-GPU run executes.)"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dbg", "rccl_two_ranks_one_gpu.py")], cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    out = p.stdout
    assert p.returncode == 0, out[-2000:]
    if "communicator refused" in out:
        for r in (0, 1):
            assert ("rank %d: communicator refused: ncclCommInitRank failed with code 5" % r) in out, out[-2000:]
        assert out.count("| exit 3") == 2, out[-2000:]
    else:  # (an RCCL that accepts two ranks on one device: then the sums must be right and RCCL must count two ranks)
        for r in (0, 1):
            assert ("rank %d: ncclAllReduce rc 0, ranks seen 2, sum 3.0" % r) in out, out[-2000:]
