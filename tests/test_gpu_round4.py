"""Round-4 parity tests of the HIP path (through the C ABI).

  the reference's own names at module level: net.Encoder(x), rbf(X, cx), costFunction(u, r, AB, C, x0, ...)   duffing.py:17-29, 540-581, 847
  the thin-plate logarithm of the device against libm over the argument range                                 vanderpol_RBF.py:21-22
  ONE fused K = 20 launch at full size against 20 one-step calls and against per-trajectory oracles           duffing.py:823-1012
  cold against warm start: where the two loops differ, WHICH one is right (KKT of the exported QP)             duffing.py:634-635, 857-861
  the native shared-model loop (kmpc_shared_rollout) against the Python loop of its stages                    Tank_System.m:170-291

Runs on the MI355X box:  python -m pytest tests -m gpu
"""
import os

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


# ------------------------------------------------------------------ the reference's names at module level
def test_module_level_costFunction_matches_the_reference_values(torch_mod):
    """koopmpc.costFunction(uSequence, r, AB, C, x0, pastu, Np, Nc, d) -- the signature of duffing.py:540 / vanderpol.py:445 -- on the
    device against the values the reference's own costFunction returned for random sequences on its loop's models (fixtures written
    by tests/golden/make_golden.py): 1e-10 relative; a batch of sequences in one call gives the same numbers."""
    import koopmpc

    for fname, lifted in (("duffing_loop.npz", False), ("vanderpol_loop.npz", True)):
        g = _load(fname)
        by_k = {}
        for row, J in zip(g["cost_in"], g["cost_out"]):
            k = int(row[0])
            A, B, Cm, psi, r = g["loop_Ap"][k], g["loop_Bp"][k], g["loop_Cp"][k], g["loop_xlift"][k], g["loop_r"][k]
            AB = np.concatenate([A, B.reshape(-1, 1)], axis=1)
            N = len(row) - 1
            Jg = koopmpc.costFunction(row[1:], r, AB, None if lifted else Cm, psi.reshape(-1, 1), 0.0, N, N, np.zeros((A.shape[0], 1)))
            assert abs(Jg - J) <= 1e-10 * abs(J), (fname, k, Jg, J)
            by_k.setdefault(k, []).append((row[1:], J))
        for k, rows in list(by_k.items())[:3]:  # several sequences for one model in one call
            A, B, Cm, psi, r = g["loop_Ap"][k], g["loop_Bp"][k], g["loop_Cp"][k], g["loop_xlift"][k], g["loop_r"][k]
            AB = np.concatenate([A, B.reshape(-1, 1)], axis=1)
            U = np.stack([u for u, _ in rows], axis=1)
            Jg = koopmpc.costFunction(U, r, AB, None if lifted else Cm, psi)
            assert np.abs(Jg - np.array([J for _, J in rows])).max() <= 1e-10 * max(abs(J) for _, J in rows)
    with pytest.raises(ValueError):
        koopmpc.costFunction(np.zeros(10), g["loop_r"][0], AB, None, psi, d=np.ones((AB.shape[0], 1)))


def test_module_level_Encoder_and_rbf(torch_mod):
    """net = AutoEncoder(weights); net.Encoder(x) (duffing.py:17-29, 847) and rbf(X, cx) (vanderpol_RBF.py:20-23) as module-level
    names: the reference's own encoder outputs / rbf outputs (fixtures) to 1e-12, with the reference's shapes for a single state."""
    import koopmpc

    g = _load("duffing_loop.npz")
    w = ko.load_mlp_weights(_load("weights_duffing.npz"))
    net = koopmpc.AutoEncoder(w)
    psi = net.Encoder(g["lift_X"].T)  # (64 states of the reference's own encoder, (L, B) out)
    assert np.abs(psi.T - g["lift_Psi"]).max() <= 1e-12 * max(1.0, np.abs(g["lift_Psi"]).max())
    x = np.array([-2.0, -2.0])
    assert net.Encoder(x).shape == (8,) and net.Encoder(x.reshape(2, 1)).shape == (8, 1)
    gr = _load("vanderpol_rbf_loop.npz")
    cx = gr["cx"]
    out = koopmpc.rbf(gr["lift_X"].T, cx)
    assert out.shape == (8, 64) and np.abs(out.T - gr["lift_Psi"]).max() <= 1e-12 * np.abs(gr["lift_Psi"]).max()
    assert koopmpc.rbf(gr["lift_X"][0], cx).shape == (cx.shape[0], 1)  # (the reference returns (L, 1) for one state)


def test_device_log_against_libm_over_the_argument_range(torch_mod, KM):
    """kmpc_log (plant_device.h: fdlibm's reduction and polynomial with the coefficients as immediates) inside the thin-plate lift,
    psi = d^2 log(d + eps): distances from 1e-7 to 1e6 against NumPy's log -- the logarithm recovered as psi / d^2 within 2 ulp of
    |log| (absolute 4e-16 where the logarithm passes through zero); +inf gives +inf and NaN gives NaN in the stand-alone lift."""
    L = 4
    cx = np.zeros((L, 2))
    d = np.concatenate([np.logspace(-7, 6, 3000), 1.0 - 1e-4 + np.linspace(-1e-6, 1e-6, 41), [1e-4, 0.5, 2.0, 1e3]])
    X = np.stack([d, np.zeros_like(d)])
    m = KM(n=2, L=L, N=2, batch=1, lift="rbf", centres=cx, rbf_eps=1e-4)
    psi = m.Encoder(X)[0]
    lg = psi / (d * d)
    want = np.log(d + 1e-4)
    err = np.abs(lg - want)
    tol = 4 * np.finfo(float).eps * np.maximum(np.abs(want), 0.25)  # (d * d and the division add an ulp each to the log's own)
    assert np.all(err <= tol), (float((err / tol).max()), d[np.argmax(err / tol)])
    Xs = np.array([[np.inf, np.nan], [0.0, 0.0]])
    ps = m.Encoder(Xs)[0]
    assert np.isposinf(ps[0]) and np.isnan(ps[1])
