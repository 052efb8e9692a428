"""Synthetic workloads for the bench and the full-size tests: random-init encoders of
the reference's architecture, initial states, and the offline EDMD model the closed loop starts
from.  Host-side set-up only -- nothing here runs per control step."""
from __future__ import annotations

import numpy as np


def random_mlp_weights(n=2, hidden=100, layers=3, L=20, seed=2024):
    """W ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) like torch.nn.Linear's default, the way the reference's
    nets were initialised (DeepLearning_KoopmanControl_Approach3.py:314-322)."""
    rng = np.random.RandomState(seed)
    dims = [n] + [hidden] * layers + [L]
    ws = []
    for i in range(len(dims) - 1):
        k = 1.0 / np.sqrt(dims[i])
        ws.append((rng.uniform(-k, k, (dims[i + 1], dims[i])), rng.uniform(-k, k, dims[i + 1])))
    return ws


def initial_states(B, seed=101, lo=-2.0, hi=2.0, n=2):
    """x0 ~ U[-2, 2]^n (data_generate.py:41), seed 101 (+ rank) as duffing.py:47."""
    rng = np.random.RandomState(seed)
    return lo + (hi - lo) * rng.rand(n, B)


def duffing_rk4(x, u, h=0.05):
    """duffing.py:255-261 on (2,M) states with (M,) inputs (used only to make offline data)."""
    f = lambda s: np.array([s[1], -0.5 * s[1] + s[0] - s[0] ** 3 + u])
    k1 = f(x); k2 = f(x + 0.5 * h * k1); k3 = f(x + 0.5 * h * k2); k4 = f(x + h * k3)
    return x + (h / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4)


def vdp_rk4(x, u, h=0.05):
    """vanderpol_RBF.py:113-118."""
    f = lambda s: np.array([2.0 * s[1], 2.0 * s[1] - 10.0 * s[0] ** 2 * s[1] - 0.8 * s[0] + u])
    k1 = f(x); k2 = f(x + 0.5 * h * k1); k3 = f(x + 0.5 * h * k2); k4 = f(x + h * k3)
    return x + (h / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4)


def offline_data(plant=duffing_rk4, n_steps=100, n_traj=100, seed=101):
    """The reference's training set (data_generate.py:17-79): 100 trajectories x 100 RK4 steps, u, x0 ~ U[-2,2]."""
    rng = np.random.RandomState(seed)
    U0 = 4.0 * rng.rand(n_steps, n_traj) - 2.0
    x = 4.0 * rng.rand(2, n_traj) - 2.0
    Xs, Ys, Us = [], [], []
    for i in range(n_steps):
        xn = plant(x, U0[i])
        Xs.append(x); Ys.append(xn); Us.append(U0[i])
        x = xn
    return np.concatenate(Xs, 1), np.concatenate(Ys, 1), np.concatenate(Us, 0)


def offline_edmd(lift, plant=duffing_rk4, n_steps=100, n_traj=100, seed=101):
    """The reference's one-off fit (duffing.py:152-177, data_generate.py:17-79): 100 x 100 random-input
    samples, K = PHIY pinv([PHIX; U]), C = X pinv(PHIX).  `lift` maps (n,M) -> (L,M) (the HIP Encoder)."""
    rng = np.random.RandomState(seed)
    U0 = 4.0 * rng.rand(n_steps, n_traj) - 2.0
    x = 4.0 * rng.rand(2, n_traj) - 2.0
    Xs, Ys, Us = [], [], []
    for i in range(n_steps):
        xn = plant(x, U0[i])
        Xs.append(x); Ys.append(xn); Us.append(U0[i][None, :])
        x = xn
    X, Y, U = np.concatenate(Xs, 1), np.concatenate(Ys, 1), np.concatenate(Us, 1)
    PX, PY = np.asarray(lift(X)), np.asarray(lift(Y))
    K = PY @ np.linalg.pinv(np.concatenate([PX, U], 0))
    C = X @ np.linalg.pinv(PX)
    return K[:, :-1].copy(), K[:, -1:].copy(), C


def tank_map(x, u, switched=False):
    """Cascaded tanks, one sample: Tank_System.m:9-10 (nominal), :194-195 (after step 100), clipped at 0 (:211)."""
    x1 = np.maximum(x[0], 0.0)
    x2 = np.maximum(x[1], 0.0)
    if switched:
        xn = np.stack([x1 - 0.53 * np.sqrt(x1) + 0.3 * u, x2 + 0.1 * np.sqrt(x1) - 0.35 * np.sqrt(x2)])
    else:
        xn = np.stack([x1 - 0.5 * np.sqrt(x1) + 0.4 * u, x2 + 0.2 * np.sqrt(x1) - 0.3 * np.sqrt(x2)])
    return np.maximum(xn, 0.0)


def tank_offline_data(n_steps=100, n_traj=100, seed=101):
    """Tank_System.m:29-49: random inputs in [-5, 5], initial levels in [0, 2]."""
    rng = np.random.RandomState(seed)
    U0 = 10.0 * rng.rand(n_steps, n_traj) - 5.0
    x = np.maximum(4.0 * rng.rand(2, n_traj) - 2.0, 0.0)
    Xs, Ys, Us = [], [], []
    for i in range(n_steps):
        xn = tank_map(x, U0[i])
        Xs.append(x); Ys.append(xn); Us.append(U0[i])
        x = xn
    return np.concatenate(Xs, 1), np.concatenate(Ys, 1), np.concatenate(Us, 0)
