#!/usr/bin/env python3
"""Dev tool: BASELINE cfg4-like closed loop on one GPU's share (cascaded tanks, 32-dim two-layer MLP lift, N = 40,
delta-u form with output = second tank level, ONE model for the batch from pooled Gram sums -- the block that is
all-reduced over ranks in the multi-GPU job --, 65536 / 8 = 8192 trajectories per GPU): steps/s and QP status.
    python tools/cfg4_timing.py [B] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "koopman-online-updated-mpc_amd"))
sys.path.insert(0, ROOT)
import numpy as np, torch
from koopmpc import KoopmanMPC
from koopmpc.synth import random_mlp_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
L, N = 32, 40
rng = np.random.RandomState(0)
w = random_mlp_weights(2, 100, 2, L, seed=9)
m = KoopmanMPC(n=2, L=L, N=N, batch=B, weights=w, layers=2, lb=-0.5, ub=0.5, umin=-8.0, umax=8.0, Qw=10.0, Rw=1e-3, P0=1e4,
               barQ0=1e4, delta_u=True, out_row0=1, out_rows=1)


def tank(x, u):  # Tank_System.m:9-10 with clipping (:211)
    x1 = np.maximum(x[0], 0.0); x2 = np.maximum(x[1], 0.0)
    return np.maximum(np.stack([x1 - 0.5 * np.sqrt(x1) + 0.4 * u, x2 + 0.2 * np.sqrt(x1) - 0.3 * np.sqrt(x2)]), 0.0)


# offline data and fit as Tank_System.m:29-49, 87-100 (on the device)
Ub = 10 * rng.rand(100, 100) - 5
xc = np.maximum(4 * rng.rand(2, 100) - 2, 0.0)
Xs, Ys, Us = [], [], []
for i in range(100):
    xn = tank(xc, Ub[i]); Xs.append(xc); Ys.append(xn); Us.append(Ub[i]); xc = xn
m.offline_fit(np.concatenate(Xs, 1), np.concatenate(Ys, 1), np.concatenate(Us, 0), ridge=1e-9)
r = torch.ones(1, N, dtype=torch.float64, device="cuda:0")  # (a device tensor: a NumPy reference is copied to the device at every step)
X = torch.tensor(np.abs(rng.rand(2, B)), dtype=torch.float64, device="cuda:0")


def run(k0, n):
    global X
    for k in range(n):
        u = m.shared_step(X, r)
        X = m.plant_step("tank", X, u, switched=(k0 + k > 100))


run(0, 40); torch.cuda.synchronize()  # (the first ~20 steps of a fresh shared model are slow: few samples, ill-conditioned QPs)
t0 = time.perf_counter()
run(40, steps); torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = m.status.cpu().numpy()
print("cfg4-like (shared model, delta-u tank) L=%d N=%d B=%d: %.2f M steps/s (%.1f us/step), last-step status!=0: %d of %d, finite %s, level median %.3f"
      % (L, N, B, B * steps / dt / 1e6, dt / steps * 1e6, int((st != 0).sum()), B, bool(torch.isfinite(X).all()), float(X[1].median())))
