#!/usr/bin/env python3
"""Dev tool: host enqueue time vs GPU time of kmpc_rollout."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import KoopmanMPC
from koopmpc.synth import initial_states, offline_edmd, random_mlp_weights
L, N, B = 20, 20, int(os.environ.get("B", 4096))
dev = torch.device("cuda:0")
w = random_mlp_weights(2, 100, 3, L)
mpc = KoopmanMPC(n=2, L=L, N=N, batch=B, weights=w)
A0, B0, C0 = offline_edmd(lambda X: mpc.Encoder(X))
mpc.set_model(A0, B0, C0)
X = torch.tensor(initial_states(B), dtype=torch.float64, device=dev)
r = torch.tensor(np.tile(np.array([[1.0], [0.0]]), (1, N)), dtype=torch.float64, device=dev)
mpc.rollout("duffing", X, r, 20); torch.cuda.synchronize()
for steps in (50, 200):
    t0 = time.perf_counter(); mpc.rollout("duffing", X, r, steps, step0=20); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("steps %d: enqueue %.1f us/step, total %.1f us/step" % (steps, (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6))
# python-level loop for comparison
t0 = time.perf_counter()
for i in range(100):
    u = mpc.step(X, r); mpc.plant_step("duffing", X, u)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("python loop: enqueue %.1f us/step, total %.1f us/step" % ((t1 - t0) / 100 * 1e6, (t2 - t0) / 100 * 1e6))
