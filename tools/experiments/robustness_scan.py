#!/usr/bin/env python3
"""Dev tool: long closed-loop rollouts at several dimension sets; reports worst QP status, iteration statistics
and finiteness (the bench only covers 220 steps of cfg2)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import KoopmanMPC
from koopmpc.synth import initial_states, offline_data, random_mlp_weights
cold_arg = len(sys.argv) > 1 and sys.argv[1] == "cold"  # every solve from clip(0)
for (L, N, B, steps, layers) in [(20, 20, 4096, 1000, 3), (8, 10, 4096, 1000, 3), (8, 30, 4096, 300, 3), (32, 40, 2048, 300, 2), (64, 50, 512, 150, 3)]:
    w = random_mlp_weights(2, 100, layers, L)
    mpc = KoopmanMPC(n=2, L=L, N=N, batch=B, weights=w, layers=layers, cold_start=cold_arg)
    mpc.offline_fit(*offline_data())
    X = torch.tensor(initial_states(B), dtype=torch.float64, device="cuda:0")
    r = torch.tensor(np.tile(np.array([[1.0], [0.0]]), (1, N)), dtype=torch.float64, device="cuda:0")
    t0 = time.perf_counter()
    mpc.rollout("duffing", X, r, steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = mpc.status.cpu().numpy(); it = mpc.iters.cpu().numpy()
    print("L=%d N=%d B=%d steps=%d: worst status %d (n!=0: %d), Newton solves/step mean %.2f, worst trajectory mean %.1f, finite %s, |x1-1| median %.3f, %.1f us/step"
          % (L, N, B, steps, st.max(), int((st != 0).sum()), it.mean() / steps, it.max() / steps, bool(torch.isfinite(X).all()),
             float((X[0] - 1).abs().median()), dt / steps * 1e6))
