#!/usr/bin/env python3
"""Dev tool: find trajectories whose QP hits the iteration cap in a rollout and report cond(H) / KKT of the result."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "koopman-online-updated-mpc_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from koopmpc import KoopmanMPC
from koopmpc.synth import initial_states, offline_data, random_mlp_weights
from oracle import koopman_oracle as ko
L, N, B = int(os.environ.get("L", 8)), int(os.environ.get("N", 30)), int(os.environ.get("B", 4096))
w = random_mlp_weights(2, 100, 3, L)
mpc = KoopmanMPC(n=2, L=L, N=N, batch=B, weights=w)
mpc.offline_fit(*offline_data())
X = torch.tensor(initial_states(B), dtype=torch.float64, device="cuda:0")
r = torch.tensor(np.tile(np.array([[1.0], [0.0]]), (1, N)), dtype=torch.float64, device="cuda:0")
nbad = 0
for k in range(int(os.environ.get("STEPS", 80))):
    Xk = X.clone()
    u = mpc.step(X, r)
    st = mpc.status.cpu().numpy()
    bad = np.nonzero(st)[0]
    if len(bad):
        psi = mpc.Encoder(Xk)
        H, f = mpc.condense(psi, r)
        for b in bad[:3]:
            Hb, fb = H[b].cpu().numpy(), f[b].cpu().numpy()
            U = mpc.Useq[:, b].cpu().numpy()
            Uo, ito = ko.qp_exact(Hb, fb, -2, 2)
            print("step %d traj %d status %d iters %d cond(H) %.2e  KKT(gpu)/|f| %.1e  |U-Uo| %.1e  J gpu-oracle %.2e (|J| %.2e)" % (
                k, b, st[b], int(mpc.iters[b]), np.linalg.cond(Hb), ko.kkt_residual(Hb, fb, -2, 2, U) / np.abs(fb).max(),
                np.abs(U - Uo).max(), (U @ Hb @ U + fb @ U) - (Uo @ Hb @ Uo + fb @ Uo), abs(Uo @ Hb @ Uo + fb @ Uo)))
        if nbad < 6:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.savez(os.path.join(ROOT, "gpurun_out", "qpfail_%d.npz" % k), H=H[bad].cpu().numpy(), f=f[bad].cpu().numpy(), U=mpc.Useq[:, bad].cpu().numpy())
        nbad += len(bad)
    X = mpc.plant_step("duffing", X, u, switched=(k >= 102))
print("total status!=0:", nbad)
