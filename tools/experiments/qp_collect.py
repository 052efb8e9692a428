"""Dev tool (GPU box): the box QPs of the bench workload's first steps as data -- H, f, warm start, Newton solves --
for tools/qp_study.py (host-side study of the projected-Newton rules).  Per-step launches through the public API."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import KoopmanMPC
from koopmpc.synth import random_mlp_weights, initial_states, offline_data
B, s0, s1 = 4096, int(sys.argv[1]), int(sys.argv[2])
keep = 512
L = N = 20
m = KoopmanMPC(n=2, L=L, N=N, batch=B, weights=random_mlp_weights(2, 100, 3, L, seed=2024))
m.offline_fit(*offline_data())
r = np.tile(np.array([[1.0], [0.0]]), (1, N))
X = torch.tensor(initial_states(B, seed=101), dtype=torch.float64, device="cuda:0").contiguous()
Hs, fs, ws, its, us = [], [], [], [], []
warm = torch.zeros(N, B, dtype=torch.float64, device="cuda:0")
for k in range(s1):
    Xk = X.clone()
    u = m.step(X, r).clone()
    if k >= s0:
        psi = m.Encoder(Xk)
        H, f = m.condense(psi, r)   # the model the solve of step k used (updated with the transition that ended in x_k)
        Hs.append(H[:keep].cpu().numpy()); fs.append(f[:keep].cpu().numpy()); ws.append(warm[:, :keep].cpu().numpy().T.copy())
        its.append(m.iters[:keep].cpu().numpy().copy()); us.append(m.Useq[:, :keep].cpu().numpy().T.copy())
    warm = m.Useq.clone()
    X = m.plant_step("duffing", X, u, switched=(k >= 102))
np.savez_compressed("gpurun_out/qp_data_%d_%d.npz" % (s0, s1), H=np.array(Hs), f=np.array(fs), warm=np.array(ws), iters=np.array(its), U=np.array(us))
print("saved", np.array(its).mean(), np.array(its).max())
