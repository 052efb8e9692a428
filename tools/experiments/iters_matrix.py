"""Dev tool (GPU box): per-step Newton-solve counts of every trajectory of the bench workload (cfg2), one launch per
step, saved as gpurun_out/iters_matrix.npz (steps x B) -- input of tools/sched_sim.py."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import KoopmanMPC
from koopmpc.synth import random_mlp_weights, initial_states, offline_data

B, steps = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 230
L = N = 20
m = KoopmanMPC(n=2, L=L, N=N, batch=B, weights=random_mlp_weights(2, 100, 3, L, seed=2024))
m.offline_fit(*offline_data())
r = np.tile(np.array([[1.0], [0.0]]), (1, N))
X = torch.tensor(initial_states(B, seed=101), dtype=torch.float64, device="cuda:0").contiguous()
out = np.zeros((steps, B), dtype=np.int16)
for k in range(steps):
    m.rollout("duffing", X, r, 1, step0=k)
    out[k] = m.iters.cpu().numpy()
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/iters_matrix.npz", iters=out)
print("mean per step:", out.mean(1)[:30].round(2).tolist(), "...", out.mean(), "max", out.max())
