"""Dev tool: Van der Pol tracking in the lifted space (vanderpol.py constants) from 256 random initial states, QP status
and state finiteness every 10 steps.  A few trajectories leave the stability region of the explicit RK4 plant
(10 x1^2 h < 2.78, i.e. |x1| < 2.36 at h = 0.05) and blow up -- in the plant, not in the controller: their status
becomes 2 (non-finite problem data) and the input handed back is the feasible start."""
import sys, os
sys.path.insert(0, "/root/repo/koopman-online-updated-mpc_amd"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
from koopmpc import KoopmanMPC
from koopmpc.synth import initial_states, offline_data, vdp_rk4
d = np.load("/root/repo/tests/golden/weights_vdp.npz")
w = [(d["W%d" % k], d["b%d" % k].reshape(-1)) for k in range(1, 5)]
B, N = 256, 10
m = KoopmanMPC(n=2, L=8, N=N, batch=B, weights=w, output="lift", lb=-6.0, ub=6.0, P0=1e5, barQ0=1e5)
m.offline_fit(*offline_data(plant=vdp_rk4))
goal = np.asarray(m.Encoder(np.array([[1.0], [0.0]])))
r = np.tile(goal.reshape(8, 1), (1, N))
X = torch.tensor(initial_states(B), dtype=torch.float64, device="cuda:0")
for chunk in range(15):
    m.rollout("vdp", X, r, 10, step0=chunk * 10, switch_step=102)
    st = m.status.cpu().numpy()
    print(chunk, "status counts", np.bincount(st, minlength=4), "finite X", bool(torch.isfinite(X).all()), "max|x|", float(X.abs().max()))
