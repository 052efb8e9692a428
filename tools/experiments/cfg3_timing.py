#!/usr/bin/env python3
"""Dev tool: BASELINE cfg3-like closed loop (Van der Pol, thin-plate RBF lift with 8 centres, tracking in the lifted
space y = psi, N = 30, box +-6) -- steps/s and QP status over a roll-out.  python tools/cfg3_timing.py [B] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import KoopmanMPC
from koopmpc.synth import initial_states, offline_data, vdp_rk4

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
out = sys.argv[3] if len(sys.argv) > 3 else "lift"
storage = not (len(sys.argv) > 4 and sys.argv[4] == "restart")  # restart: first update from K_A = 0 (duffing.py)
L, N = 8, 30
rng = np.random.RandomState(0)
X, Y, U = offline_data(plant=vdp_rk4)
# centres: a k-means-free stand-in (vanderpol_RBF.py:44-46 uses KMeans on the data): 8 data points
cx = X[:, rng.choice(X.shape[1], L, replace=False)].T.copy()
m = KoopmanMPC(n=2, L=L, N=N, batch=B, lift="rbf", centres=cx, output=out, lb=-6.0, ub=6.0, P0=1e5, barQ0=1e5)
m.offline_fit(X, Y, U, ridge=1e-9, init_rls=storage)  # storage semantics of vanderpol_RBF.py:434-438
lr = m.rbf(np.array([[1.0], [0.0]]))
lr = lr.cpu().numpy() if hasattr(lr, "cpu") else np.asarray(lr)
r = np.tile(np.reshape(lr, (L, 1)), (1, N)) if out == "lift" else np.tile(np.array([[1.0], [0.0]]), (1, N))
Xd = torch.tensor(initial_states(B), dtype=torch.float64, device="cuda:0")
print("fused:", m.rollout_is_fused())
t0 = time.time()
while time.time() - t0 < 1.0:
    m.rollout("vdp", Xd, r, 20)
    torch.cuda.synchronize()
m.reset(); m.offline_fit(X, Y, U, ridge=1e-9, init_rls=storage)
Xd.copy_(torch.tensor(initial_states(B), dtype=torch.float64, device="cuda:0"))
m.rollout("vdp", Xd, r, 20)
torch.cuda.synchronize()
t0 = time.perf_counter()
m.rollout("vdp", Xd, r, steps, step0=20)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = m.status.cpu().numpy(); it = m.iters.cpu().numpy()
print("cfg3-like L=%d N=%d B=%d y=%s: %.2f M steps/s (%.1f us/step), status!=0: %d of %d, Newton solves/step mean %.2f, finite %s, |x1-1| median %.3f"
      % (L, N, B, out, B * steps / dt / 1e6, dt / steps * 1e6, int((st != 0).sum()), B, it.mean() / steps, bool(torch.isfinite(Xd).all()),
         float((Xd[0] - 1).abs().median())))
