import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import KoopmanMPC
from koopmpc.synth import initial_states, offline_data
B, steps, L, N = 4096, 200, 20, 20
rng = np.random.RandomState(0)
X, Y, U = offline_data()
cx = X[:, rng.choice(X.shape[1], L, replace=False)].T.copy()
m = KoopmanMPC(n=2, L=L, N=N, batch=B, lift="rbf", centres=cx)
m.offline_fit(X, Y, U, ridge=1e-9)
r = np.tile(np.array([[1.0], [0.0]]), (1, N))
Xd = torch.tensor(initial_states(B), dtype=torch.float64, device="cuda:0")
print("fused:", m.rollout_is_fused())
t0 = time.time()
while time.time() - t0 < 1.0:
    m.rollout("duffing", Xd, r, 20); torch.cuda.synchronize()
m.reset(); m.offline_fit(X, Y, U, ridge=1e-9)
Xd.copy_(torch.tensor(initial_states(B), dtype=torch.float64, device="cuda:0"))
m.rollout("duffing", Xd, r, 20); torch.cuda.synchronize()
t0 = time.perf_counter()
m.rollout("duffing", Xd, r, steps, step0=20); torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = m.status.cpu().numpy(); it = m.iters.cpu().numpy()
print("rbf L=20 N=20 B=%d: %.2f M steps/s (%.1f us/step) status!=0 %d, newton/step %.2f" % (B, B*steps/dt/1e6, dt/steps*1e6, int((st!=0).sum()), it.mean()/steps))
