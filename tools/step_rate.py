"""Dev tool: a Python loop of kmpc_step + kmpc_plant_step at the bench batch (cfg2): microseconds per step; KMPC_STEP_TWO_KERNELS=1\nforces the lift kernel + step kernel route."""
import os as _os; _os.environ.setdefault("KMPC_DEBUG", "1")
import os, sys, time
sys.path.insert(0, "/root/repo/koopman-online-updated-mpc_amd")
import numpy as np, torch
from koopmpc import KoopmanMPC
from koopmpc.synth import initial_states, offline_data, random_mlp_weights
B, L, N = 4096, 20, 20
m = KoopmanMPC(n=2, L=L, N=N, batch=B, weights=random_mlp_weights(2, 100, 3, L))
m.offline_fit(*offline_data())
r = torch.tensor(np.tile(np.array([[1.0], [0.0]]), (1, N)), dtype=torch.float64, device="cuda:0")
X = torch.tensor(initial_states(B), dtype=torch.float64, device="cuda:0")
for k in range(60):
    u = m.step(X, r); X = m.plant_step("duffing", X, u, switched=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(200):
    u = m.step(X, r); X = m.plant_step("duffing", X, u, switched=False)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("kmpc_step + kmpc_plant_step loop from Python: %.1f us/step, %.1f M steps/s, status %d" % (dt / 200 * 1e6, B * 200 / dt / 1e6, int(m.status.max())))
