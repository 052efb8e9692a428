#!/usr/bin/env python3
"""Dev tool: per-step duration of the step kernel and Newton-iteration statistics along the bench rollout."""
import os, sys
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
mpc.profile(True)
rows = []
for i in range(int(os.environ.get("STEPS", 260))):
    mpc.rollout("duffing", X, r, 1, step0=i)
    torch.cuda.synchronize()
    pr = mpc.profile_read()
    it = mpc.iters.cpu().numpy()
    rows.append((i, pr["step_ms"] * 1e3, pr["lift_ms"] * 1e3, it.mean(), it.max(), int((it >= 5).sum())))
for rr in rows:
    if rr[0] < 12 or rr[0] % 10 == 0 or 100 <= rr[0] <= 112:
        print("step %3d  step_kernel %7.1f us  lift %5.1f us  iters mean %.2f max %2d  n(it>=5) %d" % rr)
a = np.array(rows)
print("mean step kernel us: steps 20-219: %.1f ; 220-259: %.1f" % (a[20:220, 1].mean(), a[220:, 1].mean()))
