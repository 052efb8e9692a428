"""Dev tool: distribution of the Newton solves per trajectory over a 20-step launch, cold start against warm start."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import KoopmanMPC
from koopmpc.synth import random_mlp_weights, initial_states, offline_data
B, L, N = 4096, 20, 20
r = np.tile(np.array([[1.0], [0.0]]), (1, N))
for cold in (False, True):
    m = KoopmanMPC(n=2, L=L, N=N, batch=B, weights=random_mlp_weights(2, 100, 3, L), cold_start=cold)
    m.offline_fit(*offline_data())
    X = torch.tensor(initial_states(B), dtype=torch.float64, device="cuda:0").contiguous()
    m.rollout("duffing", X, r, 200, step0=0)
    tot = []
    for i in range(3):
        m.iters.zero_()
        m.rollout("duffing", X, r, 20, step0=200 + 20 * i)
        torch.cuda.synchronize()
        tot.append(m.iters.cpu().numpy().astype(float) / 20)
    import time
    for steps in (200, 20):
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(3):
            m.rollout("duffing", X, r, steps, step0=400)
        torch.cuda.synchronize()
        print("   %d-step launches: %.2f us per step" % (steps, (time.time() - t0) / 3 / steps * 1e6))
    m.step(X.cpu().numpy(), r)
    U = m.Useq.cpu().numpy()
    nact = (np.abs(U) >= 2.0 - 1e-9).sum(0)
    t = tot[-1]
    print("cold" if cold else "warm", "solves per step: mean %.3f  quantiles 50/90/99/99.9/max %.2f %.2f %.2f %.2f %.2f" % ((t.mean(),) + tuple(np.percentile(t, [50, 90, 99, 99.9, 100]))))
    print("   corr between consecutive launches %.2f; trajectories above 2 per step: %d; active inputs at the end: mean %.2f, share with any %.3f" % (
        np.corrcoef(tot[-2], t)[0, 1], (t > 2).sum(), nact.mean(), (nact > 0).mean()))
    for lo, hi in ((0, 1), (1, 3), (3, 8), (8, 21)):
        sel = (nact >= lo) & (nact < hi)
        if sel.any():
            print("   %2d..%2d active inputs: %5d trajectories, %.2f solves per step" % (lo, hi - 1, sel.sum(), t[sel].mean()))
