"""Dev tool (GPU box): the barrier-free roll-out (rollout_dyn.h) against the lock-step kernel on the same inputs."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import KoopmanMPC, _ffi
from koopmpc.synth import random_mlp_weights, initial_states, offline_data
lib = _ffi.load()
L = N = 20
r = np.tile(np.array([[1.0], [0.0]]), (1, N))
w = random_mlp_weights(2, 100, 3, L, seed=2024)
def run(B, steps, group, timeout=300, log=True):
    lib.kmpc_set_rollout_schedule(group, timeout)
    m = KoopmanMPC(n=2, L=L, N=N, batch=B, weights=w)
    m.offline_fit(*offline_data())
    X = torch.tensor(initial_states(B, seed=101), dtype=torch.float64, device="cuda:0").contiguous()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = m.rollout("duffing", X, r, steps, log=log)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return X.cpu().numpy(), (out[0].cpu().numpy() if log else None), m.status.cpu().numpy(), m.iters.cpu().numpy(), dt, m.get_model()[0].cpu().numpy()
for B, steps in ((100, 12), (4096, 40)):
    X0, U0, s0, i0, t0, A0 = run(B, steps, 0)
    for G in (4, 2, 1, 3):
        X1, U1, s1, i1, t1, A1 = run(B, steps, G)
        print("B=%d steps=%d G=%d: max|dU| %.2e max|dX| %.2e max|dA| %.2e status %d/%d iters equal %s  time %.2f vs %.2f ms" % (
            B, steps, G, np.abs(U1 - U0).max(), np.abs(X1 - X0).max(), np.abs(A1 - A0).max(), s0.max(), s1.max(), (i0 == i1).mean(), t0 * 1e3, t1 * 1e3), flush=True)
