"""Dev tool: is a 20-step launch slower per step than a 200-step one because of the launch itself, or because of the pauses around it?
Back-to-back 20-step launches (no synchronisation in between) against synchronised ones and against 200-step launches."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import KoopmanMPC
from koopmpc.synth import random_mlp_weights, initial_states, offline_data
B, L, N = 4096, 20, 20
r = np.tile(np.array([[1.0], [0.0]]), (1, N))
m = KoopmanMPC(n=2, L=L, N=N, batch=B, weights=random_mlp_weights(2, 100, 3, L))
m.offline_fit(*offline_data())
X = torch.tensor(initial_states(B), dtype=torch.float64, device="cuda:0").contiguous()
m.rollout("duffing", X, r, 200, step0=0)
torch.cuda.synchronize()
def run(steps, n, sync):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.time(); e0.record()
    for i in range(n):
        m.rollout("duffing", X, r, steps, step0=400)
        if sync:
            torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    return (time.time() - t0) / n / steps * 1e6, e0.elapsed_time(e1) / n / steps * 1e3
for _ in range(2):
    for steps, n, sync in ((200, 10, False), (20, 100, False), (20, 100, True), (20, 1, True), (200, 1, True)):
        w, g = run(steps, n, sync)
        print("%3d-step launches x %3d, %s: %.2f us per step (wall), %.2f (GPU events)" % (steps, n, "synchronised" if sync else "back to back", w, g))
# how long may the GPU idle before a single launch pays for it?
def one(gap_us):
    for _ in range(5):
        m.rollout("duffing", X, r, 20, step0=400)
    torch.cuda.synchronize()
    t_end = time.perf_counter() + gap_us * 1e-6
    while time.perf_counter() < t_end:
        pass
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    m.rollout("duffing", X, r, 20, step0=400)
    e1.record(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e6, e0.elapsed_time(e1) / 20 * 1e3
for gap in (0, 20, 50, 100, 200, 500, 1000, 5000, 20000, 100000):
    rs = [one(gap) for _ in range(5)]
    print("idle %6d us before one 20-step launch: %.2f us per step wall, %.2f GPU events (median of 5)" % (gap, np.median([a for a, _ in rs]), np.median([b for _, b in rs])))
