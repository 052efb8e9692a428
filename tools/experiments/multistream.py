"""Experiment: the 4096-trajectory rollout as G independent sub-batches on G HIP streams (phases of different
sub-batches overlap on the CUs) vs one launch per step.  python tools/multistream.py [B] [steps]"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import KoopmanMPC
from koopmpc.synth import random_mlp_weights, initial_states, offline_data

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
L, N = 20, 20
w = random_mlp_weights(2, 100, 3, L)
Xo, Yo, Uo = offline_data()
r = np.tile(np.array([[1.0], [0.0]]), (1, N))
dev = torch.device("cuda:0")

def run(G, skew):
    bs = B // G
    ctrls, Xs, streams = [], [], []
    X0 = initial_states(B)
    for g in range(G):
        m = KoopmanMPC(n=2, L=L, N=N, batch=bs, weights=w)
        m.offline_fit(Xo, Yo, Uo)
        ctrls.append(m)
        Xs.append(torch.tensor(X0[:, g * bs:(g + 1) * bs].copy(), dtype=torch.float64, device=dev).contiguous())
        streams.append(torch.cuda.Stream(device=dev))
    torch.cuda.synchronize()
    # spin-up + warm-up
    t0 = time.time()
    while time.time() - t0 < 1.0:
        for g in range(G):
            with torch.cuda.stream(streams[g]):
                ctrls[g].rollout("duffing", Xs[g], r, 5, step0=0, switch_step=10**9)
        torch.cuda.synchronize()
    for g in range(G):
        ctrls[g].reset()
        Xs[g].copy_(torch.tensor(X0[:, g * bs:(g + 1) * bs].copy(), dtype=torch.float64, device=dev))
    torch.cuda.synchronize()
    for g in range(G):
        with torch.cuda.stream(streams[g]):
            ctrls[g].rollout("duffing", Xs[g], r, 20, step0=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if skew:
        # enqueue step by step, round-robin, so that the streams stay interleaved in submission order
        for k in range(0, steps, skew):
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    ctrls[g].rollout("duffing", Xs[g], r, min(skew, steps - k), step0=20 + k)
    else:
        for g in range(G):
            with torch.cuda.stream(streams[g]):
                ctrls[g].rollout("duffing", Xs[g], r, steps, step0=20)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = max(int(c.status.max().item()) for c in ctrls)
    xf = torch.cat(Xs, 1)
    print("G=%d skew=%d: %.1f us/step  %.2f M steps/s  status %d  checksum %.12f" % (G, skew, dt / steps * 1e6, B * steps / dt / 1e6, st, float(xf.sum())), flush=True)

for G, skew in ((1, 0), (2, 0), (2, 10), (4, 0), (4, 10), (8, 0), (1, 0)):
    run(G, skew)
