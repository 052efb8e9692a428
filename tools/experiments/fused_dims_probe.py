#!/usr/bin/env python3
"""Dev tool: fused roll-out vs per-step launches on the larger dimension sets (same controller, same states):
agreement of the closed loop and steps/s of both.  python tools/fused_dims_probe.py"""
import os, subprocess, sys, time
os.environ.setdefault("KMPC_DEBUG", "1")  # (the library reads KMPC_NO_FUSED_ROLLOUT and the other measurement switches only with this set)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import KoopmanMPC
from koopmpc.synth import initial_states, offline_data, random_mlp_weights

def run(L, N, B, steps, layers):
    w = random_mlp_weights(2, 100, layers, L)
    m = KoopmanMPC(n=2, L=L, N=N, batch=B, weights=w, layers=layers)
    m.offline_fit(*offline_data())
    X = torch.tensor(initial_states(B), dtype=torch.float64, device="cuda:0")
    r = torch.tensor(np.tile(np.array([[1.0], [0.0]]), (1, N)), dtype=torch.float64, device="cuda:0")
    m.rollout("duffing", X, r, 20); torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.rollout("duffing", X, r, steps, step0=20); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return m.rollout_is_fused(), B * steps / dt / 1e6, X.cpu().numpy(), int(m.status.max().item()), float(m.iters.double().mean().item()) / (steps + 20)

if __name__ == "__main__":
    if len(sys.argv) > 1:  # child: one configuration, prints a line and saves the final states
        L, N, B, steps, layers = map(int, sys.argv[1:6])
        fused, rate, X, st, it = run(L, N, B, steps, layers)
        np.save(sys.argv[6], X)
        print("L=%d N=%d B=%d layers=%d fused=%s: %.2f M steps/s, worst status %d, Newton/step %.2f" % (L, N, B, layers, fused, rate, st, it), flush=True)
    else:
        for cfg in ([(8, 30, 4096, int(os.environ["STEPS"]), 3)] if "STEPS" in os.environ else [(8, 30, 4096, 100, 3), (20, 30, 4096, 100, 3), (32, 40, 2048, 100, 2)]):
            outs = []
            for env in ({}, {"KMPC_NO_FUSED_ROLLOUT": "1"}):
                f = "/tmp/x_%d_%d_%d.npy" % (cfg[0], cfg[1], len(env))
                subprocess.run([sys.executable, __file__] + [str(c) for c in cfg] + [f], env=dict(os.environ, **env), check=True)
                outs.append(np.load(f))
            print("   fused vs per-step final states: max |dx| = %.2e" % np.abs(outs[0] - outs[1]).max(), flush=True)
