#!/usr/bin/env python3
"""Fused roll-out (plug-in) of a dimension set against the per-step launches of the same handle type and against the oracle.
python tools/dbg/plugin_sets_probe.py L N lift [B steps]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "koopman-online-updated-mpc_amd")]
import numpy as np

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    from koopmpc import KoopmanMPC
    from koopmpc.synth import duffing_rk4, initial_states, offline_edmd, random_mlp_weights
    L, N, lift, B, steps = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
    rng = np.random.RandomState(L * N)
    if lift == "mlp":
        w = random_mlp_weights(2, 100, 3, L, seed=5)
        m = KoopmanMPC(n=2, L=L, N=N, batch=B, weights=w)
    else:
        cx = 4 * rng.rand(L, 2) - 2
        m = KoopmanMPC(n=2, L=L, N=N, batch=B, lift="rbf", centres=cx)
    A0, B0, C0 = offline_edmd(lambda X: m.Encoder(X), plant=duffing_rk4)
    m.set_model(A0, B0, C0)
    r = np.tile(np.array([[1.0], [0.0]]), (1, N))
    X = torch.tensor(initial_states(B, seed=3), dtype=torch.float64, device="cuda:0").contiguous()
    U, Xl = m.rollout("duffing", X, r, steps, step0=99, switch_step=102, log=True)
    np.savez(sys.argv[7], U=U.cpu().numpy(), X=Xl.cpu().numpy(), st=m.status.cpu().numpy(), it=m.iters.cpu().numpy(), fused=m.rollout_is_fused())
    sys.exit(0)

L, N, lift = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
B = int(sys.argv[4]) if len(sys.argv) > 4 else 24
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 8
res = {}
for tag, env in (("fused", {}), ("steps", {"KMPC_NO_FUSED_ROLLOUT": "1"})):
    e = dict(os.environ, KMPC_DEBUG="1", **env)
    out = "/tmp/pp_%s.npz" % tag
    subprocess.check_call([sys.executable, __file__, "--child", str(L), str(N), lift, str(B), str(steps), out], env=e)
    res[tag] = np.load(out)
f, s = res["fused"], res["steps"]
print("(%d, %d, %s) B=%d steps=%d: fused=%s/%s  status fused %s steps %s" % (L, N, lift, B, steps, f["fused"], s["fused"], np.bincount(f["st"]), np.bincount(s["st"])))
d = np.abs(f["U"] - s["U"])
print("   max |u_fused - u_steps| per step:", ["%.1e" % v for v in d.max(1)])
print("   trajectories with status != 0 (fused):", np.nonzero(f["st"])[0][:10], " (steps):", np.nonzero(s["st"])[0][:10])
