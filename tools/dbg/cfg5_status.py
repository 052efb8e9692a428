#!/usr/bin/env python3
"""Which cfg5 trajectories leave a QP with status != 0 in the first steps after the reset, and how far their inputs are from the exact
minimiser of the QP the device built (model exported per trajectory).  python tools/dbg/cfg5_status.py [steps] [switch_step]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "koopman-online-updated-mpc_amd")]
import numpy as np
import torch

import bench
from oracle import koopman_oracle as ko

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 14
sw = int(sys.argv[2]) if len(sys.argv) > 2 else 8
name = "cfg5"
c = bench.CONFIGS[name]
w = bench.workload_inputs(name, c["L"], c["N"])
B = int(os.environ.get("B", c["B"]))
loop = bench.Loop(name, w, B, torch.float64, torch.device("cuda", 0), 0)
L, N = c["L"], c["N"]
lift_fn = lambda x: ko.mlp_lift(w["weights"], x)
r = w["ref"]
for k in range(steps):
    X = loop.X.cpu().numpy().copy()
    Ul, _ = loop.m.rollout(c["plant"], loop.X, loop.r, 1, step0=k, switch_step=sw, log=True)
    st = loop.m.status.cpu().numpy()
    it = loop.m.iters.cpu().numpy()
    bad = np.nonzero(st != 0)[0]
    print("step %2d: status != 0 on %d trajectories %s; iters mean %.2f max %d" % (k, len(bad), bad[:8].tolist(), it.mean(), it.max()), flush=True)
    if len(bad):
        A, Bm, Cm = [t.cpu().numpy() for t in loop.m.get_model()]
        Useq = None
        for b in bad[:6]:
            psi = lift_fn(X[:, b:b + 1]).reshape(-1)
            _, _, H, f, _ = ko.condense(A[b], Bm[b].reshape(L, 1), Cm[b], psi, r, N, 100.0, 1e-4)
            ev = np.linalg.eigvalsh(H)
            try:
                U, _ = ko.qp_exact(H, f, c["lb"], c["ub"])
                print("   b=%d status %d iters %d: u_gpu %.9f u_exact %.9f  diff %.2e  eig(H) %.2e .. %.2e" % (b, st[b], it[b], Ul[0, b].item(), U[0], abs(Ul[0, b].item() - U[0]), ev[0], ev[-1]))
            except Exception as e:
                print("   b=%d status %d: qp_exact failed %s, eig(H) %.2e .. %.2e" % (b, st[b], e, ev[0], ev[-1]))
        del A, Bm, Cm
