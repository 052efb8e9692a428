#!/usr/bin/env python3
"""Dev tool: set up the bench workload, then run ONE phase of the step kernel a few times (for rocprofv3 --pmc)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import KoopmanMPC
from koopmpc.synth import initial_states, offline_edmd, random_mlp_weights
phase = sys.argv[1] if len(sys.argv) > 1 else "step"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
L, N, B = int(os.environ.get("L", 20)), int(os.environ.get("N", 20)), int(os.environ.get("B", 4096))
dev = torch.device("cuda:0")
w = random_mlp_weights(2, 100, 3, L)
mpc = KoopmanMPC(n=2, L=L, N=N, batch=B, weights=w)
A0, B0, C0 = offline_edmd(lambda X: mpc.Encoder(X))
mpc.set_model(A0, B0, C0)
X = torch.tensor(initial_states(B), dtype=torch.float64, device=dev)
r = torch.tensor(np.tile(np.array([[1.0], [0.0]]), (1, N)), dtype=torch.float64, device=dev)
mpc.rollout("duffing", X, r, int(os.environ.get("WARM", 60)))
psi = mpc.Encoder(X); u = mpc.step(X, r).clone()
Xn = mpc.plant_step("duffing", X.clone(), u); psin = mpc.Encoder(Xn)
H, f = mpc.condense(psin, r)
U = torch.empty(N, B, dtype=torch.float64, device=dev); st = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty_like(st)
if os.environ.get("QP_EASY"):
    _, _, it0 = mpc.qp_solve(H, f)
    idx = torch.nonzero(it0 <= 1).flatten()
    idx = idx.repeat((B + idx.numel() - 1) // idx.numel())[:B]
    H, f = H[idx].contiguous(), f[idx].contiguous()
torch.cuda.synchronize()
print("MARK setup done")
for _ in range(reps):
    if phase == "condense":
        mpc.lib.kmpc_condense(mpc.h, mpc._p(psin), mpc._p(r), 0, mpc._p(H), mpc._p(f), B, mpc._stream())
    elif phase == "qp":
        mpc.lib.kmpc_qp_solve(mpc.h, mpc._p(H), mpc._p(f), mpc._p(U), mpc._p(st), mpc._p(it), B, mpc._stream())
    elif phase == "rls":
        mpc.lib.kmpc_rls_update(mpc.h, mpc._p(psi), mpc._p(u), mpc._p(psin), mpc._p(Xn), B, mpc._stream())
    else:
        mpc.step(Xn, r)
torch.cuda.synchronize()
