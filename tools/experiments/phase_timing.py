#!/usr/bin/env python3
"""Dev tool: time the phases of the step kernel separately (same kernel, phase masks) on the
bench workload after a closed-loop warm-up, and print the Newton-iteration histogram."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import KoopmanMPC
from koopmpc.synth import initial_states, offline_edmd, random_mlp_weights

L, N, B = int(os.environ.get("L", 20)), int(os.environ.get("N", 20)), int(os.environ.get("B", 4096))
dev = torch.device("cuda:0")
w = random_mlp_weights(2, 100, 3, L)
mpc = KoopmanMPC(n=2, L=L, N=N, batch=B, weights=w, threads=int(os.environ.get("THREADS", 0)))
A0, B0, C0 = offline_edmd(lambda X: mpc.Encoder(X))
mpc.set_model(A0, B0, C0)
X = torch.tensor(initial_states(B), dtype=torch.float64, device=dev)
r = torch.tensor(np.tile(np.array([[1.0], [0.0]]), (1, N)), dtype=torch.float64, device=dev)
warm = int(os.environ.get("WARM", 60))
mpc.rollout("duffing", X, r, warm)
torch.cuda.synchronize()

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

psi = mpc.Encoder(X)
u = mpc.step(X, r).clone()
Xn = mpc.plant_step("duffing", X.clone(), u)
psin = mpc.Encoder(Xn)
H, f = mpc.condense(psin, r)
U, st, it = mpc.qp_solve(H, f)
itc = it.cpu().numpy()
print("iters hist:", np.bincount(itc), "mean %.2f max %d" % (itc.mean(), itc.max()))
sd = mpc.state_dict()
print("lift      %8.1f us" % timeit(lambda: mpc.Encoder(X)))
print("rls       %8.1f us" % timeit(lambda: mpc.lib.kmpc_rls_update(mpc.h, mpc._p(psi), mpc._p(u), mpc._p(psin), mpc._p(Xn), B, mpc._stream())))
mpc.load_state_dict(sd)
Hb = torch.empty_like(H); fb = torch.empty_like(f)
print("condense  %8.1f us" % timeit(lambda: mpc.lib.kmpc_condense(mpc.h, mpc._p(psin), mpc._p(r), 0, mpc._p(Hb), mpc._p(fb), B, mpc._stream())))
Ub = torch.empty_like(U); stb = torch.empty_like(st); itb = torch.empty_like(it)
print("qp        %8.1f us" % timeit(lambda: mpc.lib.kmpc_qp_solve(mpc.h, mpc._p(H), mpc._p(f), mpc._p(Ub), mpc._p(stb), mpc._p(itb), B, mpc._stream())))
# qp restricted to easy problems (<=1 iteration) and to hard ones
for name, sel in (("qp it<=1", it <= 1), ("qp it>=5", it >= 5)):
    idx = torch.nonzero(sel).flatten()
    if idx.numel() == 0: continue
    reps = (B + idx.numel() - 1) // idx.numel()
    idx = idx.repeat(reps)[:B]
    Hs, fs = H[idx].contiguous(), f[idx].contiguous()
    print("%-9s %8.1f us  (B=%d replicated)" % (name, timeit(lambda: mpc.lib.kmpc_qp_solve(mpc.h, mpc._p(Hs), mpc._p(fs), mpc._p(Ub), mpc._p(stb), mpc._p(itb), B, mpc._stream())), B))
    print("          mean iters %.2f max %d" % (itb.double().mean().item(), itb.max().item()))
