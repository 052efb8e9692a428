"""Dev tool (GPU box): the shared-model QP of cfg4 around the plant switch as data (model, lifted state, u_prev, warm start,
solution, Newton solves) for a host-side study."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "koopman-online-updated-mpc_amd"))
import numpy as np, torch
import bench
c = bench.CONFIGS["cfg4"]; w = bench.workload_inputs("cfg4", c["L"], c["N"])
loop = bench.Loop("cfg4", w, 256, torch.float64, torch.device("cuda", 0), 0)
s0, s1 = 98, 125
loop.advance(s0, 0)
rec = {k: [] for k in ("A", "B", "C", "psi", "uprev", "warm", "U", "iters", "x")}
uprev = None
for k in range(s0, s1):
    warm = loop.m.Useq[:, 0].cpu().numpy().copy()
    x = loop.X[:, 0].cpu().numpy().copy()
    psi = loop.m.Encoder(loop.X[:, :1].contiguous().cpu().numpy())
    up = loop.m.U0[0].item()   # u applied at the previous step
    loop.advance(1, k)
    A, Bm, C = [t.cpu().numpy() for t in loop.m.shared_model()]
    for key, v in (("A", A), ("B", Bm), ("C", C), ("psi", np.asarray(psi).reshape(-1)), ("uprev", up), ("warm", warm),
                   ("U", loop.m.Useq[:, 0].cpu().numpy().copy()), ("iters", loop.m.iters[0].item()), ("x", x)):
        rec[key].append(v)
np.savez_compressed("gpurun_out/cfg4_qps.npz", **{k: np.array(v) for k, v in rec.items()})
print("iters", rec["iters"])
