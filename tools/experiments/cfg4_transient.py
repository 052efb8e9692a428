"""Dev tool (GPU box): cfg4 (shared model, delta-u tank) around the plant switch: per step QP time, Newton solves, status,
conditioning of the shared H."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "koopman-online-updated-mpc_amd"))
import numpy as np, torch
import bench
name = "cfg4"; c = bench.CONFIGS[name]
w = bench.workload_inputs(name, c["L"], c["N"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else c["B"]
loop = bench.Loop(name, w, B, torch.float64, torch.device("cuda", 0), 0)
s0, s1 = int(sys.argv[2]) if len(sys.argv) > 2 else 96, int(sys.argv[3]) if len(sys.argv) > 3 else 125
loop.advance(s0, 0)
torch.cuda.synchronize()
for k in range(s0, s1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); loop.advance(1, k); e1.record(); torch.cuda.synchronize()
    A, Bm, C = [t.cpu().numpy() for t in loop.m.shared_model()]
    it = loop.m.iters.cpu().numpy(); st = loop.m.status.cpu().numpy()
    ev = np.abs(np.linalg.eigvals(A))
    print("step %3d: %8.1f us  Newton solves mean %.1f max %d  status!=0: %d  |eig(A)| max %.3f  level median %.3f" % (
        k, e0.elapsed_time(e1) * 1e3, it.mean(), it.max(), int((st != 0).sum()), ev.max(), float(loop.X[1].median())), flush=True)
