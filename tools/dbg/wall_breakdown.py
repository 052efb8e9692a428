"""Dev tool (GPU box): where the wall time of a 20-step cfg2 call goes beyond its kernel -- host time until kmpc_rollout returns, time until
the stream is idle, kernel time from HIP events."""
import os as _os; _os.environ.setdefault("KMPC_DEBUG", "1")
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "koopman-online-updated-mpc_amd"))
import numpy as np, torch
import bench
c = bench.CONFIGS["cfg2"]; w = bench.workload_inputs("cfg2", c["L"], c["N"])
loop = bench.Loop("cfg2", w, 4096, torch.float64, torch.device("cuda", 0), 0)
loop.advance(200, 0); loop.advance(5, 200); torch.cuda.synchronize()
snap = (loop.m.state_to(), loop.X.clone())
res = []
for prof in (False, True):
    for rep in range(30):
        loop.m.state_from(snap[0]); loop.X.copy_(snap[1]); loop.advance(0, 205); torch.cuda.synchronize()
        if prof: loop.m.profile(True)
        t0 = time.perf_counter(); loop.advance(20, 205); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        k = loop.m.profile_read()["step_ms"] if prof else 0.0
        if prof: loop.m.profile(False)
        res.append((prof, (t1 - t0) * 1e6, (t2 - t0) * 1e6, k * 1e3))
for prof in (False, True):
    r = np.array([x[1:] for x in res if x[0] == prof][5:])
    print("profile events %s: call returns after %.1f us, stream idle after %.1f us (median of 25)%s" % (prof, np.median(r[:, 0]), np.median(r[:, 1]),
          ", kernel + place between events %.1f us" % np.median(r[:, 2]) if prof else ""))
