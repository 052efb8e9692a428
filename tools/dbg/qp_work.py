"""Dev tool (GPU box, trace build): what the box QPs of a fused cfg2 roll-out do, per trajectory and launch -- sweeps, rebuilds of
the tableau, refinement passes, iterations -- and how the workgroups' finish times follow them.
    KMPC_TRACE_LIB=libkoopmpc_devtrace.so python tools/dbg/qp_work.py [steps] [cold]"""
import os as _os; _os.environ.setdefault("KMPC_DEBUG", "1")
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import _ffi
_ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), os.environ.get("KMPC_TRACE_LIB", "libkoopmpc_trace.so"))
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cold = len(sys.argv) > 2 and sys.argv[2] == "cold"
name = "cfg2"
c = bench.CONFIGS[name]; w = bench.workload_inputs(name, c["L"], c["N"])
B, G = 4096, 16
loop = bench.Loop(name, w, B, torch.float64, torch.device("cuda", 0), 0, cold=cold)
start = int(os.environ.get("KMPC_QPW_START", c["settle"] + 5))  # first step of the traced window, counted from the RLS reset
loop.advance(start - 5, 0); loop.advance(5, start - 5); torch.cuda.synchronize()
import time
t0 = time.time()
snap = (loop.m.state_to(), loop.X.clone())
while time.time() - t0 < 1.0:
    loop.m.state_from(snap[0]); loop.X.copy_(snap[1]); loop.advance(steps, start); torch.cuda.synchronize()
loop.m.state_from(snap[0]); loop.X.copy_(snap[1]); loop.m.iters.zero_()
loop.advance(steps, start); torch.cuda.synchronize()
lib = _ffi.load()
buf = np.zeros(8192 * 32, dtype=np.uint64)
lib.kmpc_trace_read.restype = C.c_int; lib.kmpc_trace_read.argtypes = [C.c_void_p, C.c_size_t]
assert lib.kmpc_trace_read(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(8192, 32)[:B].astype(np.int64)
sw, rb, ps, it = t[:, 22], t[:, 23], t[:, 24], t[:, 25]
ls, rounds = t[:, 26] % 1000, t[:, 26] // 1000
carried, easy = t[:, 27] % 1000, t[:, 27] // 1000
body = t[:, 21] / 100.0 / steps
print("%d-step launch (%s start): per trajectory-step means: sweeps %.3f rebuilds %.3f refinement passes %.3f iterations %.3f prediction rounds %.3f backtracks %.3f"
      % (steps, "cold" if cold else "warm", sw.mean() / steps, rb.mean() / steps, ps.mean() / steps, it.mean() / steps, rounds.mean() / steps, ls.mean() / steps))
fb = t[:, 28]
print("window: steps %d..%d from the RLS reset; fall-back (active-set on global scratch) solves per trajectory-step %.4f (trajectories with any: %d)" % (start, start + steps - 1, fb.mean() / steps, int((fb > 0).sum())))
print("solves that began with a carried tableau: %.3f; 'easy' solves (one iteration, no sweep, no rebuild, no round): %.3f" % (carried.mean() / steps, easy.mean() / steps))
print("distribution over trajectories of easy steps out of %d: " % steps, np.bincount(easy.astype(int), minlength=steps + 1).tolist())
print("refinement passes per trajectory-step, histogram of the per-trajectory mean (bins of 0.5):", np.histogram(ps / steps, bins=np.arange(0, 8.5, 0.5))[0].tolist())
A = np.stack([np.ones(B), sw / steps, rb / steps, ps / steps, it / steps, rounds / steps, ls / steps, fb / steps], 1)
coef = np.linalg.lstsq(A, body, rcond=None)[0]
print("body us/step ~ %.2f + %.2f sweeps + %.2f rebuilds + %.2f passes + %.2f iterations + %.2f rounds + %.2f backtracks + %.2f fall-backs  (least squares over the waves)" % tuple(coef))
wgfin = (t[:, 18].reshape(-1, G).max(1) - t[:, 19].min()) / 100.0
order = np.argsort(wgfin)
print("workgroup finish: min %.0f p10 %.0f median %.0f p90 %.0f max %.0f us" % (wgfin.min(), np.percentile(wgfin, 10), np.median(wgfin), np.percentile(wgfin, 90), wgfin.max()))
def wgstat(v, f): return f(v.reshape(-1, G), 1)
for nm, v in (("sweeps", sw), ("rebuilds", rb), ("passes", ps), ("iterations", it)):
    print("  corr(finish, WG sum of %s) %.2f, with the WG max %.2f" % (nm, np.corrcoef(wgfin, wgstat(v, np.sum))[0, 1], np.corrcoef(wgfin, wgstat(v, np.max))[0, 1]))
for label, sel in (("fastest 10%", order[:len(order) // 10]), ("middle 20%", order[int(0.4 * len(order)):int(0.6 * len(order))]), ("slowest 10%", order[-(len(order) // 10):]), ("slowest 3", order[-3:])):
    print("  %-12s finish %.0f us | per trajectory-step: sweeps %.2f (max traj %.2f) rebuilds %.3f (max %.2f) passes %.2f (max %.2f) iterations %.2f (max %.2f)" % (
        label, wgfin[sel].mean(), sw.reshape(-1, G)[sel].mean() / steps, sw.reshape(-1, G)[sel].max() / steps, rb.reshape(-1, G)[sel].mean() / steps, rb.reshape(-1, G)[sel].max() / steps,
        ps.reshape(-1, G)[sel].mean() / steps, ps.reshape(-1, G)[sel].max() / steps, it.reshape(-1, G)[sel].mean() / steps, it.reshape(-1, G)[sel].max() / steps))
wait = t[:, 20] / 100.0 / steps
print("per wave and step (us): body median %.2f p90 %.2f max %.2f | lift + barrier wait median %.2f p10 %.2f p90 %.2f" % (np.median(body), np.percentile(body, 90), body.max(), np.median(wait), np.percentile(wait, 10), np.percentile(wait, 90)))
bw = body.reshape(-1, G)
print("per workgroup: mean over waves of body %.2f, max over waves %.2f (medians over the workgroups); corr(finish, max-wave body) %.2f, corr(finish, mean body) %.2f"
      % (np.median(bw.mean(1)), np.median(bw.max(1)), np.corrcoef(wgfin, bw.max(1))[0, 1], np.corrcoef(wgfin, bw.mean(1))[0, 1]))
for g in order[-4:]:
    print("   slow workgroup %3d finish %.0f us: wave bodies (us/step) %s | iterations/step %s | fall-backs %s" % (
        g, wgfin[g], np.round(bw[g], 1).tolist(), np.round(it.reshape(-1, G)[g] / steps, 2).tolist(), fb.reshape(-1, G)[g].tolist()))
worst = np.argsort(-body)[:12]
print("the 12 slowest waves: body us/step, sweeps, rebuilds, passes, iterations, rounds per step")
for b in worst:
    print("   traj %4d: %.1f us  %.2f %.2f %.2f %.2f %.2f" % (b, body[b], sw[b] / steps, rb[b] / steps, ps[b] / steps, it[b] / steps, rounds[b] / steps))
U = None
