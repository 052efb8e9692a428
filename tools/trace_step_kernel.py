"""In-kernel phase timeline of the per-step kernel of a bench configuration (cfg5 by default) -- the measurement build
`make -C koopman-online-updated-mpc_amd trace`, thread 0 of every workgroup stamps the 100 MHz wall clock at the phase
boundaries; prints, per segment, median / p90 / max over the first 8192 trajectories of one launch at a settled step.
    python tools/trace_step_kernel.py [config] [B] [settle]"""
import os as _os; _os.environ.setdefault("KMPC_DEBUG", "1")
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "koopman-online-updated-mpc_amd"))
sys.path.insert(0, ROOT)
import numpy as np, torch
from koopmpc import _ffi
_ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), os.environ.get("KMPC_TRACE_LIB", "libkoopmpc_trace.so"))
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
c = bench.CONFIGS[name]
B = int(sys.argv[2]) if len(sys.argv) > 2 else c["B"]
settle = int(sys.argv[3]) if len(sys.argv) > 3 else 30
w = bench.workload_inputs(name, c["L"], c["N"])
loop = bench.Loop(name, w, B, torch.float64, torch.device("cuda:0"), 0)
loop.advance(settle, 0)
torch.cuda.synchronize()
loop.advance(1, settle)
torch.cuda.synchronize()
nb = min(B, 8192)
buf = np.zeros(8192 * 32, dtype=np.uint64)
lib = _ffi.load()
lib.kmpc_trace_read.restype = C.c_int
lib.kmpc_trace_read.argtypes = [C.c_void_p, C.c_size_t]
assert lib.kmpc_trace_read(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(8192, 32)[:nb].astype(np.int64)
names = ["start", "rls: first block in LDS", "rls: C / bar_Q", "rls: inv_K_G", "rls: [A B]", "cond: init", "cond: chains",
         "cond: H, f", "qp: setup", "qp: first KKT", "qp: sweeps / refinement", "qp: direction", "qp: first Armijo", "qp: loop exit",
         "end"]
k0 = t[:, 0].min()
print("%s: B = %d, launch span %.1f us (first %d workgroups: start spread %.1f us)" % (name, B, (t[:, 14].max() - k0) / 100.0, nb, (t[:, 0].max() - k0) / 100.0))
print("%-26s %8s %8s %8s   (us, per workgroup)" % ("segment", "median", "p90", "max"))
for i in range(1, 15):
    d = (t[:, i] - t[:, i - 1]) / 100.0
    print("%-26s %8.2f %8.2f %8.2f" % (names[i], np.median(d), np.percentile(d, 90), d.max()))
tot = (t[:, 14] - t[:, 0]) / 100.0
print("%-26s %8.2f %8.2f %8.2f" % ("whole workgroup", np.median(tot), np.percentile(tot, 90), tot.max()))
its = t[:, 15]
print("Newton iterations: histogram", np.bincount(np.clip(its, 0, 20).astype(int)).tolist())
if t[:, 17].any() or t[:, 16].any():  # (four-wave solver: its counters)
    c0, c1 = (t[:, 18] & 1) != 0, (t[:, 18] & 2) != 0
    print("solves that started from a carried tableau: %.1f %%; still carried at the end: %.1f %%" % (100 * c0.mean(), 100 * c1.mean()))
    print("refinement passes: histogram", np.bincount(np.clip(t[:, 16], 0, 30).astype(int)).tolist())
    print("sweeps: histogram (by tens)", np.bincount(np.clip(t[:, 17] // 10, 0, 12).astype(int)).tolist())
    print("Newton points: histogram", np.bincount(np.clip(t[:, 19], 0, 12).astype(int)).tolist())
    seg = (t[:, 10] - t[:, 9]) / 100.0
    for nm, sel in (("started carried, stayed", c0 & c1), ("started carried, rebuilt", c0 & ~c1), ("started from 2H", ~c0)):
        if sel.any():
            print("  %-26s %5d workgroups: sweeps/refinement segment median %.1f us, whole %.1f us" % (nm, sel.sum(), np.median(seg[sel]), np.median(tot[sel])))
