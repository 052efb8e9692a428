"""Dev tool (GPU box, trace build): per-wave phase times of the fused roll-out of a bench configuration (RBF sets: the waves never
meet, a wave's phase times are its own).   KMPC_TRACE_LIB=libkoopmpc_devtrace.so python tools/dbg/trace_cfg.py cfg3 [steps]"""
import os as _os; _os.environ.setdefault("KMPC_DEBUG", "1")
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import _ffi
_ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), os.environ.get("KMPC_TRACE_LIB", "libkoopmpc_trace.so"))
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
c = bench.CONFIGS[name]; w = bench.workload_inputs(name, c["L"], c["N"])
B = min(c["B"], 8192)
loop = bench.Loop(name, w, B, torch.float64, torch.device("cuda", 0), 0)
loop.advance(c["settle"], 0); torch.cuda.synchronize()
loop.advance(steps, c["settle"]); torch.cuda.synchronize()
lib = _ffi.load()
buf = np.zeros(8192 * 32, dtype=np.uint64)
lib.kmpc_trace_read.restype = C.c_int; lib.kmpc_trace_read.argtypes = [C.c_void_p, C.c_size_t]
assert lib.kmpc_trace_read(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(8192, 32)[:B].astype(np.int64)
names = ["start", "rls: state loaded", "rls: Pz, d", "rls: P, K written", "rls: C part", "cond: init", "cond: chains",
         "cond: H, f", "qp: setup", "qp: first KKT", "qp: sweeps", "qp: direction", "qp: first Armijo", "qp: loop exit", "end"]
print("%s, B = %d: last step of a %d-step launch, per wave (us)" % (name, B, steps))
print("%-22s %8s %8s %8s" % ("segment", "median", "p90", "max"))
for i in range(1, 15):
    d = (t[:, i] - t[:, i - 1]) / 100.0
    print("%-22s %8.2f %8.2f %8.2f" % (names[i], np.median(d), np.percentile(d, 90), d.max()))
tot = (t[:, 14] - t[:, 0]) / 100.0
print("%-22s %8.2f %8.2f %8.2f" % ("step body", np.median(tot), np.percentile(tot, 90), tot.max()))
print("lift + loop top (stamp 16 -> 0): median %.2f us" % np.median((t[:, 0] - t[:, 16]) / 100.0))
