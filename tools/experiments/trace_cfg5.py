"""Dev tool (GPU box, trace build): phase stamps of the four-wave step kernel at the cfg5 sizes (L = 64, N = 50)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import _ffi
_ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), os.environ.get("KMPC_TRACE_LIB", "libkoopmpc_trace.so"))
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
c = bench.CONFIGS["cfg5"]; w = bench.workload_inputs("cfg5", c["L"], c["N"])
loop = bench.Loop("cfg5", w, B, torch.float64, torch.device("cuda", 0), 0)
loop.advance(int(sys.argv[2]) if len(sys.argv) > 2 else 30, 0); torch.cuda.synchronize()
lib = _ffi.load()
buf = np.zeros(8192 * 32, dtype=np.uint64)
lib.kmpc_trace_read.restype = C.c_int; lib.kmpc_trace_read.argtypes = [C.c_void_p, C.c_size_t]
assert lib.kmpc_trace_read(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(8192, 32)[:min(B, 8192)].astype(np.int64)
names = ["start", "rls: state in LDS", "rls: Pz, d", "rls: P, K written", "rls: C part", "cond: init", "cond: chains",
         "cond: H, f", "qp: setup", "qp: first KKT", "qp: sweeps", "qp: direction", "qp: first Armijo", "qp: loop exit", "end"]
print("%-22s %8s %8s %8s   (us, per workgroup, last step)" % ("segment", "median", "p90", "max"))
for i in range(1, 15):
    d = (t[:, i] - t[:, i - 1]) / 100.0
    print("%-22s %8.2f %8.2f %8.2f" % (names[i], np.median(d), np.percentile(d, 90), d.max()))
tot = (t[:, 14] - t[:, 0]) / 100.0
print("%-22s %8.2f %8.2f %8.2f" % ("whole step", np.median(tot), np.percentile(tot, 90), tot.max()))
print("launch span %.1f us" % ((t[:, 14].max() - t[:, 0].min()) / 100.0), "Newton solves mean %.2f" % t[:, 15].mean())
