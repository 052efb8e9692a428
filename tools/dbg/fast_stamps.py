"""Dev tool (GPU box, trace build): phase stamps of shared_fast_kernel (its last workgroup) at the cfg4 dimensions, settled loop.
    KMPC_TRACE_LIB=libkoopmpc_devtrace.so python tools/dbg/fast_stamps.py [steps]"""
import os as _os; _os.environ.setdefault("KMPC_DEBUG", "1")
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import _ffi
_ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), os.environ.get("KMPC_TRACE_LIB", "libkoopmpc_trace.so"))
import bench
c = bench.CONFIGS["cfg4"]; w = bench.workload_inputs("cfg4", c["L"], c["N"])
loop = bench.Loop("cfg4", w, c["B"], torch.float64, torch.device("cuda", 0), 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
loop.advance(steps, 0); torch.cuda.synchronize()
lib = _ffi.load()
buf = np.zeros(8, dtype=np.uint64)
lib.kmpc_fast_trace_read.restype = C.c_int; lib.kmpc_fast_trace_read.argtypes = [C.c_void_p, C.c_size_t]
assert lib.kmpc_fast_trace_read(buf.ctypes.data, buf.nbytes) == 0
t = buf.astype(np.int64)
names = ["operand loads requested", "f = F psi + f0 (+ put, barrier)", "u = T0 f, |H| 1 (+ put, barriers)", "g = 2 H u + f", "box / certificate test, outputs, plant"]
for i, nm in enumerate(names):
    print("%-40s %7.2f us" % (nm, (t[i + 1] - t[i]) / 100.0))
print("%-40s %7.2f us" % ("first stamp to last", (t[5] - t[0]) / 100.0))
