"""Dev tool (GPU box, trace build): phase stamps of shared_model_kernel for the cfg4 dimensions."""
import os as _os; _os.environ.setdefault("KMPC_DEBUG", "1")
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import _ffi
_ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), os.environ.get("KMPC_TRACE_LIB", "libkoopmpc_trace.so"))
import bench
c = bench.CONFIGS["cfg4"]; w = bench.workload_inputs("cfg4", c["L"], c["N"])
loop = bench.Loop("cfg4", w, 1024, torch.float64, torch.device("cuda", 0), 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
loop.advance(steps, 0); torch.cuda.synchronize()
lib = _ffi.load()
buf = np.zeros(16, dtype=np.uint64)
lib.kmpc_shared_trace_read.restype = C.c_int; lib.kmpc_shared_trace_read.argtypes = [C.c_void_p, C.c_size_t]
assert lib.kmpc_shared_trace_read(buf.ctypes.data, buf.nbytes) == 0
t = buf.astype(np.int64)
names = ["inverses (p sweeps)", "model products", "condense set-up", "Gamma chain (N steps)", "g_k", "H", "F, f0", "T0 = -(2H)^-1 (N sweeps)"]
for i, nm in enumerate(names):
    print("%-28s %7.2f us" % (nm, (t[i + 1] - t[i]) / 100.0))
print("total %.2f us" % ((t[8] - t[0]) / 100.0))
if t[9] > 0:
    print("Krylov stages (us since the chain started):", " ".join("%.2f" % ((t[9 + i] - t[3]) / 100.0) for i in range(6)))
