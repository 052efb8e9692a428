"""Dev tool (GPU box, trace build): where the cooperative lift of the cfg2 roll-out spends its time -- wave 0 of every workgroup stamps
the wall clock after each barrier and after each layer's products (LAST step of the launch).
    KMPC_TRACE_LIB=libkoopmpc_devtrace.so python tools/dbg/lift_timeline.py [steps]"""
import os as _os; _os.environ.setdefault("KMPC_DEBUG", "1")
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import _ffi
_ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), os.environ.get("KMPC_TRACE_LIB", "libkoopmpc_trace.so"))
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
c = bench.CONFIGS["cfg2"]; w = bench.workload_inputs("cfg2", c["L"], c["N"])
loop = bench.Loop("cfg2", w, 4096, torch.float64, torch.device("cuda", 0), 0)
loop.advance(c["settle"], 0); torch.cuda.synchronize()
import time
t0 = time.time()
while time.time() - t0 < 1.0:
    loop.advance(steps, c["settle"]); torch.cuda.synchronize()
lib = _ffi.load()
buf = np.zeros(512 * 16, dtype=np.uint64)
lib.kmpc_lift_trace_read.restype = C.c_int; lib.kmpc_lift_trace_read.argtypes = [C.c_void_p, C.c_size_t]
assert lib.kmpc_lift_trace_read(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(512, 16)[:256].astype(np.int64)
tb = np.zeros(8192 * 32, dtype=np.uint64)
lib.kmpc_trace_read.restype = C.c_int; lib.kmpc_trace_read.argtypes = [C.c_void_p, C.c_size_t]
assert lib.kmpc_trace_read(tb.ctypes.data, tb.nbytes) == 0
tt = tb.reshape(8192, 32)[:4096].astype(np.int64)
ready = tt[:, 16].reshape(256, 16)   # every wave's arrival at the lift of the last step
names = ["barrier 0 released (after the last arrival)", "layer 1 product + store", "barrier 1", "hidden layer 2 products + store", "barrier 2",
         "hidden layer 3 products + store", "barrier 3", "output layer products + store", "barrier 4"]
print("lift timeline of wave 0, last step of a %d-step launch, median / p90 over the 256 workgroups (us)" % steps)
d0 = (t[:, 0] - ready.max(1)) / 100.0
print("  %-48s %6.2f %6.2f" % (names[0], np.median(d0), np.percentile(d0, 90)))
for i in range(1, 9):
    d = (t[:, i] - t[:, i - 1]) / 100.0
    print("  %-48s %6.2f %6.2f" % (names[i], np.median(d), np.percentile(d, 90)))
tot = (t[:, 8] - ready.max(1)) / 100.0
print("  %-48s %6.2f %6.2f" % ("last arrival -> barrier 4 released", np.median(tot), np.percentile(tot, 90)))
sp = (ready.max(1) - ready.min(1)) / 100.0
print("  spread of the 16 arrivals: median %.2f p90 %.2f us" % (np.median(sp), np.percentile(sp, 90)))
