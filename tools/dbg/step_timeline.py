"""Dev tool (GPU box, trace build): the timeline of a fused cfg2 launch step by step -- per workgroup the time every step took, how
much of it the slowest wave's body was, and what the slowest workgroups' steps looked like.
    KMPC_TRACE_LIB=libkoopmpc_devtrace.so KMPC_QPW_START=5 python tools/dbg/step_timeline.py [steps]"""
import os as _os; _os.environ.setdefault("KMPC_DEBUG", "1")
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import _ffi
_ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), os.environ.get("KMPC_TRACE_LIB", "libkoopmpc_trace.so"))
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
name = "cfg2"
c = bench.CONFIGS[name]; w = bench.workload_inputs(name, c["L"], c["N"])
B, G = 4096, 16
loop = bench.Loop(name, w, B, torch.float64, torch.device("cuda", 0), 0)
start = int(os.environ.get("KMPC_QPW_START", c["settle"] + 5))
loop.advance(start - 5, 0); loop.advance(5, start - 5); torch.cuda.synchronize()
import time
t0 = time.time()
snap = (loop.m.state_to(), loop.X.clone())
while time.time() - t0 < 1.0:
    loop.m.state_from(snap[0]); loop.X.copy_(snap[1]); loop.advance(steps, start); torch.cuda.synchronize()
loop.m.state_from(snap[0]); loop.X.copy_(snap[1])
loop.advance(steps, start); torch.cuda.synchronize()
lib = _ffi.load()
buf = np.zeros(4096 * 96, dtype=np.uint32)
lib.kmpc_step_trace_read.restype = C.c_int; lib.kmpc_step_trace_read.argtypes = [C.c_void_p, C.c_size_t]
assert lib.kmpc_step_trace_read(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(4096, 32, 3)[:, :steps].astype(np.int64)
tb = np.zeros(8192 * 32, dtype=np.uint64)
lib.kmpc_trace_read.restype = C.c_int; lib.kmpc_trace_read.argtypes = [C.c_void_p, C.c_size_t]
assert lib.kmpc_trace_read(tb.ctypes.data, tb.nbytes) == 0
hw = tb.reshape(8192, 32)[:B, 29].astype(np.int64)
xcc, hwid = hw & 15, hw >> 8
cu, sh, se = (hwid >> 8) & 15, (hwid >> 12) & 1, (hwid >> 13) & 7
# NOTE: trajectories are dealt to waves by the placement: the workgroup of a trajectory is not b // 16.  Group by the time a step began:
# waves of one workgroup leave the lift's last barrier together, so (lift done) of step 0 identifies the workgroup up to ties; instead
# use the exported perm when present -- simpler: cluster by rounding the step-0 lift-done stamps
ready, lifted, done = t[:, :, 0], t[:, :, 1], t[:, :, 2]
body = (done - lifted) / 100.0            # us
# (one workgroup per CU: the CU a wave runs on identifies its workgroup)
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
ids, inv, cnt = np.unique(cuid, return_inverse=True, return_counts=True)
assert (cnt == G).all(), ("workgroups per CU", np.bincount(cnt).tolist())
groups = np.argsort(inv, kind="stable").reshape(-1, G)
key = lifted[:, 0]
spread = (key[groups].max(1) - key[groups].min(1)) / 100.0
print("window %d..%d: %d workgroups identified by the CU they run on; max spread of the step-0 lift-done stamps inside a group %.2f us" % (start, start + steps - 1, len(groups), spread.max()))
wg_step = (lifted[groups][:, :, 1:].max(1) - lifted[groups][:, :, :-1].max(1)) / 100.0   # (W, steps-1): lift-done to lift-done
wg_body_max = body[groups].max(1)         # (W, steps)
wg_body_mean = body[groups].mean(1)
fin = (done[groups].max(1).max(1) - lifted[groups][:, :, 0].min(1)) / 100.0
o = np.argsort(fin)
print("workgroup finish (from the first lift): min %.0f median %.0f p90 %.0f max %.0f us" % (fin.min(), np.median(fin), np.percentile(fin, 90), fin.max()))
print("per step: lift-to-lift median %.2f us | slowest wave's body median %.2f, mean body %.2f | (step - slowest body) median %.2f us" % (
    np.median(wg_step), np.median(wg_body_max), np.median(wg_body_mean), np.median(wg_step - wg_body_max[:, :-1])))
print("corr over workgroups (finish, sum of per-step max bodies) %.2f; (finish, sum of mean bodies) %.2f" % (
    np.corrcoef(fin, wg_body_max.sum(1))[0, 1], np.corrcoef(fin, wg_body_mean.sum(1))[0, 1]))
print("distribution of the per-step slowest body over all workgroups and steps: p50 %.1f p90 %.1f p99 %.1f max %.1f us" % tuple(np.percentile(wg_body_max, [50, 90, 99, 100])))
for lab, sel in (("fastest", o[:3]), ("median", o[len(o) // 2 - 1:len(o) // 2 + 2]), ("slowest", o[-5:])):
    for g in sel:
        print("  %-8s wg finish %5.0f us: per-step slowest body %s" % (lab, fin[g], np.round(wg_body_max[g], 0).astype(int).tolist()))

gx, gse, gcu, gsh = xcc[groups][:, 0], se[groups][:, 0], cu[groups][:, 0], sh[groups][:, 0]
print("workgroups per XCC:", np.bincount(gx, minlength=8).tolist())
for x in range(8):
    f = fin[gx == x]
    if len(f): print("  XCC %d: %3d workgroups, finish min %.0f median %.0f max %.0f | mean lift-to-lift %.2f us" % (x, len(f), f.min(), np.median(f), f.max(), wg_step[gx == x].mean()))
slow = o[-26:]
print("the 26 slowest workgroups: (xcc, se, sh, cu, finish):", [(int(gx[g]), int(gse[g]), int(gsh[g]), int(gcu[g]), int(fin[g])) for g in slow])

print("per-step lift-to-lift (us) of the slowest workgroups against the median workgroup's:")
med = np.median(wg_step, axis=0)
print("   all workgroups, median per step:", np.round(med, 0).astype(int).tolist())
for g in o[-4:]:
    print("   xcc %d finish %4.0f:" % (gx[g], fin[g]), np.round(wg_step[g], 0).astype(int).tolist())
for x in range(8):
    print("   XCC %d mean per step:" % x, np.round(wg_step[gx == x].mean(0), 0).astype(int).tolist())
