"""In-kernel phase timeline of step_kernel (measurement build `make -C koopman-online-updated-mpc_amd trace`).
Lane 0 of every workgroup stamps the 100 MHz wall clock at the phase boundaries; this prints, per segment,
the median / p90 / max duration over the 4096 trajectories of one launch at a late closed-loop step.
    python tools/trace_phases.py [B] [steps] [cold]"""
import os as _os; _os.environ.setdefault("KMPC_DEBUG", "1")
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import _ffi
_ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), os.environ.get("KMPC_TRACE_LIB", "libkoopmpc_trace.so"))
from koopmpc import KoopmanMPC
from koopmpc.synth import random_mlp_weights, initial_states, offline_data

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
L, N = 20, 20
G = int(os.environ.get("KMPC_ROLLOUT_WAVES", "16"))  # trajectories per workgroup
cold = len(sys.argv) > 3 and sys.argv[3] == "cold"
m = KoopmanMPC(n=2, L=L, N=N, batch=B, weights=random_mlp_weights(2, 100, 3, L), cold_start=cold)
m.offline_fit(*offline_data())
r = np.tile(np.array([[1.0], [0.0]]), (1, N))
X = torch.tensor(initial_states(B), dtype=torch.float64, device="cuda:0").contiguous()
import time
t0 = time.time()
X0 = X.clone()
while time.time() - t0 < 1.0:  # clock ramp
    m.rollout("duffing", X, r, 10, step0=0, switch_step=10**9)
    torch.cuda.synchronize()
m.reset()
m.offline_fit(*offline_data())
X.copy_(X0)
m.iters.zero_()
pre = int(os.environ.get("KMPC_TRACE_PRE", "0"))  # closed-loop steps in an earlier launch (the traced launch then starts settled)
if pre:
    m.rollout("duffing", X, r, pre, step0=0)
    torch.cuda.synchronize()
    m.iters.zero_()
m.rollout("duffing", X, r, steps, step0=pre)
torch.cuda.synchronize()
nb = min(B, 8192)
buf = np.zeros(8192 * 32, dtype=np.uint64)
lib = _ffi.load()
lib.kmpc_trace_read.restype = C.c_int
lib.kmpc_trace_read.argtypes = [C.c_void_p, C.c_size_t]
assert lib.kmpc_trace_read(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(8192, 32)[:nb].astype(np.int64)
names = ["start", "rls: state in LDS", "rls: Pz, d", "rls: P, K written", "rls: C part", "cond: init", "cond: chains",
         "cond: H, f", "qp: setup", "qp: first KKT", "qp: sweeps", "qp: direction", "qp: first Armijo", "qp: loop exit",
         "end"]
fused = bool((t[:, 17] > 0).any())
if fused:  # roll-out kernel: one launch for all the steps; the per-step stamps are those of the LAST step
    total = (t[:, 18].max() - t[:, 19].min()) / 100.0
    print("fused roll-out: %d steps in %.1f us after the first lift => %.2f us/step" % (steps, total, total / max(1, steps - 1)))
    wg = (np.arange(nb) // G)
    lift = (t[:, 17] - t[:, 16]) / 100.0      # wait for the workgroup's slowest wave + cooperative lift
    body = (t[:, 18] - t[:, 17]) / 100.0
    ready = t[:, 16].reshape(-1, G)
    wait = (ready.max(1, keepdims=True) - ready).reshape(-1) / 100.0  # idle time before the lift barrier
    print("last step: wait for the slowest of the workgroup  median %.2f p90 %.2f max %.2f us" % (np.median(wait), np.percentile(wait, 90), wait.max()))
    print("           lift (after the last wave arrived) median %.2f us" % np.median(((t[:, 17].reshape(-1, G).min(1) - ready.max(1)) / 100.0)))
    print("           step body median %.2f p90 %.2f max %.2f us; slowest-of-workgroup median %.2f us" % (
        np.median(body), np.percentile(body, 90), body.max(), np.median(body.reshape(-1, G).max(1))))
k0 = t[:, 0].min()
print("launch span: %.2f us; wave start spread %.2f us" % ((t[:, 14].max() - k0) / 100.0, (t[:, 0].max() - k0) / 100.0))
print("%-22s %8s %8s %8s   (us, per wave)" % ("segment", "median", "p90", "max"))
for i in range(1, 15):
    d = (t[:, i] - t[:, i - 1]) / 100.0
    print("%-22s %8.2f %8.2f %8.2f" % (names[i], np.median(d), np.percentile(d, 90), d.max()))
tot = (t[:, 14] - t[:, 0]) / 100.0
print("%-22s %8.2f %8.2f %8.2f" % ("whole wave", np.median(tot), np.percentile(tot, 90), tot.max()))
its = t[:, 15]
print("Newton iterations in this launch: histogram", np.bincount(its.astype(int)).tolist())
for k in sorted(set(its.tolist())):
    sel = its == k
    print("  it=%2d: %5d waves, wave time median %.1f max %.1f us, qp loop after first iteration median %.2f us" % (
        k, sel.sum(), np.median(tot[sel]), tot[sel].max(), np.median((t[sel, 13] - t[sel, 12]) / 100.0)))
fin = np.sort((t[:, 14] - k0) / 100.0)
print("finish-time quantiles (us since first wave start): 50%% %.1f  90%% %.1f  99%% %.1f  99.9%% %.1f  max %.1f" % tuple(
    fin[[int(nb * f) - 1 for f in (0.5, 0.9, 0.99, 0.999, 1.0)]]))
print("iters: mean %.2f max %d" % (float(m.iters.double().mean()) / (steps + 0.0), int(m.iters.max())))
if fused:
    s0 = (t[:, 19] - t[:, 19].min()) / 100.0
    print("first-step lift-done stamp since the earliest one (us): 50%% %.1f 75%% %.1f 90%% %.1f max %.1f  (late starters = workgroups that were not resident at launch)"
          % tuple(np.percentile(s0, [50, 75, 90, 100])))
if fused:
    sl, sb = t[:, 20] / 100.0 / steps, t[:, 21] / 100.0 / steps
    print("whole launch, per wave and step: wait+lift mean %.2f (median %.2f) us, body mean %.2f (median %.2f, p90 %.2f) us"
          % (sl.mean(), np.median(sl), sb.mean(), np.median(sb), np.percentile(sb, 90)))
    if t[:, 23].any():  # barrier-free kernel: time in the queue until the lift group was formed
        sq = t[:, 23] / 100.0 / steps
        print("  of which queueing for a lift group: mean %.2f (median %.2f, p90 %.2f) us; the group's lift itself: mean %.2f us"
              % (sq.mean(), np.median(sq), np.percentile(sq, 90), (sl - sq).mean()))
    if t[:, 25].any():
        print("  lift timeline since the group was formed (mean us): layer 1 done %.2f, synced %.2f | hidden 1 done %.2f, synced %.2f | rest (summed) done %.2f synced %.2f"
              % tuple((t[:, i] / 100.0 / steps).mean() for i in (24, 25, 26, 27, 28, 29)))
    print("  body us/step by wave index in the workgroup (age order):", " ".join("%.1f" % v for v in sb.reshape(-1, G).mean(0)))
    fin = (t[:, 18] - t[:, 19].min()) / 100.0
    order = np.argsort(fin)
    for lo, hi in ((0, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 1.0)):
        sel = order[int(lo * nb):int(hi * nb)]
        print("  waves finishing in quartile %.2f-%.2f: finish %.0f-%.0f us, wait+lift %.2f, body %.2f us/step"
              % (lo, hi, fin[sel].min(), fin[sel].max(), sl[sel].mean(), sb[sel].mean()))
if fused and os.environ.get("KMPC_TRACE_WG"):
    # which workgroups are slow: finish time against the Newton solves of its trajectories and against its index (placement)
    wgfin = (t[:, 18].reshape(-1, G).max(1) - t[:, 19].min()) / 100.0
    itw = t[:, 15].reshape(-1, G)
    itsum = m.iters.cpu().numpy()[:nb].reshape(-1, G)
    order = np.argsort(wgfin)
    print("workgroup finish: min %.0f median %.0f p90 %.0f max %.0f us" % (wgfin.min(), np.median(wgfin), np.percentile(wgfin, 90), wgfin.max()))
    print("corr(finish, sum of Newton solves of the WG) %.2f, corr(finish, max over its trajectories) %.2f, corr(finish, WG index %% 8 = XCD) %.2f"
          % (np.corrcoef(wgfin, itsum.sum(1))[0, 1], np.corrcoef(wgfin, itsum.max(1))[0, 1], np.corrcoef(wgfin, np.arange(len(wgfin)) % 8)[0, 1]))
    for name, sel in (("fastest 10%", order[:len(order) // 10]), ("slowest 10%", order[-(len(order) // 10):])):
        print("  %s: finish %.0f us, Newton solves per trajectory-step mean %.2f max %.2f, WG indices mod 8: %s" % (
            name, wgfin[sel].mean(), itsum[sel].mean() / steps, itsum[sel].max() / steps, np.bincount(sel % 8, minlength=8).tolist()))
    xcd = [wgfin[np.arange(len(wgfin)) % 8 == i].mean() for i in range(8)]
    print("  mean finish by WG index mod 8:", " ".join("%.0f" % v for v in xcd))
