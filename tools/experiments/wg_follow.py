"""Dev tool (trace build): do the slow workgroups follow the trajectories or the workgroup index (CU placement)?
Same batch twice, the second time with the trajectories in reversed order."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import _ffi
_ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), "libkoopmpc_trace.so")
from koopmpc import KoopmanMPC
from koopmpc.synth import random_mlp_weights, initial_states, offline_data
B, L, N, G = 4096, 20, 20, 16
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lib = _ffi.load()
lib.kmpc_trace_read.restype = C.c_int
lib.kmpc_trace_read.argtypes = [C.c_void_p, C.c_size_t]
r = np.tile(np.array([[1.0], [0.0]]), (1, N))
def run(rev):
    m = KoopmanMPC(n=2, L=L, N=N, batch=B, weights=random_mlp_weights(2, 100, 3, L))
    m.offline_fit(*offline_data())
    x0 = initial_states(B)
    if rev:
        x0 = x0[:, ::-1].copy()
    X = torch.tensor(x0, dtype=torch.float64, device="cuda:0").contiguous()
    m.rollout("duffing", X, r, 200, step0=0)
    out = []
    for i in range(3):
        m.iters.zero_()
        m.rollout("duffing", X, r, steps, step0=200 + i * steps)
        torch.cuda.synchronize()
        buf = np.zeros(8192 * 32, dtype=np.uint64)
        assert lib.kmpc_trace_read(buf.ctypes.data, buf.nbytes) == 0
        t = buf.reshape(8192, 32)[:B].astype(np.int64)
        out.append(((t[:, 18].reshape(-1, G).max(1) - t[:, 19].min()) / 100.0, m.iters.cpu().numpy().astype(float)))
    return out
a, b = run(False), run(True)
for i in range(3):
    fa, ia = a[i]; fb, ib = b[i]
    print("launch %d: corr(finish A, finish B same WG index) %.2f   corr(finish A, finish B same trajectories) %.2f   corr iters same trajectories %.2f" % (
        i, np.corrcoef(fa, fb)[0, 1], np.corrcoef(fa, fb[::-1])[0, 1], np.corrcoef(ia, ib[::-1])[0, 1]))
fa = np.mean([x[0] for x in a], 0); fb = np.mean([x[0] for x in b], 0)
print("mean of 3 launches: same index %.2f, same trajectories %.2f" % (np.corrcoef(fa, fb)[0, 1], np.corrcoef(fa, fb[::-1])[0, 1]))
ia = np.sum([x[1] for x in a], 0).reshape(-1, G)
# a linear model of the WG finish: sum of Newton solves over the WG + the per-SIMD maximum
A = np.stack([np.ones(len(fa)), ia.sum(1), np.stack([ia[:, s::4].sum(1) for s in range(4)], 1).max(1)], 1)
coef, res, *_ = np.linalg.lstsq(A, fa, rcond=None)
pred = A @ coef
print("finish ~ %.1f + %.3f x (WG Newton solves) + %.3f x (max SIMD Newton solves): R^2 %.2f; residual sd %.1f of sd %.1f us" % (
    coef[0], coef[1], coef[2], 1 - ((fa - pred) ** 2).sum() / ((fa - fa.mean()) ** 2).sum(), (fa - pred).std(), fa.std()))
