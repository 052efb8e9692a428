"""Dev tool (trace build): are the slow workgroups of a roll-out launch the same ones in the next launch?
    python tools/dbg/wg_persistence.py [steps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import _ffi
_ffi.LIB_PATH = os.path.join(os.path.dirname(_ffi.LIB_PATH), os.environ.get("KMPC_TRACE_LIB", "libkoopmpc_trace.so"))
from koopmpc import KoopmanMPC
from koopmpc.synth import random_mlp_weights, initial_states, offline_data
B, L, N, G = 4096, 20, 20, 16
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
m = KoopmanMPC(n=2, L=L, N=N, batch=B, weights=random_mlp_weights(2, 100, 3, L))
m.offline_fit(*offline_data())
r = np.tile(np.array([[1.0], [0.0]]), (1, N))
X = torch.tensor(initial_states(B), dtype=torch.float64, device="cuda:0").contiguous()
m.rollout("duffing", X, r, 200, step0=0)
torch.cuda.synchronize()
lib = _ffi.load()
lib.kmpc_trace_read.restype = C.c_int
lib.kmpc_trace_read.argtypes = [C.c_void_p, C.c_size_t]
def launch(k0):
    m.iters.zero_()
    m.rollout("duffing", X, r, steps, step0=k0)
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 32, dtype=np.uint64)
    assert lib.kmpc_trace_read(buf.ctypes.data, buf.nbytes) == 0
    t = buf.reshape(8192, 32)[:B].astype(np.int64)
    slot = np.arange(B)                  # workgroup slot of every trajectory (an experiment dealt them out by solver work: DESIGN 5)
    order = np.argsort(slot)
    fin = (t[order, 18].reshape(-1, G).max(1) - t[:, 19].min()) / 100.0
    body = t[:, 21] / 100.0 / steps
    it = m.iters.cpu().numpy().astype(float)
    print("   WG finish min/med/mean/max %.0f %.0f %.0f %.0f | corr(finish, WG iters sum) %.2f | identity placement: %s" % (
        fin.min(), np.median(fin), fin.mean(), fin.max(), np.corrcoef(fin, it[order].reshape(-1, G).sum(1))[0, 1], bool((slot == np.arange(B)).all())))
    return fin, body, it
res = [launch(200 + i * steps) for i in range(4)]
for i in range(3):
    (f0, b0, i0), (f1, b1, i1) = res[i], res[i + 1]
    print("launch %d -> %d: WG finish min/med/max %.0f %.0f %.0f | corr WG finish %.2f, corr per-trajectory body %.2f, corr iters %.2f, corr WG iters-sum %.2f" % (
        i, i + 1, f1.min(), np.median(f1), f1.max(), np.corrcoef(f0, f1)[0, 1], np.corrcoef(b0, b1)[0, 1], np.corrcoef(i0, i1)[0, 1],
        np.corrcoef(i0.reshape(-1, G).sum(1), i1.reshape(-1, G).sum(1))[0, 1]))
f, b, it = res[-1]
# how much of the WG finish does the Newton count explain?  regress finish on WG sums of (iters) and on the per-SIMD max
its = it.reshape(-1, G)
simd = np.stack([its[:, s::4].sum(1) for s in range(4)], 1)
print("corr(finish, WG iters sum) %.2f  corr(finish, max over SIMDs of its waves' iters) %.2f" % (np.corrcoef(f, its.sum(1))[0, 1], np.corrcoef(f, simd.max(1))[0, 1]))
print("iters per trajectory-step: mean %.3f; share of trajectories above 1.5: %.3f" % (it.mean() / steps, (it / steps > 1.5).mean()))
