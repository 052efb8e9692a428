"""Dev tool (GPU box): the RBF register-state set behind float32 panels against the float64 handle, step by step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import KoopmanMPC as KM
from koopmpc.synth import initial_states, offline_data, vdp_rk4
f32x = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
L, N, B = 8, 30, 512
Xo, Yo, Uo = offline_data(plant=vdp_rk4)
cx = Xo[:, np.random.RandomState(0).choice(Xo.shape[1], L, replace=False)].T.copy()
m64 = KM(n=2, L=L, N=N, batch=B, lift="rbf", centres=cx, P0=1e5, barQ0=1e5, dtype=torch.float64, device="cuda:0")
A, Bm, C = [t.cpu().numpy() for t in m64.offline_fit(f32x(Xo), f32x(Yo), f32x(Uo), ridge=1e-9)]
m32 = KM(n=2, L=L, N=N, batch=B, lift="rbf", centres=cx, P0=1e5, barQ0=1e5, dtype=torch.float32, device="cuda:0")
sys.path.insert(0, ROOT)
from oracle import koopman_oracle as ko
PX, PY = ko.rbf_lift(f32x(Xo), cx), ko.rbf_lift(f32x(Yo), cx)
Z = np.concatenate([PX, f32x(Uo)[None, :]], 0)
P = np.linalg.inv(Z @ Z.T + 1e-9 * np.eye(L + 1)); KA = PY @ Z.T
bQ = np.linalg.inv(PX @ PX.T + 1e-9 * np.eye(L)); bX = f32x(Xo) @ PX.T
for m in (m32, m64):
    m.set_model(f32x(A), f32x(Bm), f32x(C))
    m.state_init(K_A=KA, inv_K_G=P, bar_X=bX, bar_Q=bQ)
print("fused:", m32.rollout_is_fused(), m64.rollout_is_fused())
X0 = f32x(initial_states(B, seed=3))
r = np.tile(np.array([[1.0], [0.0]]), (1, N))
X32 = torch.tensor(X0, dtype=torch.float32, device="cuda:0").contiguous()
X64 = torch.tensor(X0, dtype=torch.float64, device="cuda:0").contiguous()
for k in range(15):
    U32, _ = m32.rollout("vdp", X32, r, 1, step0=k, log=True)
    U64, _ = m64.rollout("vdp", X64, r, 1, step0=k, log=True)
    print(k, "status", int(m32.status.max()), int(m64.status.max()), "du %.2e dx %.2e" % (float((U32.double() - U64).abs().max()), float((X32.double() - X64).abs().max())),
          "finite", bool(torch.isfinite(X32).all()), "nbad", int((m32.status != 0).sum()))
