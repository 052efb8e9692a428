"""Dev tool (GPU box): which trajectory / step of the (20, 20, Cx) random-model roll-out reports a QP status != 0, and what the
exact solver says about that QP."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import KoopmanMPC
from koopmpc.synth import random_mlp_weights
from oracle import koopman_oracle as ko

L, N, B, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rng = np.random.RandomState(4)
w = random_mlp_weights(2, 100, 3, L, seed=3)
A = rng.randn(L, L); A *= 0.95 / np.abs(np.linalg.eigvals(A)).max()
Bm, Cm = rng.randn(L, 1) * 0.1, rng.randn(2, L) * 0.5
r = np.tile(np.array([[1.0], [0.0]]), (1, N))
m = KoopmanMPC(n=2, L=L, N=N, batch=B, weights=w)
m.set_model(A, Bm, Cm)
X = torch.tensor(4 * rng.rand(2, B) - 2, dtype=torch.float64, device="cuda:0")
for i in range(steps):
    psi = m.lift(X) if hasattr(m, "lift") else None
    u = m.step(X, r).clone()
    st = m.status.cpu().numpy(); it = m.iters.cpu().numpy()
    print("step", i, "status max", st.max(), "iters", it.min(), it.max(), "bad", np.nonzero(st)[0].tolist())
    if st.max() != 0:
        b = int(np.nonzero(st)[0][0])
        Am, Bv, Cv = [t.cpu().numpy() for t in m.get_model()]
        Ab, Bb, Cb = Am[b], Bv[b].reshape(-1, 1), Cv[b]
        ps = ko.mlp_lift(w, X.cpu().numpy()[:, b:b + 1]).reshape(-1)
        _, _, H, f, _ = ko.condense(Ab, Bb, Cb, ps, r, N, 100.0, 1e-4)
        U, info = ko.qp_exact(H, f, -2.0, 2.0)
        print("  trajectory", b, "cond(H) %.3e" % np.linalg.cond(H), "exact U0", U[0], "gpu u", float(u[b]), "Useq diff", float(np.abs(m.Useq.cpu().numpy()[:, b] - U).max()),
              "nactive", int(np.sum(np.abs(np.abs(U) - 2) < 1e-9)), "eig min %.3e" % np.linalg.eigvalsh(H).min())
        psi = m.Encoder(X)
        Hh, fh = m.condense(psi, r)
        Uq, stq, itq = m.qp_solve(Hh, fh)
        print("  legacy qp_solve on the same H, f (cold): status", int(m.status[b]), "iters", int(m.iters[b]), "U0", float(Uq[0, b]), "diff to exact", float(np.abs(Uq[:, b].cpu().numpy() - U).max()),
              "H diff gpu/oracle %.2e" % float(np.abs(Hh[b].cpu().numpy() - H).max() / np.abs(H).max()))
        Jg = lambda v: float(v @ H @ v + f @ v)
        print("  J(exact) %.6e J(gpu step) %.6e J(legacy) %.6e" % (Jg(U), Jg(m.Useq.cpu().numpy()[:, b]), Jg(Uq[:, b].cpu().numpy())))
    X = m.plant_step("duffing", X, u, switched=(98 + i >= 102))
