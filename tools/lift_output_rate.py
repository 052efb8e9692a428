"""Dev tool: fused roll-out rate of the lifted-output sets (y = psi, q = L: vanderpol.py dimensions L = 8, N = 10, MLP lift; cfg3's
L = 8, N = 30 with the RBF lift) -- offline fit on Van der Pol data, closed loop, K-step launches.
    [KMPC_LIB=...] python tools/lift_output_rate.py [B]"""
import os as _os; _os.environ.setdefault("KMPC_DEBUG", "1")
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "koopman-online-updated-mpc_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from koopmpc import KoopmanMPC
from koopmpc.synth import random_mlp_weights
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.RandomState(5)
for name, L, N, kw in (("mlp  L=8 N=10 y=psi", 8, 10, dict(weights=random_mlp_weights(2, 100, 3, 8, seed=3))),
                       ("rbf  L=8 N=30 y=psi", 8, 30, dict(lift="rbf", centres=4 * rng.rand(8, 2) - 2))):
    m = KoopmanMPC(n=2, L=L, N=N, batch=B, output="lift", lb=-2.0, ub=2.0, **kw)
    w = bench.workload_inputs("cfg3", 8, 30)
    m.offline_fit(*w["data"], ridge=1e-9)
    X = torch.tensor(bench.initial_states_for("cfg3", B, 101), dtype=torch.float64, device="cuda:0").contiguous()
    # reference in the lifted space: psi of the target state, constant over the horizon
    xt = torch.tensor(np.tile(np.array([[1.0], [0.0]]), (1, 16)), dtype=torch.float64, device="cuda:0")
    psi_t = (m.Encoder(xt) if "weights" in kw else m.rbf(xt))[:, :1].cpu().numpy()
    r = np.tile(psi_t, (1, N))
    m.rollout("vdp", X, r, 60, step0=0)
    torch.cuda.synchronize()
    for K in (20, 200):
        m.rollout("vdp", X, r, K, step0=60)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 5 if K == 20 else 2
        for _ in range(reps):
            m.rollout("vdp", X, r, K, step0=60)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        print("%-22s B=%d K=%-3d %.3f ms per launch, %.1f us/step, %.1f M steps/s, worst status %d, finite %s" % (
            name, B, K, dt * 1e3, dt / K * 1e6, B * K / dt / 1e6, int(m.status.max()), bool(torch.isfinite(X).all())))
