#!/usr/bin/env python3
"""The closed loop of the reference's vanderpol_RBF.py (:348-526), batched: thin-plate RBF dictionary on KMeans centres
of the offline data (:20-23, 44-46), y = C x, bounds +-2 (:218), and the "storage" update -- least squares over the
stored offline + online samples (:434-438) -- as the RLS continued from the offline Gram (matrix-inversion lemma).
--plant duffing is the reference's duffing_RBF.py: the same script with the Duffing training data (:40), the Duffing
plant (:118, 342) and its parameter switch after iteration 101 (:505-506).

    python -m koopmpc.scripts.vanderpol_RBF --batch 16384 --steps 200 --horizon 30
    python -m koopmpc.scripts.vanderpol_RBF --plant duffing --batch 4096 --steps 200          # duffing_RBF.py
"""
import argparse

import numpy as np
import torch

from koopmpc import KoopmanMPC
from koopmpc.synth import initial_states, offline_data, vdp_rk4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--Nrbf", type=int, default=8)           # vanderpol_RBF.py:43
    ap.add_argument("--horizon", type=int, default=10)
    ap.add_argument("--bound", type=float, default=2.0)      # vanderpol_RBF.py:218
    ap.add_argument("--plant", default="vdp", choices=["vdp", "duffing"])  # duffing: duffing_RBF.py:40, 118, 342, 506
    ap.add_argument("--switch-step", type=int, default=None,
                    help="first step with the switched plant parameters (default: none for vdp as before, 102 for duffing: "
                         "duffing_RBF.py:505-506 switches after iteration 101)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    X, Y, U = offline_data(plant=vdp_rk4) if a.plant == "vdp" else offline_data()  # (data_generator.duffing_generate(), duffing_RBF.py:40)
    try:  # cx = KMeans(n_clusters=Nrbf).fit(X.T).cluster_centers_   (vanderpol_RBF.py:44-46)
        from sklearn.cluster import KMeans

        cx = KMeans(n_clusters=a.Nrbf, n_init=10, random_state=101).fit(X.T).cluster_centers_
    except Exception:  # no scikit-learn: data points as centres
        cx = X[:, np.random.RandomState(101).choice(X.shape[1], a.Nrbf, replace=False)].T.copy()
    B, N = a.batch, a.horizon
    mpc = KoopmanMPC(n=2, L=a.Nrbf, N=N, batch=B, lift="rbf", centres=cx, lb=-a.bound, ub=a.bound, P0=1e5, barQ0=1e5)
    mpc.offline_fit(X, Y, U, ridge=1e-9, init_rls=True)       # offline fit (:78-103) + storage semantics (:434-438)
    r = np.concatenate([np.ones((1, N)), np.zeros((1, N))], axis=0)
    x0 = np.tile(np.array([[-2.0], [-2.0]]), (1, B)) if B == 1 else initial_states(B)
    x_loc = torch.tensor(x0, dtype=torch.float64, device=mpc.device)
    sw = a.switch_step if a.switch_step is not None else (10 ** 9 if a.plant == "vdp" else 102)
    logUloc, logXloc = mpc.rollout(a.plant, x_loc, r, a.steps, step0=0, switch_step=sw, log=True)
    torch.cuda.synchronize()
    print("fused roll-out: %s; worst QP status %d, mean Newton solves/step %.2f" % (mpc.rollout_is_fused(), int(mpc.status.max()),
                                                                              float(mpc.iters.double().mean()) / a.steps))
    print("x_loc[:, 0] after %d steps:" % a.steps, x_loc[:, 0].cpu().numpy(), " u_loc:", float(logUloc[-1, 0]))
    if a.out:
        np.savez(a.out, logXloc=logXloc.cpu().numpy(), logUloc=logUloc.cpu().numpy(), cx=cx)


if __name__ == "__main__":
    main()
