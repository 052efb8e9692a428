#!/usr/bin/env python3
"""The closed loop of the reference's Tank_System.m (:170-291), batched: cascaded tanks, encoder Encoder_Tank.m
(2-100-100-10) or a random one, delta-u MPC on the augmented model (:110-113, 265-268), output = second tank level
(Cy = [0 1]), |du| <= 0.5, u in [-8, 8] (:182-188), Q = 10, R = 1e-3 (:117-118), parameter switch after step 100
(:193-196).  --shared: ONE model for the batch from pooled Gram sums (the multi-GPU mode: the Gram block is
all-reduced over the process group between the two stages of every step).

    python -m koopmpc.scripts.tank --weights tests/golden/weights_tank.npz --batch 4096 --steps 150
"""
import argparse

import numpy as np
import torch

from koopmpc import KoopmanMPC
from koopmpc.synth import random_mlp_weights, tank_offline_data


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", default=None, help=".npz with W1..W3, b1..b3 (Weights/Tank_New.mat)")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--Nlift", type=int, default=10)
    ap.add_argument("--horizon", type=int, default=20)       # Tank_System.m:116
    ap.add_argument("--shared", action="store_true")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    if a.weights:
        d = np.load(a.weights)
        weights = [(d["W%d" % k], d["b%d" % k].reshape(-1)) for k in range(1, 4)]
        Nlift = weights[-1][0].shape[0]
    else:
        Nlift, weights = a.Nlift, random_mlp_weights(2, 100, 2, a.Nlift)
    B, N = a.batch, a.horizon
    mpc = KoopmanMPC(n=2, L=Nlift, N=N, batch=B, weights=weights, layers=2, lb=-0.5, ub=0.5, umin=-8.0, umax=8.0, Qw=10.0,
                     Rw=1e-3, P0=1e4, barQ0=1e4, delta_u=True, out_row0=1, out_rows=1, c_skip_first=not a.shared)
    mpc.offline_fit(*tank_offline_data(), ridge=1e-9)        # Tank_System.m:82-100
    r = np.ones((1, N))                                       # Yr: second tank level 1
    x0 = np.zeros((2, B)) if B == 1 else np.abs(np.random.RandomState(101).rand(2, B))   # Tank_System.m:124 starts at 0
    x_loc = torch.tensor(x0, dtype=torch.float64, device=mpc.device)
    if a.shared:
        logU = []
        for k in range(a.steps):
            u = mpc.shared_step(x_loc, r)
            logU.append(u.clone())
            x_loc = mpc.plant_step("tank", x_loc, u, switched=(k > 100))
        logUloc = torch.stack(logU)
        logXloc = None
    else:
        logUloc, logXloc = mpc.rollout("tank", x_loc, r, a.steps, step0=0, switch_step=102, log=True)
    torch.cuda.synchronize()
    print("worst QP status of the last step %d; second level: median %.3f; u: %.3f" % (int(mpc.status.max()), float(x_loc[1].median()),
                                                                                         float(logUloc[-1, 0])))
    if a.out:
        np.savez(a.out, logUloc=logUloc.cpu().numpy(), **({"logXloc": logXloc.cpu().numpy()} if logXloc is not None else {}))


if __name__ == "__main__":
    main()
