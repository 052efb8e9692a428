#!/usr/bin/env python3
"""The with-update closed loop of the reference's vanderpol.py (:738-951), batched over B trajectories on one MI355X:
tracking in the lifted space (y = psi, r_k = Encoder([1, 0]) re-lifted for the whole horizon, vanderpol.py:456-459,
668-675, 756-763), bounds +-6 (:542-544), RLS starts at 1e5 I (:875, 888), plant switch after iteration 101 (:923-931).

    python -m koopmpc.scripts.vanderpol --weights tests/golden/weights_vdp.npz --batch 4096 --steps 300
"""
import argparse

import numpy as np
import torch

from koopmpc import KoopmanMPC
from koopmpc.synth import initial_states, offline_data, random_mlp_weights, vdp_rk4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", default=None, help=".npz with W1..W4, b1..b4 (VDP_Revise_2/Good_VDP.mat)")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--Nlift", type=int, default=8)
    ap.add_argument("--horizon", type=int, default=10)
    ap.add_argument("--model", default=None, help=".npz with A0, B0, C0: the offline model instead of fitting one here")
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-update", action="store_true", help="the comparison loop WITHOUT the online update (vanderpol.py:645-722: logX, logU)")
    a = ap.parse_args()
    if a.weights:
        d = np.load(a.weights)
        weights = [(d["W%d" % k], d["b%d" % k].reshape(-1)) for k in range(1, 5)]
        Nlift = weights[-1][0].shape[0]
    else:
        Nlift, weights = a.Nlift, random_mlp_weights(2, 100, 3, a.Nlift)
    B, N = a.batch, a.horizon
    mpc = KoopmanMPC(n=2, L=Nlift, N=N, batch=B, weights=weights, output="lift", lb=-6.0, ub=6.0, P0=1e5, barQ0=1e5)
    if a.model:
        d = np.load(a.model)
        mpc.set_model(d["A0"], d["B0"], d["C0"])                                # vanderpol.py:726-728
    else:
        mpc.offline_fit(*offline_data(plant=vdp_rk4))                          # vanderpol.py:150-175
    goal = mpc.Encoder(np.array([[1.0], [0.0]]))                               # check_goal, vanderpol.py:668-675
    goal = goal.cpu().numpy() if torch.is_tensor(goal) else np.asarray(goal)
    r = np.tile(goal.reshape(Nlift, 1), (1, N))                                # the same lifted reference at every stage
    x0 = np.tile(np.array([[-2.0], [-2.0]]), (1, B)) if B == 1 else initial_states(B)
    x_loc = torch.tensor(x0, dtype=torch.float64, device=mpc.device)
    if a.no_update:
        mpc.set_online_update(False)
    logUloc, logXloc = mpc.rollout("vdp", x_loc, r, a.steps, step0=0, switch_step=102, log=True)
    torch.cuda.synchronize()
    print("fused roll-out: %s; worst QP status %d, mean Newton solves/step %.2f" % (mpc.rollout_is_fused(), int(mpc.status.max()),
                                                                              float(mpc.iters.double().mean()) / a.steps))
    print("x_loc[:, 0] after %d steps:" % a.steps, x_loc[:, 0].cpu().numpy(), " u_loc:", float(logUloc[-1, 0]))
    if a.out:
        np.savez(a.out, logXloc=logXloc.cpu().numpy(), logUloc=logUloc.cpu().numpy())


if __name__ == "__main__":
    main()
