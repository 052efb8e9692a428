#!/usr/bin/env python3
"""The with-update closed loop of the reference's duffing.py (:823-1012), batched over B trajectories on one
MI355X.  Same structure and names (r, xlift -> implicit in step, u_loc, x_loc, logXloc, logUloc); the
per-step arithmetic runs in libkoopmpc's HIP kernels.

    python -m koopmpc.scripts.duffing --weights tests/golden/weights_duffing.npz --batch 4096 --steps 300

--weights: an .npz with W1..W4, b1..b4 (the content of Revise_2/duffing_weights.mat); without it a
random-init encoder of the same architecture is used (koopmpc.synth).
"""
import argparse

import numpy as np
import torch

from koopmpc import KoopmanMPC
from koopmpc.synth import initial_states, offline_data, random_mlp_weights


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", default=None)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)       # maxStep (duffing.py:629 uses 10000)
    ap.add_argument("--Nlift", type=int, default=8)          # duffing.py:66
    ap.add_argument("--horizon", type=int, default=10)       # MPCHorizon = ControlHorizon, duffing.py:632-633
    ap.add_argument("--model", default=None, help=".npz with A0, B0, C0: the offline model instead of fitting one here")
    ap.add_argument("--out", default=None, help="np.savez the logs here (logXloc, logUloc, like duffing.py:1015)")
    ap.add_argument("--no-update", action="store_true", help="the comparison loop WITHOUT the online update (duffing.py:738-805: logX, logU)")
    a = ap.parse_args()

    if a.weights:
        d = np.load(a.weights)
        weights = [(d["W%d" % k], d["b%d" % k].reshape(-1)) for k in range(1, 5)]
        Nlift = weights[-1][0].shape[0]
    else:
        Nlift = a.Nlift
        weights = random_mlp_weights(2, 100, 3, Nlift)
    B, N = a.batch, a.horizon
    mpc = KoopmanMPC(n=2, L=Nlift, N=N, batch=B, weights=weights, lb=-2.0, ub=2.0)  # bounds duffing.py:636
    if a.model:
        d = np.load(a.model)
        mpc.set_model(d["A0"], d["B0"], d["C0"])    # duffing.py:811-813 (Aloc_d, Bloc_d, Cloc_d = A, B, C)
    else:
        mpc.offline_fit(*offline_data())        # duffing.py:152-177 (fit) on the device

    init = np.array([-2.0, -2.0])                                                   # duffing.py:649
    x0 = np.tile(init[:, None], (1, B)) if B == 1 else initial_states(B)
    x_loc = torch.tensor(x0, dtype=torch.float64, device=mpc.device)
    r = np.concatenate([np.ones((1, N)), np.zeros((1, N))], axis=0)                 # duffing.py:834-842
    # the loop body duffing.py:847-992, enqueued from C++; plant parameters switch after iteration 101
    if a.no_update:
        mpc.set_online_update(False)
    logUloc, logXloc = mpc.rollout("duffing", x_loc, r, a.steps, step0=0, switch_step=102, log=True)
    torch.cuda.synchronize()
    print("worst QP status %d, mean Newton solves/step %.2f" % (int(mpc.status.max()), float(mpc.iters.double().mean()) / a.steps))
    print("x_loc[:, 0] after %d steps:" % a.steps, x_loc[:, 0].cpu().numpy(), " u_loc:", float(logUloc[-1, 0]))
    if a.out:
        np.savez(a.out, logXloc=logXloc.cpu().numpy(), logUloc=logUloc.cpu().numpy())


if __name__ == "__main__":
    main()
