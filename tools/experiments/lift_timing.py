#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "koopman-online-updated-mpc_amd"))
import numpy as np, torch
from koopmpc import KoopmanMPC
from koopmpc.synth import random_mlp_weights
w = random_mlp_weights(2, 100, 3, 20)
mpc = KoopmanMPC(n=2, L=20, N=20, batch=16, weights=w)
for B in (16, 256, 4096, 16384, 65536, 262144):
    X = torch.rand(2, B, dtype=torch.float64, device="cuda:0")
    Psi = torch.empty(20, B, dtype=torch.float64, device="cuda:0")
    f = lambda: mpc.lib.kmpc_lift(mpc.h, mpc._p(X), mpc._p(Psi), B, mpc._stream())
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("B=%7d  %8.1f us   %.2f TFLOP/s" % (B, us, 2 * (2 * 100 + 2 * 100 * 100 + 100 * 20) * B / us / 1e6))
