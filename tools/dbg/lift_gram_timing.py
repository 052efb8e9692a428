#!/usr/bin/env python3
"""cfg4's first launch of a step, in pieces: the stand-alone encoder (kmpc_lift), lift + Gram sums (kmpc_shared_local_gram, which
also launches the reduce) -- HIP-event times over many calls.  python tools/dbg/lift_gram_timing.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "koopman-online-updated-mpc_amd")]
import numpy as np
import torch

import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
c = bench.CONFIGS["cfg4"]
w = bench.workload_inputs("cfg4", c["L"], c["N"])
loop = bench.Loop("cfg4", w, B, torch.float64, torch.device("cuda", 0), 0)
m = loop.m
loop.advance(5, 0)
X = loop.X


def timeit(fn, reps=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


Psi = torch.empty(c["L"], B, dtype=torch.float64, device="cuda:0")
print("B = %d" % B)
print("kmpc_lift (lift_coop_kernel<.., false, 1>)      %.2f us per call (back to back)" % timeit(lambda: m.lib.kmpc_lift(m.h, m._p(X), m._p(Psi), B, m._stream())))
print("kmpc_shared_local_gram (lift+Gram, reduce)     %.2f us per call" % timeit(lambda: m.shared_local_gram(X)))
