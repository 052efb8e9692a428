"""File-format compatibility with the reference's artefacts.

* encoder weights: the `.mat` files next to the MATLAB encoders (`Revise_2/duffing_weights.mat`,
  `VDP_Revise_2/Good_VDP.mat`, `Weights/Tank_New.mat`; W1..Wk (out x in), b1..bk (1 x out),
  Encoder_Duffing.m:2-6) and the `model_weights.mat` duffing.py:61-64 writes.
* closed-loop logs: the `.mat` the Python scripts dump for the paper's MATLAB plotting code
  (`scio.savemat('DuffingPlotrealtime.mat', {'logXloc', 'logUloc', 'logR', 'tspan', ...})`, duffing.py:1015).
"""
from __future__ import annotations

import numpy as np


def load_encoder_mat(path):
    """-> [(W1, b1), ..., (Wk, bk)] for KoopmanMPC(weights=...); b is flattened from the 1 x k row."""
    import scipy.io as sio

    d = sio.loadmat(path)
    out, k = [], 1
    while "W%d" % k in d:
        out.append((np.ascontiguousarray(d["W%d" % k], dtype=np.float64),
                    np.ascontiguousarray(d["b%d" % k], dtype=np.float64).reshape(-1)))
        k += 1
    if not out:
        raise ValueError("%s holds no W1/b1 ... arrays" % path)
    return out


def save_encoder_mat(path, weights):
    """The inverse of load_encoder_mat (same keys and shapes as duffing.py:61-64)."""
    import scipy.io as sio

    d = {}
    for k, (W, b) in enumerate(weights, start=1):
        d["W%d" % k] = np.asarray(W, dtype=np.float64)
        d["b%d" % k] = np.asarray(b, dtype=np.float64).reshape(1, -1)
    sio.savemat(path, d)


def save_closed_loop_mat(path, logXloc, logUloc, r=None, h=0.05, traj=0, extra=None):
    """Write one trajectory of a rollout with the reference's variable names (duffing.py:1015):
    logXloc (2 x T), logUloc (1 x T), logR (2 x T), tspan (T,).  `logXloc` / `logUloc` are the
    (steps, n, B) / (steps, B) logs of KoopmanMPC.rollout(log=True) (tensors or arrays)."""
    import scipy.io as sio

    X = np.asarray(logXloc.cpu() if hasattr(logXloc, "cpu") else logXloc, dtype=np.float64)
    U = np.asarray(logUloc.cpu() if hasattr(logUloc, "cpu") else logUloc, dtype=np.float64)
    T = U.shape[0]
    d = {"logXloc": X[:, :, traj].T.copy(), "logUloc": U[:, traj].reshape(1, T).copy(), "tspan": h * np.arange(T)}
    if r is not None:
        r = np.asarray(r, dtype=np.float64)
        d["logR"] = np.tile(r[:, :1], (1, T))
    if extra:
        d.update(extra)
    sio.savemat(path, d)
