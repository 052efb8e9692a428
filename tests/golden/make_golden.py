#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by EXECUTING the Python reference.

Runs only in the build container (needs /root/reference); nothing here is
imported by the product, the tests or the bench.  The reference scripts are
flat experiment files, so each one is read, patched IN MEMORY (never written
to disk, never copied into the repo) and executed as ``__main__`` in a worker
subprocess:

  * stub module ``lmi_sdp``           (imported at duffing.py:13, never used)
  * TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD  (whole-module pickle at duffing.py:57)
  * ``maxStep = 10000`` -> ``--steps`` (duffing.py:629)
  * ``f_update(0, x, u)`` -> ``f_update(0, x, np.ravel(u))`` -- numpy >= 1.24
    rejects the ragged array the plant lambda builds from a (1,1) input
    (duffing.py:255,785,871); the arithmetic is unchanged
  * one ``__snap__(...)`` call appended to the body of the with-update loop
    (after duffing.py:990) that copies the loop's live variables

What is stored is DATA only: inputs and the values the reference computed.

    python tests/golden/make_golden.py            # regenerate everything
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------
# worker: executes one patched reference script as __main__
# --------------------------------------------------------------------------
def _patch_common(src, steps):
    src, k = re.subn(r"^maxStep = 10000\s*$", "maxStep = %d" % steps, src, flags=re.M)
    assert k == 1, "maxStep patch"
    src, k = re.subn(r"f_update\(0, x_loc, u_loc\)", "f_update(0, x_loc, np.ravel(u_loc))", src)
    assert k >= 1
    src, k = re.subn(r"f_update\(0, x_origin, u\)", "f_update(0, x_origin, np.ravel(u))", src)
    assert k >= 1
    return src


_SNAP_LINE = (
    "    __snap__(i=i, xlift=xlift, u_loc=u_loc, x_loc=x_loc, ylift=ylift, K_A=K_A, inv_K_G=inv_K_G,"
    " K_ext=K_ext, C_prev=C_prev, Ap=Ap, Bp=Bp, Cp=Cp, r=r,"
    " Useq=result_loc.x, J=result_loc.fun, nfev=result_loc.nfev, **__extra__(locals()))\n"
)


def _insert_snapshot(src):
    """Append the snapshot call right after the (un-commented) C_error.append line of the
    with-update loop -- the last statement of the loop body before the plant switch."""
    lines = src.split("\n")
    idx = [i for i, l in enumerate(lines) if l.startswith("    C_error.append(")]
    assert len(idx) == 1, idx
    lines.insert(idx[0] + 1, _SNAP_LINE.rstrip("\n"))
    return "\n".join(lines)


def worker(script, steps, out):
    import types

    os.environ["TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD"] = "1"
    os.environ["MPLBACKEND"] = "Agg"
    stub = types.ModuleType("lmi_sdp")
    stub.LMI_PD = stub.LMI_NSD = object
    sys.modules["lmi_sdp"] = stub
    sys.path.insert(0, REF)

    work = tempfile.mkdtemp(prefix="kmpc_golden_")
    for f in os.listdir(REF):
        if f.endswith(".pkl"):
            os.symlink(os.path.join(REF, f), os.path.join(work, f))
    os.chdir(work)

    src = open(os.path.join(REF, script)).read()
    src = _patch_common(src, steps)
    src = _insert_snapshot(src)

    snaps = []

    def __snap__(**kw):
        snaps.append({k: np.array(v, dtype=float).copy() for k, v in kw.items()})

    def __extra__(loc):
        # bar_X / bar_Q only exist in the RLS scripts (the *_RBF.py files use the storage method)
        out = {k: loc[k] for k in ("bar_X", "bar_Q") if k in loc}
        if "XU_EX" in loc:
            # the *_RBF.py "storage" update (vanderpol_RBF.py:434-438) refits from ALL stored samples at every step; what it
            # needs of the 10 000+ stored columns are their Gram sums -- data computed from the reference's own arrays
            out.update(stor_KG=loc["K_G"], stor_GX=loc["X_EX"] @ loc["X_EX"].T, stor_XXE=loc["X"] @ loc["X_EX"].T,
                       stor_M=float(loc["X_EX"].shape[1]))
        return out

    g = globals()
    g["__snap__"] = __snap__
    g["__extra__"] = __extra__
    # the script must BE __main__: its pickle resolves __main__.AutoEncoder
    exec(compile(src, script, "exec"), g)

    res = {}
    keys = snaps[0].keys()
    for k in keys:
        res["loop_" + k] = np.stack([s[k] for s in snaps])
    # offline model and closed-loop logs (duffing.py:152-177, 738-805, 823-1012)
    A0 = g["A"]; B0 = g["B"]; C0 = g["C"]
    res["A0"] = np.array(A0, float); res["B0"] = np.array(B0, float); res["C0"] = np.array(C0, float)
    for k in ("logX", "logU", "logXloc", "logUloc", "logXlift", "logXLOClift"):
        if k in g:
            res[k] = np.array(g[k], float)
    if "cx" in g:
        res["cx"] = np.array(g["cx"], float)

    # known-answer lifts straight from the reference's encoder / rbf
    rng = np.random.RandomState(7)
    XS = np.concatenate([np.array([[0.0, 0.0], [-2.0, -2.0], [1.0, 0.0]]), 4.0 * rng.rand(61, 2) - 2.0])
    if "net" in g:
        import torch

        with torch.no_grad():
            PS = np.stack([g["net"].Encoder(torch.tensor(x)).numpy() for x in XS])
    else:
        PS = np.stack([g["rbf"](x, g["cx"]).reshape(-1) for x in XS])
    res["lift_X"] = XS
    res["lift_Psi"] = PS

    # the reference's costFunction on random input sequences, for a few of the loop's models
    cf = g["costFunction"]
    Np = int(g["MPCHorizon"])
    L = res["loop_xlift"].shape[1]
    cost_in, cost_out = [], []
    for s in snaps[:: max(1, len(snaps) // 12)]:
        AB = np.concatenate([s["Ap"], s["Bp"]], axis=1)
        for _ in range(3):
            u = 4.0 * rng.rand(Np) - 2.0
            args = [u, s["r"], AB, s["Cp"], s["xlift"].reshape(L, 1), 0.0, Np, Np, np.zeros((L, 1))]
            J = float(cf(*args))
            cost_in.append(np.concatenate([[s["i"]], u]))
            cost_out.append(J)
    res["cost_in"] = np.array(cost_in)
    res["cost_out"] = np.array(cost_out)
    res["bounds"] = np.array(g["bounds"], float)
    res["h"] = np.array(g["h"], float)
    np.savez_compressed(out, **res)
    print("wrote", out, {k: v.shape for k, v in res.items()})


# --------------------------------------------------------------------------
# driver
# --------------------------------------------------------------------------
def weights():
    """Encoder weights: plain float64 arrays out of the reference's .mat files
    (Encoder_Duffing.m:2, Encoder_VDP.m:2, Encoder_Tank.m:2)."""
    import scipy.io as sio

    for name, rel in (
        ("duffing", "Revise_2/duffing_weights.mat"),
        ("vdp", "VDP_Revise_2/Good_VDP.mat"),
        ("tank", "Weights/Tank_New.mat"),
    ):
        d = sio.loadmat(os.path.join(REF, rel))
        out = {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in d.items() if not k.startswith("__")}
        np.savez_compressed(os.path.join(HERE, "weights_%s.npz" % name), **out)
        print("weights", name, sorted(out))
    d = sio.loadmat(os.path.join(REF, "VDP_Revise_2/NN_Encoder.mat"))
    np.savez_compressed(
        os.path.join(HERE, "vdp_nn_encoder_first200.npz"),
        X_Collection=d["X_Collection"][:, :200],
        X_Collection_NO=d["X_Collection_NO"][:, :200],
        U_Collection=d["U_Collection"][:, :200],
    )


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worker")
    ap.add_argument("--steps", type=int, default=130)
    ap.add_argument("--out")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    if a.worker:
        worker(a.worker, a.steps, a.out)
        return
    if not a.only or a.only == "weights":
        weights()
    for script, out, steps in (
        ("duffing.py", "duffing_loop.npz", 130),
        ("vanderpol.py", "vanderpol_loop.npz", 130),
        ("vanderpol_RBF.py", "vanderpol_rbf_loop.npz", 40),
    ):
        if a.only and a.only not in script:
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--worker", script, "--steps", str(steps),
               "--out", os.path.join(HERE, out)]
        log = os.path.join(tempfile.gettempdir(), "golden_" + script + ".log")
        with open(log, "w") as fh:
            rc = subprocess.call(cmd, stdout=fh, stderr=subprocess.STDOUT)
        print(script, "rc", rc, "log", log)
        if rc:
            print(open(log).read()[-3000:])
            sys.exit(rc)


if __name__ == "__main__":
    main()
