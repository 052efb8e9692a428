"""Roll-out plug-ins from Python: fill the kernel cache ahead of the first controller of a dimension set.

libkoopmpc.so holds the fused roll-out kernel for the dimension sets BASELINE.json and the reference's scripts use; a controller
of any other set gets its kernel when it is created (csrc/rollout_plugin.hip: kernel cache on disk, else hipcc on
csrc/rollout_jit.hip, 4-8 s).  `prebuild` does that step without a device -- `__graft_entry__.build()` calls it for DEFAULT_SETS
so that the objects travel with the tree."""
from __future__ import annotations

import ctypes as C

from . import _ffi

# (n, L, N, out_rows, lift, effective hidden width, batch, dtype)
#   the MATLAB twin of the reference: liftFun = [x; Encoder(x)] - [0; Encoder(0)], L = 10, N = 10 (Koopman_update.m:67, 70, 113) --
#   lift_offset "x_psi0" carries x through the encoder on 2 n extra hidden units: 104
DEFAULT_SETS = [
    (2, 10, 10, 0, "mlp", 104, 64, "f64"),
    (2, 10, 10, 0, "mlp", 104, 4096, "f64"),
    (2, 10, 10, 0, "mlp", 100, 4096, "f64"),
]


def prebuild(sets=None, verbose=False):
    """Make (or find) the plug-ins of the given configurations; returns [(set, code, text)] with code as kmpc_rollout_plugin_status."""
    lib = _ffi.load()
    out = []
    for st in (DEFAULT_SETS if sets is None else sets):
        n, L, N, out_rows, lift, hidden, batch, dtype = st
        kind = {"mlp": _ffi.KMPC_LIFT_MLP, "rbf": _ffi.KMPC_LIFT_RBF_PY, "rbf_matlab": _ffi.KMPC_LIFT_RBF_MATLAB}[lift]
        buf = C.create_string_buffer(1024)
        code = int(lib.kmpc_rollout_plugin_prebuild(n, L, N, out_rows, kind, hidden, batch, _ffi.KMPC_F64 if dtype == "f64" else _ffi.KMPC_F32, buf, len(buf)))
        text = buf.value.decode("utf-8", "replace")
        if verbose:
            print("plug-in %s: %d %s" % (st, code, text))
        out.append((st, code, text))
    return out
