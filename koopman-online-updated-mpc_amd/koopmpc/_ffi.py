"""ctypes binding of libkoopmpc.so (include/koopmpc.h).

There is no CPU implementation behind this module: if the HIP library has not been built
(``python __graft_entry__.py`` / ``make -C koopman-online-updated-mpc_amd``) importing it raises.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# (KMPC_LIB: a measurement build of the same sources, e.g. libkoopmpc_dev.so / libkoopmpc_trace.so)
LIB_PATH = os.environ.get("KMPC_LIB") or os.path.join(os.path.dirname(_PKG_DIR), "libkoopmpc.so")

KMPC_F32, KMPC_F64 = 0, 1
KMPC_LIFT_MLP, KMPC_LIFT_RBF_PY, KMPC_LIFT_RBF_MATLAB = 0, 1, 2
KMPC_OUT_CX, KMPC_OUT_LIFT = 0, 1
KMPC_PLANT_DUFFING, KMPC_PLANT_VDP, KMPC_PLANT_TANK = 0, 1, 2
KMPC_PLANT_RK4_MATLAB = 16
KMPC_LIFT_OFFSET_NONE, KMPC_LIFT_OFFSET_PSI0, KMPC_LIFT_OFFSET_X_PSI0 = 0, 1, 2


class KmpcConfig(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("m", C.c_int32), ("L", C.c_int32), ("N", C.c_int32),
        ("hidden", C.c_int32), ("layers", C.c_int32), ("lift_kind", C.c_int32),
        ("output_kind", C.c_int32), ("dtype", C.c_int32), ("batch", C.c_int32),
        ("qp_max_iter", C.c_int32), ("threads", C.c_int32),
        ("delta_u", C.c_int32), ("out_row0", C.c_int32), ("out_rows", C.c_int32), ("c_skip_first", C.c_int32),
        ("cold_start", C.c_int32), ("lift_offset", C.c_int32),
        ("lam", C.c_double), ("P0", C.c_double), ("barQ0", C.c_double),
        ("Qw", C.c_double), ("Rw", C.c_double), ("lb", C.c_double), ("ub", C.c_double),
        ("rbf_eps", C.c_double), ("umin", C.c_double), ("umax", C.c_double),
    ]


# name -> (restype, argtypes); every symbol include/koopmpc.h declares
_VP, _I, _I64, _D = C.c_void_p, C.c_int, C.c_int64, C.c_double
_DP = C.POINTER(C.c_double)
SIGNATURES = {
    "kmpc_create": (_I, [C.POINTER(KmpcConfig), C.POINTER(_VP)]),
    "kmpc_destroy": (_I, [_VP]),
    "kmpc_last_error": (C.c_char_p, [_VP]),
    "kmpc_version": (_I, []),
    "kmpc_set_encoder": (_I, [_VP, _I, _DP, _DP, _I, _I]),
    "kmpc_set_encoder_layer": (_I, [_VP, _I, _DP, _DP, _I, _I]),
    "kmpc_set_centres": (_I, [_VP, _DP, _I, _I]),
    "kmpc_set_model": (_I, [_VP, _DP, _DP, _DP]),
    "kmpc_set_terminal_weight": (_I, [_VP, _DP]),
    "kmpc_solve_dare": (_I, [_VP, _VP, _DP, _D, _I, _D, _I, _I, _VP, _VP, _VP, _VP]),
    "kmpc_terminal_from_dare": (_I, [_VP, _DP, _D, _I, _D, _I, _DP, C.POINTER(C.c_int32), _VP]),
    "kmpc_set_terminal_refresh": (_I, [_VP, _I, _DP, _D, _I, _D]),
    "kmpc_rollout_is_fused": (_I, [_VP]),
    "kmpc_rollout_plugin_status": (_I, [_VP, C.c_char_p, _I]),
    "kmpc_rollout_plugin_prebuild": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, C.c_char_p, _I]),
    "kmpc_set_rollout_workgroup": (_I, [_I]),
    "kmpc_rank_by_work": (_I, [_VP, _I, _VP, _VP]),
    "kmpc_reset": (_I, [_VP, _VP]),
    "kmpc_state_init": (_I, [_VP, _D, _D, _VP]),
    "kmpc_state_init_from": (_I, [_VP, _DP, _DP, _DP, _DP, _VP]),
    "kmpc_lift": (_I, [_VP, _VP, _VP, _I, _VP]),
    "kmpc_rls_update": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _VP]),
    "kmpc_get_model": (_I, [_VP, _VP, _VP, _VP, _VP]),
    "kmpc_condense": (_I, [_VP, _VP, _VP, _I, _VP, _VP, _I, _VP]),
    "kmpc_condense_cost": (_I, [_VP, _VP, _VP, _I, _VP, _VP, _VP, _I, _VP]),
    "kmpc_mpc_solve": (_I, [_VP, _VP, _VP, _VP, _I, _VP, _VP, _I, _D, _D, _D, _D, _DP, _VP, _VP, _VP, _VP, _VP, _I, _VP]),
    "kmpc_qp_solve": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _VP]),
    "kmpc_step": (_I, [_VP, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP]),
    "kmpc_offline_fit": (_I, [_VP, _VP, _VP, _VP, _I, _D, _I, _VP, _VP, _VP, _VP]),
    "kmpc_generate_and_fit": (_I, [_VP, _I, _VP, _VP, _I, _I, _D, _D, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "kmpc_gram_elems": (_I64, [_VP]),
    "kmpc_allreduce_gram": (_I, [_VP, _VP, _VP, _VP]),
    "kmpc_shared_local_gram": (_I, [_VP, _VP, _VP, _VP]),
    "kmpc_gram_accumulate": (_I, [_VP, _VP, _VP, _VP]),
    "kmpc_shared_solve": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "kmpc_shared_solve_plant": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP, _I, _D, _VP]),
    "kmpc_shared_rollout": (_I, [_VP, _I, _VP, _VP, _I, _I, _I, _D, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "kmpc_shared_get_model": (_I, [_VP, _VP, _VP, _VP, _VP]),
    "kmpc_set_applied_input": (_I, [_VP, _VP, _I, _VP]),
    "kmpc_set_online_update": (_I, [_VP, _I]),
    "kmpc_plant_step": (_I, [_VP, _I, _VP, _VP, _D, _I, _I, _VP]),
    "kmpc_rollout": (_I, [_VP, _I, _VP, _VP, _I, _I, _I, _I, _D, _VP, _VP, _VP, _VP, _VP]),
    "kmpc_state_bytes": (_I64, [_VP]),
    "kmpc_state_export": (_I, [_VP, _VP, _I64]),
    "kmpc_state_import": (_I, [_VP, _VP, _I64]),
    "kmpc_profile_enable": (_I, [_VP, _I]),
    "kmpc_profile_read": (_I, [_VP, _DP, C.POINTER(_I64), _I]),
    "kmpc_algorithmic_bytes_per_step": (_I64, [_VP]),
}

_lib = None


def load():
    """Load libkoopmpc.so (once).  Raises if it has not been built -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: libkoopmpc.so and torch must share ONE HIP runtime (torch's bundled
    # libamdhip64 has the same SONAME; whichever is loaded first serves both), otherwise the
    # device pointers torch hands out are not valid for our launches.
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "koopmpc: %s is missing. Build the HIP library first (python __graft_entry__.py, or "
            "make -C koopman-online-updated-mpc_amd); koopmpc has no CPU path." % LIB_PATH
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export the symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class KmpcError(RuntimeError):
    pass


def check(lib, handle, rc, what):
    if rc != 0:
        msg = lib.kmpc_last_error(handle)
        raise KmpcError("%s failed (%d): %s" % (what, rc, (msg or b"").decode("utf-8", "replace")))
