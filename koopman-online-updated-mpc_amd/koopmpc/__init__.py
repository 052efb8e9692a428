"""koopmpc -- MI355X-native batched Koopman online-updated MPC step (lift -> RLS-EDMD -> condensed QP -> box-QP).

Host mirror of the reference's call surface over libkoopmpc.so (hand-written HIP, gfx950).
"""
from . import _ffi  # noqa: F401
from .sharding import shard_range, max_over_ranks  # noqa: F401


def __getattr__(name):
    # KoopmanMPC pulls in torch + the HIP library; import it lazily so host-only helpers
    # (sharding, the ctypes signature table) stay importable on a machine without a GPU.
    if name in ("KoopmanMPC", "solve_DARE", "dlqr", "AutoEncoder", "rbf", "costFunction", "Koopman_update", "koopman_update", "mpc_solve", "mpc_step"):
        from . import api

        return getattr(api, name)
    raise AttributeError(name)
