"""Multi-GPU layout of the path: trajectories are independent units, so the
batch is cut into contiguous slices, one per rank (one process per GPU); every rank owns the
RLS state of its slice and no collective sits on the step path.  The only exchange is the
max-over-ranks of the timed region in bench.py."""
from __future__ import annotations


def shard_range(total, rank, world):
    """Contiguous slice [lo, hi) of `total` trajectories owned by `rank` of `world`
    (sizes differ by at most one; earlier ranks take the remainder)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def force_collectives():
    """KMPC_FORCE_COLLECTIVES=1: run the collectives of the path on a one-rank process group too (a rehearsal of the RCCL calls on a
    one-GPU box: communicator set-up, dtype, stream order -- `bench.py --force-process-group`)."""
    import os

    return os.environ.get("KMPC_FORCE_COLLECTIVES", "") not in ("", "0")


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float over the default process group (identity without one)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force_collectives()):
        return float(value)
    if dist.get_backend() == "gloo":
        device = None  # (gloo reduces host tensors)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
