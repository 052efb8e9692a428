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


class NcclCommunicator:
    """An RCCL communicator of this process's own for the native shared-model loop (kmpc_shared_rollout): rank 0 draws the unique id,
    torch.distributed's default process group carries it to the other ranks (any backend: it is 128 bytes), every rank joins with
    ncclCommInitRank.  Without a process group: a one-rank communicator.  The RCCL is the one that lives in the process (torch's own
    copy when it exports the symbols, else the system librccl) -- the one kmpc_shared_rollout resolves ncclAllReduce from."""

    def __init__(self, device=None):
        import ctypes as C

        import torch
        import torch.distributed as dist

        proc = C.CDLL(None)
        try:
            proc.ncclCommInitRank
            self.rccl = proc
        except AttributeError:
            self.rccl = C.CDLL("librccl.so", mode=C.RTLD_GLOBAL)

        class UID(C.Structure):
            _fields_ = [("b", C.c_char * 128)]

        world, rank = 1, 0
        if dist.is_available() and dist.is_initialized():
            world, rank = dist.get_world_size(), dist.get_rank()
        uid = UID()
        if rank == 0:
            rc = self.rccl.ncclGetUniqueId(C.byref(uid))
            if rc != 0:
                raise RuntimeError("ncclGetUniqueId failed with code %d" % rc)
        if world > 1:
            raw = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).clone()
            if dist.get_backend() != "gloo":
                raw = raw.to(device if device is not None else "cuda")
            dist.broadcast(raw, src=0)
            C.memmove(C.byref(uid), bytes(raw.cpu().numpy().tobytes()), 128)
        if device is not None:
            torch.cuda.set_device(device)
        comm = C.c_void_p()
        self.rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UID, C.c_int]
        rc = self.rccl.ncclCommInitRank(C.byref(comm), world, uid, rank)
        if rc != 0:
            raise RuntimeError("ncclCommInitRank failed with code %d" % rc)
        self.handle, self.world, self.rank = comm, world, rank

    def destroy(self):
        import ctypes as C

        if getattr(self, "handle", None):
            self.rccl.ncclCommDestroy.argtypes = [C.c_void_p]
            self.rccl.ncclCommDestroy(self.handle)
            self.handle = None
