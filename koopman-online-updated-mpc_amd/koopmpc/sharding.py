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


def exchange_unique_id(idb, device=None):
    """Rank 0's 128-byte RCCL unique id to every rank of the default process group, validated: every rank must hold the same bytes
    before anyone calls ncclCommInitRank (which blocks until all ranks have joined -- a rank with another id would hang the job, so a
    mismatch raises on EVERY rank instead).  idb: the id on rank 0, ignored elsewhere.  Returns the 128 bytes."""
    import hashlib
    import os

    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    on_dev = dist.get_backend() != "gloo"
    raw = torch.frombuffer(bytearray(idb if rank == 0 else bytes(128)), dtype=torch.uint8).clone()
    if on_dev:
        raw = raw.to(device if device is not None else "cuda")
    dist.broadcast(raw, src=0)
    got = bytes(raw.cpu().numpy().tobytes())
    if os.environ.get("KMPC_TEST_CORRUPT_UID_RANK") == str(rank):  # (test hook: tests/test_host_cpu.py)
        got = bytes([got[0] ^ 0xFF]) + got[1:]
    h = int.from_bytes(hashlib.sha256(got).digest()[:7], "little")
    mine = torch.tensor([h], dtype=torch.int64)
    if on_dev:
        mine = mine.to(raw.device)
    seen = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(seen, mine)
    hs = [int(v.item()) for v in seen]
    if any(v != hs[0] for v in hs):
        raise RuntimeError("RCCL unique id differs between the ranks (rank %d holds %x; all: %s)" % (rank, h, [hex(v) for v in hs]))
    return got


class NcclCommunicator:
    """An RCCL communicator of this process's own for the native shared-model loop (kmpc_shared_rollout): rank 0 draws the unique id,
    torch.distributed's default process group carries it to the other ranks (any backend: it is 128 bytes), every rank checks that all
    ranks hold the SAME id (a hash all-gather: a rank that joined with another id would hang ncclCommInitRank for everybody) and
    joins with ncclCommInitRank.  Without a process group: a one-rank communicator.  The RCCL is the one that lives in the process
    (torch's own copy when it exports the symbols, else the system librccl) -- the one kmpc_shared_rollout resolves ncclAllReduce from.
    Use `process_communicator()` for the one communicator a process needs.  Teardown is EXPLICIT: `with NcclCommunicator(...) as c:`
    or `destroy()` / `destroy_process_communicator()` -- behind a barrier, with the streams that carry its collectives drained, on every
    rank (ncclCommDestroy: device buffers, proxy threads, xGMI channels).  Garbage collection does NOT destroy a live communicator:
    ranks reach their finalisers at unrelated moments (possibly with collectives still queued, possibly after HIP itself has shut
    down), so `__del__` only warns about the leak."""

    def __init__(self, device=None):
        import ctypes as C

        import torch
        import torch.distributed as dist

        self.handle = None
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
            C.memmove(C.byref(uid), exchange_unique_id(bytes(uid) if rank == 0 else None, device), 128)
        if device is not None:
            torch.cuda.set_device(device)
        comm = C.c_void_p()
        self.rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UID, C.c_int]
        rc = self.rccl.ncclCommInitRank(C.byref(comm), world, uid, rank)
        if rc != 0:
            raise RuntimeError("ncclCommInitRank failed with code %d" % rc)
        self.handle, self.world, self.rank = comm, world, rank
        self.device = None if device is None else torch.device(device)
        n = self.count()
        if n != world:
            self.destroy()
            raise RuntimeError("NcclCommunicator: RCCL reports %d ranks, the process group has %d" % (n, world))

    def count(self):
        """ncclCommCount: the number of ranks RCCL itself sees in this communicator."""
        import ctypes as C

        n = C.c_int(-1)
        self.rccl.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        rc = self.rccl.ncclCommCount(self.handle, C.byref(n))
        if rc != 0:
            raise RuntimeError("ncclCommCount failed with code %d" % rc)
        return int(n.value)

    def destroy(self):
        import ctypes as C

        if getattr(self, "handle", None):
            self.rccl.ncclCommDestroy.argtypes = [C.c_void_p]
            self.rccl.ncclCommDestroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.destroy()
        return False

    def __del__(self):
        if getattr(self, "handle", None):
            try:
                import warnings

                warnings.warn("NcclCommunicator of %d rank(s) was never destroyed: call destroy() / destroy_process_communicator() "
                              "behind a barrier before the process ends" % getattr(self, "world", 1), ResourceWarning)
            except Exception:  # (interpreter shutdown: the warnings machinery may be gone already)
                pass


_process_comm = None


def process_communicator(device=None):
    """THE RCCL communicator of this process (created on first use, shared by every controller: a communicator pins device buffers,
    proxy threads and xGMI channels for as long as it lives -- one per process is all the path needs)."""
    global _process_comm
    if _process_comm is None or _process_comm.handle is None:
        _process_comm = NcclCommunicator(device)
    elif device is not None and _process_comm.device is not None:
        import torch

        want = torch.device(device)
        have = _process_comm.device
        if want.type == "cuda" and have.type == "cuda" and want.index is not None and have.index is not None and want.index != have.index:
            raise RuntimeError("process_communicator: this process's RCCL communicator lives on %s, asked for %s "
                               "(one process drives one GPU; destroy_process_communicator() first)" % (have, want))
    return _process_comm


def destroy_process_communicator():
    global _process_comm
    if _process_comm is not None:
        _process_comm.destroy()
        _process_comm = None
