"""Dev tool (GPU box): can RCCL run TWO ranks on the ONE GPU of a pool box?  Two child processes, gloo for the unique-id hand-over,
koopmpc.sharding.NcclCommunicator (ncclCommInitRank with world = 2, both on cuda:0), then ncclAllReduce of a Gram-sized float64 block.
Prints what RCCL answers -- success with the right sums, or its refusal."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, ctypes as C
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "koopman-online-updated-mpc_amd"))
import torch, torch.distributed as dist
rank = int(sys.argv[1])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK=str(rank), WORLD_SIZE="2")
dist.init_process_group("gloo")
from koopmpc.sharding import NcclCommunicator
torch.cuda.set_device(0)
try:
    comm = NcclCommunicator(torch.device("cuda", 0))
except Exception as e:
    print("rank %%d: communicator refused: %%s" %% (rank, e)); sys.exit(3)
x = torch.full((2211,), float(rank + 1), dtype=torch.float64, device="cuda:0")
ar = comm.rccl.ncclAllReduce
ar.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
rc = ar(x.data_ptr(), x.data_ptr(), x.numel(), 8, 0, comm.handle, torch.cuda.current_stream().cuda_stream)  # ncclDouble = 8, ncclSum = 0
torch.cuda.synchronize()
print("rank %%d: ncclAllReduce rc %%d, ranks seen %%d, sum %%s (expected 3.0)" %% (rank, rc, comm.count(), float(x[0])))
comm.destroy(); dist.destroy_process_group()
''' % (ROOT, ROOT)
import socket
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
for k, v in [a.split("=", 1) for a in sys.argv[1:]]:
    env[k] = v
ps = [subprocess.Popen([sys.executable, "-c", CHILD, str(r), str(port)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
for p in ps:
    try:
        out = p.communicate(timeout=120)[0]
    except subprocess.TimeoutExpired:
        p.kill(); out = "timeout (killed)"
    print("\n".join(l for l in out.splitlines() if "amdgpu.ids" not in l)[-1500:], "| exit", p.returncode)
