"""Does our RCCL transport work in a process that has imported torch (ROCm build, which ships its own librccl) and initialised a
gloo group first -- the order bench.py uses for N > 1?  One rank, one GPU: communicator creation, all-reduce, send-to-self, and
which librccl the process ended up with."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
dist.init_process_group("gloo")
import petsc_dev_amd as pda  # noqa: E402
from petsc_dev_amd import dist as PD  # noqa: E402
from petsc_dev_amd import petsc as P  # noqa: E402
L = P.lib(); k = pda.load_kernels()
print("torch", torch.__version__, "cuda available:", torch.cuda.is_available())
uid = C.create_string_buffer(128)
assert k.mi355x_comm_get_unique_id(uid) == 0
for which in ("reductions", "halo"):
    d = C.c_void_p()
    rc = k.mi355x_comm_init_rank(C.byref(d), 1, 0, uid.raw)
    assert rc == 0, k.mi355x_comm_error_string(rc)
    why = PD._rccl_self_test(k, d, 0, 1)
    print(which, "communicator: self test", "ok" if not why else why)
    k.mi355x_comm_destroy(d)
    assert k.mi355x_comm_get_unique_id(uid) == 0
print("librccl mapped:", sorted({l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l}))
dist.destroy_process_group()
