"""The RCCL transport BETWEEN GPUs (grouped ncclSend/ncclRecv halo on the halo stream, ncclAllReduce of the scalar
reductions on the compute stream, each on its own communicator): what replaces VecScatterBegin_1/End_1
(src/vec/vec/utils/vpscat.h:14-233) and MatMult_MPIAIJ's choreography (src/mat/impls/aij/mpi/mpiaij.c:1102-1116)
when every rank has a GPU of its own.  These tests need >= 2 visible devices and skip themselves on a one-GPU box,
where tests/test_multirank_gpu.py runs the same program over the host-staged transport (MI355X_STAGED=1): the
assertions are the same lines, bit for bit."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ndevices():
    import petsc_dev_amd as pda
    n = C.c_int()
    pda.load_kernels().mi355x_device_count(C.byref(n))
    return n.value


def launch(nranks, script, *args, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("MI355X_STAGED", None)                  # the real thing: RCCL over xGMI, one rank per GPU
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(args)
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("nranks", [2, 3, 4, 8])
def test_mpiaij_over_rccl(built, nranks):
    if ndevices() < nranks:
        pytest.skip("needs %d GPUs, %d visible" % (nranks, ndevices()))
    # 8 ranks: the partition shape of BASELINE.json configs[2] (wide planes, interior ranks with two neighbours)
    r = launch(nranks, os.path.join(ROOT, "tests", "tools", "rank2_trial.py"), *(("8", "cfg3") if nranks == 8 else ("12",)), port=29540 + nranks)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "host-staged" not in out, out[-3000:]    # a silent fallback is a failure here
    for k in range(nranks):
        assert "rank %d/%d: transport=rccl rccl_ranks=%d rccl_communicators=2" % (k, nranks, nranks) in out, out[-3000:]
        assert "rank %d/%d: MatMult bitexact=True MatMultTranspose=True norm=True" % (k, nranks) in out, out[-3000:]
        assert "rank %d/%d: irregular MatMult bitexact=True MatMultTranspose bitexact=True" % (k, nranks) in out, out[-3000:]
        assert "rank %d/%d: MatDiagonalScale + MatScale then MatMult bitexact=True" % (k, nranks) in out, out[-3000:]
        if nranks == 2:   # the reference's own 2-rank golden, default preconditioner (block Jacobi + ILU(0))
            assert "rank %d/2: golden ex2_2.out (GMRES + bjacobi + ILU(0), 2 ranks) its=7 ok=True" % k in out, out[-3000:]
            assert "rank %d/2: golden ex5_5.out (two systems, GMRES + bjacobi + ILU(0), 2 ranks) ok=True" % k in out, out[-3000:]


def test_bench_two_gpus_over_rccl(built):
    if ndevices() < 2:
        pytest.skip("needs 2 GPUs, %d visible" % ndevices())
    r = launch(2, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--grid-n", "96", port=29547)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 10 and d["scaling"] == "weak" and d["value"] > 0
    assert "RCCL halo" in d["config"]["workload"] and "HOST-STAGED" not in d["config"]["workload"]
    assert d["config"]["transport"] == "rccl" and d["config"]["rccl_ranks"] == 2 and d["config"]["rccl_communicators"] == 2
