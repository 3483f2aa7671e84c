"""N>1 host logic over real processes and the gloo backend (CPU, world_size 2, 4 and -- in the partition shape of
BASELINE.json configs[2] -- 8)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 4])
def test_mpiaij_setup_gloo(built, world):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29540 + world), os.path.join(ROOT, "tests", "tools", "gloo_setup_check.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="1"))
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    for k in range(world):
        assert "rank %d/%d: MPIAIJ set-up matches the oracle" % (k, world) in out, out[-3000:]


def test_mpiaij_setup_gloo_eight_ranks_in_the_partition_of_configs2(built):
    """BASELINE.json configs[2] = P7(512) in 8 z-slabs: every rank owns whole planes, interior ranks exchange one plane with
    each of two neighbours, ranks 0 and 7 with one.  The same partition at reduced size (P7(16,16,16): 2 planes of 256 rows per
    rank) over 8 real processes: diagonal / off-diagonal split, garray, compacted B.j and the scatter's to / from lists equal the
    oracle's on every rank, ghost counts and neighbour counts as SURVEY 8(d) states them."""
    world = 8
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", "29558", os.path.join(ROOT, "tests", "tools", "gloo_setup_check.py"), "cfg3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, OMP_NUM_THREADS="1"))
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    for k in range(world):
        nb = 1 if k in (0, world - 1) else 2
        assert "rank %d/%d: MPIAIJ set-up matches the oracle (ec=%d, %d send / %d recv neighbours)" % (k, world, nb * 256, nb, nb) in out, out[-3000:]
