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


def test_bench_gpus_2_started_plainly_never_reports_one_gpu(built):
    """`python bench.py --gpus 2` (no torch.distributed.run around it: the shape of the driver's N=1 command with another N) must
    start two ranks itself or fail; it must never fall through to the one-GPU path and print an n_gpus: 1 line.  Here there is no
    GPU: the two ranks start (gloo rendezvous on 127.0.0.1), find no device and exit non-zero; the parent relays that."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MI355X_STAGED")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid-n", "8", "--steps", "1", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0, (r.stdout + r.stderr)[-2000:]
    assert "starting 2 ranks" in r.stderr
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            assert json.loads(line).get("n_gpus") != 1, line[:300]
    # and a world size that contradicts --gpus is refused, not reinterpreted
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--grid-n", "8"],
                       env=dict(env, WORLD_SIZE="2", RANK="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE=2 but --gpus 4" in (r.stdout + r.stderr)
