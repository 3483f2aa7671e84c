"""N>1 host logic over real processes and the gloo backend (CPU, world_size 2 and 4)."""
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
