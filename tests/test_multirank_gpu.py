"""The N>1 path end to end on ONE GPU: two processes (torch.distributed.run, gloo) share the card, so RCCL
cannot be used (it rejects two ranks on one device) and the host-staged transport (device -> host -> gloo
isend/irecv -> host -> device in place of ncclSend/ncclRecv; host all-reduce in place of ncclAllReduce) carries
the halo and the reductions.  Pack/unpack kernels, contiguity shortcuts, offsets and the unpack order are the
RCCL path's own.  Everything else -- MatCreateMPIAIJWithArrays, diagonal/off-diagonal SpMV, compressed-row
off-diagonal block, MatMultTranspose with reverse/ADD scatter, parallel dots and norms, KSPCG over MPI vectors --
is the code the 8-GPU run executes.  Results are compared with the sequential oracle on every rank."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("nranks", [2, 3])
def test_mpiaij_two_ranks_one_gpu(built, nranks):
    env = dict(os.environ, MI355X_STAGED="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
           "--master-addr", "127.0.0.1", "--master-port", str(29520 + nranks), os.path.join(ROOT, "tests", "tools", "rank2_trial.py"), "12"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    for k in range(nranks):
        assert "rank %d/%d: transport=host-staged rccl_ranks=0 rccl_communicators=0" % (k, nranks) in out, out[-3000:]
        assert "rank %d/%d: MatMult bitexact=True MatMultTranspose=True norm=True" % (k, nranks) in out, out[-3000:]
        assert "rank %d/%d: irregular MatMult bitexact=True MatMultTranspose bitexact=True" % (k, nranks) in out, out[-3000:]
        assert "rank %d/%d: irregular MatMult through the column-tiled kernel bitexact=True MatMultTranspose bitexact=True" % (k, nranks) in out, out[-3000:]
        assert "rank %d/%d: plain KSPSolve_CG with the update calls deferred == launched one by one: True" % (k, nranks) in out, out[-3000:]
        assert "rank %d/%d: VecScatter INSERT / ADD / MAX, forward and reverse, bitexact=True" % (k, nranks) in out, out[-3000:]
        assert "rank %d/%d: MatDiagonalScale + MatScale then MatMult bitexact=True" % (k, nranks) in out, out[-3000:]
        if nranks == 3:   # the reference's 3-rank golden of mat/tests/ex5 with -test_diagonalscale
            assert "rank %d/3: golden ex5_33.out (MatMult, MatMultTranspose, MatGetDiagonal, MatDiagonalScale of MPIAIJ, 3 ranks) ok=True" % k in out, out[-3000:]
        if nranks == 2:   # the reference's own 2-rank golden, default preconditioner (block Jacobi + ILU(0))
            assert "rank %d/2: golden ex2_2.out (GMRES + bjacobi + ILU(0), 2 ranks) its=7 ok=True" % k in out, out[-3000:]
            assert "rank %d/2: golden ex5_5.out (two systems, GMRES + bjacobi + ILU(0), 2 ranks) ok=True" % k in out, out[-3000:]
            assert "rank %d/2: golden ex16_1.out (four right-hand sides, one KSP, GMRES + bjacobi + ILU(0), 2 ranks) ok=True" % k in out, out[-3000:]
            assert "rank %d/2: golden ex40.out (default GMRES, PCNONE, 2 ranks) ok=True" % k in out, out[-3000:]
            assert "rank %d/2: golden ex7_1.out (block Jacobi, a different solver on every block, 2 ranks) ok=True" % k in out, out[-3000:]


def test_mpiaij_four_ranks_in_the_partition_of_configs2(built):
    """BASELINE.json configs[2] (P7(512) in 8 z-slabs over RCCL) rehearsed in ITS partition shape on the one GPU a box has:
    planes twice as wide, a quarter as many per rank (P7(2n, 2n, n/4 * ranks)), so that interior ranks exchange a plane with
    TWO neighbours and the edge ranks with one -- over the host-staged transport, with FOUR ranks (the GPU boxes admit at most
    six processes on a card, and the test runner and the launcher are two of them; the 8-rank partition itself is checked on the
    CPU by test_distributed_gloo.py, and over RCCL by test_rccl_multigpu.py where eight GPUs exist).  Every rank: MatMult and MatMultTranspose bit-exact against the
    MPIAIJ-ordered oracle, norms, CG + Jacobi history against the oracle."""
    nranks = 4
    env = dict(os.environ, MI355X_STAGED="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
           "--master-addr", "127.0.0.1", "--master-port", "29527", os.path.join(ROOT, "tests", "tools", "rank2_trial.py"), "8", "cfg3"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    for k in range(nranks):
        assert "rank %d/%d: transport=host-staged rccl_ranks=0 rccl_communicators=0" % (k, nranks) in out, out[-3000:]
        assert "rank %d/%d: MatMult bitexact=True MatMultTranspose=True norm=True" % (k, nranks) in out, out[-3000:]
        assert "rank %d/%d: irregular MatMult bitexact=True MatMultTranspose bitexact=True" % (k, nranks) in out, out[-3000:]


@pytest.mark.parametrize("wide", [False, True])
def test_bench_two_ranks_rehearsal(built, wide):
    """bench.py's N>1 flow (torch.distributed.run launch, z-slab MatMPIAIJ, max-over-ranks timing, one JSON line from
    rank 0) rehearsed with two ranks on the one GPU through the host-staged transport (MI355X_STAGED=1)."""
    import json
    env = dict(os.environ, MI355X_STAGED="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29534" if wide else "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--grid-n", "40"]
    if wide:                      # the slab shape bench.py uses on 8 GPUs (planes twice as wide, a quarter as many per rank)
        cmd.append("--wide-planes")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["config"]["rows_per_gpu"] == 40 ** 3 and d["value"] > 0 and d["roofline"]["avg_launch_ms"] > 0
    assert "rowblock_pat_kernel" in d["roofline_spmv"]["kernel"]     # the headline streams the value array of the slab's diagonal block (row-pattern kernel)
    assert "cghipmi355x" in d["config"]["workload"] and 0 < d["roofline"]["frac"] < 1 and 0 < d["ksp_hbm_frac"] < 1
    assert d["roofline_spmv"]["avg_launch_ms"] > 0 and d["roofline_cg_update"]["avg_launch_ms"] > 0 and d["roofline_cg_update"]["launches_timed"] == 6
    assert d["config"]["transport"] == "host-staged" and "HOST-STAGED" in d["config"]["workload"]
    assert ("P7(80,80,20)" if wide else "P7(40,40,80)") in d["config"]["workload"]


def test_bench_started_plainly_with_gpus_2_starts_two_ranks(built):
    """`python3 bench.py --gpus 2` with NO launcher around it (the shape of the driver's bench command): the process starts the two
    ranks itself before touching the GPU, relays rank 0's line and reports n_gpus == 2 -- never the one-GPU path.  Two ranks on the
    one card of this box: the host-staged rehearsal transport has to be asked for (MI355X_STAGED=1); without it the run refuses."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MI355X_STAGED="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--grid-n", "40"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["transport"] == "host-staged" and "HOST-STAGED" in d["config"]["workload"] and "P7(40,40,80)" in d["config"]["workload"]
    mg = d["multi_gpu"]                               # the exchange step's own figures
    assert mg["halo_bytes_per_neighbour_per_spmv"] == 8 * 40 * 40 and mg["halo_neighbours_busiest_rank"] == 1 and mg["spmv_products_timed"] == 6
    assert mg["halo_stream_busy_ms_per_spmv"] > 0 and 0.0 <= mg["halo_overlap_frac"] <= 1.0 and mg["halo_exposed_ms_per_spmv"] >= 0
    assert len(mg["halo_ms_per_rank"]) == 2 and mg["transport"] == "host-staged"
    s = d["strong"]                                   # the strong-scaling point rides along: the cube P7(80) in two slabs
    assert s["scaling"] == "strong" and s["n_gpus"] == 2 and s["rows_total"] == 80 ** 3 and s["rows_per_gpu"] == 80 ** 3 // 2 and s["value"] > 0
    assert s["multi_gpu"]["halo_bytes_per_neighbour_per_spmv"] == 8 * 80 * 80
    ck = d["checks"]                                  # the N-rank line checks itself: the halo-carrying product against the row sums on every rank
    assert ck["mpiaij_spmv_times_ones_equals_row_sums_on_every_rank"] is True and 0.0 < ck["true_residual_after_K_steps"] < 1.0
    # not asked for: two ranks on one card are refused, with no line at all
    env.pop("MI355X_STAGED")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")], (r.stdout + r.stderr)[-2000:]
    assert "MI355X_STAGED=1" in (r.stdout + r.stderr)
