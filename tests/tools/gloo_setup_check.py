"""world_size-N gloo check of the host-side N>1 logic (no GPU needed): real processes, real collectives.
Every rank builds its MatMPIAIJ piece of a z-slab 7-point operator and compares all integer outputs
(diagonal/off-diagonal split, garray, VecScatter to/from lists, layout) with the sequential oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import petsc_dev_amd  # noqa: F401
    from petsc_dev_amd import petsc as P
    from petsc_dev_amd import dist as PD
    import orc
    from test_host_cpu import mpiaij_pieces, check_against_oracle
    comm = PD.torch_comm(device_comm=False)
    nx, ny, n = 7, 5, 4
    if len(sys.argv) > 1 and sys.argv[1] == "cfg3":
        # BASELINE.json configs[2]'s partition at reduced size: P7(2k, 2k, k/4 * world) in z-slabs of k/4 planes of (2k)^2 rows
        # (k = 256: P7(512) on 8 ranks, 64 planes of 512^2 per rank, 262 144 ghosts per neighbour); here k = 8
        nx, ny, n = 16, 16, 2
    nz = n * world
    mloc = nx * ny * n
    ai, aj, aa = P.gen_poisson7(nx, ny, nz, rank * mloc, (rank + 1) * mloc)
    A = P.Mat.from_csr_mpi(ai, aj, aa, mloc, P.PETSC_DECIDE, P.PETSC_DECIDE, comm=comm)   # global sizes by all-reduce
    got = mpiaij_pieces(P, A.h)
    gi, gj, ga = orc.gen_p7(nx, ny, nz)
    ranges = np.arange(world + 1, dtype=np.int32) * mloc
    # check this rank's piece against the oracle's view of all ranks
    fake = [None] * world
    fake[rank] = got
    ref = [orc.mpiaij_split(int(ranges[r]), int(ranges[r + 1]), int(ranges[r]), int(ranges[r + 1]), gi, gj, ga) for r in range(world)]
    (di, dj, da), (oi, oj, oa), garray, lists = got
    assert np.array_equal(di, ref[rank]["ad_i"]) and np.array_equal(dj, ref[rank]["ad_j"]) and np.array_equal(da, ref[rank]["ad_a"])
    assert np.array_equal(oi, ref[rank]["bo_i"]) and np.array_equal(oj, ref[rank]["bo_j"]) and np.array_equal(oa, ref[rank]["bo_a"])
    assert np.array_equal(garray, ref[rank]["garray"])
    sc = orc.scatter_create(world, rank, ranges, [r["garray"] for r in ref])
    for key in sc:
        assert np.array_equal(lists[key], sc[key]), (rank, key)
    M = P.i32(); N = P.i32()
    P.lib().MatGetSize(A.h, M, N)
    assert M.value == mloc * world and N.value == mloc * world
    if len(sys.argv) > 1 and sys.argv[1] == "cfg3":       # interior ranks: two z-neighbours, one plane of ghosts from each; edge ranks: one
        nb = 1 if rank in (0, world - 1) else 2
        assert garray.size == nb * nx * ny and lists["sprocs"].size == nb and lists["rprocs"].size == nb, (rank, garray.size)
        assert int(oi[-1]) == nb * nx * ny                 # B_o: one entry per ghost column (524 288 / 262 144 at full size)
    print("rank %d/%d: MPIAIJ set-up matches the oracle (ec=%d, %d send / %d recv neighbours)" % (rank, world, garray.size, lists["sprocs"].size, lists["rprocs"].size), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
