#!/usr/bin/env python3
"""bench.py -- KSP CG + Jacobi on the 3-D 7-point Poisson operator, 256^3 rows per GPU
(BASELINE.json configs[1] at N=1; configs[2] = 512^3 in 8 z-slabs at N=8; weak scaling in between).

A step = one CG iteration of the reference's KSPSolve_CG op sequence over the HIPMI355X Vec/Mat types:
1 SpMV (MatMult_SeqAIJ / MatMult_MPIAIJ with RCCL halo), 1 Jacobi apply, 2 dots, 1 norm, 2 axpy, 1 aypx.
value = iterations/s x global unknowns (aggregates over ranks under weak scaling); ksp_its_per_sec and
spmv_gbps carry BASELINE.json's two quantities as absolute numbers.  The two kernels that make up three quarters of an
iteration -- the SpMV and the fused CG update -- are both timed with HIP events on the compute stream inside the timed solve
(roofline_spmv, roofline_cg_update); roofline is whichever of the two took longer.  csr_streaming: the same solve with
the matrix's value array streamed (value patterns switched off), measured in the same run.  cpu_baseline: the oracle's
C restatement of the same solve on the host cores, bounded sample (rank 0, N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--grid-n", dest="n", type=int, default=256, help="grid points per side per GPU (rows per GPU = n^3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-its", type=int, default=20)
    ap.add_argument("--wide-planes", action="store_true", help="development: (2n)x(2n)x(n/4) rows per GPU instead of n^3 (N=8 is then the cube P7(2n))")
    ap.add_argument("--ksp-opts", default="", help="extra options-database string (development: e.g. '-ksp_cg_fused 2')")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    import numpy as np
    dist = None
    if world > 1:
        import torch  # noqa: F401  (device runtime first, then our libraries bind to the same one)
        import torch.distributed as dist
        dist.init_process_group("gloo")
    import petsc_dev_amd as pda  # noqa: F401
    from petsc_dev_amd import petsc as P
    L = P.lib()
    if world > 1:
        from petsc_dev_amd import dist as PD
        # MI355X_STAGED=1: rehearsal of the N>1 flow with several ranks on ONE GPU (RCCL refuses that): host-staged transport
        comm = PD.torch_comm(device_comm=os.environ.get("MI355X_STAGED", "0") != "1")
    else:
        comm = L.COMM_SELF

    if args.ksp_opts:                        # before the matrix exists: -mat_* options are read when it is first used
        L.PetscOptionsInsertString(args.ksp_opts.encode())
    n = args.n
    nx, ny, nz = n, n, n * world            # z-slabs: rank r owns planes [r*n, (r+1)*n)
    if args.wide_planes or (world == 8 and n == 256):
        # same rows per GPU, planes twice as wide: (2n) x (2n) x (n/4) per rank.  On 8 GPUs that is the cube P7(512) =
        # BASELINE.json configs[2] exactly (2 MiB halo per neighbour); measured cost of the wider planes on one GPU: +3 %
        nx, ny, nz = 2 * n, 2 * n, (n // 4) * world
    mloc = nx * ny * (nz // world)
    rs, re_ = rank * mloc, (rank + 1) * mloc
    t0 = time.time()
    ai, aj, aa = P.gen_poisson7(nx, ny, nz, rs, re_)
    if world > 1:
        A = P.Mat.from_csr_mpi(ai, aj, aa, mloc, mloc * world, mloc * world, comm=comm)
    else:
        A = P.Mat.from_csr(ai, aj, aa, comm=comm)
    nnz_loc = int(aj.size)
    u = P.Vec.create(mloc, N=mloc * world, comm=comm)
    L.VecSet(u.h, 1.0)
    b, x = u.duplicate(), u.duplicate()
    A.mult(u, b)                                         # b = A * 1
    setup_s = time.time() - t0

    ksp = P.KSP(comm=comm)
    ksp.set_operators(A)
    ksp.set_type("cg")
    ksp.set_pc_type("jacobi")
    if args.ksp_opts:
        ksp.set_from_options()

    # the dominant kernel: SpMV of the (diagonal block of the) matrix
    if world > 1:
        Ad = C.c_void_p()
        L.MatMPIAIJGetSeqAIJ(A.h, C.byref(Ad), None, None)
        m_, i_, j_, a_ = C.c_int(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        L.MatSeqAIJGetArrays(Ad, C.byref(m_), C.byref(i_), C.byref(j_), C.byref(a_))
        nnz_k = int(np.ctypeslib.as_array(C.cast(i_, C.POINTER(C.c_int)), (mloc + 1,))[mloc])
        timed = Ad
    else:
        nnz_k = nnz_loc
        timed = A.h
    spmv_bytes = 12 * nnz_k + 4 * (mloc + 1) + 8 * mloc + 8 * mloc

    def barrier():
        if dist is not None:
            dist.barrier()

    k = pda.load_kernels()

    def timed_solve():
        """W untimed iterations (the first call also uploads the matrix and builds the Jacobi diagonal), then exactly K timed ones"""
        ksp.set_tolerances(rtol=0.0, abstol=1e-300, dtol=1e300, max_it=max(args.warmup, 1))
        ksp.solve(b, x)
        k.mi355x_device_synchronize()
        ev = 0 if os.environ.get("BENCH_NO_SPMV_EVENTS") else 1                      # (development: cost of the event pairs)
        L.MatHIPMI355XSetTiming(timed, ev)
        L.VecHIPMI355XSetCGUpdateTiming(ev)
        ksp.set_tolerances(rtol=0.0, abstol=1e-300, dtol=1e300, max_it=args.steps)
        barrier()
        k.mi355x_device_synchronize()
        t0 = time.perf_counter()
        ksp.solve(b, x)
        k.mi355x_device_synchronize()
        barrier()
        dt = time.perf_counter() - t0
        assert ksp.its == args.steps, "solver stopped after %d of %d iterations (reason %d)" % (ksp.its, args.steps, ksp.reason)
        nl, tms, nu, ums = C.c_int(), C.c_double(), C.c_int(), C.c_double()
        L.MatHIPMI355XGetTiming(timed, C.byref(nl), C.byref(tms))
        L.VecHIPMI355XGetCGUpdateTiming(C.byref(nu), C.byref(ums))
        L.MatHIPMI355XSetTiming(timed, 0)
        L.VecHIPMI355XSetCGUpdateTiming(0)
        spmv_ms = tms.value / max(nl.value, 1)
        upd_ms = ums.value / max(nu.value, 1)
        if dist is not None:
            import torch
            t = torch.tensor([dt, spmv_ms, upd_ms], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt, spmv_ms, upd_ms = float(t[0]), float(t[1]), float(t[2])
        return dt, spmv_ms, nl.value, upd_ms, nu.value

    dt, spmv_ms, spmv_launches, upd_ms, upd_launches = timed_solve()

    its_per_s = args.steps / dt
    unknowns = mloc * world
    value = its_per_s * unknowns / 1e6
    if os.environ.get("BENCH_NO_SPMV_EVENTS"):
        print("no-events run: %.5f ms/step" % (dt / args.steps * 1e3), flush=True)
        return
    spmv_gbps_one = spmv_bytes / (spmv_ms * 1e-3) / 1e9
    cg_bytes = spmv_bytes + 136 * mloc                                   # unfused CG+Jacobi op sequence
    noff = C.c_int(0)
    L.MatHIPMI355XGetIndexCompression(timed, C.byref(noff))
    npat = C.c_int()
    L.MatHIPMI355XGetRowPatterns(timed, C.byref(npat))
    nvpat = C.c_int()
    L.MatHIPMI355XGetValuePatterns(timed, C.byref(nvpat))

    def streamed_kernel_name():
        if npat.value:
            return "spmv_csr_rowblock_pat_kernel (CSR values + a %d-list row-pattern dictionary: one 4-byte word per row instead of column indices and row pointer; 'achieved' uses the CSR algorithmic bytes)" % npat.value
        if noff.value:
            return "spmv_csr_rowblock_idx8_kernel (CSR + 1-byte offset-dictionary column indices, %d offsets; 'achieved' uses the CSR algorithmic bytes)" % noff.value
        return "spmv_csr_rowblock_kernel"

    if nvpat.value:
        kernel_name = ("spmv_csr_valpat_kernel (constant-coefficient operator: %d distinct rows {offsets, values} in a dictionary, 2 bytes per row, "
                       "the value array is not read; 'achieved' uses the CSR algorithmic bytes)" % nvpat.value)
    elif npat.value:
        kernel_name = "spmv_csr_rowblock_pat_kernel (CSR values + a %d-list row-pattern dictionary: one 4-byte word per row instead of column indices and row pointer; 'achieved' uses the CSR algorithmic bytes)" % npat.value
    elif noff.value:
        kernel_name = "spmv_csr_rowblock_idx8_kernel (CSR + 1-byte offset-dictionary column indices, %d offsets; 'achieved' uses the CSR algorithmic bytes)" % noff.value
    else:
        kernel_name = "spmv_csr_rowblock_kernel"
    if world > 1:
        tr = PD.transport_report(comm)                   # asked of the C library: what the halo and the reductions travelled over
    else:
        tr = {"transport": "single", "rccl_ranks": 0, "rccl_communicators": 0}
    staged = world > 1 and tr["transport"] != "rccl"
    if world > 1 and not staged:
        assert tr["rccl_ranks"] == world, "RCCL reports %d ranks, launched %d" % (tr["rccl_ranks"], world)
    out = {
        "metric": "KSP CG+Jacobi iterations/s x unknowns (3-D 7-pt Poisson, %d^3 rows per GPU); ksp_its_per_sec and spmv_gbps are BASELINE.json's two quantities" % n,
        "value": round(value, 3), "unit": "Mdof-it/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "3D 7-pt Poisson P7(%d,%d,%d) = %d rows, %s, KSPCG + PCJACOBI, b = A*1, x0 = 0, exactly K iterations"
                   % (nx, ny, nz, unknowns, "MatSeqAIJ on 1 GPU" if world == 1 else "MatMPIAIJ in %d z-slabs, %s" % (world, "HOST-STAGED halo and reductions (one-GPU rehearsal, not a measurement)" if staged else "RCCL halo")),
                   "rows_per_gpu": mloc, "nnz_per_gpu": nnz_loc, "parallelism": "row-block dp%d" % world,
                   "transport": tr["transport"], "rccl_ranks": tr["rccl_ranks"], "rccl_communicators": tr["rccl_communicators"]},
        "ksp_its_per_sec": round(its_per_s, 2),
        "spmv_gbps": round(spmv_gbps_one * world, 1),
        "ksp_gbps": round(cg_bytes * its_per_s * world / 1e9, 1),
        "ksp_hbm_frac": round(cg_bytes * its_per_s / 8e12, 4),
        "ksp_gbps_basis": "SURVEY 8(d) algorithmic bytes of the reference's op-by-op iteration (SpMV + 17 vector passes); the fused CG update moves 13 passes",
        "setup_s": round(setup_s, 2),
    }
    BASIS = ("achieved/frac price the launch in the reference's bytes (SURVEY 8d: 12 B per nonzero + 4 B per row + x + y for the CSR "
             "product; the op-by-op vector passes for the update); a kernel that moves fewer bytes than the reference's sequence can "
             "exceed 1.0 on that basis -- traffic / traffic_frac are the bytes it really moved")
    # the SpMV (the kernel SURVEY 8d prices) ...
    r_spmv = {"bound": "hbm", "kernel": kernel_name, "achieved": round(spmv_gbps_one, 1), "peak": 8000.0,
              "unit": "GB/s", "frac": round(spmv_gbps_one / 8000.0, 4), "traffic": None, "basis": BASIS,
              "algorithmic_bytes_per_launch": spmv_bytes, "avg_launch_ms": round(spmv_ms, 5), "launches_timed": spmv_launches}
    if nvpat.value:
        r_spmv["limiter"] = ("not HBM: with the value array out of the way the launch moves ~0.36 GB (ideal 0.30: x once, y once, 2 B per row) and is bound "
                             "by the latency chain row word -> table -> gathers -> store at full occupancy (DESIGN.md section 4, 'Value patterns'); traffic_frac says how far from the memory roof it runs")
    # ... and the fused CG update (VecAXPY, VecAXPY, PCApply_Jacobi, VecNorm, VecTDot of cg.c:206-232 in one sweep): the
    # reference's five operations make 12 vector passes, the kernel 8 (reads x p r w d, writes x r z)
    upd_bytes = 12 * 8 * mloc
    upd_gbps = upd_bytes / (upd_ms * 1e-3) / 1e9 if upd_ms > 0 else 0.0
    r_upd = {"bound": "hbm", "kernel": "reduce_kernel<4, 0, CGUpdateDevF> (x += a p, r -= a w, z = r .* d, z'z, z'r, r'r in one sweep; 'achieved' uses the "
                                       "12 vector passes of the reference's five operations, the kernel makes 8)",
             "achieved": round(upd_gbps, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(upd_gbps / 8000.0, 4), "traffic": None, "basis": BASIS,
             "algorithmic_bytes_per_launch": upd_bytes, "avg_launch_ms": round(upd_ms, 5), "launches_timed": upd_launches}

    # HBM traffic of the two kernels from the committed rocprofv3 PMC passes of this command (separate
    # FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE x2 on gfx950, calibration in profiles/r01_fetch_calibration.md)
    try:
        import csv
        import hashlib
        cnt = {}
        pmc_csv = os.path.join(ROOT, "profiles", "bench_pmc_summary.csv")
        # the counters are only quoted while they describe THESE kernels: the summary's first line carries the hashes of the
        # kernel sources it was collected with (tests/tools/pmc_summary.py --stamp); a stale file is ignored
        head = open(pmc_csv).readline()
        for src in ("spmv_csr.hip", "vec_kernels.hip"):
            h16 = hashlib.sha256(open(os.path.join(ROOT, "petsc-dev_amd", "csrc", src), "rb").read()).hexdigest()[:16]
            if ("%s sha256/16 = %s" % (src, h16)) not in head:
                raise RuntimeError("stale PMC summary")
        spmv_tag = "spmv_csr_valpat_kernel<0>" if nvpat.value else "spmv_csr_rowblock"
        with open(pmc_csv) as f:
            f.readline()
            for row in csv.DictReader(f):
                kn = row["kernel"]
                if spmv_tag in kn and ("kernel<0," in kn or "kernel<0>" in kn):                # the y = A x instantiation (ADD == 0)
                    cnt[("spmv", row["counter"])] = float(row["avg_value_KB"])
                elif "CGUpdateDevF" in kn:
                    cnt[("upd", row["counter"])] = float(row["avg_value_KB"])
        src_note = "profiles/bench_pmc_summary.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of this command with this kernel source; bytes per launch)"
        if n == 256 and world == 1:
            for key, r in (("spmv", r_spmv), ("upd", r_upd)):
                if (key, "FETCH_SIZE") in cnt and (key, "WRITE_SIZE") in cnt and r["avg_launch_ms"] > 0:
                    r["traffic"] = int((2.0 * cnt[(key, "FETCH_SIZE")] + cnt[(key, "WRITE_SIZE")]) * 1024)
                    # what the kernel really moved per second: frac uses the reference's algorithmic bytes as the contract asks,
                    # traffic_frac the measured bytes
                    r["traffic_gbps"] = round(r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9, 1)
                    r["traffic_frac"] = round(r["traffic_gbps"] / 8000.0, 4)
                    r["kernel"] = r["kernel"].replace("'achieved' uses the", "'traffic' is what the kernel actually moved, 'achieved' uses the")
                    r["traffic_source"] = src_note
            if r_spmv["traffic"] is not None:
                # the whole iteration in bytes really moved: the SpMV's measured traffic + the 13 vector passes of the fused CG
                # iteration (AYPX 3, dot 2, fused update 8), next to ksp_gbps / ksp_hbm_frac which price the reference's op-by-op bytes
                moved = r_spmv["traffic"] + 13 * 8 * mloc
                out["ksp_moved_gbps"] = round(moved / (out["ms_per_step"] * 1e-3) / 1e9, 1)
                out["ksp_moved_frac"] = round(out["ksp_moved_gbps"] / 8000.0, 4)
    except Exception:
        pass
    # roofline = the dominant kernel of the step: whichever of the two took longer per launch (one launch of each per step)
    dom = r_upd if upd_ms > spmv_ms else r_spmv
    out["roofline"] = dict(dom)
    out["roofline"]["dominant"] = ("the fused CG update: %.4f ms per step against %.4f ms for the SpMV" % (upd_ms, spmv_ms)) if dom is r_upd else \
                                  ("the SpMV: %.4f ms per step against %.4f ms for the fused CG update" % (spmv_ms, upd_ms))
    out["roofline_spmv"] = r_spmv
    out["roofline_cg_update"] = r_upd

    # the same solve with the value array streamed (value patterns off): what the path does for an operator with varying
    # coefficients, and the configuration BASELINE.json's spmv_gbps is defined on; same matrix, same run
    if nvpat.value and world == 1:
        L.MatHIPMI355XSetValuePatterns(timed, 0)
        dt2, spmv_ms2, _, upd_ms2, _ = timed_solve()
        L.MatHIPMI355XSetValuePatterns(timed, 1)
        g2 = spmv_bytes / (spmv_ms2 * 1e-3) / 1e9
        out["csr_streaming"] = {"what": "the same %d iterations with -mat_hipmi355x_value_patterns 0: the SpMV streams the value array (%s)" % (args.steps, streamed_kernel_name().split(" (")[0]),
                                "ksp_its_per_sec": round(args.steps / dt2, 2), "value": round(args.steps / dt2 * unknowns / 1e6, 3),
                                "ms_per_step": round(dt2 / args.steps * 1e3, 5), "spmv_avg_launch_ms": round(spmv_ms2, 5),
                                "spmv_gbps": round(g2, 1), "spmv_frac": round(g2 / 8000.0, 4), "cg_update_avg_launch_ms": round(upd_ms2, 5)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import orc
        # the same workload, bounded sample: the first cpu_its iterations of the same solve on one host core
        bref = orc.spmv(ai, aj, aa, np.ones(mloc))
        t0 = time.perf_counter()
        _, _, cits, _ = orc.ksp_solve(ai, aj, aa, bref, ksp="cg", pc="jacobi", rtol=0.0, abstol=1e-300, dtol=1e300, max_it=args.cpu_its)
        cdt = time.perf_counter() - t0
        # all the host cores this job may use, one thread per block of rows (the reference's one-MPI-rank-per-core
        # arrangement inside one process, oracle/cpu_baseline_mt.c)
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        try:                                             # a container's CPU share (cgroup v2 quota), if there is one
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if quota != "max":
                cores = max(1, min(cores, int(int(quota) / int(period))))
        except Exception:
            pass
        cores = min(cores, 16)                           # the GPU box's CPU share for one GPU
        mt_its = max(args.cpu_its, 3 * args.cpu_its)
        mdt, _, _ = orc.cg_jacobi_mt(ai, aj, aa, bref, mt_its, cores)
        out["cpu_baseline"] = {"value": round(mt_its / mdt * unknowns / 1e6, 3), "unit": "Mdof-it/s", "cores": cores, "kind": "port",
                               "its_per_sec": round(mt_its / mdt, 4), "its_per_sec_1core": round(cits / cdt, 4),
                               "sample": "first %d CG+Jacobi iterations of the same P7(%d) solve on %d host threads (one per block of rows, the "
                                         "reference's rank-per-core arrangement in one process; C restatement of the reference CPU path, gcc -O2); "
                                         "its_per_sec_1core: first %d iterations by the sequential oracle on one core" % (mt_its, n, cores, cits)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
