#!/usr/bin/env python3
"""bench.py -- KSP CG + Jacobi on the 3-D 7-point Poisson operator, 256^3 rows per GPU
(BASELINE.json configs[1] at N=1; configs[2] = 512^3 in 8 z-slabs at N=8; weak scaling in between).

A step = one CG iteration (the reference's KSPSolve_CG recurrence, cg.c:180-281) over the HIPMI355X Vec/Mat types:
1 SpMV (MatMult_SeqAIJ / MatMult_MPIAIJ with RCCL halo), 1 Jacobi apply, 2 dots, 1 norm, 2 axpy, 1 aypx.

Three configurations of the same solve are timed in the same run (N=1; on several GPUs only the first):
  headline          -ksp_type cghipmi355x (the plug-in's registered CG, fused sweeps), the matrix's VALUE ARRAY STREAMED
                    (-mat_hipmi355x_value_patterns 0): what every matrix gets, constant coefficients or not.  value,
                    ms_per_step, roofline, spmv_gbps, ksp_* describe THIS leg.
  value_patterns    the same solver with the library's default for this matrix: P7 has constant coefficients, its rows come
                    from a 27-entry dictionary and the product does not read the value array (the product carries the same bits).
  op_by_op          -ksp_type cg: the plain restatement of PETSc's own KSPSolve_CG -- what an UNCHANGED PETSc program drives over
                    the same types (value array streamed).  The Vec type recognises the update calls of that sequence and runs
                    them as one fused sweep (same bits); ksp_its_per_sec_every_call_its_own_kernel: with that switched off.

Every fraction in the line is BYTES MOVED / time / 8 TB/s: PMC-measured bytes per launch (profiles/bench_pmc_summary.csv,
quoted only while its stamp matches the kernel sources) or, failing that, the kernel's own byte model.  What the same time
would be worth in the reference's algorithmic bytes (CSR 12 B/nnz + ..., 17 vector passes) is reported
separately as effective_gbps_vs_reference_bytes and never as a fraction of the roof.
cpu_baseline: the oracle's C restatement of the same solve on the host cores, bounded sample (rank 0, N=1 only).
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

PEAK_GBPS = 8000.0          # MI355X HBM3E (MI355X_MICROARCH.md)
# the pool's host driver only does dmabuf IPC: without this RCCL's peer-memory set-up between the ranks' GPUs fails
# (hipIpcGetMemHandle: invalid argument); must be in the environment before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


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
    ap.add_argument("--headline-only", action="store_true", help="development: skip the value_patterns and op_by_op legs")
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

    def make_ksp(ksp_type):
        k_ = P.KSP(comm=comm)
        k_.set_operators(A)
        k_.set_type(ksp_type)
        k_.set_pc_type("jacobi")
        if args.ksp_opts:
            k_.set_from_options()
        return k_

    # the kernel the matrix part of a step runs: SpMV of the (diagonal block of the) matrix
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
    spmv_ref_bytes = 12 * nnz_k + 4 * (mloc + 1) + 8 * mloc + 8 * mloc     # the reference CSR's algorithmic bytes

    def barrier():
        if dist is not None:
            dist.barrier()

    k = pda.load_kernels()

    def timed_solve(ksp):
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
        return {"dt": dt, "spmv_ms": spmv_ms, "spmv_launches": nl.value, "upd_ms": upd_ms, "upd_launches": nu.value}

    # ---- which SpMV kernel the analysis chose for this matrix, and what each one moves per launch (its own byte model) ----
    noff, npat, nvpat = C.c_int(0), C.c_int(0), C.c_int(0)
    L.MatHIPMI355XSetValuePatterns(timed, 1)
    ksp_fused = make_ksp("cghipmi355x")
    ksp_fused.set_tolerances(rtol=0.0, abstol=1e-300, dtol=1e300, max_it=1)
    ksp_fused.solve(b, x)                                # first use: upload + analysis
    L.MatHIPMI355XGetIndexCompression(timed, C.byref(noff))
    L.MatHIPMI355XGetRowPatterns(timed, C.byref(npat))
    L.MatHIPMI355XGetValuePatterns(timed, C.byref(nvpat))

    def spmv_kernel(value_patterns, with_dot):
        """(kernel name for the JSON, tag in rocprofv3's kernel names, modelled bytes moved per launch: matrix stream + x once + y once)"""
        vec = 8 * mloc + 8 * mloc
        inst = "<0, true>" if with_dot else "<0, false>"          # the instance that also leaves p'w (the registered CG on one GPU) / the plain product
        also = "; p'w from the same pass" if with_dot else ""
        if value_patterns and nvpat.value:
            return ("spmv_csr_valpat_kernel (%d distinct rows {offsets, values} in a dictionary, 2 bytes per row; the value array is not read%s)" % (nvpat.value, also),
                    "spmv_csr_valpat_kernel" + inst, 2 * mloc + vec)
        if npat.value:
            return ("spmv_csr_rowblock_pat_kernel (CSR values + a %d-list row-pattern dictionary: 8 B per nonzero + one 4-byte word per row%s)" % (npat.value, also),
                    "spmv_csr_rowblock_pat_kernel" + inst, 8 * nnz_k + 4 * mloc + vec)
        if noff.value:
            return ("spmv_csr_rowblock_idx8_kernel (CSR values + 1-byte offset-dictionary column indices, %d offsets)" % noff.value,
                    "spmv_csr_rowblock_idx8_kernel<0,", 9 * nnz_k + 4 * (mloc + 1) + vec)
        return ("spmv_csr_rowblock_kernel (plain CSR)", "spmv_csr_rowblock_kernel<0,", 12 * nnz_k + 4 * (mloc + 1) + vec)

    # ---- PMC-measured bytes per launch, while the committed summary describes THESE kernel sources ----
    pmc = {}
    pmc_note = None
    try:
        import csv
        import hashlib
        pmc_csv = os.path.join(ROOT, "profiles", "bench_pmc_summary.csv")
        head = open(pmc_csv).readline()
        for src in ("spmv_csr.hip", "vec_kernels.hip"):
            h16 = hashlib.sha256(open(os.path.join(ROOT, "petsc-dev_amd", "csrc", src), "rb").read()).hexdigest()[:16]
            if ("%s sha256/16 = %s" % (src, h16)) not in head:
                raise RuntimeError("stale PMC summary")
        if n == 256 and world == 1:
            with open(pmc_csv) as f:
                f.readline()
                for row in csv.DictReader(f):
                    pmc[(row["kernel"], row["counter"])] = float(row["avg_value_KB"])
            pmc_note = ("profiles/bench_pmc_summary.csv: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of this command with these "
                        "kernel sources; bytes per launch = 2 x FETCH_SIZE (gfx950 counts 64-B units as 32) + WRITE_SIZE")
    except Exception:
        pmc = {}

    def pmc_bytes(tag):
        f = [v for (kn, c), v in pmc.items() if tag in kn and c == "FETCH_SIZE"]
        w = [v for (kn, c), v in pmc.items() if tag in kn and c == "WRITE_SIZE"]
        return int((2.0 * f[0] + w[0]) * 1024) if len(f) == 1 and len(w) == 1 else None

    def kernel_roofline(name, tag, model_bytes, ref_bytes, ms, launches):
        """one kernel against the HBM roof in bytes it MOVED (PMC if stamped, else its byte model)"""
        measured = pmc_bytes(tag)
        moved = measured if measured is not None else model_bytes
        gbps = moved / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return {"bound": "hbm", "kernel": name, "achieved": round(gbps, 1), "peak": PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / PEAK_GBPS, 4),
                "traffic": measured, "bytes_moved_per_launch": moved, "bytes_moved_basis": "pmc" if measured is not None else "kernel byte model",
                "bytes_model_per_launch": model_bytes, "avg_launch_ms": round(ms, 5), "launches_timed": launches,
                "reference_algorithmic_bytes_per_launch": ref_bytes,
                "effective_gbps_vs_reference_bytes": round(ref_bytes / (ms * 1e-3) / 1e9, 1) if ms > 0 else 0.0,
                **({"traffic_source": pmc_note} if measured is not None else {})}

    def leg(ksp, value_patterns, vec_passes, what, with_dot=False, host_scalar_update=False):
        """time one configuration; its/s, and the step / its two main kernels in bytes moved"""
        L.MatHIPMI355XSetValuePatterns(timed, 1 if value_patterns else 0)
        t = timed_solve(ksp)
        its = args.steps / t["dt"]
        name, tag, model = spmv_kernel(value_patterns, with_dot)
        r_spmv = kernel_roofline(name, tag, model, spmv_ref_bytes, t["spmv_ms"], t["spmv_launches"])
        step_moved = r_spmv["bytes_moved_per_launch"] + vec_passes * 8 * mloc
        out = {"what": what, "ksp_its_per_sec": round(its, 2), "value": round(its * mloc * world / 1e6, 3), "ms_per_step": round(t["dt"] / args.steps * 1e3, 5),
               "vector_passes_per_step": vec_passes, "bytes_moved_per_step": step_moved,
               "ksp_gbps": round(step_moved * its * world / 1e9, 1), "ksp_hbm_frac": round(step_moved * its / 1e9 / PEAK_GBPS, 4),
               "spmv": r_spmv}
        if t["upd_launches"]:
            # fused CG update: x += a p, r -= a w, z = d .* r, z'z, z'r, r'r in one sweep = 8 vector passes (reads x p r w d, writes x r z);
            # the reference's five calls (VecAXPY x2, PCApply_Jacobi, VecNorm, VecTDot; cg.c:206-232) make 12
            # (the registered solver's sweep takes its step length from device memory: CGUpdateDevF; the sweep the Vec type runs for the
            # plain KSPSolve_CG gets it from the host with the call: CGUpdateF; the same 8 passes)
            kname, ktag = (("reduce_kernel<3, 0, CGUpdateF>", "CGUpdateF>") if host_scalar_update else ("reduce_kernel<4, 0, CGUpdateDevF>", "CGUpdateDevF"))
            out["cg_update"] = kernel_roofline(kname + " (fused CG update, 8 vector passes)", ktag, 8 * 8 * mloc, 12 * 8 * mloc,
                                               t["upd_ms"], t["upd_launches"])
        return out, t

    # ---- leg 1, the headline: registered fused CG, value array streamed ----
    # the registered CG (-ksp_cg_fused 4, its default): on one GPU p'w comes out of the SpMV pass -> AYPX 3 + fused update 8 = 11 vector
    # passes beside the SpMV; on a parallel matrix p'w is its own reduction (2 more passes)
    fused_dot = world == 1
    fused_passes = 11 if fused_dot else 13
    head, th = leg(ksp_fused, False, fused_passes, "-ksp_type cghipmi355x -mat_hipmi355x_value_patterns 0: fused CG sweeps, the SpMV streams the value array"
                   + (" and leaves p'w" if fused_dot else ""), fused_dot)
    if os.environ.get("BENCH_NO_SPMV_EVENTS"):
        print("no-events run: %.5f ms/step" % head["ms_per_step"], flush=True)
        return
    its_per_s = head["ksp_its_per_sec"]
    unknowns = mloc * world
    if world > 1:
        tr = PD.transport_report(comm)                   # asked of the C library: what the halo and the reductions travelled over
    else:
        tr = {"transport": "single", "rccl_ranks": 0, "rccl_communicators": 0}
    staged = world > 1 and tr["transport"] != "rccl"
    if world > 1 and not staged:
        assert tr["rccl_ranks"] == world, "RCCL reports %d ranks, launched %d" % (tr["rccl_ranks"], world)
    cg_ref_bytes = spmv_ref_bytes + 136 * mloc           # the reference's unfused CG+Jacobi iteration (SpMV + 17 vector passes)
    out = {
        "metric": "KSP CG+Jacobi iterations/s x unknowns (3-D 7-pt Poisson, %d^3 rows per GPU); ksp_its_per_sec and spmv_gbps are BASELINE.json's two quantities" % n,
        "value": head["value"], "unit": "Mdof-it/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "3D 7-pt Poisson P7(%d,%d,%d) = %d rows, %s, CG + PCJACOBI (-ksp_type cghipmi355x), value array streamed, b = A*1, x0 = 0, exactly K iterations"
                   % (nx, ny, nz, unknowns, "MatSeqAIJ on 1 GPU" if world == 1 else "MatMPIAIJ in %d z-slabs, %s" % (world, "HOST-STAGED halo and reductions (one-GPU rehearsal, not a measurement)" if staged else "RCCL halo")),
                   "rows_per_gpu": mloc, "nnz_per_gpu": nnz_loc, "parallelism": "row-block dp%d" % world,
                   "transport": tr["transport"], "rccl_ranks": tr["rccl_ranks"], "rccl_communicators": tr["rccl_communicators"]},
        "ksp_its_per_sec": its_per_s,
        "spmv_gbps": round(head["spmv"]["achieved"] * world, 1),
        "spmv_gbps_basis": "bytes the SpMV kernel moved (%s) / its average launch time inside the timed solve, summed over GPUs" % head["spmv"]["bytes_moved_basis"],
        "ksp_gbps": head["ksp_gbps"], "ksp_hbm_frac": head["ksp_hbm_frac"],
        "ksp_gbps_basis": "bytes moved per step = SpMV (as above) + %d vector passes of the fused iteration (AYPX 3, fused update 8%s)" % (fused_passes, "" if fused_dot else ", dot 2"),
        "effective_gbps_vs_reference_bytes": {"spmv": round(head["spmv"]["effective_gbps_vs_reference_bytes"] * world, 1),
                                              "ksp": round(cg_ref_bytes * its_per_s * world / 1e9, 1),
                                              "what": "the same times priced in the REFERENCE's algorithmic bytes (SURVEY 8d: CSR 12 B/nnz + 4 B/row + x + y = %d B per SpMV; "
                                                      "SpMV + 17 vector passes = %d B per op-by-op iteration); not a fraction of any roof" % (spmv_ref_bytes, cg_ref_bytes)},
        "setup_s": round(setup_s, 2),
    }
    # roofline = the dominant kernel of the headline step: whichever of the SpMV and the fused update took longer per launch
    upd = head.get("cg_update")
    dom_is_upd = bool(upd) and upd["avg_launch_ms"] > head["spmv"]["avg_launch_ms"]
    out["roofline"] = dict(upd if dom_is_upd else head["spmv"])
    out["roofline"]["dominant"] = "%s: %.4f ms per step against %.4f ms for %s" % (
        ("the fused CG update", upd["avg_launch_ms"], head["spmv"]["avg_launch_ms"], "the SpMV") if dom_is_upd else
        ("the SpMV", head["spmv"]["avg_launch_ms"], upd["avg_launch_ms"] if upd else 0.0, "the fused CG update"))
    out["roofline_spmv"] = head["spmv"]
    if upd:
        out["roofline_cg_update"] = upd
    out["legs"] = {"headline": {kk: vv for kk, vv in head.items() if kk not in ("spmv", "cg_update")}}

    if world == 1 and not args.headline_only:
        # ---- leg 2: the library's default for THIS matrix (constant coefficients: rows from a dictionary) ----
        if nvpat.value:
            vp, _ = leg(ksp_fused, True, fused_passes, "-ksp_type cghipmi355x, value patterns on (library default; P7 is a constant-coefficient operator): the SpMV does not read the value array; "
                        "the same products bit for bit, p'w summed per workgroup (iterates agree with the headline's to rounding)", fused_dot)
            out["legs"]["value_patterns"] = vp
        # ---- leg 3: what an unchanged PETSc program drives: KSPSolve_CG op by op (VecAYPX, MatMult, VecTDot, VecAXPY x2, PCApply, VecNorm, VecTDot) ----
        # The Vec type notes KSPSolve_CG's update calls instead of launching them one by one and runs them as the fused sweep when the norm
        # is asked for (host/vechip.c, "deferred element-wise operations"; same bits): SpMV + AYPX 3 + TDot(p,w) 2 + sweep 8 = 13 vector
        # passes.  With -vec_hipmi355x_defer 0 every call is a kernel of its own: 17 passes.
        ksp_plain = make_ksp("cg")
        setdef = L.raw("VecHIPMI355XSetDeferral")
        setdef(1)
        ob, _ = leg(ksp_plain, False, 13, "-ksp_type cg: the plain KSPSolve_CG call sequence of an unchanged program (value array streamed); the Vec type recognises its "
                    "update calls (VecAXPY x2, PCApply_Jacobi, VecNorm, VecTDot) and runs them as one fused sweep: SpMV + 13 vector passes", host_scalar_update=True)
        setdef(0)
        t1 = timed_solve(ksp_plain)
        ob["ksp_its_per_sec_every_call_its_own_kernel"] = round(args.steps / t1["dt"], 2)      # -vec_hipmi355x_defer 0: 17 vector passes
        setdef(1)
        if nvpat.value:
            L.MatHIPMI355XSetValuePatterns(timed, 1)
            t2 = timed_solve(ksp_plain)
            ob["ksp_its_per_sec_with_value_patterns"] = round(args.steps / t2["dt"], 2)
        setdef(-1)
        out["legs"]["op_by_op"] = ob
        L.MatHIPMI355XSetValuePatterns(timed, 1)

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
