#!/usr/bin/env python3
"""bench.py -- KSP CG + Jacobi on the 3-D 7-point Poisson operator, 256^3 rows per GPU
(BASELINE.json configs[1] at N=1; configs[2] = 512^3 in 8 z-slabs at N=8; weak scaling in between).

A step = one CG iteration (the reference's KSPSolve_CG recurrence, cg.c:180-281) over the HIPMI355X Vec/Mat types:
1 SpMV (MatMult_SeqAIJ / MatMult_MPIAIJ with RCCL halo), 1 Jacobi apply, 2 dots, 1 norm, 2 axpy, 1 aypx.

Launch.  `python bench.py --gpus N` is a complete command: for N > 1 and no launcher environment (WORLD_SIZE unset) this process
starts `python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>` as a CHILD before it has imported
torch or loaded any library that touches the GPU, relays the ranks' output and exits with the child's code (and non-zero when no
JSON line with n_gpus == N came back).  Started by torch.distributed.run itself (the driver's N>1 form) it is one of the ranks.
It never prints an n_gpus: 1 line for --gpus N > 1.

Three configurations of the same solve are timed in the same run (N=1; on several GPUs only the first):
  headline          -ksp_type cghipmi355x (the plug-in's registered CG, fused sweeps), the matrix's VALUE ARRAY STREAMED
                    (-mat_hipmi355x_value_patterns 0): what every matrix gets, constant coefficients or not.  value,
                    ms_per_step, roofline, spmv_gbps, ksp_* describe THIS leg.
  value_patterns    the same solver with the library's default for this matrix: P7 has constant coefficients, its rows come
                    from a 27-entry dictionary and the product does not read the value array (the product carries the same bits).
  op_by_op          -ksp_type cg: the plain restatement of PETSc's own KSPSolve_CG -- what an UNCHANGED PETSc program drives over
                    the same types (value array streamed).  The Vec type recognises the update calls of that sequence and runs
                    them as one fused sweep (same bits); ksp_its_per_sec_every_call_its_own_kernel: with that switched off.
After the timed legs, outside any timed region, the run CHECKS ITSELF (N=1): one headline SpMV against the value-pattern SpMV bit
for bit and against the row sums the operator has by construction, and the headline solve's x against the op_by_op solve's x
(`checks` in the line; a failed check exits non-zero and prints no line).

Two series.  `value` is WEAK scaling (n^3 rows per GPU).  `strong` in the same line is the STRONG-scaling point of SURVEY 8(d)
config 3: the cube P7(2n) (= P7(512) at the default n) split into N z-slabs, headline solver only; on 8 GPUs the two coincide.

N > 1 additionally reports `multi_gpu`: bytes per halo message, busy time of the halo stream per SpMV, how much of it lies inside
the diagonal-block SpMV (overlap), what the compute stream still waits for, halo GB/s per link, microseconds per scalar all-reduce.

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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--grid-n", dest="n", type=int, default=256, help="grid points per side per GPU (rows per GPU = n^3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-its", type=int, default=20)
    ap.add_argument("--wide-planes", action="store_true", help="development: (2n)x(2n)x(n/4) rows per GPU instead of n^3 (N=8 is then the cube P7(2n))")
    ap.add_argument("--ksp-opts", default="", help="extra options-database string (development: e.g. '-ksp_cg_fused 2')")
    ap.add_argument("--headline-only", action="store_true", help="development: skip the value_patterns and op_by_op legs and the self-checks")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="which series `value` reports: weak = n^3 rows per GPU (default; the strong point still rides along as `strong`); "
                         "strong = the cube P7(2n) split over the GPUs, nothing else")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling point")
    ap.add_argument("--no-stream", action="store_true", help="skip the STREAM-style legs (what the memory system gives the library's own copy / triad / read-only kernels)")
    return ap.parse_args(argv)


def launch_ranks(args):
    """--gpus N > 1 without a launcher: start the N ranks as a child (torch.distributed.run), relay, exit with its code.
    Nothing here has touched the GPU (no torch, no HIP library), so the children start on a clean device."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    try:
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=dict(os.environ))
    except OSError as e:
        print("[bench] could not start the ranks: %s" % e, file=sys.stderr, flush=True)
        return 1
    ok = False
    for line in p.stdout:
        if line.startswith("{"):
            try:
                ok = ok or json.loads(line).get("n_gpus") == args.gpus
            except ValueError:
                pass
        sys.stdout.write(line)
        sys.stdout.flush()
    rc = p.wait()
    if rc == 0 and not ok:
        print("[bench] the ranks exited 0 but printed no JSON line with n_gpus == %d" % args.gpus, file=sys.stderr, flush=True)
        return 1
    return rc


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
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
    k = pda.load_kernels()
    staged_asked = os.environ.get("MI355X_STAGED", "0") == "1"
    if world > 1:
        ndev = C.c_int()
        k.mi355x_device_count(C.byref(ndev))
        if ndev.value < world and not staged_asked:
            # several ranks on one card is a rehearsal, never a measurement: it has to be asked for
            raise SystemExit("--gpus %d but %d device(s) visible (MI355X_STAGED=1 rehearses the N>1 flow on fewer cards over the host-staged transport)" % (world, ndev.value))
        from petsc_dev_amd import dist as PD
        # MI355X_STAGED=1: rehearsal of the N>1 flow with several ranks on ONE GPU (RCCL refuses that): host-staged transport
        comm = PD.torch_comm(device_comm=not staged_asked)
    else:
        comm = L.COMM_SELF

    if args.ksp_opts:                        # before the matrix exists: -mat_* options are read when it is first used
        L.PetscOptionsInsertString(args.ksp_opts.encode())

    def barrier():
        if dist is not None:
            dist.barrier()

    if world > 1:
        tr = PD.transport_report(comm)                   # asked of the C library: what the halo and the reductions travel over
    else:
        tr = {"transport": "single", "rccl_ranks": 0, "rccl_communicators": 0}
    staged = world > 1 and tr["transport"] != "rccl"
    if world > 1 and not staged:
        assert tr["rccl_ranks"] == world, "RCCL reports %d ranks, launched %d" % (tr["rccl_ranks"], world)
    if staged and not staged_asked:
        raise SystemExit("RCCL could not be set up between the %d ranks (see the message above) and MI355X_STAGED=1 was not asked for" % world)

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
        if args.n == 256 and world == 1:
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

    def kernel_roofline(name, tag, model_bytes, ref_bytes, ms, launches, use_pmc):
        """one kernel against the HBM roof in bytes it MOVED (PMC if stamped, else its byte model)"""
        measured = pmc_bytes(tag) if use_pmc else None
        moved = measured if measured is not None else model_bytes
        gbps = moved / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return {"bound": "hbm", "kernel": name, "achieved": round(gbps, 1), "peak": PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / PEAK_GBPS, 4),
                "traffic": measured, "bytes_moved_per_launch": moved, "bytes_moved_basis": "pmc" if measured is not None else "kernel byte model",
                "bytes_model_per_launch": model_bytes, "avg_launch_ms": round(ms, 5), "launches_timed": launches,
                "reference_algorithmic_bytes_per_launch": ref_bytes,
                "effective_gbps_vs_reference_bytes": round(ref_bytes / (ms * 1e-3) / 1e9, 1) if ms > 0 else 0.0,
                **({"traffic_source": pmc_note} if measured is not None else {})}

    class Case:
        """P7(nx, ny, nz) in `world` z-slabs: operator, right-hand side b = A 1, the timed solves"""

        def __init__(self, nx, ny, nz, use_pmc):
            self.nx, self.ny, self.nz, self.use_pmc = nx, ny, nz, use_pmc
            self.mloc = mloc = nx * ny * (nz // world)
            rs, re_ = rank * mloc, (rank + 1) * mloc
            t0 = time.time()
            self.csr = ai, aj, aa = P.gen_poisson7(nx, ny, nz, rs, re_)
            if world > 1:
                self.A = P.Mat.from_csr_mpi(ai, aj, aa, mloc, mloc * world, mloc * world, comm=comm)
            else:
                self.A = P.Mat.from_csr(ai, aj, aa, comm=comm)
            self.nnz_loc = int(aj.size)
            self.u = P.Vec.create(mloc, N=mloc * world, comm=comm)
            L.VecSet(self.u.h, 1.0)
            self.b, self.x = self.u.duplicate(), self.u.duplicate()
            self.A.mult(self.u, self.b)                                         # b = A * 1
            self.setup_s = time.time() - t0
            # the kernel the matrix part of a step runs: SpMV of the (diagonal block of the) matrix
            if world > 1:
                Ad = C.c_void_p()
                L.MatMPIAIJGetSeqAIJ(self.A.h, C.byref(Ad), None, None)
                m_, i_, j_, a_ = C.c_int(), C.c_void_p(), C.c_void_p(), C.c_void_p()
                L.MatSeqAIJGetArrays(Ad, C.byref(m_), C.byref(i_), C.byref(j_), C.byref(a_))
                self.nnz_k = int(np.ctypeslib.as_array(C.cast(i_, C.POINTER(C.c_int)), (mloc + 1,))[mloc])
                self.timed = Ad
            else:
                self.nnz_k = self.nnz_loc
                self.timed = self.A.h
            self.spmv_ref_bytes = 12 * self.nnz_k + 4 * (mloc + 1) + 8 * mloc + 8 * mloc     # the reference CSR's algorithmic bytes
            self.noff, self.npat, self.nvpat = C.c_int(0), C.c_int(0), C.c_int(0)

        def make_ksp(self, ksp_type):
            k_ = P.KSP(comm=comm)
            k_.set_operators(self.A)
            k_.set_type(ksp_type)
            k_.set_pc_type("jacobi")
            if args.ksp_opts:
                k_.set_from_options()
            return k_

        def first_use(self, ksp):
            """upload + analysis; which SpMV kernel the analysis chose for this matrix"""
            L.MatHIPMI355XSetValuePatterns(self.timed, 1)
            ksp.set_tolerances(rtol=0.0, abstol=1e-300, dtol=1e300, max_it=1)
            ksp.solve(self.b, self.x)
            L.MatHIPMI355XGetIndexCompression(self.timed, C.byref(self.noff))
            L.MatHIPMI355XGetRowPatterns(self.timed, C.byref(self.npat))
            L.MatHIPMI355XGetValuePatterns(self.timed, C.byref(self.nvpat))

        def timed_solve(self, ksp):
            """W untimed iterations (the first call also uploads the matrix and builds the Jacobi diagonal), then exactly K timed ones"""
            ksp.set_tolerances(rtol=0.0, abstol=1e-300, dtol=1e300, max_it=max(args.warmup, 1))
            ksp.solve(self.b, self.x)
            k.mi355x_device_synchronize()
            ev = 0 if os.environ.get("BENCH_NO_SPMV_EVENTS") else 1                      # (development: cost of the event pairs)
            if world > 1:
                L.MatMPIAIJHIPMI355XSetHaloTiming(self.A.h, ev)                           # diagonal-block product AND halo exchange
            else:
                L.MatHIPMI355XSetTiming(self.timed, ev)
            L.VecHIPMI355XSetCGUpdateTiming(ev)
            ksp.set_tolerances(rtol=0.0, abstol=1e-300, dtol=1e300, max_it=args.steps)
            barrier()
            k.mi355x_device_synchronize()
            t0 = time.perf_counter()
            ksp.solve(self.b, self.x)
            k.mi355x_device_synchronize()
            barrier()
            dt = time.perf_counter() - t0
            assert ksp.its == args.steps, "solver stopped after %d of %d iterations (reason %d)" % (ksp.its, args.steps, ksp.reason)
            nl, tms, nu, ums = C.c_int(), C.c_double(), C.c_int(), C.c_double()
            L.MatHIPMI355XGetTiming(self.timed, C.byref(nl), C.byref(tms))
            L.VecHIPMI355XGetCGUpdateTiming(C.byref(nu), C.byref(ums))
            halo = None
            if world > 1:
                nh, hms, oms, xms, sb, nb = C.c_int(), C.c_double(), C.c_double(), C.c_double(), C.c_double(), C.c_int()
                L.MatMPIAIJHIPMI355XGetHaloTiming(self.A.h, C.byref(nh), C.byref(hms), C.byref(oms), C.byref(xms), C.byref(sb), C.byref(nb))
                halo = [hms.value / max(nh.value, 1), oms.value / max(nh.value, 1), xms.value / max(nh.value, 1), sb.value, float(nb.value), float(nh.value)]
                L.MatMPIAIJHIPMI355XSetHaloTiming(self.A.h, 0)
            else:
                L.MatHIPMI355XSetTiming(self.timed, 0)
            L.VecHIPMI355XSetCGUpdateTiming(0)
            spmv_ms = tms.value / max(nl.value, 1)
            upd_ms = ums.value / max(nu.value, 1)
            out = {"spmv_launches": nl.value, "upd_launches": nu.value}
            if dist is not None:
                import torch
                t = torch.tensor([dt, spmv_ms, upd_ms], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt, spmv_ms, upd_ms = float(t[0]), float(t[1]), float(t[2])
                # the halo figures of the rank whose halo stream was busy longest (an interior rank: two neighbours)
                g = [torch.zeros(6, dtype=torch.float64) for _ in range(world)]
                dist.all_gather(g, torch.tensor(halo, dtype=torch.float64))
                out["halo_per_rank"] = [[float(v) for v in r_] for r_ in g]
            out.update({"dt": dt, "spmv_ms": spmv_ms, "upd_ms": upd_ms})
            return out

        def spmv_kernel(self, value_patterns, with_dot):
            """(kernel name for the JSON, tag in rocprofv3's kernel names, modelled bytes moved per launch: matrix stream + x once + y once)"""
            mloc, nnz_k = self.mloc, self.nnz_k
            vec = 8 * mloc + 8 * mloc
            inst = "<0, true>" if with_dot else "<0, false>"          # the instance that also leaves p'w (the registered CG on one GPU) / the plain product
            also = "; p'w from the same pass" if with_dot else ""
            if value_patterns and self.nvpat.value:
                return ("spmv_csr_valpat_kernel (%d distinct rows {offsets, values} in a dictionary, 2 bytes per row; the value array is not read%s)" % (self.nvpat.value, also),
                        "spmv_csr_valpat_kernel" + inst, 2 * mloc + vec)
            if self.npat.value:
                return ("spmv_csr_rowblock_pat_kernel (CSR values + a %d-list row-pattern dictionary: 8 B per nonzero + one 4-byte word per row%s)" % (self.npat.value, also),
                        "spmv_csr_rowblock_pat_kernel" + inst, 8 * nnz_k + 4 * mloc + vec)
            if self.noff.value:
                return ("spmv_csr_rowblock_idx8_kernel (CSR values + 1-byte offset-dictionary column indices, %d offsets)" % self.noff.value,
                        "spmv_csr_rowblock_idx8_kernel<0,", 9 * nnz_k + 4 * (mloc + 1) + vec)
            return ("spmv_csr_rowblock_kernel (plain CSR)", "spmv_csr_rowblock_kernel<0,", 12 * nnz_k + 4 * (mloc + 1) + vec)

        def leg(self, ksp, value_patterns, vec_passes, what, with_dot=False, host_scalar_update=False):
            """time one configuration; its/s, and the step / its two main kernels in bytes moved"""
            mloc = self.mloc
            L.MatHIPMI355XSetValuePatterns(self.timed, 1 if value_patterns else 0)
            t = self.timed_solve(ksp)
            its = args.steps / t["dt"]
            name, tag, model = self.spmv_kernel(value_patterns, with_dot)
            r_spmv = kernel_roofline(name, tag, model, self.spmv_ref_bytes, t["spmv_ms"], t["spmv_launches"], self.use_pmc)
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
                                                   t["upd_ms"], t["upd_launches"], self.use_pmc)
            return out, t

        def workload(self):
            return ("3D 7-pt Poisson P7(%d,%d,%d) = %d rows, %s, CG + PCJACOBI (-ksp_type cghipmi355x), value array streamed, b = A*1, x0 = 0, exactly K iterations"
                    % (self.nx, self.ny, self.nz, self.mloc * world, "MatSeqAIJ on 1 GPU" if world == 1 else "MatMPIAIJ in %d z-slabs, %s"
                       % (world, "HOST-STAGED halo and reductions (one-GPU rehearsal, not a measurement)" if staged else "RCCL halo")))

        def multi_gpu(self, t, spmv_ms):
            """what the N>1 line says about the exchange step (the busiest rank's figures)"""
            rows = t["halo_per_rank"]
            busiest = max(range(world), key=lambda r_: rows[r_][0])
            hms, oms, xms, sbytes, nbr, nprod = rows[busiest]
            per_link = sbytes / max(nbr, 1.0)
            return {"rccl_ranks": tr["rccl_ranks"], "transport": tr["transport"],
                    "halo_bytes_per_neighbour_per_spmv": int(per_link), "halo_neighbours_busiest_rank": int(nbr), "busiest_rank": busiest,
                    "halo_stream_busy_ms_per_spmv": round(hms, 5), "halo_inside_diagonal_spmv_ms": round(oms, 5),
                    "halo_overlap_frac": round(oms / hms, 4) if hms > 0 else None,
                    "halo_exposed_ms_per_spmv": round(xms, 5), "diagonal_spmv_ms": round(spmv_ms, 5),
                    "halo_gbps_per_link": round(per_link / (hms * 1e-3) / 1e9, 2) if hms > 0 else None,
                    "halo_gbps_basis": "bytes one neighbour receives from the busiest rank per SpMV / that rank's halo-stream busy time (pack, grouped ncclSend/ncclRecv "
                                       "to all its neighbours, unpack): a lower bound of the link rate; xGMI is ~153 GB/s per link and direction",
                    "spmv_products_timed": int(nprod),
                    "halo_ms_per_rank": [round(r_[0], 5) for r_ in rows], "halo_exposed_ms_per_rank": [round(r_[2], 5) for r_ in rows]}

    n = args.n
    fused_dot = world == 1
    # the registered CG (-ksp_cg_fused 4, its default): on one GPU p'w comes out of the SpMV pass -> AYPX 3 + fused update 8 = 11 vector
    # passes beside the SpMV; on a parallel matrix p'w is its own reduction (2 more passes)
    fused_passes = 11 if fused_dot else 13

    def strong_dims():
        """SURVEY 8(d) config 3, strong series: the cube P7(2n) (P7(512) at the default n) in `world` z-slabs"""
        return 2 * n, 2 * n, 2 * n

    def weak_dims():
        nx, ny, nz = n, n, n * world            # z-slabs: rank r owns planes [r*n, (r+1)*n)
        if args.wide_planes or (world == 8 and n == 256):
            # same rows per GPU, planes twice as wide: (2n) x (2n) x (n/4) per rank.  On 8 GPUs that is the cube P7(512) =
            # BASELINE.json configs[2] exactly (2 MiB halo per neighbour); measured cost of the wider planes on one GPU: +3 %
            nx, ny, nz = 2 * n, 2 * n, (n // 4) * world
        return nx, ny, nz

    def headline_of(case):
        ksp = case.make_ksp("cghipmi355x")
        case.first_use(ksp)
        head, th = case.leg(ksp, False, fused_passes, "-ksp_type cghipmi355x -mat_hipmi355x_value_patterns 0: fused CG sweeps, the SpMV streams the value array"
                            + (" and leaves p'w" if fused_dot else ""), fused_dot)
        return ksp, head, th

    def strong_point(case, head, th):
        d = {"what": "strong-scaling point (SURVEY 8d config 3): the cube P7(%d) = %d rows split into %d z-slab(s), headline solver" % (case.nx, case.mloc * world, world),
             "scaling": "strong", "n_gpus": world, "value": head["value"], "unit": "Mdof-it/s", "ksp_its_per_sec": head["ksp_its_per_sec"], "ms_per_step": head["ms_per_step"],
             "rows_total": case.mloc * world, "rows_per_gpu": case.mloc, "setup_s": round(case.setup_s, 2),
             "spmv": {kk: head["spmv"][kk] for kk in ("kernel", "achieved", "frac", "avg_launch_ms", "bytes_moved_per_launch", "bytes_moved_basis")},
             "ksp_hbm_frac": head["ksp_hbm_frac"]}
        if "cg_update" in head:
            d["cg_update"] = {kk: head["cg_update"][kk] for kk in ("achieved", "frac", "avg_launch_ms", "bytes_moved_per_launch")}
        if world > 1:
            d["multi_gpu"] = case.multi_gpu(th, head["spmv"]["avg_launch_ms"])
        return d

    strong_possible = (2 * n) % world == 0 and (2 * n) // world >= 1
    if args.scaling == "strong":
        if not strong_possible:
            raise SystemExit("--scaling strong: %d planes do not split over %d ranks" % (2 * n, world))
        case = Case(*strong_dims(), use_pmc=False)
        _, head, th = headline_of(case)
        out = {"metric": "KSP CG+Jacobi iterations/s x unknowns (3-D 7-pt Poisson P7(%d), STRONG scaling: the same cube on every N); ksp_its_per_sec and spmv_gbps are BASELINE.json's two quantities" % (2 * n),
               "value": head["value"], "unit": "Mdof-it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": case.workload(), "rows_per_gpu": case.mloc, "nnz_per_gpu": case.nnz_loc, "parallelism": "row-block dp%d" % world, **tr},
               "ksp_its_per_sec": head["ksp_its_per_sec"], "spmv_gbps": round(head["spmv"]["achieved"] * world, 1),
               "roofline": head["spmv"], "strong": strong_point(case, head, th), "setup_s": round(case.setup_s, 2)}
        if world > 1:
            out["multi_gpu"] = case.multi_gpu(th, head["spmv"]["avg_launch_ms"])
        if rank == 0:
            print(json.dumps(out), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ------------------------------------------------------------------ weak series (the line's `value`) ------------------------------------------------------------------
    case = Case(*weak_dims(), use_pmc=True)
    mloc, unknowns = case.mloc, case.mloc * world
    ksp_fused, head, th = headline_of(case)
    if os.environ.get("BENCH_NO_SPMV_EVENTS"):
        print("no-events run: %.5f ms/step" % head["ms_per_step"], flush=True)
        return
    its_per_s = head["ksp_its_per_sec"]
    cg_ref_bytes = case.spmv_ref_bytes + 136 * mloc           # the reference's unfused CG+Jacobi iteration (SpMV + 17 vector passes)
    out = {
        "metric": "KSP CG+Jacobi iterations/s x unknowns (3-D 7-pt Poisson, %d^3 rows per GPU); ksp_its_per_sec and spmv_gbps are BASELINE.json's two quantities" % n,
        "value": head["value"], "unit": "Mdof-it/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": case.workload(),
                   "rows_per_gpu": mloc, "nnz_per_gpu": case.nnz_loc, "parallelism": "row-block dp%d" % world,
                   "transport": tr["transport"], "rccl_ranks": tr["rccl_ranks"], "rccl_communicators": tr["rccl_communicators"]},
        "ksp_its_per_sec": its_per_s,
        "spmv_gbps": round(head["spmv"]["achieved"] * world, 1),
        "spmv_gbps_basis": "bytes the SpMV kernel moved (%s) / its average launch time inside the timed solve, summed over GPUs" % head["spmv"]["bytes_moved_basis"],
        "ksp_gbps": head["ksp_gbps"], "ksp_hbm_frac": head["ksp_hbm_frac"],
        "ksp_gbps_basis": "bytes moved per step = SpMV (as above) + %d vector passes of the fused iteration (AYPX 3, fused update 8%s)" % (fused_passes, "" if fused_dot else ", dot 2"),
        "effective_gbps_vs_reference_bytes": {"spmv": round(head["spmv"]["effective_gbps_vs_reference_bytes"] * world, 1),
                                              "ksp": round(cg_ref_bytes * its_per_s * world / 1e9, 1),
                                              "what": "the same times priced in the REFERENCE's algorithmic bytes (SURVEY 8d: CSR 12 B/nnz + 4 B/row + x + y = %d B per SpMV; "
                                                      "SpMV + 17 vector passes = %d B per op-by-op iteration); not a fraction of any roof" % (case.spmv_ref_bytes, cg_ref_bytes)},
        "setup_s": round(case.setup_s, 2),
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
    if world > 1:
        mg = case.multi_gpu(th, head["spmv"]["avg_launch_ms"])
        sync_us, b2b_us = C.c_double(), C.c_double()
        L.PetscCommDeviceAllreduceLatency(comm, 50, C.byref(sync_us), C.byref(b2b_us))       # collective: every rank calls it
        mg["allreduce_us_per_reduction"] = {"host_sync_after_each": round(sync_us.value, 2), "queued_back_to_back": round(b2b_us.value, 2),
                                            "what": "one-double ncclAllReduce on the compute stream's communicator, 50 repetitions; the fused CG makes 2 per iteration "
                                                    "(p'w; the update's three sums in one), KSPSolve_CG op by op 3" if not staged else "host-staged transport: not measured"}
        out["multi_gpu"] = mg
        # ---- self-checks of the N-rank line (every rank asserts; outside the timed region): the parallel product with its halo
        # exchange applied to 1 must give the operator's row sums on this rank's rows -- a halo that brought the wrong planes, or none,
        # cannot pass --, and the solve must have reduced the true residual b - A x, recomputed with separate calls
        ai_, aj_, aa_ = case.csr
        ychk = case.u.duplicate()
        case.A.mult(case.u, ychk)
        rows_ok = torch.tensor([1.0 if np.array_equal(ychk.array(), np.add.reduceat(aa_, ai_[:-1].astype(np.int64))) else 0.0], dtype=torch.float64)
        dist.all_reduce(rows_ok, op=dist.ReduceOp.MIN)                                     # every rank learns whether ALL passed: rank 0 prints no line otherwise
        assert rows_ok.item() == 1.0, "A*1 through MatMult_MPIAIJ differs from the operator's row sums on some rank"
        case.A.mult(case.x, ychk)
        L.VecAYPX(ychk.h, -1.0, case.b.h)
        nr_, nb_ = C.c_double(), C.c_double()
        L.VecNorm(ychk.h, P.NORM_2, C.byref(nr_))                                          # collective
        L.VecNorm(case.b.h, P.NORM_2, C.byref(nb_))
        assert nr_.value / nb_.value < 1.0, "the %d-rank solve does not reduce the residual" % world
        out["checks"] = {"mpiaij_spmv_times_ones_equals_row_sums_on_every_rank": True, "true_residual_after_K_steps": nr_.value / nb_.value}
        del ychk

    if world == 1 and not args.headline_only:
        nvpat = case.nvpat
        timed = case.timed
        # ---- leg 2: the library's default for THIS matrix (constant coefficients: rows from a dictionary) ----
        if nvpat.value:
            vp, _ = case.leg(ksp_fused, True, fused_passes, "-ksp_type cghipmi355x, value patterns on (library default; P7 is a constant-coefficient operator): the SpMV does not read the value array; "
                             "the same products bit for bit, p'w summed per workgroup (iterates agree with the headline's to rounding)", fused_dot)
            out["legs"]["value_patterns"] = vp
        # ---- leg 3: what an unchanged PETSc program drives: KSPSolve_CG op by op (VecAYPX, MatMult, VecTDot, VecAXPY x2, PCApply, VecNorm, VecTDot) ----
        # The Vec type notes KSPSolve_CG's update calls instead of launching them one by one and runs them as the fused sweep when the norm
        # is asked for (host/vechip.c, "deferred element-wise operations"; same bits): SpMV + AYPX 3 + TDot(p,w) 2 + sweep 8 = 13 vector
        # passes.  With -vec_hipmi355x_defer 0 every call is a kernel of its own: 17 passes.
        ksp_plain = case.make_ksp("cg")
        setdef = L.raw("VecHIPMI355XSetDeferral")
        setdef(1)
        ob, _ = case.leg(ksp_plain, False, 13, "-ksp_type cg: the plain KSPSolve_CG call sequence of an unchanged program (value array streamed); the Vec type recognises its "
                         "update calls (VecAXPY x2, PCApply_Jacobi, VecNorm, VecTDot) and runs them as one fused sweep: SpMV + 13 vector passes", host_scalar_update=True)
        x_plain = case.x.array().copy()                    # the op_by_op solve's iterate after exactly K steps (checked below)
        setdef(0)
        t1 = case.timed_solve(ksp_plain)
        ob["ksp_its_per_sec_every_call_its_own_kernel"] = round(args.steps / t1["dt"], 2)      # -vec_hipmi355x_defer 0: 17 vector passes
        setdef(1)
        if nvpat.value:
            L.MatHIPMI355XSetValuePatterns(timed, 1)
            t2 = case.timed_solve(ksp_plain)
            ob["ksp_its_per_sec_with_value_patterns"] = round(args.steps / t2["dt"], 2)
        setdef(-1)
        out["legs"]["op_by_op"] = ob

        # ---- self-checks, outside every timed region: a wrong kernel must not be able to print a number ----
        checks = {}
        ai, aj, aa = case.csr
        # (1) one product by the headline kernel (value array streamed) against the value-pattern kernel, bit for bit, on x_i = sin(0.37 i) + 1
        #     (SURVEY 8d config 2's SpMV vector), and against what the operator gives by construction: (A 1)_i = 6 - (number of neighbours of i)
        xs = P.Vec.from_array(np.sin(0.37 * np.arange(mloc, dtype=np.float64)) + 1.0, comm=comm)
        ys = xs.duplicate()
        L.MatHIPMI355XSetValuePatterns(timed, 0)
        case.A.mult(xs, ys)
        y_stream = ys.array().copy()
        case.A.mult(case.u, ys)
        ones_stream = ys.array().copy()
        rowsum = np.add.reduceat(aa, ai[:-1].astype(np.int64))                       # exact: small integers
        assert np.array_equal(ones_stream, rowsum), "headline SpMV kernel: A*1 differs from the operator's row sums"
        checks["headline_spmv_times_ones_equals_row_sums"] = True
        if nvpat.value:
            L.MatHIPMI355XSetValuePatterns(timed, 1)
            case.A.mult(xs, ys)
            same = bool(np.array_equal(ys.array().view(np.uint64), y_stream.view(np.uint64)))
            assert same, "headline SpMV (value array streamed) and value-pattern SpMV differ in bits"
            checks["headline_spmv_equals_value_pattern_spmv_bitwise"] = True
        # (2) the headline solve's x after K steps against the op_by_op solve's x after K steps (the headline's products come from the
        #     <0, true> instance, the one that also leaves p'w and that `roofline` quotes; the op_by_op solve's from <0, false> + VecTDot)
        L.MatHIPMI355XSetValuePatterns(timed, 0)
        ksp_fused.set_tolerances(rtol=0.0, abstol=1e-300, dtol=1e300, max_it=args.steps)
        ksp_fused.solve(case.b, case.x)
        x_head = case.x.array()
        rel = float(np.max(np.abs(x_head - x_plain)) / np.max(np.abs(x_plain)))
        checks["headline_x_vs_op_by_op_x_max_rel_diff_after_K_steps"] = rel
        checks["tolerance"] = 1e-10
        assert rel <= 1e-10, "headline CG iterate differs from the op-by-op KSPSolve_CG iterate by %g relative after %d steps" % (rel, args.steps)
        # (3) and it solves the system it was given: |b - A x| / |b| after K steps, recomputed with separate calls
        rv = case.u.duplicate()
        case.A.mult(case.x, rv)
        L.VecAYPX(rv.h, -1.0, case.b.h)
        nr, nb = C.c_double(), C.c_double()
        L.VecNorm(rv.h, P.NORM_2, C.byref(nr))
        L.VecNorm(case.b.h, P.NORM_2, C.byref(nb))
        checks["true_residual_after_K_steps"] = nr.value / nb.value
        assert nr.value / nb.value < 1.0, "the headline solve does not reduce the residual"
        out["checks"] = checks
        L.MatHIPMI355XSetValuePatterns(timed, 1)
        del xs, ys, rv

    if world == 1 and not args.no_stream and not args.headline_only:
        # ---- what the memory system gives plain streams in THIS run (SURVEY 8d "Roofline": achievable copy / triad rate next to the peak):
        # the library's own VecCopy (1 read, 1 write), VecWAXPY (2 reads, 1 write) and VecDot (2 reads) on vectors of the workload's
        # size (134 MB each at 256^3: they partly live in the 256 MB Infinity Cache, as the solve's do) and of 1 GiB (nothing cached);
        # every call a kernel of its own (no noted operations), wall clock over 20 calls between two device synchronisations
        setdef = L.raw("VecHIPMI355XSetDeferral")
        setdef(0)
        stream = {}
        for label, nn in (("workload_size", mloc), ("1GiB", 1 << 27)):
            va = P.Vec.create(nn, comm=comm); vb = va.duplicate(); vc = va.duplicate()
            L.VecSet(va.h, 1.0); L.VecSet(vb.h, 2.0); L.VecSet(vc.h, 0.0)
            nrm = C.c_double()
            res = {"n": int(nn)}
            for name, nbytes, fn in (("copy", 16 * nn, lambda: L.VecCopy(va.h, vc.h)), ("triad", 24 * nn, lambda: L.VecWAXPY(vc.h, 3.0, va.h, vb.h)),
                                     ("read", 16 * nn, lambda: L.VecDot(va.h, vb.h, C.byref(nrm)))):
                for _ in range(3):
                    fn()
                k.mi355x_device_synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    fn()
                k.mi355x_device_synchronize()
                res[name + "_gbps"] = round(nbytes * 20 / (time.perf_counter() - t0) / 1e9, 1)
            stream[label] = res
            del va, vb, vc
        setdef(-1)
        stream["what"] = ("VecCopy / VecWAXPY / VecDot of this library, 20 calls each between device synchronisations; 'read' (the dot) includes one host wait per call "
                          "(the sum comes back): a lower bound of the read rate")
        out["stream"] = stream
        ach = max(stream["workload_size"]["copy_gbps"], stream["workload_size"]["triad_gbps"])
        for key in ("roofline", "roofline_spmv", "roofline_cg_update"):
            if key in out and out[key].get("achieved"):
                out[key]["frac_of_achievable_stream"] = round(out[key]["achieved"] / ach, 4)
                out[key]["achievable_stream_gbps"] = ach

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import orc
        ai, aj, aa = case.csr
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
        del bref

    # ------------------------------------------------------------------ strong series: the cube P7(2n) on these N GPUs ------------------------------------------------------------------
    if not args.no_strong and not args.headline_only and strong_possible:
        if (case.nx, case.ny, case.nz) == strong_dims():
            out["strong"] = strong_point(case, head, th)     # 8 GPUs at the default n: the weak case IS the cube
            out["strong"]["what"] += " (the weak-scaling case of this run: the two series meet here)"
        else:
            del ksp_fused
            if world == 1 and not args.headline_only:
                del ksp_plain
            case.A.destroy(); case.u.destroy(); case.b.destroy(); case.x.destroy()
            case.csr = None
            del case
            import gc
            gc.collect()
            scase = Case(*strong_dims(), use_pmc=False)
            _, shead, sth = headline_of(scase)
            out["strong"] = strong_point(scase, shead, sth)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
