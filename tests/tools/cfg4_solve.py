"""Config 4 end to end (BASELINE.json configs[3]: unstructured FEM, GMRES(30) + block Jacobi with ILU(0) sub-blocks; on one
rank that is GMRES(30) + ILU(0), the reference's default) on the stand-ins of tests/problems.py.
  python3 tests/tools/cfg4_solve.py [fem|irr] [sub_pc=ilu|jacobi]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems as pb  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "fem"
    sub = sys.argv[2] if len(sys.argv) > 2 else "ilu"
    nblocks = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # -pc_bjacobi_blocks on the one rank
    import petsc_dev_amd as pda
    from petsc_dev_amd import petsc as P
    L = P.lib(); k = pda.load_kernels()
    t0 = time.time()
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    from cfg4_spmv import cached                    # the generated CSR is kept under $CFG4_CACHE (default /tmp) between the tools of one run
    ai, aj, aa = cached(which, pb.gen_fem3 if which == "fem" else pb.gen_irr)
    n = ai.size - 1
    print("%s: n=%d nnz=%d (%.1f/row) generated in %.1fs" % (which, n, aj.size, aj.size / n, time.time() - t0), flush=True)
    A = P.Mat.from_csr(ai, aj, aa)
    x = P.Vec.create(n, comm=L.COMM_SELF); L.VecSet(x.h, 1.0)
    b = x.duplicate(); u = x.duplicate()
    A.mult(x, b)
    ksp = P.KSP(comm=L.COMM_SELF)
    ksp.set_operators(A); ksp.set_type("gmreshipmi355x")
    extra = os.environ.get("CFG4_OPTS", "")   # e.g. "-pc_factor_hipmi355x_trisolve_order level" or "-pc_factor_hipmi355x_trisolve_nodes 0"
    L.PetscOptionsClear()
    if sub == "ilu" and nblocks == 1:      # one rank: block Jacobi with one block and ILU(0) inside IS PCILU (the reference's default on one rank)
        ksp.set_pc_type("ilu")
        L.PetscOptionsInsertString(extra.encode())
    else:
        ksp.set_pc_type("bjacobi")
        L.PetscOptionsInsertString(("-sub_pc_type %s -pc_bjacobi_blocks %d %s" % (sub, nblocks, extra)).encode())
    ksp.set_from_options()
    ksp.set_tolerances(rtol=0.0, abstol=1e-300, dtol=1e300, max_it=35)
    t0 = time.time()
    ksp.solve(b, u)                 # PCSetUp (factorisation, level analysis, upload) happens here
    k.mi355x_device_synchronize()
    print("set-up + 35 warm-up iterations: %.2f s" % (time.time() - t0), flush=True)
    if sub == "ilu" and nblocks == 1:
        pc = C.c_void_p(); L.KSPGetPC(ksp.h, C.byref(pc))
        nl, nu = C.c_int(), C.c_int(); L.PCILUGetLevels_HIPMI355X(pc, C.byref(nl), C.byref(nu))
        nn, nnl, nnu = C.c_int(), C.c_int(), C.c_int(); L.PCILUGetNodeInfo_HIPMI355X(pc, C.byref(nn), C.byref(nnl), C.byref(nnu))
        print("ILU(0) plans: %s" % ("node-blocked%s, %d nodes, node levels L=%d U=%d" % (" (whole dependency nodes as columns)" if nn.value < 0 else "", abs(nn.value), nnl.value, nnu.value) if nn.value else "row-granular"), flush=True)
        for _ in range(3):
            L.raw("PCApply")(pc, b.h, u.h)
        k.mi355x_device_synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            L.raw("PCApply")(pc, b.h, u.h)
        k.mi355x_device_synchronize()
        t = (time.perf_counter() - t0) / 10
        print("ILU(0): levels L=%d U=%d, %.1f rows per level; PCApply %.3f ms = %.2f us per level" % (nl.value, nu.value, n / max(nl.value, 1), t * 1e3, t * 1e6 / (nl.value + nu.value)), flush=True)
    L.PetscOptionsClear()
    steps = 90
    ksp.set_tolerances(rtol=0.0, abstol=1e-300, dtol=1e300, max_it=steps)
    k.mi355x_device_synchronize()
    t0 = time.perf_counter()
    ksp.solve(b, u)
    k.mi355x_device_synchronize()
    dt = time.perf_counter() - t0
    print("%s GMRES(30) + bjacobi(%s, %d block%s): %d its in %.3f s = %.1f it/s (%.3f ms per iteration)" % (which, sub, nblocks, "" if nblocks == 1 else "s", ksp.its, dt, ksp.its / dt, dt / ksp.its * 1e3), flush=True)
    ksp.set_tolerances(rtol=1e-8, abstol=1e-50, dtol=1e5, max_it=2000)
    L.VecSet(u.h, 0.0)
    t0 = time.perf_counter()
    ksp.solve(b, u)
    k.mi355x_device_synchronize()
    dt = time.perf_counter() - t0
    print("%s to rtol 1e-8: its=%d reason=%d in %.3f s, error=%g" % (which, ksp.its, ksp.reason, dt, np.linalg.norm(u.array() - 1.0) / np.sqrt(n)), flush=True)


if __name__ == "__main__":
    main()
