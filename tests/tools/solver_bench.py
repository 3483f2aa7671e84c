"""Iterations/s of the three north-star solvers with Jacobi on P7(n) on one GPU (development aid; the contract line is bench.py)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


PRODUCT = {"cg": "cghipmi355x", "gmres": "gmreshipmi355x", "bcgs": "bcgshipmi355x"}   # the plug-in's own solvers; "-ksp_*_fused 0" selects the plain types


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    extra = sys.argv[3] if len(sys.argv) > 3 else ""
    only = sys.argv[4].split(",") if len(sys.argv) > 4 else None      # e.g. gmres:jacobi,cg:none
    import petsc_dev_amd as pda
    from petsc_dev_amd import petsc as P
    L = P.lib()
    k = pda.load_kernels()
    ai, aj, aa = P.gen_poisson7(n, n, n)
    if extra:                                # -mat_* options are read when the matrix is first used
        L.PetscOptionsInsertString(extra.encode())
    A = P.Mat.from_csr(ai, aj, aa)
    N = n ** 3
    u = P.Vec.create(N, comm=L.COMM_SELF); L.VecSet(u.h, 1.0)
    b, x = u.duplicate(), u.duplicate()
    A.mult(u, b)
    for ksp_t, pc_t in (("cg", "jacobi"), ("groppcg", "jacobi"), ("gmres", "jacobi"), ("bcgs", "jacobi"), ("cg", "none"), ("gmres", "none")):
        if only and ("%s:%s" % (ksp_t, pc_t)) not in only:
            continue
        ksp = P.KSP(comm=L.COMM_SELF)
        ksp.set_operators(A)
        L.PetscOptionsClear()
        L.PetscOptionsInsertString(("-ksp_type %s -pc_type %s %s" % (PRODUCT.get(ksp_t, ksp_t) if "fused 0" not in extra else ksp_t, pc_t, extra)).encode())
        ksp.set_from_options()
        ksp.set_tolerances(rtol=0.0, abstol=1e-300, dtol=1e300, max_it=40)
        L.VecSet(x.h, 0.0); ksp.solve(b, x)
        ksp.set_tolerances(rtol=0.0, abstol=1e-300, dtol=1e300, max_it=steps)
        L.VecSet(x.h, 0.0)
        k.mi355x_device_synchronize()
        t0 = time.perf_counter()
        ksp.solve(b, x)
        k.mi355x_device_synchronize()
        dt = time.perf_counter() - t0
        print("%-7s + %-7s: %4d its  %8.3f ms/it  %8.1f it/s  (reason %d)" % (ksp_t, pc_t, ksp.its, dt / max(ksp.its, 1) * 1e3, ksp.its / dt, ksp.reason), flush=True)
        L.PetscOptionsClear()


if __name__ == "__main__":
    main()
