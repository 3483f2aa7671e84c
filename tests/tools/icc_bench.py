"""ICC(0) on P7(n) on one GPU (development aid): set-up, PCApply time, and CG to rtol 1e-8 with ICC(0) against CG with Jacobi --
iterations and wall time of the solves.  icc_bench.py N"""
import ctypes as C
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import petsc_dev_amd as pda
from petsc_dev_amd import petsc as P
import orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
L = P.lib(); k = pda.load_kernels()
ai, aj, aa = P.gen_poisson7(n, n, n)
A = P.Mat.from_csr(ai, aj, aa)
N = n ** 3
b = P.Vec.from_array(np.sin(0.1 * np.arange(N)), comm=L.COMM_SELF); x = b.duplicate()
ksp = P.KSP(comm=L.COMM_SELF); ksp.set_operators(A)
pc = C.c_void_p(); L.KSPGetPC(ksp.h, C.byref(pc)); L.PCSetType(pc, b"icc")
t0 = time.time(); L.raw("PCSetUp")(pc); print("P7(%d): ICC(0) set-up (host factorisation + two row-form systems + levels + upload) %.2f s" % (n, time.time() - t0), flush=True)
nl, nu, ns = C.c_int(), C.c_int(), C.c_int(); L.PCICCGetInfo_HIPMI355X(pc, C.byref(nl), C.byref(nu), C.byref(ns))
for _ in range(3):
    L.raw("PCApply")(pc, b.h, x.h)
k.mi355x_device_synchronize()
t0 = time.perf_counter()
for _ in range(10):
    L.raw("PCApply")(pc, b.h, x.h)
k.mi355x_device_synchronize()
t = (time.perf_counter() - t0) / 10
print("levels %d + %d, shifts %d ; PCApply %.3f ms (%.2f us per level)" % (nl.value, nu.value, ns.value, t * 1e3, t * 1e6 / (nl.value + nu.value)), flush=True)
if n <= 128:
    f, _ = orc.icc0_factor(ai, aj, aa)
    t0 = time.perf_counter(); ref = orc.icc0_solve(f, np.sin(0.1 * np.arange(N))); tc = time.perf_counter() - t0
    print("oracle (1 core) solve %.3f ms ; bit-exact: %s" % (tc * 1e3, np.array_equal(x.array().view(np.uint64), ref.view(np.uint64))), flush=True)
u = P.Vec.create(N, comm=L.COMM_SELF); L.VecSet(u.h, 1.0)
A.mult(u, b)
cases = [("icc", "-pc_type icc"), ("jacobi", "-pc_type jacobi")]
for nb in [int(v) for v in os.environ.get("ICC_BENCH_BLOCKS", "64,4096").split(",") if v]:      # several ICC(0) blocks on the one rank, solved as one system
    cases.append(("bjacobi/icc x%d" % nb, "-pc_type bjacobi -pc_bjacobi_blocks %d -sub_pc_type icc" % nb))
for pct, popt in cases:
    ks = P.KSP(comm=L.COMM_SELF); ks.set_operators(A)
    L.PetscOptionsClear(); L.PetscOptionsInsertString(("-ksp_type cg " + popt).encode())
    ks.set_from_options()
    ks.set_tolerances(rtol=1e-8, abstol=1e-50, dtol=1e5, max_it=20000)
    L.VecSet(x.h, 0.0); ks.solve(b, x)            # includes the set-up (block Jacobi reads its options there)
    L.PetscOptionsClear()
    L.VecSet(x.h, 0.0)
    k.mi355x_device_synchronize(); t0 = time.perf_counter()
    ks.solve(b, x)
    k.mi355x_device_synchronize(); dt = time.perf_counter() - t0
    err = np.abs(x.array() - 1.0).max()
    print("CG + %-18s: %5d iterations to rtol 1e-8 in %8.2f ms (%.3f ms per iteration), max error %.2e" % (pct, ks.its, dt * 1e3, dt * 1e3 / max(ks.its, 1), err), flush=True)
