"""ILU(0) application time on P7(n): ilu_bench.py N [syncfree|level] -- sync-free two-launch solves or level-scheduled launches captured in a hipGraph, vs the CPU restatement."""
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
dims = [int(v) for v in os.environ.get("ILU_BENCH_DIMS", "%d,%d,%d" % (n, n, n)).split(",")]   # e.g. 16,16,4096: few rows per level
ai, aj, aa = P.gen_poisson7(*dims)
A = P.Mat.from_csr(ai, aj, aa)
N = dims[0] * dims[1] * dims[2]
b = P.Vec.from_array(np.sin(0.1 * np.arange(N)), comm=L.COMM_SELF); x = b.duplicate()
ksp = P.KSP(comm=L.COMM_SELF); ksp.set_operators(A)
pc = C.c_void_p(); L.KSPGetPC(ksp.h, C.byref(pc)); L.PCSetType(pc, b"ilu")
if len(sys.argv) > 2:
    L.PetscOptionsInsertString(("-pc_factor_hipmi355x_trisolve " + sys.argv[2]).encode())
t0 = time.time(); L.raw("PCSetUp")(pc); print("P7(%d): ILU(0) set-up (host factorisation + levels + upload) %.2f s" % (n, time.time() - t0))
nl, nu = C.c_int(), C.c_int(); L.PCILUGetLevels_HIPMI355X(pc, C.byref(nl), C.byref(nu))
for _ in range(3):
    L.raw("PCApply")(pc, b.h, x.h)
k.mi355x_device_synchronize()
t0 = time.perf_counter()
for _ in range(10):
    L.raw("PCApply")(pc, b.h, x.h)
k.mi355x_device_synchronize()
t = (time.perf_counter() - t0) / 10
sf, ab = C.c_int(), C.c_int(); L.PCILUGetSolver_HIPMI355X(pc, C.byref(sf), C.byref(ab))
print("levels L=%d U=%d ; %s ; PCApply %.3f ms  (%.2f us per level) aborted=%d" % (nl.value, nu.value, "sync-free, 2 launches" if sf.value else "level launches (hipGraph)", t * 1e3, t * 1e6 / (nl.value + nu.value), ab.value))
if n <= 128:
    f = orc.ilu0_factor(ai, aj, aa)
    t0 = time.perf_counter(); ref = orc.ilu0_solve(f, np.sin(0.1 * np.arange(N))); tc = time.perf_counter() - t0
    print("oracle (1 core) solve %.3f ms ; bit-exact: %s" % (tc * 1e3, np.array_equal(x.array().view(np.uint64), ref.view(np.uint64))))
