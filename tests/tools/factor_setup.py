"""Set-up and application time of PCILU / PCICC on P7(n) (7-point Poisson, BASELINE configs[1]'s operator) or the FEM stand-in; the
matrix is used once before the preconditioner is set up, so that its own upload does not count.  PETSC_HIPMI355X_SETUP_TIMING=1 and
MI355X_TRISOLVE_TIMING=1 print where the set-up time goes.
  python3 tests/tools/factor_setup.py [p7:256|fem] [ilu|icc]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems as pb  # noqa: E402


def main():
    prob = sys.argv[1] if len(sys.argv) > 1 else "p7:256"
    pct = sys.argv[2] if len(sys.argv) > 2 else "ilu"
    import petsc_dev_amd as pda
    from petsc_dev_amd import petsc as P
    L = P.lib(); k = pda.load_kernels()
    if prob.startswith("p7"):
        m = int(prob.split(":")[1]) if ":" in prob else 256
        ai, aj, aa = P.gen_poisson7(m, m, m)
    else:
        ai, aj, aa = pb.gen_fem3()
    n = ai.size - 1
    A = P.Mat.from_csr(ai, aj, aa)
    b = P.Vec.from_array(np.sin(0.1 * np.arange(n)), comm=L.COMM_SELF); x = b.duplicate()
    t0 = time.time(); A.mult(b, x); k.mi355x_device_synchronize()
    print("first product (upload of the matrix, analysis of its pattern and values, SpMV plan): %.3f s" % (time.time() - t0), flush=True)
    ksp = P.KSP(comm=L.COMM_SELF); ksp.set_operators(A)
    pc = C.c_void_p(); L.KSPGetPC(ksp.h, C.byref(pc)); L.PCSetType(pc, pct.encode())
    L.PetscOptionsInsertString(os.environ.get("FEM_OPTS", "").encode())
    t0 = time.time(); L.raw("PCSetUp")(pc); k.mi355x_device_synchronize(); tset = time.time() - t0
    for _ in range(3):
        L.raw("PCApply")(pc, b.h, x.h)
    k.mi355x_device_synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        L.raw("PCApply")(pc, b.h, x.h)
    k.mi355x_device_synchronize()
    t = (time.perf_counter() - t0) / 10
    print("%s n=%d nnz=%d, pc %s: set-up %.3f s, PCApply %.3f ms" % (prob, n, aj.size, pct, tset, t * 1e3), flush=True)


if __name__ == "__main__":
    main()
