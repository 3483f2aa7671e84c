"""ILU(0) application time on the FEM-like stand-in of BASELINE configs[3] (tests/problems.py gen_fem3): the node-blocked sync-free
triangular solves (development aid: MI355X_KERNELS_LIB / MI355X_TRISOLVE_AHEAD / MI355X_TRISOLVE_SLEEP / FEM_OPTS select variants).
  python3 tests/tools/fem_ilu_apply.py [ex ey ez]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems as pb  # noqa: E402


def main():
    dims = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else [130, 130, 67]
    import petsc_dev_amd as pda
    from petsc_dev_amd import petsc as P
    L = P.lib(); k = pda.load_kernels()
    ai, aj, aa = pb.gen_fem3(*dims)
    n = ai.size - 1
    A = P.Mat.from_csr(ai, aj, aa)
    b = P.Vec.from_array(np.sin(0.1 * np.arange(n)), comm=L.COMM_SELF); x = b.duplicate()
    ksp = P.KSP(comm=L.COMM_SELF); ksp.set_operators(A)
    pc = C.c_void_p(); L.KSPGetPC(ksp.h, C.byref(pc)); L.PCSetType(pc, b"ilu")
    L.PetscOptionsInsertString(os.environ.get("FEM_OPTS", "").encode())
    t0 = time.time(); L.raw("PCSetUp")(pc); tset = time.time() - t0
    nl, nu = C.c_int(), C.c_int(); L.PCILUGetLevels_HIPMI355X(pc, C.byref(nl), C.byref(nu))
    nn, nnl, nnu = C.c_int(), C.c_int(), C.c_int(); L.PCILUGetNodeInfo_HIPMI355X(pc, C.byref(nn), C.byref(nnl), C.byref(nnu))
    for _ in range(3):
        L.raw("PCApply")(pc, b.h, x.h)
    k.mi355x_device_synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        L.raw("PCApply")(pc, b.h, x.h)
    k.mi355x_device_synchronize()
    t = (time.perf_counter() - t0) / 10
    sf, ab = C.c_int(), C.c_int(); L.PCILUGetSolver_HIPMI355X(pc, C.byref(sf), C.byref(ab))
    lev = (nnl.value + nnu.value) if nn.value else (nl.value + nu.value)
    print("fem %s n=%d: set-up %.2f s; %s, %d levels; PCApply %.3f ms = %.2f us per level; aborted=%d  [%s lib=%s ahead=%s sleep=%s]"
          % ("x".join(map(str, dims)), n, tset, ("node plans (%d nodes%s)" % (abs(nn.value), ", block columns" if nn.value < 0 else "")) if nn.value else "row plans", lev, t * 1e3, t * 1e6 / max(lev, 1), ab.value,
             os.environ.get("FEM_OPTS", ""), os.path.basename(os.environ.get("MI355X_KERNELS_LIB", "default")), os.environ.get("MI355X_TRISOLVE_AHEAD", "-"), os.environ.get("MI355X_TRISOLVE_SLEEP", "-")), flush=True)


if __name__ == "__main__":
    main()
