"""Where does a level of the sync-free triangular solve spend its time?  Needs the trace build of the kernel library:
  bash petsc-dev_amd/csrc/variants/build_tri_trace.sh
  MI355X_KERNELS_LIB=$PWD/petsc-dev_amd/csrc/variants/libmi355x_kernels_tritrace.so python tests/tools/tri_trace.py fem"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems as pb  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "fem"
    import petsc_dev_amd as pda
    from petsc_dev_amd import petsc as P
    L = P.lib(); k = pda.load_kernels()
    if which.startswith("p7:"):
        m = int(which.split(":")[1]); ai, aj, aa = P.gen_poisson7(m, m, m)
    else:
        ai, aj, aa = pb.gen_fem3() if which == "fem" else P.gen_poisson7(16, 16, 4096)
    n = ai.size - 1
    A = P.Mat.from_csr(ai, aj, aa)
    b = P.Vec.create(n, comm=L.COMM_SELF); L.VecSet(b.h, 1.0); u = b.duplicate()
    ksp = P.KSP(comm=L.COMM_SELF); ksp.set_operators(A)
    pc = C.c_void_p(); L.KSPGetPC(ksp.h, C.byref(pc)); L.PCSetType(pc, b"ilu")
    L.raw("PCSetUp")(pc)
    nslots = 16000000                                   # lower solve: slots [0, 8e6), upper: [8e6, 16e6); 4 per slice
    buf = C.c_void_p(); k.mi355x_malloc(C.byref(buf), 8 * nslots)
    h = C.c_void_p(); k.mi355x_handle_create(C.byref(h))
    k.mi355x_memset(h, buf, 0, 8 * nslots) if hasattr(k, "mi355x_memset") else None
    for _ in range(2):
        L.raw("PCApply")(pc, b.h, u.h)
    k.mi355x_device_synchronize()
    k.mi355x_trisolve_debug_trace.argtypes = [C.c_void_p]
    assert k.mi355x_trisolve_debug_trace(buf) == 0
    L.raw("PCApply")(pc, b.h, u.h)
    k.mi355x_device_synchronize()
    t = np.zeros(nslots, dtype=np.int64)
    k.mi355x_memcpy_d2h(h, t.ctypes.data, buf, 8 * nslots); k.mi355x_handle_synchronize(h)
    t = t.reshape(-1, 4)
    used = t[:, 3] > 0
    t = t[used]
    order = np.argsort(t[:, 3])
    t = t[order].astype(np.float64) * 0.01          # 100 MHz -> microseconds
    t -= t[:, 0].min()
    print("%d slices traced; span %.1f us" % (t.shape[0], t[:, 3].max()))
    d_store = np.diff(t[:, 3])
    print("between consecutive stores: median %.2f us, mean %.2f us" % (np.median(d_store), d_store.mean()))
    print("slice start -> last batch looked at: median %.2f us" % np.median(t[:, 1] - t[:, 0]))
    print("last batch looked at -> last dependency in: median %.2f us (the wait)" % np.median(t[:, 2] - t[:, 1]))
    print("last dependency in -> stored: median %.3f us" % np.median(t[:, 3] - t[:, 2]))
    print("how long before its own store a slice started: median %.2f us, 10%% %.2f us" % (np.median(t[:, 3] - t[:, 0]), np.percentile(t[:, 3] - t[:, 0], 10)))
    for i in range(2000, 2012):
        print("  slice %d: start %.2f  lastlook %.2f  dep_in %.2f  stored %.2f" % (i, t[i, 0], t[i, 1], t[i, 2], t[i, 3]))


if __name__ == "__main__":
    main()
