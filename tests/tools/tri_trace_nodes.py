"""Where does a node level of the node-blocked sync-free triangular solve spend its time?  Trace build of the kernel library:
  bash petsc-dev_amd/csrc/variants/build_tri_trace.sh
  MI355X_KERNELS_LIB=$PWD/petsc-dev_amd/csrc/variants/libmi355x_kernels_tritrace.so python tests/tools/tri_trace_nodes.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems as pb  # noqa: E402


def main():
    import petsc_dev_amd as pda
    from petsc_dev_amd import petsc as P
    L = P.lib(); k = pda.load_kernels()
    ai, aj, aa = pb.gen_fem3()
    n = ai.size - 1
    A = P.Mat.from_csr(ai, aj, aa)
    b = P.Vec.create(n, comm=L.COMM_SELF); L.VecSet(b.h, 1.0); u = b.duplicate()
    ksp = P.KSP(comm=L.COMM_SELF); ksp.set_operators(A)
    pc = C.c_void_p(); L.KSPGetPC(ksp.h, C.byref(pc)); L.PCSetType(pc, b"ilu")
    L.PetscOptionsInsertString(os.environ.get("FEM_OPTS", "-pc_factor_hipmi355x_trisolve_order level").encode())
    L.raw("PCSetUp")(pc)
    nslots = 16000000
    buf = C.c_void_p(); k.mi355x_malloc(C.byref(buf), 8 * nslots)
    h = C.c_void_p(); k.mi355x_handle_create(C.byref(h))
    z = np.zeros(nslots, dtype=np.int64)
    k.mi355x_memcpy_h2d(h, buf, z.ctypes.data, 8 * nslots); k.mi355x_handle_synchronize(h)
    for _ in range(2):
        L.raw("PCApply")(pc, b.h, u.h)
    k.mi355x_device_synchronize()
    k.mi355x_trisolve_debug_trace.argtypes = [C.c_void_p]
    assert k.mi355x_trisolve_debug_trace(buf) == 0
    L.raw("PCApply")(pc, b.h, u.h)
    k.mi355x_device_synchronize()
    t = np.zeros(nslots, dtype=np.int64)
    k.mi355x_memcpy_d2h(h, t.ctypes.data, buf, 8 * nslots); k.mi355x_handle_synchronize(h)
    raw = t[:8000000].reshape(-1, 8).astype(np.float64) * 0.01     # the lower solve, slice by slice
    s0 = int(os.environ.get("TRACE_SLICE", "4000"))
    base = raw[s0:s0 + 24][raw[s0:s0 + 24, 0] > 0][:, 0].min()
    print("slices %d.. in slice order (us, relative): start b0 b1 b2 b3 b4 lastdep stored" % s0)
    for i in range(s0, s0 + 24):
        print("  %6d: " % i + "  ".join("%8.2f" % (v - base if v > 0 else -1) for v in raw[i]))
    # how far ahead of the solve's front does a slice start?  (slices before it that are not stored yet when it starts)
    ok = raw[:, 7] > 0
    ns_ = int(np.nonzero(ok)[0].max()) + 1
    st, en = raw[:ns_, 0], raw[:ns_, 7]
    depth = []
    for i in range(2000, min(ns_, 7000), 37):
        lo_ = max(0, i - 600)
        depth.append(int(np.sum(en[lo_:i] > st[i])))
    print("slices before a slice that are still unfinished when it starts: median %d, 10%% %d, 90%% %d" % (np.median(depth), np.percentile(depth, 10), np.percentile(depth, 90)))
    t = t[:8000000].reshape(-1, 8)                     # the lower solve
    t = t[t[:, 7] > 0].astype(np.float64) * 0.01      # 100 MHz -> microseconds
    t -= t[:, 0].min()
    full = t[(t[:, 5] > 0)]                            # slices with five batches
    print("%d slices traced (%d with >= 5 batches); span %.1f us" % (t.shape[0], full.shape[0], t[:, 7].max()))
    o = np.argsort(full[:, 7]); f = full[o]
    print("between consecutive stores of full slices: median %.2f us" % np.median(np.diff(f[:, 7])))
    names = ["start->b0", "b0->b1", "b1->b2", "b2->b3", "b3->b4", "b4->lastdep", "lastdep->stored"]
    for i, nm in enumerate(names):
        print("  %-16s median %.2f us   10%% %.2f   90%% %.2f" % (nm, np.median(f[:, i + 1] - f[:, i]), np.percentile(f[:, i + 1] - f[:, i], 10), np.percentile(f[:, i + 1] - f[:, i], 90)))
    print("  start->stored    median %.2f us" % np.median(f[:, 7] - f[:, 0]))
    for i in range(1000, 1012):
        print("  slice: " + "  ".join("%.2f" % v for v in f[i]))


if __name__ == "__main__":
    main()
