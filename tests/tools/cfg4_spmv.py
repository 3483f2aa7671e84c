"""Config-4 SpMV probe (BASELINE.json configs[3]): MatMult of the irregular stand-ins through the product path.

  python3 tests/tools/cfg4_spmv.py irr|fem [reps]

  irr  SURVEY 8(d)'s IRR stand-in for Flan_1565 (tests/problems.py: gen_irr)
  fem  unstructured-FEM-like stand-in: 3 dof per node, jittered-grid nodes connected to their ~26 nearest
       neighbours, nodes renumbered by reverse Cuthill-McKee (tests/problems.py: gen_fem3)

The generated CSR is cached under $CFG4_CACHE (default /tmp) so that the separate rocprofv3 --pmc passes of one
gpurun call do not regenerate it.  Prints ms per MatMult (HIP events on the compute stream) and GB/s of the CSR
algorithmic bytes 12 nnz + 4(m+1) + 16 m."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def cached(name, gen):
    d = os.environ.get("CFG4_CACHE", "/tmp")
    path = os.path.join(d, "cfg4_%s.npz" % name)
    if os.path.exists(path):
        z = np.load(path)
        return z["ai"], z["aj"], z["aa"]
    ai, aj, aa = gen()
    try:
        np.savez(path, ai=ai, aj=aj, aa=aa)
    except OSError:
        pass
    return ai, aj, aa


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "irr"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    import problems
    import petsc_dev_amd as pda  # noqa: F401
    from petsc_dev_amd import petsc as P
    L = P.lib()
    t0 = time.time()
    if which == "irr":
        ai, aj, aa = cached("irr", problems.gen_irr)
    else:
        ai, aj, aa = cached("fem", problems.gen_fem3)
    n = ai.size - 1
    print("%s: n=%d nnz=%d (%.1f/row, max %d) ready in %.1fs" % (which, n, aj.size, aj.size / n, np.diff(ai).max(), time.time() - t0), flush=True)
    if os.environ.get("PETSC_OPTIONS_EXTRA"):
        L.PetscOptionsInsertString(os.environ["PETSC_OPTIONS_EXTRA"].encode())
    A = P.Mat.from_csr(ai, aj, aa)
    x = P.Vec.from_array(np.sin(0.37 * np.arange(n)) + 1.0, comm=L.COMM_SELF)
    y = x.duplicate()
    t0 = time.time()
    A.mult(x, y)
    y.array()
    print("first MatMult (analysis + upload) %.2fs" % (time.time() - t0), flush=True)
    nodes, groups, shared = C.c_int(), C.c_int(), C.c_int()
    L.MatHIPMI355XGetInodeInfo(A.h, C.byref(nodes), C.byref(groups), C.byref(shared))
    print("  inode check: %d nodes; device plan: %d groups, %d shared column indices (%.2f per nonzero)" % (nodes.value, groups.value, shared.value, shared.value / max(aj.size, 1)), flush=True)
    B = 12 * aj.size + 4 * (n + 1) + 16 * n
    L.MatHIPMI355XSetTiming(A.h, 1)
    for _ in range(reps):
        A.mult(x, y)
    nl, tms = C.c_int(), C.c_double()
    L.MatHIPMI355XGetTiming(A.h, C.byref(nl), C.byref(tms))
    L.MatHIPMI355XSetTiming(A.h, 0)
    t = tms.value / nl.value * 1e-3
    print("%s MatMult: %.4f ms  %.1f GB/s of CSR-algorithmic bytes (%d B) = %.3f of 8 TB/s" % (which, t * 1e3, B / t / 1e9, B, B / t / 8e12), flush=True)
    if os.environ.get("CFG4_TRANSPOSE"):                 # the cached transpose's product (wall clock over `reps` products, one host wait at the end)
        t0 = time.time()
        L.MatMultTranspose(A.h, x.h, y.h)
        y.array()
        print("first MatMultTranspose (transpose built, uploaded) %.2fs" % (time.time() - t0), flush=True)
        t0 = time.time()
        for _ in range(reps):
            L.MatMultTranspose(A.h, x.h, y.h)
        y.array()
        tt = (time.time() - t0) / reps
        print("%s MatMultTranspose: %.4f ms  = %.3f of 8 TB/s (wall clock incl. one read-back of y per %d products)" % (which, tt * 1e3, B / tt / 8e12, reps), flush=True)
    if n <= 200000:
        import orc
        ref = orc.spmv(ai, aj, aa, np.sin(0.37 * np.arange(n)) + 1.0)
        err = np.max(np.abs(y.array() - ref))
        print("  max |err| vs oracle %g, bit-exact %s" % (err, np.array_equal(y.array().view(np.uint64), ref.view(np.uint64))))


if __name__ == "__main__":
    main()
