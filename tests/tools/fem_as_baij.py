"""Feasibility: the FEM stand-in (3 dof per node, full 3x3 couplings) stored as BAIJ bs = 3 and multiplied by the BCSR row-block kernel,
against its AIJ product (grouped rows of the inode kernel).   python3 tests/tools/fem_as_baij.py"""
import ctypes as C
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))


def main():
    import problems
    from cfg4_spmv import cached
    from petsc_dev_amd import petsc as P
    L = P.lib()
    ai, aj, aa = cached("fem", problems.gen_fem3)
    n = ai.size - 1
    S = sp.csr_matrix((aa, aj, ai), shape=(n, n))
    B = sp.bsr_matrix(S, blocksize=(3, 3))
    B.sort_indices()
    print("fem: n=%d nnz=%d; as BSR(3): %d blocks, %d stored values (%.3f of them explicit zeros)" % (n, aj.size, B.indices.size, B.data.size, 1.0 - aj.size / B.data.size), flush=True)
    x = np.sin(0.37 * np.arange(n)) + 1.0
    Bmat = 12 * aj.size + 4 * (n + 1) + 16 * n

    def time_mult(A, label):
        vx = P.Vec.from_array(x, comm=L.COMM_SELF); vy = vx.duplicate()
        A.mult(vx, vy); y = vy.array().copy()
        L.MatHIPMI355XSetTiming(A.h, 1)
        for _ in range(30):
            A.mult(vx, vy)
        nl, tms = C.c_int(), C.c_double()
        L.MatHIPMI355XGetTiming(A.h, C.byref(nl), C.byref(tms))
        L.MatHIPMI355XSetTiming(A.h, 0)
        t = tms.value / nl.value * 1e-3
        print("%-34s %.4f ms = %.3f of 8 TB/s by the CSR-algorithmic bytes" % (label, t * 1e3, Bmat / t / 8e12), flush=True)
        return y

    L.PetscOptionsInsertString(b"-mat_hipmi355x_blocked 0")
    y0 = time_mult(P.Mat.from_csr(ai, aj, aa), "AIJ (grouped-row kernel):")
    L.PetscOptionsClear()
    yb = time_mult(P.Mat.from_csr(ai, aj, aa), "AIJ (blocked companion, default):")
    assert np.array_equal(yb, y0)
    ba = np.ascontiguousarray(B.data.transpose(0, 2, 1)).reshape(-1)       # blocks column-major: [block][column][row]
    y1 = time_mult(P.Mat.from_bsr(3, B.indptr.astype(np.int32), B.indices.astype(np.int32), ba), "BAIJ bs = 3 (row-block BCSR):")
    scale = np.abs(S) @ np.abs(x)
    print("max |difference| / (sum |a x|): %.2e" % np.max(np.abs(y1 - y0) / scale), flush=True)


if __name__ == "__main__":
    main()
