"""MatSetValuesBatch: the reference's loop of MatSetValues (first assembly) against the device-side value assembly through
the cached map (every later assembly with the same connectivity).  Hex8 connectivity on an ne^3 element grid, scalar dof."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ne = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    import petsc_dev_amd as pda
    from petsc_dev_amd import petsc as P
    L = P.lib()
    k = pda.load_kernels()
    nn1 = ne + 1
    e = np.arange(ne ** 3)
    ei, ej, ek = e % ne, (e // ne) % ne, e // (ne * ne)
    base = ei + nn1 * ej + nn1 * nn1 * ek
    offs = np.array([0, 1, nn1, nn1 + 1, nn1 * nn1, nn1 * nn1 + 1, nn1 * nn1 + nn1, nn1 * nn1 + nn1 + 1])
    rows = (base[:, None] + offs[None, :]).astype(np.int32)
    nb, bs = rows.shape
    nn = nn1 ** 3
    rng = np.random.default_rng(1)
    ke = rng.standard_normal((bs, bs)); ke = ke + ke.T + 16 * np.eye(bs)
    v = np.ascontiguousarray(np.broadcast_to(ke, (nb, bs, bs)) * (1.0 + 0.1 * rng.random(nb))[:, None, None])
    A = P.Mat(); L.MatCreate(L.COMM_SELF, C.byref(A.h))
    L.MatSetSizes(A.h, nn, nn, nn, nn); L.MatSetType(A.h, b"seqaijhipmi355x")
    nnz = np.full(nn, 27, dtype=np.int32)
    L.MatSeqAIJSetPreallocation(A.h, 27, nnz.ctypes.data_as(C.c_void_p))
    rp, vp_ = rows.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)
    t0 = time.perf_counter()
    L.MatSetValuesBatch(A.h, nb, bs, rp, vp_); L.MatAssemblyBegin(A.h, 0); L.MatAssemblyEnd(A.h, 0)
    t1 = time.perf_counter()
    x = P.Vec.create(nn, comm=L.COMM_SELF); L.VecSet(x.h, 1.0); y = x.duplicate()
    A.mult(x, y); k.mi355x_device_synchronize()
    t2 = time.perf_counter()
    L.MatZeroEntries(A.h)
    L.MatSetValuesBatch(A.h, nb, bs, rp, vp_); L.MatAssemblyBegin(A.h, 0); L.MatAssemblyEnd(A.h, 0)   # builds the map
    k.mi355x_device_synchronize()
    t3 = time.perf_counter()
    L.MatZeroEntries(A.h)
    L.MatSetValuesBatch(A.h, nb, bs, rp, vp_); L.MatAssemblyBegin(A.h, 0); L.MatAssemblyEnd(A.h, 0)   # map reused
    k.mi355x_device_synchronize()
    t4 = time.perf_counter()
    print("hex8 %d^3 elements: %d blocks of %dx%d = %.1f M contributions, %d rows" % (ne, nb, bs, bs, nb * bs * bs / 1e6, nn))
    print("  loop of MatSetValues (reference default, host)      %8.3f s" % (t1 - t0))
    print("  device assembly incl. building the map (host sort)  %8.3f s" % (t3 - t2))
    print("  device assembly, map reused                         %8.3f s   (%.0f M contributions/s incl. %.0f MB of values over PCIe and the host mirror refresh)"
          % (t4 - t3, nb * bs * bs / (t4 - t3) / 1e6, v.nbytes / 1e6))


if __name__ == "__main__":
    main()
