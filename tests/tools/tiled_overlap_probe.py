"""Do the column-tiled product's two parts run beside each other when they are given a stream each?  Staged part on one handle's
stream, remainder on another's, 20 products queued on both without a dependency between the streams; wall time against the same work
on one stream.   python3 tests/tools/tiled_overlap_probe.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))


def main():
    import problems
    import tiled
    from cfg4_spmv import cached
    from gpu import Dev
    dev = Dev()
    k = dev.k
    h2 = C.c_void_p()
    dev.chk(k.mi355x_handle_create(C.byref(h2)))
    ai, aj, aa = cached("irr", problems.gen_irr)
    m = n = ai.size - 1
    x = np.sin(0.37 * np.arange(n)) + 1.0
    daa = dev.put(np.concatenate((aa, [0.0, 0.0])))
    dx, dy, dy2 = dev.put(x), dev.alloc(8 * m), dev.alloc(8 * m)
    tp = tiled.build(k, ai, aj, n, 1024)
    dev.chk(k.mi355x_spmv_tiled_upload(dev.h, tp, daa))
    k.mi355x_spmv_tiled_drop_host(tp)
    dev.sync()
    reps = 20

    def run(two):
        for _ in range(3):
            k.mi355x_spmv_tiled_parts(dev.h, tp, dx, None, dy, 1)
            k.mi355x_spmv_tiled_parts(h2 if two else dev.h, tp, dx, None, dy2, 2)
        k.mi355x_handle_synchronize(dev.h); k.mi355x_handle_synchronize(h2)
        t0 = time.perf_counter()
        for _ in range(reps):
            k.mi355x_spmv_tiled_parts(dev.h, tp, dx, None, dy, 1)
            k.mi355x_spmv_tiled_parts(h2 if two else dev.h, tp, dx, None, dy2, 2)
        k.mi355x_handle_synchronize(dev.h); k.mi355x_handle_synchronize(h2)
        return (time.perf_counter() - t0) / reps * 1e3

    for two in (False, True, False, True):
        print("%s: %.4f ms per product (wall clock over %d products, both parts)" % ("two streams" if two else "one stream ", run(two), reps), flush=True)


if __name__ == "__main__":
    main()
