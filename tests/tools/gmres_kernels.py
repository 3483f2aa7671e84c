"""Back-to-back times of the many-stream kernels of a GMRES(30) step at n = 2^24 (development aid)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu import Dev  # noqa: E402
from kbench import timeit  # noqa: E402


def main():
    n = 1 << 24
    dev = Dev()
    k = dev.k
    vs = [dev.alloc(8 * n) for _ in range(33)]
    for p in vs:
        k.mi355x_vec_set(dev.h, n, 1e-3, p)
    ds = C.c_void_p(k.mi355x_handle_device_scratch(dev.h))
    coef = dev.put(np.full(32, 1e-9))
    for nv in (1, 4, 8, 16, 30):
        tab = dev.ptr_table(vs[:nv])
        x = vs[32]
        alpha = np.full(nv, 1e-9)
        t1 = timeit(dev, lambda: k.mi355x_vec_mdot(dev.h, n, nv, x, tab, ds))
        t2 = timeit(dev, lambda: k.mi355x_vec_maxpy(dev.h, n, nv, alpha.ctypes.data_as(C.POINTER(C.c_double)), tab, x))
        t3 = timeit(dev, lambda: k.mi355x_vec_maxpy_dev_norm2(dev.h, n, nv, coef, -1.0, tab, x, ds))
        gb = 8.0 * n / 1e9
        print("nv %2d: mdot %.4f ms %5.0f GB/s | maxpy %.4f ms %5.0f GB/s | maxpy+norm %.4f ms %5.0f GB/s   (ideal bytes: nv+1, nv+2 vectors)"
              % (nv, t1 * 1e3, (nv + 1) * gb / t1, t2 * 1e3, (nv + 2) * gb / t2, t3 * 1e3, (nv + 2) * gb / t3), flush=True)
    t = timeit(dev, lambda: k.mi355x_vec_scale_rnorm_dev(dev.h, n, coef, vs[0]))
    print("scale_rnorm_dev %.4f ms %5.0f GB/s" % (t * 1e3, 2 * 8.0 * n / 1e9 / t))
    t = timeit(dev, lambda: k.mi355x_vec_norm(dev.h, n, 2, vs[0], ds))
    print("norm2 %.4f ms %5.0f GB/s" % (t * 1e3, 8.0 * n / 1e9 / t))


if __name__ == "__main__":
    main()
