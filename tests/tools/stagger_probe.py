"""Does the relative placement of equally sized, power-of-two-length vectors matter for the streaming kernels?
triad / cg_update on vectors carved out of one allocation with a given gap between consecutive vectors."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu import Dev  # noqa: E402
from kbench import timeit  # noqa: E402


def main():
    n = 1 << 24
    dev = Dev()
    k = dev.k
    for gap in (0, 256, 4096, 8192 + 256, 65536 + 4096 + 256, (1 << 20) + 8192):
        stride = n * 8 + gap
        base = dev.alloc(stride * 8 + 64)
        v = [C.c_void_p(base.value + i * stride) for i in range(8)]
        for p in v:
            k.mi355x_vec_set(dev.h, n, 1.0, p)
        out = dev.host_scratch()
        t1 = timeit(dev, lambda: k.mi355x_stream_triad(dev.h, n, 0.5, v[0], v[1], v[2]))
        t2 = timeit(dev, lambda: k.mi355x_vec_cg_update(dev.h, n, 1e-9, v[0], v[1], v[2], v[3], v[4], v[5], out))
        t3 = timeit(dev, lambda: k.mi355x_vec_aypx(dev.h, n, 0.5, v[0], v[1]))
        t4 = timeit(dev, lambda: k.mi355x_vec_dot(dev.h, n, v[0], v[1], out))
        print("gap %8d B: triad %.4f ms (%.0f GB/s)  cg_update %.4f ms (%.0f GB/s)  aypx %.4f ms  dot %.4f ms"
              % (gap, t1 * 1e3, 24 * n / t1 / 1e9, t2 * 1e3, 64 * n / t2 / 1e9, t3 * 1e3, t4 * 1e3), flush=True)
        dev.free(base)


if __name__ == "__main__":
    main()
