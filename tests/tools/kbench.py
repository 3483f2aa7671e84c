"""Kernel micro-benchmark (development aid, not the contract bench): GB/s of the SpMV and Vec kernels."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
from gpu import Dev  # noqa: E402


def timeit(dev, fn, reps=20, warm=3):
    k = dev.k
    e0, e1 = C.c_void_p(), C.c_void_p()
    k.mi355x_event_create(C.byref(e0)); k.mi355x_event_create(C.byref(e1))
    for _ in range(warm):
        fn()
    dev.sync()
    k.mi355x_event_record(e0, dev.h)
    for _ in range(reps):
        fn()
    k.mi355x_event_record(e1, dev.h)
    k.mi355x_event_synchronize(e1)
    ms = C.c_float()
    k.mi355x_event_elapsed_ms(e0, e1, C.byref(ms))
    return ms.value / reps * 1e-3


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = Dev()
    k = dev.k
    buf = C.create_string_buffer(256)
    k.mi355x_device_name(buf, 256)
    print("device:", buf.value.decode(), flush=True)
    t0 = time.time()
    ai, aj, aa = orc.gen_p7(N, N, N)
    m = ai.size - 1
    nnz = aj.size
    print("P7(%d): m=%d nnz=%d gen %.1fs" % (N, m, nnz, time.time() - t0), flush=True)
    dai, daj, daa = dev.put(ai), dev.put(aj), dev.put(aa)
    x = np.sin(0.37 * np.arange(m)) + 1.0
    dx = dev.put(x)
    dy = dev.alloc(8 * m)
    dz = dev.alloc(8 * m)
    plan = C.c_void_p()
    t0 = time.time()
    dev.chk(k.mi355x_spmv_plan_create(dev.h, m, ai.ctypes.data, None, C.byref(plan)))
    print("plan %.3fs" % (time.time() - t0), flush=True)
    t = timeit(dev, lambda: k.mi355x_spmv_csr(dev.h, plan, dai, daj, daa, dx, dy))
    B = 12 * nnz + 4 * (m + 1) + 8 * m + 8 * m
    print("spmv_csr      %8.3f ms  %8.1f GB/s (algorithmic %d B)  frac of 8TB/s %.3f" % (t * 1e3, B / t / 1e9, B, B / t / 8e12), flush=True)
    y = dev.get(dy, m)
    if N <= 128:
        ref = orc.spmv(ai, aj, aa, x)
        print("  bit-exact vs oracle:", np.array_equal(y.view(np.uint64), ref.view(np.uint64)))
    if len(sys.argv) > 2 and sys.argv[2] == "spmv":
        return
    out = dev.host_scratch()
    tests = [
        ("triad", 24, lambda: k.mi355x_stream_triad(dev.h, m, 0.5, dx, dy, dz)),
        ("copy", 16, lambda: k.mi355x_vec_copy(dev.h, m, dx, dz)),
        ("set", 8, lambda: k.mi355x_vec_set(dev.h, m, 1.0, dz)),
        ("axpy", 24, lambda: k.mi355x_vec_axpy(dev.h, m, 0.5, dx, dz)),
        ("aypx", 24, lambda: k.mi355x_vec_aypx(dev.h, m, 0.5, dx, dz)),
        ("pwmult", 24, lambda: k.mi355x_vec_pointwise_mult(dev.h, m, dx, dy, dz)),
        ("dot", 16, lambda: k.mi355x_vec_dot(dev.h, m, dx, dy, out)),
        ("norm2", 8, lambda: k.mi355x_vec_norm(dev.h, m, 1, dx, out)),
    ]
    for name, bpe, fn in tests:
        t = timeit(dev, fn)
        print("%-12s %8.3f ms  %8.1f GB/s" % (name, t * 1e3, bpe * m / t / 1e9), flush=True)
    # mdot / maxpy with 8 and 30 vectors
    for nv in (8, 30):
        if 8 * m * (nv + 6) > 200e9:
            continue
        dys = [dev.alloc(8 * m) for _ in range(nv)]
        for p in dys:
            k.mi355x_vec_set(dev.h, m, 0.25, p)
        tab = dev.ptr_table(dys)
        t = timeit(dev, lambda: k.mi355x_vec_mdot(dev.h, m, nv, dx, tab, out), reps=5)
        print("mdot(%2d)     %8.3f ms  %8.1f GB/s" % (nv, t * 1e3, 8 * (nv + 1) * m / t / 1e9), flush=True)
        al = np.full(nv, 1e-3)
        t = timeit(dev, lambda: k.mi355x_vec_maxpy(dev.h, m, nv, al.ctypes.data_as(C.POINTER(C.c_double)), tab, dz), reps=5)
        print("maxpy(%2d)    %8.3f ms  %8.1f GB/s" % (nv, t * 1e3, 8 * (nv + 2) * m / t / 1e9), flush=True)
        for p in dys:
            dev.free(p)


if __name__ == "__main__":
    main()
