"""Value-pattern SpMV alone on P7(N) (development aid): ms per launch, bytes the kernel has to move, check against the streamed kernel.
  MI355X_KERNELS_LIB=.../variants/libmi355x_kernels_NAME.so python tests/tools/vpat_bench.py [N]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
from gpu import Dev  # noqa: E402
from kbench import timeit  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    tag = os.path.basename(os.environ.get("MI355X_KERNELS_LIB", "default"))
    dev = Dev(); k = dev.k
    ai, aj, aa = orc.gen_p7(N, N, N)
    m = ai.size - 1
    dai, daj, daa = dev.put(ai), dev.put(aj), dev.put(aa)
    dx = dev.put(np.sin(0.37 * np.arange(m)) + 1.0)
    dy, dz = dev.alloc(8 * m), dev.alloc(8 * m)
    plan = C.c_void_p()
    dev.chk(k.mi355x_spmv_plan_create(dev.h, m, ai.ctypes.data, None, C.byref(plan)))
    dev.chk(k.mi355x_spmv_plan_compress_indices(dev.h, plan, ai.ctypes.data, aj.ctypes.data))
    t_pat = timeit(dev, lambda: k.mi355x_spmv_csr(dev.h, plan, dai, daj, daa, dx, dy), reps=50)
    nv = C.c_int()
    mode = sys.argv[2] if len(sys.argv) > 2 else ""
    t_val, same = 1.0, None
    if mode != "probeonly":       # (probe builds of the kernel library must not run the real operator: their indexing is only valid for the synthetic ones)
        dev.chk(k.mi355x_spmv_plan_value_patterns(dev.h, plan, ai.ctypes.data, aj.ctypes.data, aa.ctypes.data, C.byref(nv)))
        t_val = timeit(dev, lambda: k.mi355x_spmv_csr(dev.h, plan, dai, daj, daa, dx, dz), reps=50)
        same = np.array_equal(dev.get(dy, m).view(np.uint64), dev.get(dz, m).view(np.uint64))
    if mode in ("probe", "probeonly"):
        # where the time goes: the same instruction stream with every gather of a row pointed at its diagonal entry (one L1
        # line per wavefront instead of up to seven), and the same with 1 entry per row (instruction count / 7)
        for name, per_row in (("7 entries per row, all at the diagonal", 7), ("1 entry per row", 1)):
            ai2 = (np.arange(m + 1, dtype=np.int64) * per_row).astype(np.int32)
            aj2 = np.repeat(np.arange(m, dtype=np.int32), per_row)
            aa2 = np.tile(np.array([6.0, -1.0, -1.0, -1.0, -1.0, -1.0, -1.0][:per_row]), m)
            plan2 = C.c_void_p()
            dev.chk(k.mi355x_spmv_plan_create(dev.h, m, ai2.ctypes.data, None, C.byref(plan2)))
            dev.chk(k.mi355x_spmv_plan_value_patterns(dev.h, plan2, ai2.ctypes.data, aj2.ctypes.data, aa2.ctypes.data, C.byref(nv)))
            assert nv.value == 1
            t2 = timeit(dev, lambda: k.mi355x_spmv_csr(dev.h, plan2, dai, daj, daa, dx, dz), reps=50)
            print("  probe: %-44s %.4f ms" % (name, t2 * 1e3), flush=True)
            dev.chk(k.mi355x_spmv_plan_destroy(plan2))
    need = 2 * m + 16 * m
    print("%-44s P7(%d) %d kinds of rows: value patterns %.4f ms (%.0f GB/s of 2+8+8 B/row), streamed values %.4f ms, same bits %s"
          % (tag, N, nv.value, t_val * 1e3, need / t_val / 1e9, t_pat * 1e3, same), flush=True)


if __name__ == "__main__":
    main()
