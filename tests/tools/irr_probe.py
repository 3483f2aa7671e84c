"""Where does the irregular-matrix SpMV spend its time?  Same row-length distribution, three column patterns."""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
from gpu import Dev
from kbench import timeit


def gen(n, band, far, seed=1):
    rng = np.random.default_rng(seed)
    lens = np.clip(np.exp(rng.normal(np.log(73) - 0.18, 0.6, n)), 3, 400).astype(np.int64)
    tot = int(lens.sum())
    rows = np.repeat(np.arange(n, dtype=np.int64), lens)
    cols = np.clip(rows + rng.integers(-band, band + 1, tot), 0, n - 1)
    if far > 0:
        f = rng.random(tot) < far
        cols = np.where(f, rng.integers(0, n, tot), cols)
    key = np.unique(rows * n + cols)
    rows = key // n
    cols = (key - rows * n).astype(np.int32)
    ai = np.zeros(n + 1, dtype=np.int64); np.add.at(ai, rows + 1, 1)
    return np.cumsum(ai).astype(np.int32), cols, np.ones(cols.size)


dev = Dev(); k = dev.k
n = 1564794
for name, band, far in (("band64", 64, 0.0), ("band50000", 50000, 0.0), ("band50000+20%far", 50000, 0.2)):
    ai, aj, aa = gen(n, band, far)
    dai, daj, daa = dev.put(ai), dev.put(aj), dev.put(aa)
    dx = dev.put(np.ones(n)); dy = dev.alloc(8 * n)
    plan = C.c_void_p()
    dev.chk(k.mi355x_spmv_plan_create(dev.h, n, ai.ctypes.data, None, C.byref(plan)))
    t = timeit(dev, lambda: k.mi355x_spmv_csr(dev.h, plan, dai, daj, daa, dx, dy))
    B = 12 * aj.size + 4 * (n + 1) + 16 * n
    nb = C.c_int(); k.mi355x_spmv_plan_info(plan, C.byref(nb), None, None)
    print("%-18s nnz=%d blocks=%d  %.3f ms  %.1f GB/s" % (name, aj.size, nb.value, t * 1e3, B / t / 1e9), flush=True)
    for p in (dai, daj, daa, dx, dy):
        dev.free(p)
