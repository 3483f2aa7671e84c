"""Column-tiled SpMV on the config-4 stand-ins, through the kernel library: ms per product for the row-block kernel, the tiled product,
its staged part and its remainder alone; stage_min sweep.   python3 tests/tools/tiled_probe.py irr|fem [stage_min ...]"""
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
    which = sys.argv[1] if len(sys.argv) > 1 else "irr"
    smins = [int(a) for a in sys.argv[2:]] or [1024]
    import problems
    import tiled
    from cfg4_spmv import cached
    from gpu import Dev
    from bench_configs import event_time
    dev = Dev()
    k = dev.k
    ai, aj, aa = cached(which, problems.gen_irr if which == "irr" else problems.gen_fem3)
    m = n = ai.size - 1
    nnz = aj.size
    B = 12 * nnz + 4 * (m + 1) + 16 * m
    print("%s: n=%d nnz=%d geometry %s" % (which, n, nnz, tiled.geometry(k)), flush=True)
    lpn = C.c_double()
    k.mi355x_spmv_tiled_probe(m, ai.ctypes.data, aj.ctypes.data, C.byref(lpn))
    print("  lines of x per nonzero in 32-row groups: %.3f" % lpn.value, flush=True)
    x = np.sin(0.37 * np.arange(n)) + 1.0
    dai, daj = dev.put(ai), dev.put(np.concatenate((aj, np.zeros(4, np.int32))))
    daa = dev.put(np.concatenate((aa, [0.0, 0.0])))
    dx, dy = dev.put(x), dev.alloc(8 * m)
    plan = C.c_void_p()
    dev.chk(k.mi355x_spmv_plan_create(dev.h, m, ai.ctypes.data, None, C.byref(plan)))
    t = event_time(k, dev.h, lambda: k.mi355x_spmv_csr(dev.h, plan, dai, daj, daa, dx, dy), 20)
    ref = dev.get(dy, m)
    print("row-block kernel:            %.4f ms  %.3f of 8 TB/s (CSR-algorithmic bytes)" % (t * 1e3, B / t / 8e12), flush=True)
    for smin in smins:
        t0 = time.time()
        tp = tiled.build(k, ai, aj, n, smin)
        tb = time.time() - t0
        inf = tiled.info(k, tp)
        dev.chk(k.mi355x_spmv_tiled_upload(dev.h, tp, daa))
        k.mi355x_spmv_tiled_drop_host(tp)
        dev.sync()
        only = os.environ.get("TILED_PROBE_WHICH")                # (counters of one part alone: every launch of the process is that part)
        ts = [event_time(k, dev.h, lambda w=w: k.mi355x_spmv_tiled_parts(dev.h, tp, dx, None, dy, int(only) if only else w), 20) for w in (0, 1, 2)]
        tr = event_time(k, dev.h, lambda: k.mi355x_spmv_tiled_refresh_values(dev.h, tp, daa), 5)
        print("  values refreshed from the CSR array on the device (one permutation gather): %.3f ms" % (tr * 1e3), flush=True)
        dev.chk(k.mi355x_spmv_tiled(dev.h, tp, dx, None, dy))
        err = np.max(np.abs(dev.get(dy, m) - ref) / (np.abs(ref) + 1.0))
        print("tiled stage_min %6d: build %.2fs staged %.1f%% in %d pairs / %d blocks (%.1f%% of the stored entries are padding) | both %.4f ms = %.3f of 8 TB/s ; staged part %.4f ms ; remainder %.4f ms ; max rel diff vs row-block %.2g"
              % (smin, tb, 100.0 * inf["staged"] / nnz, inf["pairs"], inf["blocks"], 100.0 - 100.0 * (inf["staged"] + inf["remainder"]) / max(inf["blocks"] * 128, 1), ts[0] * 1e3, B / ts[0] / 8e12, ts[1] * 1e3, ts[2] * 1e3, err), flush=True)
        k.mi355x_spmv_tiled_destroy(tp)


if __name__ == "__main__":
    main()
