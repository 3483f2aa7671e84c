"""Config-5 probe (BASELINE.json configs[4]: BCSR 3-D elasticity, 128^3 nodes, 3 dof per node): the same operator as
  bs3      BAIJ bs = 3, row-block streaming FMA kernel          (MatMult_SeqBAIJ_3, baij2.c:331)
  bs3x/g   the same kernel with x staged in LDS once per block / gathered once per value (A-B of the two forms; bs4x likewise)
  bs4      zero-padded to 4x4 blocks, row-block streaming FMA   (MatMult_SeqBAIJ_4, baij2.c:387)
  bs4mfma  zero-padded to 4x4 blocks, matrix cores, 16-B loads  (v_mfma_f64_4x4x4_4b_f64)
  bs4mfma8 ... 8-B loads
  wave3/4  one wavefront per block row, FMA (no analysis)
all timed interleaved in ONE process with HIP events (rounds x variants), GB/s of each storage's own algorithmic bytes
(8 bs^2 + 4) nnzb + 4 (mbs + 1) + 16 bs mbs, and checked against each other.
  python3 tests/tools/cfg5_baij.py [nodes_per_side=128] [rounds=5]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
from gpu import Dev  # noqa: E402
from bench_configs import gen_baij27  # noqa: E402


def main():
    nn = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
    dev = Dev(); k = dev.k
    t0 = time.time()
    bi, bj, ba3 = gen_baij27(nn, bs=3)
    mbs, nnzb = bi.size - 1, bj.size
    b3 = ba3.reshape(nnzb, 3, 3)                       # [blk][col][row] (column-major blocks)
    b4 = np.zeros((nnzb, 4, 4)); b4[:, :3, :3] = b3
    ba4 = b4.ravel()
    x3 = np.sin(0.1 * np.arange(mbs * 3))
    x4 = np.zeros((mbs, 4)); x4[:, :3] = x3.reshape(mbs, 3); x4 = x4.ravel()
    print("27-point elasticity shape: %d^3 nodes, mbs=%d nnzb=%d, generated in %.1fs" % (nn, mbs, nnzb, time.time() - t0), flush=True)
    dbi, dbj = dev.put(bi), dev.put(bj)
    da3, da4 = dev.put(ba3), dev.put(ba4)
    dx3, dx4 = dev.put(x3), dev.put(x4)
    dy3, dy4 = dev.alloc(8 * mbs * 3), dev.alloc(8 * mbs * 4)
    plans = {}
    for bs in (3, 4):
        p = C.c_void_p()
        sc = (bi.astype(np.int64) * bs * bs)
        assert sc[-1] < 2 ** 31
        sc = sc.astype(np.int32)
        dev.chk(k.mi355x_spmv_plan_create(dev.h, mbs, sc.ctypes.data, None, C.byref(p)))
        plans[bs] = p
    variants = {
        "bs3": (3, lambda: k.mi355x_spmv_bsr_planned(dev.h, plans[3], 3, dbi, dbj, da3, dx3, dy3)),
        "bs3x": (3, lambda: k.mi355x_spmv_bsr_planned_form(dev.h, plans[3], 3, 1, dbi, dbj, da3, dx3, dy3)),
        "bs3g": (3, lambda: k.mi355x_spmv_bsr_planned_form(dev.h, plans[3], 3, 0, dbi, dbj, da3, dx3, dy3)),
        "bs4": (4, lambda: k.mi355x_spmv_bsr_planned(dev.h, plans[4], 4, dbi, dbj, da4, dx4, dy4)),
        "bs4x": (4, lambda: k.mi355x_spmv_bsr_planned_form(dev.h, plans[4], 4, 1, dbi, dbj, da4, dx4, dy4)),
        "bs4mfma": (4, lambda: k.mi355x_spmv_bsr4_mfma(dev.h, mbs, 0, dbi, dbj, da4, dx4, dy4)),
        "bs4mfma8": (4, lambda: k.mi355x_spmv_bsr4_mfma(dev.h, mbs, 1, dbi, dbj, da4, dx4, dy4)),
        "wave3": (3, lambda: k.mi355x_spmv_bsr(dev.h, mbs, 3, dbi, dbj, da3, dx3, dy3)),
        "wave4": (4, lambda: k.mi355x_spmv_bsr(dev.h, mbs, 4, dbi, dbj, da4, dx4, dy4)),
    }
    if only:
        variants = {n: v for n, v in variants.items() if n in only}
    ref3 = None
    for name, (bs, fn) in variants.items():           # correctness first: all variants compute the same product
        dev.chk(fn())
        y = dev.get(dy3 if bs == 3 else dy4, mbs * bs).reshape(mbs, bs)[:, :3].ravel()
        if ref3 is None:
            ref3 = y
        err = np.max(np.abs(y - ref3)) / np.max(np.abs(ref3))
        print("  %-9s max rel diff vs first variant %.2e" % (name, err), flush=True)
        assert err < 1e-12
    e0, e1 = C.c_void_p(), C.c_void_p()
    k.mi355x_event_create(C.byref(e0)); k.mi355x_event_create(C.byref(e1))
    times = {n: [] for n in variants}
    for r in range(rounds):
        for name, (bs, fn) in variants.items():
            for _ in range(2):
                fn()
            k.mi355x_event_record(e0, dev.h)
            for _ in range(10):
                fn()
            k.mi355x_event_record(e1, dev.h)
            k.mi355x_event_synchronize(e1)
            ms = C.c_float(); k.mi355x_event_elapsed_ms(e0, e1, C.byref(ms))
            times[name].append(ms.value / 10)
    for name, (bs, fn) in variants.items():
        B = (8 * bs * bs + 4) * nnzb + 4 * (mbs + 1) + 16 * bs * mbs
        t = np.array(times[name])
        print("%-9s bs=%d  median %.4f ms (min %.4f)  %.0f GB/s of its own %.3f GB = %.3f of 8 TB/s ; as bs=3 bytes: %.3f" %
              (name, bs, np.median(t), t.min(), B / np.median(t) / 1e6, B / 1e9, B / np.median(t) / 1e6 / 8000,
               ((72 + 4) * nnzb + 4 * (mbs + 1) + 48 * mbs) / np.median(t) / 1e6 / 8000), flush=True)
    # what the same process streams from the same 4 GB value array with nothing else to do: a read-only pass (sum of squares) and a copy of its first half onto its second (read + write, same total bytes)
    nval = nnzb * 9
    out = dev.host_scratch()                        # the handle's pinned result slot (what the reductions write)
    half = nval // 2
    for label, fn, nbytes in (("read-only stream (VecNorm kernel over the bs=3 value array)", lambda: k.mi355x_vec_norm(dev.h, nval, 1, da3, out), 8 * nval),
                              ("copy stream (first half of the value array onto the bs=4 array)", lambda: k.mi355x_vec_copy(dev.h, half, da3, da4), 16 * half)):
        ts = []
        for r in range(rounds):
            fn(); fn()
            k.mi355x_event_record(e0, dev.h)
            for _ in range(10):
                fn()
            k.mi355x_event_record(e1, dev.h)
            k.mi355x_event_synchronize(e1)
            ms = C.c_float(); k.mi355x_event_elapsed_ms(e0, e1, C.byref(ms))
            ts.append(ms.value / 10)
        t = np.median(np.array(ts))
        print("%-70s %.4f ms  %.0f GB/s = %.3f of 8 TB/s" % (label, t, nbytes / t / 1e6, nbytes / t / 1e6 / 8000), flush=True)


if __name__ == "__main__":
    main()
