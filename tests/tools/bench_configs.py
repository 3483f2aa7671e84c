"""Secondary measurements (BASELINE.json configs[3], configs[4]); development aid, not the contract bench.

  irr   GMRES(30) + block Jacobi (sub: preonly + Jacobi) on the synthetic irregular stand-in for Flan_1565
        (n = 1 564 794, mean ~73 nnz/row, log-normal row lengths clipped to [3,400],
        banded +-50 000 with 20 % long-range columns, diagonally dominant, seed 12345) -- SpMV GB/s, its/s
  baij  3-D elasticity-like BAIJ bs=3 on 128^3 nodes (27-block stencil) -- MatMult_SeqBAIJ_3 GB/s
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


from problems import gen_irr  # noqa: E402


def gen_baij27(nn=128, bs=3, seed=5):
    rng = np.random.default_rng(seed)
    N = nn ** 3
    idx = np.arange(N, dtype=np.int64)
    i = idx % nn; j = (idx // nn) % nn; k = idx // (nn * nn)
    cols, valid = [], []
    for dk in (-1, 0, 1):
        for dj in (-1, 0, 1):
            for di in (-1, 0, 1):
                ok = (i + di >= 0) & (i + di < nn) & (j + dj >= 0) & (j + dj < nn) & (k + dk >= 0) & (k + dk < nn)
                cols.append(idx + di + nn * dj + nn * nn * dk)
                valid.append(ok)
    cols = np.stack(cols, 1); valid = np.stack(valid, 1)
    bi = np.concatenate(([0], np.cumsum(valid.sum(1)))).astype(np.int32)
    bj = cols[valid].astype(np.int32)
    ba = rng.standard_normal(bj.size * bs * bs)
    return bi, bj, ba


def event_time(k, h, fn, reps, warm=3):
    e0, e1 = C.c_void_p(), C.c_void_p()
    k.mi355x_event_create(C.byref(e0)); k.mi355x_event_create(C.byref(e1))
    for _ in range(warm):
        fn()
    k.mi355x_event_record(e0, h)
    for _ in range(reps):
        fn()
    k.mi355x_event_record(e1, h)
    k.mi355x_event_synchronize(e1)
    ms = C.c_float()
    k.mi355x_event_elapsed_ms(e0, e1, C.byref(ms))
    return ms.value / reps * 1e-3


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "irr"
    import petsc_dev_amd as pda
    from petsc_dev_amd import petsc as P
    L = P.lib()
    k = pda.load_kernels()
    if which == "irr":
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 1564794
        t0 = time.time()
        ai, aj, aa = gen_irr(n)
        print("IRR: n=%d nnz=%d (%.1f/row, max %d) generated in %.1fs" % (n, aj.size, aj.size / n, np.diff(ai).max(), time.time() - t0), flush=True)
        A = P.Mat.from_csr(ai, aj, aa)
        x = P.Vec.create(n, comm=L.COMM_SELF); L.VecSet(x.h, 1.0)
        b = x.duplicate(); u = x.duplicate()
        A.mult(x, b)
        nnz, m = aj.size, n
        B = 12 * nnz + 4 * (m + 1) + 16 * m
        L.MatHIPMI355XSetTiming(A.h, 1)
        for _ in range(50):
            A.mult(x, b)
        nl, tms = C.c_int(), C.c_double()
        L.MatHIPMI355XGetTiming(A.h, C.byref(nl), C.byref(tms))
        L.MatHIPMI355XSetTiming(A.h, 0)
        t = tms.value / nl.value * 1e-3
        print("IRR spmv: %.3f ms  %.1f GB/s algorithmic (%.3f of 8 TB/s)" % (t * 1e3, B / t / 1e9, B / t / 8e12), flush=True)
        ksp = P.KSP(comm=L.COMM_SELF)
        ksp.set_operators(A); ksp.set_type("gmreshipmi355x"); ksp.set_pc_type("bjacobi")
        L.PetscOptionsClear(); L.PetscOptionsInsertString(b"-sub_pc_type jacobi")   # all-device sub-solve
        ksp.set_from_options()
        ksp.set_tolerances(rtol=0.0, abstol=1e-300, dtol=1e300, max_it=10)
        ksp.solve(b, u)                 # PCSetUp (reads -sub_* for the block solver) happens here
        L.PetscOptionsClear()
        steps = 120
        ksp.set_tolerances(rtol=0.0, abstol=1e-300, dtol=1e300, max_it=steps)
        k.mi355x_device_synchronize()
        t0 = time.perf_counter()
        ksp.solve(b, u)
        k.mi355x_device_synchronize()
        dt = time.perf_counter() - t0
        # algorithmic bytes of GMRES(30) iteration k: SpMV + 24n + 8(k+2)n + 8(k+3)n + 24n, averaged over a cycle
        kk = np.arange(30)
        vecb = np.mean(24 * n + 8 * (kk + 2) * n + 8 * (kk + 3) * n + 24 * n)
        print("IRR GMRES(30)+bjacobi(jacobi): %d its in %.3f s = %.1f it/s ; %.1f GB/s algorithmic" % (ksp.its, dt, ksp.its / dt, (B + vecb) * ksp.its / dt / 1e9), flush=True)
        ksp.set_tolerances(rtol=1e-8, abstol=1e-50, dtol=1e5, max_it=2000)
        ksp.record_history()
        ksp.solve(b, u)
        print("IRR converged: its=%d reason=%d final=%g error=%g" % (ksp.its, ksp.reason, ksp.rnorm, np.linalg.norm(u.array() - 1.0)), flush=True)
    elif which == "aij27":
        # the same elasticity shape stored as plain AIJ (81 nonzeros per row, 135 distinct offsets -> idx8 applies)
        nn = int(sys.argv[2]) if len(sys.argv) > 2 else 96
        bs = 3
        bi, bj, ba = gen_baij27(nn)
        mbs = bi.size - 1
        cnt = np.diff(bi)
        ai = np.concatenate(([0], np.cumsum(np.repeat(cnt * bs, bs)))).astype(np.int64)
        assert ai[-1] < 2 ** 31
        # row (br, r): for each block j of block row br, columns bj*3 + 0..2; values ba[blk*9 + c*3 + r] (column-major blocks)
        blk_row = np.repeat(np.arange(mbs), cnt)
        aj = np.empty(bj.size * 9, dtype=np.int32); aa = np.empty(bj.size * 9)
        start = np.repeat(bi[:-1].astype(np.int64), cnt)           # first block of the block row, per block
        pos_in_row = np.arange(bj.size) - start
        for r in range(bs):
            base = ai[blk_row * bs + r] + pos_in_row * bs
            for c in range(bs):
                aj[base + c] = bj * bs + c
                aa[base + c] = ba[np.arange(bj.size) * 9 + c * 3 + r]
        ai = ai.astype(np.int32)
        m = mbs * bs
        print("AIJ27: m=%d nnz=%d (%.1f/row)" % (m, aj.size, aj.size / m), flush=True)
        A = P.Mat.from_csr(ai, aj, aa)
        x = P.Vec.from_array(np.sin(0.1 * np.arange(m)), comm=L.COMM_SELF)
        y = x.duplicate()
        A.mult(x, y)
        B = 12 * aj.size + 4 * (m + 1) + 16 * m
        L.MatHIPMI355XSetTiming(A.h, 1)
        for _ in range(20):
            A.mult(x, y)
        nl, tms = C.c_int(), C.c_double()
        L.MatHIPMI355XGetTiming(A.h, C.byref(nl), C.byref(tms))
        t = tms.value / nl.value * 1e-3
        print("AIJ (elasticity shape) spmv: %.3f ms  %.1f GB/s CSR-algorithmic (%.3f of 8 TB/s)" % (t * 1e3, B / t / 1e9, B / t / 8e12), flush=True)
        Ab = P.Mat.from_bsr(bs, bi, bj, ba)
        yb = x.duplicate(); Ab.mult(x, yb)
        ya, ybb = y.array(), yb.array()
        print("  max |AIJ - BAIJ| = %g (scale %g)" % (np.max(np.abs(ya - ybb)), np.max(np.abs(ya))), flush=True)
    else:
        nn = int(sys.argv[2]) if len(sys.argv) > 2 else 128
        bs = int(sys.argv[3]) if len(sys.argv) > 3 else 3      # 4: the 3-dof blocks zero-padded to 4x4 (BASELINE configs[4])
        t0 = time.time()
        bi, bj, ba = gen_baij27(nn, bs=bs)
        mbs = bi.size - 1
        print("BAIJ27: mbs=%d nnzb=%d generated in %.1fs" % (mbs, bj.size, time.time() - t0), flush=True)
        A = P.Mat.from_bsr(bs, bi, bj, ba)
        x = P.Vec.from_array(np.sin(0.1 * np.arange(mbs * bs)), comm=L.COMM_SELF)
        y = x.duplicate()
        A.mult(x, y)
        B = (8 * bs * bs + 4) * bj.size + 4 * (mbs + 1) + 16 * bs * mbs
        L.MatHIPMI355XSetTiming(A.h, 1)
        for _ in range(20):
            A.mult(x, y)
        nl, tms = C.c_int(), C.c_double()
        L.MatHIPMI355XGetTiming(A.h, C.byref(nl), C.byref(tms))
        t = tms.value / nl.value * 1e-3
        print("BAIJ bs=%d spmv: %.3f ms  %.1f GB/s algorithmic (%.3f of 8 TB/s)" % (bs, t * 1e3, B / t / 1e9, B / t / 8e12), flush=True)
        if nn <= 32:
            import orc
            ref = orc.spmv_bsr(bs, bi, bj, ba, np.sin(0.1 * np.arange(mbs * bs)))
            print("  max |err| vs oracle: %g" % np.max(np.abs(y.array() - ref)))


if __name__ == "__main__":
    main()
