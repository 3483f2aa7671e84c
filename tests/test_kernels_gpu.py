"""Parity of every HIP kernel against the oracle, through the C ABI (include/mi355x_kernels.h).

Element-wise kernels, MAXPY and the row-sequential SpMV paths must be BIT-EXACT; reductions use a
fixed tree and are checked to |err| <= 1e-13 * sum|terms| (stated tolerance, fp64)."""
import ctypes as C

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu

SIZES = [0, 1, 2, 3, 63, 64, 65, 255, 1000, 4097, 65536 + 3, 1_000_001]


@pytest.fixture(scope="module")
def dev(built):
    from gpu import Dev
    d = Dev()
    yield d
    d.free_all()


def rnd(n, seed):
    return np.random.default_rng(seed).standard_normal(n)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


def assert_bitexact(a, b):
    assert a.shape == b.shape
    assert np.array_equal(bits(a), bits(b)), "max abs diff %g" % (np.max(np.abs(a - b)) if a.size else 0)


@pytest.mark.parametrize("n", SIZES)
def test_elementwise_bitexact(dev, n):
    k = dev.k
    x, y, z = rnd(n, 1), rnd(n, 2), rnd(n, 3)
    dx, dy, dz = dev.put(x), dev.put(y), dev.put(z)
    dw = dev.alloc(8 * max(n, 2))

    def reset():
        for p, a in ((dx, x), (dy, y), (dz, z)):
            if n:
                dev.chk(k.mi355x_memcpy_h2d(dev.h, p, a.ctypes.data, a.nbytes))

    for alpha in (0.0, 1.0, -1.0, 0.37):
        reset(); dev.chk(k.mi355x_vec_axpy(dev.h, n, alpha, dx, dy))
        r = y.copy(); orc.vec_axpy(r, alpha, x); assert_bitexact(dev.get(dy, n), r)
        reset(); dev.chk(k.mi355x_vec_aypx(dev.h, n, alpha, dx, dy))
        r = y.copy(); orc.vec_aypx(r, alpha, x); assert_bitexact(dev.get(dy, n), r)
        reset(); dev.chk(k.mi355x_vec_waxpy(dev.h, n, alpha, dx, dy, dw))
        r = np.zeros(n); orc.vec_waxpy(r, alpha, x, y); assert_bitexact(dev.get(dw, n), r)
        reset(); dev.chk(k.mi355x_vec_scale(dev.h, n, alpha, dx))
        r = x.copy(); orc.vec_scale(r, alpha); assert_bitexact(dev.get(dx, n), r)
        for beta in (0.0, 1.0, -2.5):
            reset(); dev.chk(k.mi355x_vec_axpby(dev.h, n, alpha, beta, dx, dy))
            r = y.copy(); orc.vec_axpby(r, alpha, beta, x); assert_bitexact(dev.get(dy, n), r)
            for gamma in (0.0, 1.0, 0.5):
                reset(); dev.chk(k.mi355x_vec_axpbypcz(dev.h, n, alpha, beta, gamma, dx, dy, dz))
                r = z.copy(); orc.vec_axpbypcz(r, alpha, beta, gamma, x, y); assert_bitexact(dev.get(dz, n), r)
    reset(); dev.chk(k.mi355x_vec_pointwise_mult(dev.h, n, dx, dy, dw))
    r = np.zeros(n); orc.vec_pointwise_mult(r, x, y); assert_bitexact(dev.get(dw, n), r)
    # aliasing cases of VecPointwiseMult_Seq (w==x, w==y)
    reset(); dev.chk(k.mi355x_vec_pointwise_mult(dev.h, n, dx, dy, dx)); assert_bitexact(dev.get(dx, n), r)
    reset(); dev.chk(k.mi355x_vec_pointwise_mult(dev.h, n, dx, dy, dy)); assert_bitexact(dev.get(dy, n), r)
    reset(); dev.chk(k.mi355x_vec_pointwise_divide(dev.h, n, dx, dy, dw))
    r = np.zeros(n); orc.vec_pointwise_divide(r, x, y); assert_bitexact(dev.get(dw, n), r)
    reset(); dev.chk(k.mi355x_vec_set(dev.h, n, 3.25, dw)); assert_bitexact(dev.get(dw, n), np.full(n, 3.25))
    reset(); dev.chk(k.mi355x_vec_copy(dev.h, n, dx, dw)); assert_bitexact(dev.get(dw, n), x)
    reset(); dev.chk(k.mi355x_vec_swap(dev.h, n, dx, dy))
    assert_bitexact(dev.get(dx, n), y); assert_bitexact(dev.get(dy, n), x)
    xz = x.copy(); xz[::7] = 0.0
    if n:
        dev.chk(k.mi355x_memcpy_h2d(dev.h, dx, xz.ctypes.data, xz.nbytes))
    dev.chk(k.mi355x_vec_reciprocal(dev.h, n, dx))
    r = xz.copy(); orc.vec_reciprocal(r); assert_bitexact(dev.get(dx, n), r)
    for p in (dx, dy, dz, dw):
        dev.free(p)


def test_unaligned_views(dev):
    """Vectors that start 8 bytes off a 16-byte boundary take the scalar path; same bits."""
    k = dev.k
    n = 10001
    x, y = rnd(n + 1, 4), rnd(n + 1, 5)
    dx, dy = dev.put(x), dev.put(y)
    ox, oy = C.c_void_p(dx.value + 8), C.c_void_p(dy.value + 8)
    dev.chk(k.mi355x_vec_axpy(dev.h, n, 0.7, ox, oy))
    r = y[1:].copy(); orc.vec_axpy(r, 0.7, x[1:].copy())
    assert_bitexact(dev.get(oy, n), r)
    dev.chk(k.mi355x_vec_dot(dev.h, n, ox, oy, dev.host_scratch()))
    got = dev.scalar_out()[0]
    ref = orc.vec_dot(x[1:].copy(), r)
    assert abs(got - ref) <= 1e-13 * np.sum(np.abs(x[1:] * r))
    dev.free(dx); dev.free(dy)


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("nv", [1, 2, 3, 4, 5, 7, 8, 9, 12, 30, 31])
def test_maxpy_bitexact(dev, n, nv):
    if n > 70000 and nv not in (3, 30):
        pytest.skip("large size covered for nv=3,30")
    k = dev.k
    x = rnd(n, 10)
    ys = [rnd(n, 100 + j) for j in range(nv)]
    alpha = rnd(nv, 11)
    dx = dev.put(x)
    dys = [dev.put(v) for v in ys]
    tab = dev.ptr_table(dys)
    dev.chk(k.mi355x_vec_maxpy(dev.h, n, nv, alpha.ctypes.data_as(C.POINTER(C.c_double)), tab, dx))
    r = x.copy(); orc.vec_maxpy(r, alpha, ys)
    assert_bitexact(dev.get(dx, n), r)
    dev.free(dx)
    for p in dys:
        dev.free(p)


@pytest.mark.parametrize("n", [1, 2, 255, 4097, 70001, 1 << 20])
@pytest.mark.parametrize("nv", [1, 2, 3, 4, 5, 8, 15, 16, 17, 29, 30, 31, 32])
def test_gmres_fused_sweeps_carry_the_bits_of_the_separate_calls(dev, n, nv):
    """mi355x_vec_maxpy_dev_norm2 == VecMAXPY (coefficients negated as borthog2.c:63) + VecNorm's sum of squares;
    mi355x_vec_scale_rnorm_dev == VecNormalize's VecScale(1/norm) (rvector.c:308-314)"""
    if n > 70001 and nv not in (3, 17, 30):
        pytest.skip("large size covered for nv = 3, 17, 30")
    k = dev.k
    x = rnd(n, 10)
    ys = [rnd(n, 100 + j) for j in range(nv)]
    h = rnd(nv, 11)
    dx, dx2 = dev.put(x), dev.put(x)
    dys = [dev.put(v) for v in ys]
    tab = dev.ptr_table(dys)
    dh = dev.put(h)
    out = dev.alloc(16)
    # separate calls: host coefficients -h, then the norm
    mh = -h
    dev.chk(k.mi355x_vec_maxpy(dev.h, n, nv, mh.ctypes.data_as(C.POINTER(C.c_double)), tab, dx))
    dev.chk(k.mi355x_vec_norm(dev.h, n, 2, dx, dev.host_scratch()))
    ref_n2 = dev.scalar_out()[0]
    ref_x = dev.get(dx, n)
    # fused
    dev.chk(k.mi355x_vec_maxpy_dev_norm2(dev.h, n, nv, dh, -1.0, tab, dx2, out))
    dev.sync()
    got_n2 = dev.get(out, 1)
    assert_bitexact(dev.get(dx2, n), ref_x)
    assert_bitexact(got_n2, np.array([ref_n2]))
    dev.chk(k.mi355x_vec_scale_rnorm_dev(dev.h, n, out, dx2))
    dev.sync()
    assert_bitexact(dev.get(dx2, n), ref_x * (1.0 / np.sqrt(ref_n2)))
    for p in [dx, dx2, dh, out] + dys:
        dev.free(p)


def test_scale_rnorm_dev_special_cases(dev):
    """a zero norm and a norm of exactly one leave the vector alone (rvector.c:309-311)"""
    k = dev.k
    x = rnd(1001, 5)
    dx = dev.put(x)
    for n2 in (0.0, 1.0):
        dn = dev.put(np.array([n2]))
        dev.chk(k.mi355x_vec_scale_rnorm_dev(dev.h, x.size, dn, dx)); dev.sync()
        assert_bitexact(dev.get(dx, x.size), x)
        dev.free(dn)
    dn = dev.put(np.array([4.0]))
    dev.chk(k.mi355x_vec_scale_rnorm_dev(dev.h, x.size, dn, dx)); dev.sync()
    assert_bitexact(dev.get(dx, x.size), x * 0.5)
    dev.free(dn); dev.free(dx)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 257, 4095, 4097, 70001, 1 << 20, 5000001, (1 << 25) + 3, (1 << 25) + (1 << 23) + 2048 + 1])
def test_reductions_equal_the_oracle_in_the_device_summation_order(dev, n):
    """orc.device_reduction_order() restates reduce_kernel's tree (per-lane strided pair sums -> shuffle-down tree -> four
    wavefronts in order -> one workgroup over the per-workgroup sums): dot, norms, dot+norm2 and MDot are then bit-identical.
    The two largest sizes are vectors of >= 256 MiB: the geometry of contiguous runs of tiles (one / several tiles per workgroup)
    and streaming loads."""
    k = dev.k
    x, y = rnd(n, 20), rnd(n, 21)
    ys = [rnd(n, 200 + j) for j in range(5)]
    dx, dy = dev.put(x), dev.put(y)
    dys = [dev.put(v) for v in ys]
    with orc.device_reduction_order():
        ref_dot = orc.vec_dot(x, y)
        ref_n2 = orc.vec_norm(x, 1); ref_n1 = orc.vec_norm(x, 0)      # oracle types: 0 = NORM_1, 1 = NORM_2
        ref_dn = orc.vec_dotnorm2(x, y)
        ref_md = orc.vec_mdot(x, ys)
    dev.chk(k.mi355x_vec_dot(dev.h, n, dx, dy, dev.host_scratch()))
    assert_bitexact(dev.scalar_out(1), np.array([ref_dot]))
    dev.chk(k.mi355x_vec_norm(dev.h, n, 2, dx, dev.host_scratch()))
    assert_bitexact(np.sqrt(dev.scalar_out(1)), np.array([ref_n2]))
    dev.chk(k.mi355x_vec_norm(dev.h, n, 0, dx, dev.host_scratch()))
    assert_bitexact(dev.scalar_out(1), np.array([ref_n1]))
    dev.chk(k.mi355x_vec_dotnorm2(dev.h, n, dx, dy, dev.host_scratch()))
    assert_bitexact(dev.scalar_out(2), np.array(ref_dn))
    dev.chk(k.mi355x_vec_mdot(dev.h, n, 5, dx, dev.ptr_table(dys), dev.host_scratch()))
    assert_bitexact(dev.scalar_out(5), np.array(ref_md))
    for p in [dx, dy] + dys:
        dev.free(p)


def test_elementwise_and_fused_update_streaming_forms_bitexact(dev):
    """Vectors of >= 256 MiB take the non-temporal forms of the element-wise kernels and of the fused CG update (one tile per
    workgroup; contiguous runs of tiles): the same bits as the reference's loops / as the separate calls."""
    k = dev.k
    n = (1 << 25) + 7
    x, y, z = rnd(n, 31), rnd(n, 32), rnd(n, 33)
    dx, dy, dz = dev.put(x), dev.put(y), dev.put(z)
    dw = dev.alloc(8 * n)
    dev.chk(k.mi355x_vec_axpy(dev.h, n, 0.37, dx, dy))
    r = y.copy(); orc.vec_axpy(r, 0.37, x); assert_bitexact(dev.get(dy, n), r); y = r
    dev.chk(k.mi355x_vec_aypx(dev.h, n, -1.25, dx, dy))
    r = y.copy(); orc.vec_aypx(r, -1.25, x); assert_bitexact(dev.get(dy, n), r); y = r
    dev.chk(k.mi355x_vec_waxpy(dev.h, n, 2.0, dx, dy, dw))
    r = np.zeros(n); orc.vec_waxpy(r, 2.0, x, y); assert_bitexact(dev.get(dw, n), r)
    dev.chk(k.mi355x_vec_pointwise_mult(dev.h, n, dx, dy, dw))
    assert_bitexact(dev.get(dw, n), x * y)
    dev.chk(k.mi355x_vec_copy(dev.h, n, dx, dw))
    assert_bitexact(dev.get(dw, n), x)
    # fused CG update against the separate calls (x += a p; r -= a w; z = r .* d; sums in the device order)
    p_, w_, d_ = rnd(n, 34), rnd(n, 35), 1.0 + 0.1 * rnd(n, 36)
    dp, dwv, dd = dev.put(p_), dev.put(w_), dev.put(d_)
    xs, rs = x.copy(), z.copy()
    dev.chk(k.mi355x_memcpy_h2d(dev.h, dz, rs.ctypes.data, rs.nbytes))
    dzz = dev.alloc(8 * n)
    dev.chk(k.mi355x_vec_cg_update(dev.h, n, 0.61, dp, dwv, dd, dx, dz, dzz, dev.host_scratch()))
    sums = dev.scalar_out(3)
    orc.vec_axpy(xs, 0.61, p_); orc.vec_axpy(rs, -0.61, w_)
    zs = rs * d_
    assert_bitexact(dev.get(dx, n), xs); assert_bitexact(dev.get(dz, n), rs); assert_bitexact(dev.get(dzz, n), zs)
    with orc.device_reduction_order():
        ref = np.array([orc.vec_dot(zs, zs), orc.vec_dot(zs, rs), orc.vec_dot(rs, rs)])
    assert_bitexact(sums, ref)
    dev.free_all()


@pytest.mark.parametrize("n", SIZES)
def test_reductions(dev, n):
    k = dev.k
    x, y = rnd(n, 20), rnd(n, 21)
    dx, dy = dev.put(x), dev.put(y)
    out = dev.host_scratch()
    tol = 1e-13
    dev.chk(k.mi355x_vec_dot(dev.h, n, dx, dy, out))
    got = dev.scalar_out()[0]
    assert abs(got - orc.vec_dot(x, y)) <= tol * np.sum(np.abs(x * y)) + 0.0
    # run-to-run reproducible (fixed tree)
    dev.chk(k.mi355x_vec_dot(dev.h, n, dx, dy, out))
    assert dev.scalar_out()[0] == got
    dev.chk(k.mi355x_vec_norm(dev.h, n, 1, dx, out))
    assert abs(np.sqrt(dev.scalar_out()[0]) - orc.vec_norm(x, 1)) <= tol * max(np.linalg.norm(x), 1e-300) * 4
    dev.chk(k.mi355x_vec_norm(dev.h, n, 0, dx, out))
    assert abs(dev.scalar_out()[0] - orc.vec_norm(x, 0)) <= tol * np.sum(np.abs(x))
    dev.chk(k.mi355x_vec_norm(dev.h, n, 3, dx, out))
    assert dev.scalar_out()[0] == orc.vec_norm(x, 3)  # max is exact
    dev.chk(k.mi355x_vec_norm(dev.h, n, 4, dx, out))
    g = dev.scalar_out(2)
    r = orc.vec_norm(x, 4)
    assert abs(g[0] - r[0]) <= tol * np.sum(np.abs(x)) and abs(np.sqrt(g[1]) - r[1]) <= 4 * tol * max(r[1], 1e-300)
    dev.chk(k.mi355x_vec_dotnorm2(dev.h, n, dx, dy, out))
    g = dev.scalar_out(2)
    dp, nm = orc.vec_dotnorm2(x, y)
    assert abs(g[0] - dp) <= tol * np.sum(np.abs(x * y)) and abs(g[1] - nm) <= tol * np.sum(y * y)
    dev.free(dx); dev.free(dy)


def test_norm_inf_nan(dev):
    k = dev.k
    x = rnd(5000, 30)
    x[1234] = np.nan
    dx = dev.put(x)
    dev.chk(k.mi355x_vec_norm(dev.h, x.size, 3, dx, dev.host_scratch()))
    assert np.isnan(dev.scalar_out()[0]) and np.isnan(orc.vec_norm(x, 3))
    dev.free(dx)


@pytest.mark.parametrize("n", [0, 1, 5, 1000, 65537, 300_001])
@pytest.mark.parametrize("nv", [1, 2, 3, 5, 8, 9, 17, 31])
def test_mdot(dev, n, nv):
    k = dev.k
    x = rnd(n, 40)
    ys = [rnd(n, 200 + j) for j in range(nv)]
    dx = dev.put(x)
    dys = [dev.put(v) for v in ys]
    dev.chk(k.mi355x_vec_mdot(dev.h, n, nv, dx, dev.ptr_table(dys), dev.host_scratch()))
    got = dev.scalar_out(nv)
    ref = orc.vec_mdot(x, ys)
    for j in range(nv):
        assert abs(got[j] - ref[j]) <= 1e-13 * np.sum(np.abs(x * ys[j]))
    dev.free(dx)
    for p in dys:
        dev.free(p)


# ---------------------------------------------------------------- SpMV
def upload_csr(dev, ai, aj, aa):
    return dev.put(ai), dev.put(aj if aj.size else np.zeros(1, np.int32)), dev.put(aa if aa.size else np.zeros(1))


def make_plan(dev, ai, rows=None):
    plan = C.c_void_p()
    rp = rows.ctypes.data if rows is not None else None
    dev.chk(dev.k.mi355x_spmv_plan_create(dev.h, ai.size - 1, ai.ctypes.data, rp, C.byref(plan)))
    return plan


def random_csr(m, n, rowlen, seed, sort=True):
    rng = np.random.default_rng(seed)
    lens = rowlen(rng, m).astype(np.int64)
    lens = np.minimum(lens, n)
    ai = np.zeros(m + 1, dtype=np.int32)
    ai[1:] = np.cumsum(lens)
    aj = np.empty(int(ai[-1]), dtype=np.int32)
    for r in range(m):
        c = rng.choice(n, size=int(lens[r]), replace=False)
        aj[ai[r]:ai[r + 1]] = np.sort(c) if sort else c
    aa = rng.standard_normal(aj.size)
    return ai, aj, aa


def run_spmv(dev, ai, aj, aa, x, y0=None, rows=None, compress=False, group=False, pairsum=None, dscale=None, patterns=None, values=False):
    k = dev.k
    dai, daj, daa = upload_csr(dev, ai, aj, aa)
    dx = dev.put(x)
    plan = make_plan(dev, ai, rows)
    if group:       # node sizes as Mat_CheckInode finds them (the oracle's restatement), then the device-side grouping
        nodes, ns = orc.check_inode(ai, aj)
        ns = np.ascontiguousarray(ns, dtype=np.int32)
        dev.chk(k.mi355x_spmv_plan_group_rows(dev.h, plan, ai.ctypes.data, aj.ctypes.data, nodes, ns.ctypes.data))
        ng, ngj = C.c_int(), C.c_long()
        k.mi355x_spmv_plan_group_info(plan, C.byref(ng), C.byref(ngj), None)
        run_spmv.last_groups = (nodes, ng.value, ngj.value)
    if pairsum is not None:
        dev.chk(k.mi355x_spmv_plan_set_pairsum(plan, int(pairsum)))
    if compress:
        dev.chk(k.mi355x_spmv_plan_compress_indices(dev.h, plan, ai.ctypes.data, aj.ctypes.data))
        nt = C.c_int()
        k.mi355x_spmv_plan_is_compressed(plan, C.byref(nt))
        run_spmv.last_ntab = nt.value
        npat = C.c_int()                       # row-pattern kernel: patterns=False forces the per-nonzero (idx8) kernel
        dev.chk(k.mi355x_spmv_plan_use_patterns(plan, -1 if patterns is None else int(patterns), C.byref(npat)))
        run_spmv.last_npat = npat.value
    if values:      # value patterns: whole rows (offsets + values) from a dictionary, the device value array is not read
        nv = C.c_int()
        aa_h = np.ascontiguousarray(aa, dtype=np.float64)
        dev.chk(k.mi355x_spmv_plan_value_patterns(dev.h, plan, ai.ctypes.data, aj.ctypes.data, aa_h.ctypes.data, C.byref(nv)))
        run_spmv.last_nvpat = nv.value
        if nv.value:                       # prove it: poison the device values the other kernels would stream
            dev.free(daa)
            daa = dev.put(np.full(max(aa.size, 1) + 2, np.nan))
    if dscale is not None:                                       # y = d .* (A x) in the product's epilogue
        m_out = ai.size - 1
        dy = dev.put(np.full(m_out, 7.0))
        dd = dev.put(dscale)
        dev.chk(k.mi355x_spmv_csr_scaled(dev.h, plan, dai, daj, daa, dx, dd, dy))
        dev.sync(); dev.free(dd)
    elif y0 is None:
        m_out = ai.size - 1
        dy = dev.put(np.full(m_out, 7.0))
        dev.chk(k.mi355x_spmv_csr(dev.h, plan, dai, daj, daa, dx, dy))
    else:
        m_out = y0.size
        dy = dev.put(y0)
        dev.chk(k.mi355x_spmv_csr_add(dev.h, plan, dai, daj, daa, dx, dy, dy))
    y = dev.get(dy, m_out)
    dev.chk(k.mi355x_spmv_plan_destroy(plan))
    for p in (dai, daj, daa, dx, dy):
        dev.free(p)
    return y


@pytest.mark.parametrize("dims", [(5, 4, 3), (16, 16, 16), (33, 17, 9), (64, 64, 40)])
def test_spmv_p7_bitexact(dev, dims):
    ai, aj, aa = orc.gen_p7(*dims)
    n = ai.size - 1
    x = np.sin(0.37 * np.arange(n)) + 1.0
    assert_bitexact(run_spmv(dev, ai, aj, aa, x), orc.spmv(ai, aj, aa, x))
    y0 = rnd(n, 50)
    assert_bitexact(run_spmv(dev, ai, aj, aa, x, y0=y0), orc.spmv_add(ai, aj, aa, x, y0))


@pytest.mark.parametrize("threads", [1, 5])
@pytest.mark.parametrize("dims", [(5, 4, 3), (33, 17, 9), (64, 64, 40)])
def test_spmv_index_compression_bitexact(dev, dims, threads, monkeypatch):
    """offset-dictionary index compression (col = row + table[idx8]): same bits as the plain kernel and the oracle; the analysis
    in chunks of rows on several host threads (tables merged in chunk order, as for large matrices) finds the same dictionaries"""
    monkeypatch.setenv("MI355X_ANALYSIS_THREADS", str(threads))
    ai, aj, aa = orc.gen_p7(*dims)
    aa = aa * (1.0 + 0.01 * np.cos(np.arange(aa.size)))
    n = ai.size - 1
    x = np.sin(0.37 * np.arange(n)) + 1.0
    y0 = rnd(n, 51)
    for patterns in (False, True):             # the per-nonzero byte kernel, then the row-pattern kernel (2 bytes per row)
        got = run_spmv(dev, ai, aj, aa, x, compress=True, patterns=patterns)
        assert run_spmv.last_ntab == 7 and 1 <= run_spmv.last_npat <= 27
        assert_bitexact(got, orc.spmv(ai, aj, aa, x))
        assert_bitexact(run_spmv(dev, ai, aj, aa, x, y0=y0, compress=True, patterns=patterns), orc.spmv_add(ai, aj, aa, x, y0))
        assert_bitexact(run_spmv(dev, ai, aj, aa, x, compress=True, patterns=patterns, pairsum=1), orc.spmv_inode(ai, aj, aa, x))


@pytest.mark.parametrize("threads", [1, 3])
@pytest.mark.parametrize("dims", [(5, 4, 3), (33, 17, 9), (64, 64, 40), (1, 1, 1), (700, 3, 1)])
def test_spmv_value_patterns_bitexact(dev, dims, threads, monkeypatch):
    """constant-coefficient operator: whole rows (offsets and values) come from the dictionary, the value array is not read
    (it is NaN on the device here); y = Ax, y = y0 + Ax, y = d .* (Ax) and the inode summation order carry the oracle's bits;
    one host thread or the chunked analysis"""
    monkeypatch.setenv("MI355X_ANALYSIS_THREADS", str(threads))
    ai, aj, aa = orc.gen_p7(*dims)
    n = ai.size - 1
    x = np.sin(0.37 * np.arange(n)) + 1.0
    y0 = rnd(n, 70); d = rnd(n, 71)
    got = run_spmv(dev, ai, aj, aa, x, values=True)
    assert 1 <= run_spmv.last_nvpat <= 27
    assert_bitexact(got, orc.spmv(ai, aj, aa, x))
    assert_bitexact(run_spmv(dev, ai, aj, aa, x, y0=y0, values=True), orc.spmv_add(ai, aj, aa, x, y0))
    assert_bitexact(run_spmv(dev, ai, aj, aa, x, dscale=d, values=True), orc.spmv(ai, aj, aa, x) * d)
    assert_bitexact(run_spmv(dev, ai, aj, aa, x, values=True, pairsum=1), orc.spmv_inode(ai, aj, aa, x))
    # the same with the other analyses present: value patterns take precedence over index compression / row patterns
    assert_bitexact(run_spmv(dev, ai, aj, aa, x, compress=True, values=True), orc.spmv(ai, aj, aa, x))


def test_spmv_value_patterns_shapes_and_refusals(dev):
    """a 9-point stencil with two coefficient regions, stored zeros of either sign (compared as bits, so two entries) and
    empty rows; varying coefficients and compressed-row plans are refused and the value array is streamed as before;
    dropping the dictionary returns to the streamed values"""
    import scipy.sparse as sp
    nx, ny = 41, 23
    n = nx * ny
    rows, cols, vals = [], [], []
    for j in range(ny):
        for i in range(nx):
            r = i + nx * j
            if r % 13 == 7:
                continue                                   # a row without entries
            for dj in (-1, 0, 1):
                for di in (-1, 0, 1):
                    if 0 <= i + di < nx and 0 <= j + dj < ny:
                        rows.append(r); cols.append(i + di + nx * (j + dj))
                        c = 8.0 if (di == 0 and dj == 0) else (-1.0 if (di == 0 or dj == 0) else -0.5)
                        vals.append(c * (3.0 if j >= ny // 2 else 1.0))
    A = sp.csr_matrix((vals, (rows, cols)), shape=(n, n)); A.sort_indices()
    ai, aj, aa = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy()
    aa[ai[5]] = -0.0; aa[ai[6]] = 0.0                     # rows 5 and 6: same offsets, values differ in the sign of a zero
    x = rnd(n, 72)
    got = run_spmv(dev, ai, aj, aa, x, values=True)
    kinds = {(tuple(aj[ai[r]:ai[r + 1]] - r), aa[ai[r]:ai[r + 1]].tobytes()) for r in range(n)}
    assert run_spmv.last_nvpat == len(kinds) and len(kinds) >= 12
    assert_bitexact(got, orc.spmv(ai, aj, aa, x))
    # varying coefficients: refused after a few rows, the kernel streams the values
    aav = aa * (1.0 + 0.01 * np.cos(np.arange(aa.size)))
    got = run_spmv(dev, ai, aj, aav, x, values=True)
    assert run_spmv.last_nvpat == 0
    assert_bitexact(got, orc.spmv(ai, aj, aav, x))
    # switched off / dropped: back to the value stream
    k = dev.k
    ai7, aj7, aa7 = orc.gen_p7(9, 8, 7)
    n7 = ai7.size - 1
    x7 = rnd(n7, 73)
    dai, daj, daa = upload_csr(dev, ai7, aj7, aa7 * 2.0)     # device values are TWICE the ones the dictionary is derived from
    dx = dev.put(x7); dy = dev.put(np.zeros(n7))
    plan = make_plan(dev, ai7, None)
    nv = C.c_int()
    dev.chk(k.mi355x_spmv_plan_value_patterns(dev.h, plan, ai7.ctypes.data, aj7.ctypes.data, aa7.ctypes.data, C.byref(nv)))
    assert nv.value > 0
    dev.chk(k.mi355x_spmv_csr(dev.h, plan, dai, daj, daa, dx, dy))
    assert_bitexact(dev.get(dy, n7), orc.spmv(ai7, aj7, aa7, x7))               # dictionary
    dev.chk(k.mi355x_spmv_plan_drop_value_patterns(plan))
    dev.chk(k.mi355x_spmv_plan_use_value_patterns(plan, -1, C.byref(nv)))
    assert nv.value == 0
    dev.chk(k.mi355x_spmv_csr(dev.h, plan, dai, daj, daa, dx, dy))
    assert_bitexact(dev.get(dy, n7), orc.spmv(ai7, aj7, aa7 * 2.0, x7))         # streamed device values
    dev.chk(k.mi355x_spmv_plan_use_value_patterns(plan, 0, None))               # switched off: the analysis declines
    dev.chk(k.mi355x_spmv_plan_value_patterns(dev.h, plan, ai7.ctypes.data, aj7.ctypes.data, aa7.ctypes.data, C.byref(nv)))
    assert nv.value == 0
    dev.chk(k.mi355x_spmv_plan_destroy(plan))
    for p in (dai, daj, daa, dx, dy):
        dev.free(p)
    # compressed-row plan: not applicable
    rows_nz = np.flatnonzero(np.diff(ai) > 0).astype(np.int32)
    aic = np.concatenate(([0], np.cumsum(np.diff(ai)[rows_nz]))).astype(np.int32)
    planc = make_plan(dev, aic, rows_nz)
    dev.chk(k.mi355x_spmv_plan_value_patterns(dev.h, planc, ai.ctypes.data, aj.ctypes.data, aa.ctypes.data, C.byref(nv)))
    assert nv.value == 0
    dev.chk(k.mi355x_spmv_plan_destroy(planc))


def test_spmv_row_patterns_other_shapes(dev):
    """row-pattern analysis: a 2-D 9-point stencil with empty rows sprinkled in and rows of 0..9 entries (dictionary), a banded
    matrix whose boundary rows need more table than there is (declined: per-nonzero bytes), and the ex5 rectangular shape"""
    import scipy.sparse as sp
    nx, ny = 37, 29
    n = nx * ny
    rows, cols = [], []
    for j in range(ny):
        for i in range(nx):
            r = i + nx * j
            if r % 11 == 5:
                continue                                   # a row without entries
            for dj in (-1, 0, 1):
                for di in (-1, 0, 1):
                    if 0 <= i + di < nx and 0 <= j + dj < ny:
                        rows.append(r); cols.append(i + di + nx * (j + dj))
    A = sp.csr_matrix((rnd(len(rows), 60), (rows, cols)), shape=(n, n)); A.sort_indices()
    ai, aj, aa = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy()
    x = rnd(n, 61); y0 = rnd(n, 62); d = rnd(n, 63)
    got = run_spmv(dev, ai, aj, aa, x, compress=True)
    assert run_spmv.last_ntab == 9 and run_spmv.last_npat >= 9
    assert_bitexact(got, orc.spmv(ai, aj, aa, x))
    assert_bitexact(run_spmv(dev, ai, aj, aa, x, y0=y0, compress=True), orc.spmv_add(ai, aj, aa, x, y0))
    assert_bitexact(run_spmv(dev, ai, aj, aa, x, dscale=d, compress=True), orc.spmv(ai, aj, aa, x) * d)
    # banded, 81 offsets: 81 boundary lists of 41..80 offsets do not fit the table -> no dictionary, idx8 kernel
    nb = 3000
    rows_ = [np.arange(max(0, r - 40), min(nb, r + 41)) for r in range(nb)]
    aib = np.concatenate(([0], np.cumsum([c.size for c in rows_]))).astype(np.int32)
    ajb = np.concatenate(rows_).astype(np.int32)
    run_spmv(dev, aib, ajb, rnd(ajb.size, 64), rnd(nb, 65), compress=True)
    assert run_spmv.last_ntab == 81 and run_spmv.last_npat == 0


def test_spmv_index_compression_other_shapes(dev):
    """banded matrix with long rows (several lanes per row, 81 offsets), a dense row longer than the LDS stage, and a
    random matrix with > 256 distinct offsets (analysis declines, plain CSR is used)"""
    n = 3000
    rows_ = []
    for r in range(n):
        c = np.arange(max(0, r - 40), min(n, r + 41))
        rows_.append(c)
    lens = np.array([c.size for c in rows_])
    ai = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    aj = np.concatenate(rows_).astype(np.int32)
    aa = rnd(aj.size, 52)
    x = rnd(n, 53)
    got = run_spmv(dev, ai, aj, aa, x, compress=True)
    assert run_spmv.last_ntab == 81
    scale = np.zeros(n); np.add.at(scale, np.repeat(np.arange(n), lens), np.abs(aa * x[aj]))
    assert np.all(np.abs(got - orc.spmv(ai, aj, aa, x)) <= 1e-12 * scale)
    # one dense row of 2500 entries in an otherwise diagonal matrix: 2500 offsets > 256 -> declined
    ai2 = np.concatenate(([0], np.cumsum(np.where(np.arange(n) == 7, 2500, 1)))).astype(np.int32)
    aj2 = np.concatenate([np.arange(2500) if r == 7 else np.array([r]) for r in range(n)]).astype(np.int32)
    aa2 = rnd(aj2.size, 54)
    got = run_spmv(dev, ai2, aj2, aa2, x, compress=True)
    assert run_spmv.last_ntab == 0
    sc2 = np.zeros(n); np.add.at(sc2, np.repeat(np.arange(n), np.diff(ai2)), np.abs(aa2 * x[aj2]))
    assert np.all(np.abs(got - orc.spmv(ai2, aj2, aa2, x)) <= 1e-12 * sc2)
    # Toeplitz-like long row: row 0 holds columns 0..2499 and every row r holds r..r+199 (200 offsets <= 256), so the
    # compressed kernel's whole-workgroup path runs for row 0
    n3 = 2600
    cols3 = [np.arange(0, 2500)] + [np.arange(r, min(n3, r + 200)) for r in range(1, n3)]
    ai3 = np.concatenate(([0], np.cumsum([c.size for c in cols3]))).astype(np.int32)
    aj3 = np.concatenate(cols3).astype(np.int32)
    aa3 = rnd(aj3.size, 55); x3 = rnd(n3, 56)
    got = run_spmv(dev, ai3, aj3, aa3, x3, compress=True)
    assert run_spmv.last_ntab == 0    # 2500 offsets in row 0
    sc3 = np.zeros(n3); np.add.at(sc3, np.repeat(np.arange(n3), np.diff(ai3)), np.abs(aa3 * x3[aj3]))
    assert np.all(np.abs(got - orc.spmv(ai3, aj3, aa3, x3)) <= 1e-12 * sc3)


def test_spmv_with_dot_byproduct(dev):
    """mi355x_spmv_csr_dot + mi355x_spmv_dot_finish: y carries the bits of mi355x_spmv_csr, and x'y (one value per row
    block, summed in block order) equals the oracle's dot to 1e-13 * sum|x_r y_r|.  Shapes: P7 (one lane per row),
    banded long rows with every 7th row empty, 200-entry rows (several lanes per row, ten rows per block); an
    uncompressed plan must be refused (801) without launching anything."""
    k = dev.k
    cases = []
    ai, aj, aa = orc.gen_p7(33, 17, 9)
    cases.append((ai, aj, aa * (1.0 + 0.01 * np.cos(np.arange(aa.size)))))
    n = 3000
    rows_ = [np.arange(max(0, r - 40), min(n, r + 41)) if r % 7 else np.arange(0) for r in range(n)]   # every 7th row empty
    cases.append((np.concatenate(([0], np.cumsum([c.size for c in rows_]))).astype(np.int32), np.concatenate(rows_).astype(np.int32), None))
    n3 = 2600
    cols3 = [np.arange(r, min(n3, r + 200)) for r in range(n3)]
    cases.append((np.concatenate(([0], np.cumsum([c.size for c in cols3]))).astype(np.int32), np.concatenate(cols3).astype(np.int32), None))
    for ci, (ai, aj, aa) in enumerate(cases):
        if aa is None:
            aa = rnd(aj.size, 70 + ci)
        m = ai.size - 1
        x = rnd(m, 80 + ci)
        dai, daj, daa = upload_csr(dev, ai, aj, aa)
        dx = dev.put(x); dy = dev.put(np.full(m, 7.0)); dy2 = dev.put(np.full(m, 9.0)); dout = dev.alloc(64)
        plan = make_plan(dev, ai, None)
        rc = k.mi355x_spmv_csr_dot(dev.h, plan, dai, daj, daa, dx, dy)
        assert rc == 801                                              # no index compression yet: refused
        dev.chk(k.mi355x_spmv_plan_compress_indices(dev.h, plan, ai.ctypes.data, aj.ctypes.data))
        nt = C.c_int(); k.mi355x_spmv_plan_is_compressed(plan, C.byref(nt))
        if nt.value == 0:
            assert k.mi355x_spmv_csr_dot(dev.h, plan, dai, daj, daa, dx, dy) == 801
        else:
            dev.chk(k.mi355x_spmv_csr_dot(dev.h, plan, dai, daj, daa, dx, dy))
            dev.chk(k.mi355x_spmv_dot_finish(dev.h, plan, dout))
            dev.chk(k.mi355x_spmv_csr(dev.h, plan, dai, daj, daa, dx, dy2))
            y, y2 = dev.get(dy, m), dev.get(dy2, m)
            assert_bitexact(y, y2)
            got = dev.get(dout, 1)[0]
            assert abs(got - orc.vec_dot(x, y)) <= 1e-13 * np.sum(np.abs(x * y)), (ci, got)
        dev.chk(k.mi355x_spmv_plan_destroy(plan))
        for q in (dai, daj, daa, dx, dy, dy2, dout):
            dev.free(q)
        assert ci != 0 or nt.value == 7
    # the same by-product out of the pattern kernels: row patterns (case 0 above has them: values streamed, offsets from a dictionary)
    # and value patterns (constant coefficients: the value array is not read; several rows per lane, ragged last workgroup)
    for dims in ((33, 17, 9), (40, 40, 3)):
        ai, aj, aa = orc.gen_p7(*dims)
        m = ai.size - 1
        x = rnd(m, 90)
        dai, daj, daa = upload_csr(dev, ai, aj, aa)
        dx = dev.put(x); dy = dev.put(np.full(m, 7.0)); dy2 = dev.put(np.full(m, 9.0)); dout = dev.alloc(64)
        plan = make_plan(dev, ai, None)
        nv = C.c_int()
        dev.chk(k.mi355x_spmv_plan_value_patterns(dev.h, plan, ai.ctypes.data, aj.ctypes.data, aa.ctypes.data, C.byref(nv)))
        assert nv.value > 0
        yes = C.c_int(); dev.chk(k.mi355x_spmv_plan_dot_available(plan, daa, C.byref(yes))); assert yes.value == 1
        dev.chk(k.mi355x_spmv_csr_dot(dev.h, plan, dai, daj, daa, dx, dy))
        dev.chk(k.mi355x_spmv_dot_finish(dev.h, plan, dout))
        dev.chk(k.mi355x_spmv_csr(dev.h, plan, dai, daj, daa, dx, dy2))
        y, y2 = dev.get(dy, m), dev.get(dy2, m)
        assert_bitexact(y, y2); assert_bitexact(y, orc.spmv(ai, aj, aa, x))
        got = dev.get(dout, 1)[0]
        assert abs(got - orc.vec_dot(x, y)) <= 1e-13 * np.sum(np.abs(x * y)), (dims, got)
        dev.chk(k.mi355x_spmv_csr_dot(dev.h, plan, dai, daj, daa, dx, dy))        # deterministic
        dev.chk(k.mi355x_spmv_dot_finish(dev.h, plan, dout))
        assert dev.get(dout, 1)[0] == got
        dev.chk(k.mi355x_spmv_plan_destroy(plan))
        for q in (dai, daj, daa, dx, dy, dy2, dout):
            dev.free(q)


def test_spmv_short_rows_bitexact(dev):
    """rows of 0..16 nonzeros incl. empty rows, rectangular: one lane per row, reference summation order"""
    ai, aj, aa = random_csr(3001, 2000, lambda rng, m: rng.integers(0, 17, m), 60)
    x = rnd(2000, 61)
    assert_bitexact(run_spmv(dev, ai, aj, aa, x), orc.spmv(ai, aj, aa, x))


def test_spmv_irregular(dev):
    """log-normal row lengths (SURVEY 8d config 4 stand-in, scaled down) + a few rows longer than the LDS stage"""
    def rl(rng, m):
        l = np.clip(np.exp(rng.normal(np.log(60), 0.6, m)), 3, 400)
        l[::97] = 3000  # > 2048: whole-workgroup path
        l[5] = 0
        return l
    ai, aj, aa = random_csr(1500, 6000, rl, 62)
    x = rnd(6000, 63)
    got = run_spmv(dev, ai, aj, aa, x)
    ref = orc.spmv(ai, aj, aa, x)
    scale = np.zeros(ai.size - 1)
    np.add.at(scale, np.repeat(np.arange(ai.size - 1), np.diff(ai)), np.abs(aa * x[aj]))
    assert np.all(np.abs(got - ref) <= 1e-12 * np.maximum(scale, 1e-300))  # BASELINE.md tolerance
    y0 = rnd(ai.size - 1, 64)
    got = run_spmv(dev, ai, aj, aa, x, y0=y0)
    ref = orc.spmv_add(ai, aj, aa, x, y0)
    assert np.all(np.abs(got - ref) <= 1e-12 * np.maximum(scale + np.abs(y0), 1e-300))


def grouped_csr(nnodes, ncols, seed, maxdof=5, maxlen=40, empty=0.05):
    """rows in runs of 1..maxdof sharing one random column list (what a multi-dof FEM matrix looks like to Mat_CheckInode)"""
    rng = np.random.default_rng(seed)
    ai = [0]; aj = []
    for _ in range(nnodes):
        dof = int(rng.integers(1, maxdof + 1))
        ln = 0 if rng.random() < empty else int(rng.integers(1, maxlen + 1))
        cols = np.sort(rng.choice(ncols, size=min(ln, ncols), replace=False))
        for _ in range(dof):
            aj.extend(cols.tolist()); ai.append(len(aj))
    ai = np.array(ai, dtype=np.int32); aj = np.array(aj, dtype=np.int32)
    return ai, aj, rng.standard_normal(aj.size)


@pytest.mark.parametrize("shape", ["groups16", "fem3", "long"])
def test_spmv_grouped_rows(dev, shape):
    """the inode form (mi355x_spmv_plan_group_rows): one stored column list per group of identical rows.  Row blocks of
    short rows (<= 16 nonzeros per row on average: one lane per row) carry the bits of MatMult_SeqAIJ_Inode with the pair
    summation and those of MatMult_SeqAIJ with the plain one -- MatMult and MatMultAdd; longer rows are summed by
    several lanes + a tree and agree within BASELINE.md's 1e-12 * sum|a_ij x_j|"""
    import problems as pb
    if shape == "groups16":
        ai, aj, aa = grouped_csr(6000, 3000, 71, maxlen=16)     # groups of 1..5 rows, rows of 0..16 entries
    elif shape == "fem3":
        ai, aj, aa = pb.gen_fem3(12, 12, 7)                     # 3 dof per node, ~60 nonzeros per row
    else:
        ai, aj, aa = grouped_csr(300, 5000, 72, maxdof=4, maxlen=950, empty=0.0)   # rows up to the shared-index cap (960)
    m = ai.size - 1
    ncols = int(aj.max()) + 1 if aj.size else 1
    x = rnd(max(ncols, m), 73)
    y0 = rnd(m, 74)
    exact = shape == "groups16"
    scale = np.zeros(m); np.add.at(scale, np.repeat(np.arange(m), np.diff(ai)), np.abs(aa * x[aj]))
    for pairsum, ref, ref_add in ((1, orc.spmv_inode(ai, aj, aa, x), orc.matmult(ai, aj, aa, x, y0)[0]),
                                  (0, orc.spmv(ai, aj, aa, x), orc.spmv_add(ai, aj, aa, x, y0))):
        got = run_spmv(dev, ai, aj, aa, x, group=True, pairsum=pairsum)
        nodes, ng, ngj = run_spmv.last_groups
        assert nodes > 0 and ng >= nodes and 3 * ngj <= 2 * aj.size     # grouped, and it pays
        got_add = run_spmv(dev, ai, aj, aa, x, y0=y0, group=True, pairsum=pairsum)
        if exact:
            assert_bitexact(got, ref)
            assert_bitexact(got_add, ref_add)
        else:
            assert np.all(np.abs(got - ref) <= 1e-12 * np.maximum(scale, 1e-300))
            assert np.all(np.abs(got_add - ref_add) <= 1e-12 * np.maximum(scale + np.abs(y0), 1e-300))
    if exact:   # the ungrouped kernels with pair summation: same bits (index-compressed / compressed-row plans of such matrices)
        assert_bitexact(run_spmv(dev, ai, aj, aa, x, pairsum=1), orc.spmv_inode(ai, aj, aa, x))
        assert_bitexact(run_spmv(dev, ai, aj, aa, x, y0=y0, pairsum=1), orc.matmult(ai, aj, aa, x, y0)[0])


@pytest.mark.parametrize("shape", ["stencil", "stencil_patterns", "irregular", "groups16", "fem3", "longrow", "emptyrows"])
def test_spmv_with_diagonal_scaling_epilogue(dev, shape):
    """mi355x_spmv_csr_scaled: y = d .* (A x) is the product followed by PCApply_Jacobi's VecPointwiseMult (jacobi.c:266),
    bit for bit, in every SpMV kernel (plain, index-compressed, grouped rows, long row, rows without entries)"""
    import problems as pb
    kw = {}
    if shape == "stencil":
        ai, aj, aa = orc.gen_p7(13, 11, 9); kw = dict(compress=True, patterns=False)
    elif shape == "stencil_patterns":
        ai, aj, aa = orc.gen_p7(13, 11, 9); kw = dict(compress=True, patterns=True)
    elif shape == "irregular":
        ai, aj, aa = random_csr(3000, 3000, lambda rng, m: rng.integers(0, 40, m), 81)
    elif shape == "groups16":
        ai, aj, aa = grouped_csr(6000, 3000, 71, maxlen=16); kw = dict(group=True, pairsum=1)
    elif shape == "fem3":
        ai, aj, aa = pb.gen_fem3(10, 9, 6); kw = dict(group=True, pairsum=1)
    elif shape == "longrow":
        ai, aj, aa = random_csr(40, 9000, lambda rng, m: np.where(np.arange(m) == 7, 6000, rng.integers(0, 9, m)), 82)
    else:
        ai, aj, aa = random_csr(2000, 2000, lambda rng, m: np.where(rng.random(m) < 0.7, 0, rng.integers(1, 9, m)), 83)
    m = ai.size - 1
    ncols = int(aj.max()) + 1 if aj.size else 1
    x = rnd(max(ncols, m), 84)
    d = rnd(m, 85)
    plain = run_spmv(dev, ai, aj, aa, x, **kw)
    assert_bitexact(run_spmv(dev, ai, aj, aa, x, dscale=d, **kw), plain * d)


def test_spmv_grouping_declines_when_it_would_not_pay(dev):
    ai, aj, aa = orc.gen_p7(9, 8, 7)                            # no two rows share a pattern
    x = rnd(ai.size - 1, 75)
    nodes, ns = orc.check_inode(ai, aj)
    assert nodes == 0
    ns1 = np.ones(ai.size - 1, dtype=np.int32)
    k = dev.k
    plan = make_plan(dev, ai)
    dev.chk(k.mi355x_spmv_plan_group_rows(dev.h, plan, ai.ctypes.data, aj.ctypes.data, ai.size - 1, ns1.ctypes.data))
    ng = C.c_int(-1)
    k.mi355x_spmv_plan_group_info(plan, C.byref(ng), None, None)
    assert ng.value == 0                                        # singleton groups share nothing: plan left alone
    bad = np.array([2, 2], dtype=np.int32)                      # node sizes that do not add up to m
    assert k.mi355x_spmv_plan_group_rows(dev.h, plan, ai.ctypes.data, aj.ctypes.data, 2, bad.ctypes.data) != 0
    dev.chk(k.mi355x_spmv_plan_destroy(plan))
    assert_bitexact(run_spmv(dev, ai, aj, aa, x), orc.spmv(ai, aj, aa, x))


def test_spmv_compressed_rows(dev):
    """off-diagonal-block shape: most rows empty; compressed-row plan touches only listed rows"""
    m, n = 5000, 700
    rng = np.random.default_rng(70)
    rows = np.sort(rng.choice(m, size=300, replace=False)).astype(np.int32)
    cai, aj, aa = random_csr(300, n, lambda r, mm: r.integers(1, 4, mm), 71)
    x = rnd(n, 72)
    y0 = rnd(m, 73)
    got = run_spmv(dev, cai, aj, aa, x, y0=y0, rows=rows)
    # oracle on the uncompressed matrix
    ai = np.zeros(m + 1, dtype=np.int32)
    ai[rows + 1] = np.diff(cai)
    ai = np.cumsum(ai).astype(np.int32)
    assert_bitexact(got, orc.spmv_add(ai, aj, aa, x, y0))


def test_spmv_transpose_via_explicit_transpose(dev):
    ai, aj, aa = random_csr(700, 900, lambda rng, m: rng.integers(0, 12, m), 80)
    x = rnd(700, 81)
    ti, tj, ta = orc.csr_transpose(ai, aj, aa, 900)
    assert_bitexact(run_spmv(dev, ti, tj, ta, x), orc.spmv_t(ai, aj, aa, x, 900))
    z = rnd(900, 82)
    assert_bitexact(run_spmv(dev, ti, tj, ta, x, y0=z), orc.spmv_t_add(ai, aj, aa, x, z, 900))


def test_get_diagonal(dev):
    ai, aj, aa = random_csr(2000, 2000, lambda rng, m: rng.integers(0, 9, m), 90)
    dai, daj, daa = upload_csr(dev, ai, aj, aa)
    dd = dev.alloc(8 * 2000)
    dev.chk(dev.k.mi355x_csr_get_diagonal(dev.h, 2000, dai, daj, daa, dd))
    assert_bitexact(dev.get(dd, 2000), orc.get_diagonal(ai, aj, aa))


@pytest.mark.parametrize("bs", [1, 2, 3, 4, 5])
def test_spmv_bsr(dev, bs):
    mbs, nbs = 400, 500
    ai, aj, _ = random_csr(mbs, nbs, lambda rng, m: rng.integers(0, 30, m), 95 + bs)
    aa = rnd(aj.size * bs * bs, 96)
    x = rnd(nbs * bs, 97)
    dai, daj, daa = upload_csr(dev, ai, aj, aa)
    dx = dev.put(x)
    dy = dev.alloc(8 * mbs * bs)
    dev.chk(dev.k.mi355x_spmv_bsr(dev.h, mbs, bs, dai, daj, daa, dx, dy))
    got = dev.get(dy, mbs * bs)
    ref = orc.spmv_bsr(bs, ai, aj, aa, x)
    assert np.allclose(got, ref, rtol=0, atol=1e-12 * 30 * bs * 10)
    if bs > 1:   # row-block streaming variant, plan over the value stream
        sc = (ai.astype(np.int64) * bs * bs).astype(np.int32)
        plan = make_plan(dev, sc)
        dev.chk(dev.k.mi355x_vec_set(dev.h, mbs * bs, 7.0, dy))
        dev.chk(dev.k.mi355x_spmv_bsr_planned(dev.h, plan, bs, dai, daj, daa, dx, dy))
        assert np.allclose(dev.get(dy, mbs * bs), ref, rtol=0, atol=1e-12 * 30 * bs * 10)


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("shape", ["ragged", "stencil", "wide"])
def test_spmv_bsr4_mfma(dev, variant, shape):
    """bs = 4 on the matrix cores (v_mfma_f64_4x4x4_4b_f64 as multiply + 4-lane reduction): exact on small-integer data
    (every lane map right: any swapped row, column or block shows up as an integer difference), BASELINE.md tolerance
    on random data; block rows of 0 .. 70 blocks incl. counts that are not multiples of the 4 / 8 blocks per step"""
    bs = 4
    if shape == "ragged":
        mbs, nbs = 700, 900
        ai, aj, _ = random_csr(mbs, nbs, lambda rng, m: rng.integers(0, 40, m), 301)
    elif shape == "stencil":
        import problems as pb
        (_, _, _), (ai, aj, _) = pb.elasticity_like(7, 6, 5, dof=1)
        mbs = nbs = ai.size - 1
    else:
        mbs, nbs = 64, 400
        ai, aj, _ = random_csr(mbs, nbs, lambda rng, m: np.where(np.arange(m) % 5 == 0, 70, rng.integers(1, 9, m)), 302)
    rng = np.random.default_rng(303)
    dx_host = rng.integers(-3, 4, nbs * bs).astype(np.float64)
    for exact in (True, False):
        aa = rng.integers(-4, 5, aj.size * 16).astype(np.float64) if exact else rnd(aj.size * 16, 304)
        x = dx_host if exact else rnd(nbs * bs, 305)
        dai, daj, daa = upload_csr(dev, ai, aj, aa)
        dx = dev.put(x); dy = dev.put(np.full(mbs * bs, 7.0))
        dev.chk(dev.k.mi355x_spmv_bsr4_mfma(dev.h, mbs, variant, dai, daj, daa, dx, dy))
        got = dev.get(dy, mbs * bs)
        ref = orc.spmv_bsr(bs, ai, aj, aa, x)
        if exact:
            assert np.array_equal(got, ref)
        else:
            blk_row = np.repeat(np.arange(mbs), np.diff(ai))
            absx = np.abs(aa.reshape(-1, 4, 4)) * np.abs(x.reshape(-1, 4)[aj])[:, :, None]      # [blk][col][row]
            scale = np.zeros((mbs, 4)); np.add.at(scale, blk_row, absx.sum(axis=1))
            assert np.all(np.abs(got - ref) <= 1e-12 * np.maximum(scale.ravel(), 1e-300))
        for q in (dai, daj, daa, dx, dy):
            dev.free(q)


def test_spmv_bsr_wide_block_row(dev):
    """a block row wider than the LDS stage (> 2046 values) takes the whole-workgroup path"""
    bs, mbs, nbs = 3, 40, 600
    ai, aj, _ = random_csr(mbs, nbs, lambda rng, m: np.where(np.arange(m) % 7 == 3, 300, rng.integers(0, 20, m)), 123)
    aa = rnd(aj.size * bs * bs, 124)
    x = rnd(nbs * bs, 125)
    dai, daj, daa = upload_csr(dev, ai, aj, aa)
    dx = dev.put(x); dy = dev.alloc(8 * mbs * bs)
    plan = make_plan(dev, (ai.astype(np.int64) * bs * bs).astype(np.int32))
    dev.chk(dev.k.mi355x_spmv_bsr_planned(dev.h, plan, bs, dai, daj, daa, dx, dy))
    assert np.allclose(dev.get(dy, mbs * bs), orc.spmv_bsr(bs, ai, aj, aa, x), rtol=0, atol=1e-10)


@pytest.mark.parametrize("bs", [2, 3, 5, 8])
def test_spmv_bsr_few_blocks_per_row(dev, bs):
    """block rows with 0, 1 or 2 blocks: a row block of <= 256 block rows then holds more point rows (<= 256*bs) than
    the workgroup has lanes -- every one of them must still be written (block-diagonal BAIJ is the extreme case)"""
    mbs = 1500
    rng = np.random.default_rng(200 + bs)
    cnt = rng.integers(0, 3, mbs); cnt[::5] = 1
    ai = np.concatenate(([0], np.cumsum(cnt))).astype(np.int32)
    aj = np.concatenate([np.sort(rng.choice(mbs, c, replace=False)) for c in cnt]).astype(np.int32)
    aa = rnd(aj.size * bs * bs, 201)
    x = rnd(mbs * bs, 202)
    dai, daj, daa = upload_csr(dev, ai, aj, aa)
    dx = dev.put(x); dy = dev.put(np.full(mbs * bs, 7.0))
    plan = make_plan(dev, (ai.astype(np.int64) * bs * bs).astype(np.int32))
    dev.chk(dev.k.mi355x_spmv_bsr_planned(dev.h, plan, bs, dai, daj, daa, dx, dy))
    ref = orc.spmv_bsr(bs, ai, aj, aa, x)
    got = dev.get(dy, mbs * bs)
    assert np.allclose(got, ref, rtol=0, atol=1e-12 * 2 * bs * 10)   # baij2.c groups a block's products before adding
    assert not np.any(got == 7.0)                        # every point row written
    dev.chk(dev.k.mi355x_spmv_plan_destroy(plan))
    for q in (dai, daj, daa, dx, dy):
        dev.free(q)


def test_pack_unpack(dev):
    k = dev.k
    n = 50000
    x = rnd(n, 110)
    idx = np.random.default_rng(111).permutation(n)[:20000].astype(np.int32)
    dx, didx = dev.put(x), dev.put(idx)
    dbuf = dev.alloc(8 * idx.size)
    dev.chk(k.mi355x_pack(dev.h, idx.size, didx, dx, dbuf))
    assert_bitexact(dev.get(dbuf, idx.size), x[idx])
    y = rnd(n, 112)
    dy = dev.put(y)
    dev.chk(k.mi355x_unpack_add(dev.h, idx.size, didx, dbuf, dy))
    r = y.copy(); r[idx] = r[idx] + x[idx]
    assert_bitexact(dev.get(dy, n), r)
    dev.chk(k.mi355x_unpack_insert(dev.h, idx.size, didx, dbuf, dy))
    r[idx] = x[idx]
    assert_bitexact(dev.get(dy, n), r)
    dev.chk(k.mi355x_unpack_insert(dev.h, 100, None, dbuf, dy))
    r[:100] = x[idx][:100]
    assert_bitexact(dev.get(dy, n), r)


@pytest.mark.parametrize("n", SIZES)
def test_cg_update_matches_separate_kernels_bitwise(dev, n):
    """mi355x_vec_cg_update (x += a p; r -= a w; z = r.*d; z'z, z'r in one sweep) must leave exactly the bits of the
    five separate launches it replaces (axpy, axpy, pointwise mult, norm, dot), vectors AND reduction results, and
    the vectors must equal the oracle's element-wise loops."""
    k = dev.k
    a = 0.731
    p, w, d, x, r = rnd(n, 11), rnd(n, 12), 1.0 / (2.0 + rnd(n, 13) ** 2), rnd(n, 14), rnd(n, 15)
    dp, dw, dd = dev.put(p), dev.put(w), dev.put(d)
    x1, r1, z1 = dev.put(x), dev.put(r), dev.alloc(8 * max(n, 2))
    x2, r2, z2 = dev.put(x), dev.put(r), dev.alloc(8 * max(n, 2))
    hs = dev.host_scratch()
    dev.chk(k.mi355x_vec_axpy(dev.h, n, a, dp, x1))
    dev.chk(k.mi355x_vec_axpy(dev.h, n, -a, dw, r1))
    dev.chk(k.mi355x_vec_pointwise_mult(dev.h, n, r1, dd, z1))
    dev.chk(k.mi355x_vec_norm(dev.h, n, 2, z1, hs)); zz = dev.scalar_out(1)[0]
    dev.chk(k.mi355x_vec_dot(dev.h, n, z1, r1, hs)); zr = dev.scalar_out(1)[0]
    dev.chk(k.mi355x_vec_norm(dev.h, n, 2, r1, hs)); rr = dev.scalar_out(1)[0]
    dev.chk(k.mi355x_vec_cg_update(dev.h, n, a, dp, dw, dd, x2, r2, z2, hs))
    out = dev.scalar_out(3)
    for u, v in ((x1, x2), (r1, r2), (z1, z2)):
        assert_bitexact(dev.get(u, n), dev.get(v, n))
    assert_bitexact(np.array([zz, zr, rr]), out)
    xo, ro = x.copy(), r.copy()
    orc.vec_axpy(xo, a, p); orc.vec_axpy(ro, -a, w)
    zo = np.zeros(n); orc.vec_pointwise_mult(zo, ro, d)
    assert_bitexact(dev.get(x2, n), xo); assert_bitexact(dev.get(r2, n), ro); assert_bitexact(dev.get(z2, n), zo)
    # z aliasing w, as KSPSolve_CG uses it (cg.c:122: W = Z)
    dev.chk(k.mi355x_memcpy_h2d(dev.h, x2, x.ctypes.data, x.nbytes)) if n else None
    dev.chk(k.mi355x_memcpy_h2d(dev.h, r2, r.ctypes.data, r.nbytes)) if n else None
    dev.chk(k.mi355x_vec_cg_update(dev.h, n, a, dp, dw, dd, x2, r2, dw, hs))
    assert_bitexact(dev.scalar_out(3), out)
    assert_bitexact(dev.get(dw, n), zo); assert_bitexact(dev.get(r2, n), ro); assert_bitexact(dev.get(x2, n), xo)
    for q in (dp, dw, dd, x1, r1, z1, x2, r2, z2):
        dev.free(q)


@pytest.mark.parametrize("n", [0, 1, 2, 255, 4097, 1_000_001])
def test_cg_update_dev_scalar_form(dev, n):
    """mi355x_vec_cg_update_dev: a = beta / dpi formed on the device from a device-resident dpi.  Same bits as the
    host-scalar form (IEEE division on both sides); dpi comes back in out[2]; when a break-down test of cg.c:196-199
    fires (dpi NaN/Inf/0, or a sign change against dpiold) the vectors are left untouched."""
    k = dev.k
    beta, dpi = 0.83, 1.37
    a = beta / dpi
    p, w, d, x, r = rnd(n, 21), rnd(n, 22), 1.0 / (2.0 + rnd(n, 23) ** 2), rnd(n, 24), rnd(n, 25)
    dp, dw, dd = dev.put(p), dev.put(w), dev.put(d)
    x1, r1, z1 = dev.put(x), dev.put(r), dev.alloc(8 * max(n, 2))
    x2, r2, z2 = dev.put(x), dev.put(r), dev.alloc(8 * max(n, 2))
    ddpi = dev.put(np.array([dpi, 0.0]))
    hs = dev.host_scratch()
    dev.chk(k.mi355x_vec_cg_update(dev.h, n, a, dp, dw, dd, x1, r1, z1, hs)); ref = dev.scalar_out(3)
    dev.chk(k.mi355x_vec_cg_update_dev(dev.h, n, beta, ddpi, 0.5, 1, dp, dw, dd, x2, r2, z2, hs, 0)); out = dev.scalar_out(4)
    assert_bitexact(out[:3], ref)
    assert out[3] == dpi
    for u, v in ((x1, x2), (r1, r2), (z1, z2)):
        assert_bitexact(dev.get(u, n), dev.get(v, n))
    # refused updates: nothing is written
    xb, rb, zb = dev.get(x2, n), dev.get(r2, n), dev.get(z2, n)
    for bad, dpiold, chk in ((0.0, 1.0, 0), (np.nan, 1.0, 0), (np.inf, 1.0, 0), (-1.0, 2.0, 1), (1.0, -2.0, 1)):
        dev.chk(k.mi355x_memcpy_h2d(dev.h, ddpi, np.array([bad]).ctypes.data, 8)); dev.sync()
        dev.chk(k.mi355x_vec_cg_update_dev(dev.h, n, beta, ddpi, dpiold, chk, dp, dw, dd, x2, r2, z2, hs, 0)); out = dev.scalar_out(4)
        assert (np.isnan(out[3]) and np.isnan(bad)) or out[3] == bad
        assert out[0] == 0.0 and out[1] == 0.0 and out[2] == 0.0
        assert_bitexact(dev.get(x2, n), xb); assert_bitexact(dev.get(r2, n), rb); assert_bitexact(dev.get(z2, n), zb)
    # a sign change is only a break-down when the caller asks for the test (first iteration: check_sign = 0)
    dev.chk(k.mi355x_memcpy_h2d(dev.h, ddpi, np.array([-1.0]).ctypes.data, 8)); dev.sync()
    dev.chk(k.mi355x_vec_cg_update_dev(dev.h, n, beta, ddpi, 2.0, 0, dp, dw, dd, x2, r2, z2, hs, 0)); out = dev.scalar_out(4)
    if n:
        assert not np.array_equal(dev.get(x2, n), xb)
    # result kept on the device AND handed to the host by the same launch; then p = z + (z'r / den) p on the device
    dev.chk(k.mi355x_memcpy_h2d(dev.h, ddpi, np.array([dpi]).ctypes.data, 8)); dev.sync()
    for q_, a_ in ((x2, x), (r2, r)):
        if n:
            dev.chk(k.mi355x_memcpy_h2d(dev.h, q_, a_.ctypes.data, a_.nbytes))
    dres = dev.alloc(64)
    dev.chk(k.mi355x_vec_cg_update_dev(dev.h, n, beta, ddpi, 0.5, 1, dp, dw, dd, x2, r2, z2, dres, 1))
    dev.chk(k.mi355x_handle_wait_result(dev.h))
    addr = k.mi355x_handle_host_scratch(dev.h)
    polled = np.ctypeslib.as_array((C.c_double * 4).from_address(addr)).copy()
    assert_bitexact(polled[:3], ref); assert polled[3] == dpi
    assert_bitexact(dev.get(dres, 4), polled)
    den = 0.77
    dev.chk(k.mi355x_vec_aypx_dev(dev.h, n, C.c_void_p(dres.value + 8), den, z2, dp))       # p <- z + (zr/den) p
    pref = p.copy(); orc.vec_aypx(pref, polled[1] / den, dev.get(z2, n))
    assert_bitexact(dev.get(dp, n), pref)
    # publish: device values -> pinned scratch + completion number
    dev.chk(k.mi355x_handle_publish(dev.h, dres, 4)); dev.chk(k.mi355x_handle_wait_result(dev.h))
    assert_bitexact(np.ctypeslib.as_array((C.c_double * 4).from_address(addr)).copy(), polled)
    for q in (dp, dw, dd, x1, r1, z1, x2, r2, z2, ddpi, dres):
        dev.free(q)


@pytest.mark.parametrize("n", [0, 1, 3, 255, 4097, 1_000_001])
def test_bcgs_fused_kernels_match_separate_kernels_bitwise(dev, n):
    """mi355x_vec_pmult_dot / pmult_dotnorm2 / bcgs_update against the separate launches they replace
    (pointwise mult + dot; pointwise mult + dotnorm2; axpbypcz + waxpy + norm + dot): vectors and sums bit for bit,
    including the identity preconditioner (d = NULL) and alpha = 1 / omega = 0, 1, -1 special forms."""
    k = dev.k
    x, d, y, s_ = rnd(n, 31), 1.0 / (2.0 + rnd(n, 32) ** 2), rnd(n, 33), rnd(n, 34)
    dx, dd, dy, ds = dev.put(x), dev.put(d), dev.put(y), dev.put(s_)
    w1, w2 = dev.alloc(8 * max(n, 2)), dev.alloc(8 * max(n, 2))
    hs = dev.host_scratch()
    for dptr in (dd, None):
        if dptr is None:
            dev.chk(k.mi355x_vec_copy(dev.h, n, dx, w1))
        else:
            dev.chk(k.mi355x_vec_pointwise_mult(dev.h, n, dx, dptr, w1))
        dev.chk(k.mi355x_vec_dot(dev.h, n, w1, dy, hs)); ref = dev.scalar_out(1)
        dev.chk(k.mi355x_vec_pmult_dot(dev.h, n, dx, dptr, dy, w2, hs)); got = dev.scalar_out(1)
        assert_bitexact(got, ref); assert_bitexact(dev.get(w2, n), dev.get(w1, n))
        dev.chk(k.mi355x_vec_dotnorm2(dev.h, n, ds, w1, hs)); ref = dev.scalar_out(2)
        dev.chk(k.mi355x_vec_pmult_dotnorm2(dev.h, n, dx, dptr, ds, w2, hs)); got = dev.scalar_out(2)
        assert_bitexact(got, ref); assert_bitexact(dev.get(w2, n), dev.get(w1, n))
    p, t, rp, xx = rnd(n, 35), rnd(n, 36), rnd(n, 37), rnd(n, 38)
    dp, dt, drp = dev.put(p), dev.put(t), dev.put(rp)
    x1, x2, r1, r2 = dev.put(xx), dev.put(xx), dev.alloc(8 * max(n, 2)), dev.alloc(8 * max(n, 2))
    for alpha, omega in ((0.37, -1.21), (1.0, 0.5), (0.2, 1.0), (0.2, -1.0), (0.3, 0.0)):
        for q in (x1, x2):
            if n:
                dev.chk(k.mi355x_memcpy_h2d(dev.h, q, xx.ctypes.data, xx.nbytes))
        dev.chk(k.mi355x_vec_axpbypcz(dev.h, n, alpha, omega, 1.0, dp, ds, x1))
        dev.chk(k.mi355x_vec_waxpy(dev.h, n, -omega, dt, ds, r1))
        dev.chk(k.mi355x_vec_norm(dev.h, n, 2, r1, hs)); rr = dev.scalar_out(1)[0]
        dev.chk(k.mi355x_vec_dot(dev.h, n, r1, drp, hs)); rho = dev.scalar_out(1)[0]
        dev.chk(k.mi355x_vec_bcgs_update(dev.h, n, alpha, omega, dp, ds, dt, drp, x2, r2, hs)); got = dev.scalar_out(2)
        assert_bitexact(got, np.array([rr, rho]))
        assert_bitexact(dev.get(x2, n), dev.get(x1, n)); assert_bitexact(dev.get(r2, n), dev.get(r1, n))
    for q in (dx, dd, dy, ds, w1, w2, dp, dt, drp, x1, x2, r1, r2):
        dev.free(q)


def _tri_levels(n, rp, rl, cj):
    lev = np.zeros(n, dtype=np.int32)
    for i in range(n):
        d = cj[rp[i]:rp[i] + rl[i]]
        lev[i] = (lev[d].max() + 1) if d.size else 0
    return lev


def _tri_plan_arrays(dev, plan):
    out = []
    for which, dt in ((0, np.int32), (1, np.int32), (2, np.int32), (3, np.int32), (4, np.uint64), (5, np.uint64), (6, np.uint64), (7, np.uint8), (8, np.int32), (9, np.int32)):
        nb = C.c_size_t()
        dev.chk(dev.k.mi355x_trisolve_debug_get(plan, which, None, 0, C.byref(nb)))
        a = np.empty(nb.value // np.dtype(dt).itemsize, dtype=dt)
        if a.nbytes:
            dev.chk(dev.k.mi355x_trisolve_debug_get(plan, which, a.ctypes.data, a.nbytes, C.byref(nb)))
        out.append(a)
    return out


@pytest.mark.parametrize("shape", ["wide", "ragged", "chain", "tiny", "empty_rows", "n1", "n64", "n65", "longrow", "deepchain"])
@pytest.mark.parametrize("kind", ["lower", "upper", "upper_scaled"])
def test_trisolve_row_plans_built_on_the_device_equal_the_host_built_plans(dev, shape, kind, monkeypatch):
    """csrc/trisolve_build.hip lays a row plan out on the device (radix sort of the rows by (level, longer first), positions,
    per-slice words, scan of the slice widths, fill kernel); MI355X_TRISOLVE_BUILD=host is the host threads' route.  Every array of
    the plan -- slice offsets, (length, sub-step) words, position <-> row maps, sliced-ELL column positions and values, diagonals,
    scales, sub-steps, level extents -- must be identical, padding included; the entries may sit anywhere inside the caller's
    column / value arrays (the device route uploads only the range the rows name)."""
    k = dev.k
    rng = np.random.default_rng(hash((shape, kind)) % (1 << 31))
    n = {"wide": 40000, "ragged": 9000, "chain": 700, "tiny": 5, "empty_rows": 3000, "n1": 1, "n64": 64, "n65": 65, "longrow": 4000, "deepchain": 70000}[shape]
    # strictly lower-triangular dependency structure (for an upper solve the caller hands over reversed roles; the plan only needs
    # "dependencies come earlier in level order", so a lower structure serves both kinds here)
    rl = np.zeros(n, dtype=np.int32)
    cols = []
    for i in range(n):
        if shape == "wide":
            cand = sorted(c for c in (i - 7000, i - 200, i - 1) if c >= 0 and not (c == i - 1 and i % 200 == 0))   # a 3-D stencil's lower part
        elif shape == "ragged":
            m = int(rng.integers(0, 40)) if i % 11 else 0
            cand = sorted(set(int(c) for c in rng.integers(max(0, i - 3000), max(i, 1), size=m) if c < i)) if i else []
        elif shape in ("chain", "deepchain"):
            cand = [i - 1] if i and i % 3 else ([i - 2] if i > 1 else [])
        elif shape in ("n1", "n64", "n65"):
            cand = [c for c in (i - 1, i - 9) if c >= 0 and i % 4]
        elif shape == "longrow":                            # a few rows with thousands of entries among short ones
            cand = list(range(0, i, 1 if i in (1500, 3999) else max(i, 1))) if i in (1500, 3999) else ([i - 1] if i % 2 else [])
        elif shape == "tiny":
            cand = list(range(i))
        else:
            cand = [] if i % 2 else sorted(set(int(c) for c in rng.integers(0, max(i, 1), size=3) if c < i))
        rl[i] = len(cand); cols.append(cand)
    pad = 17                                                  # the rows' entries do not start at the arrays' first element
    rp = (pad + np.concatenate(([0], np.cumsum(rl)[:-1]))).astype(np.int32)
    nz = int(rl.sum())
    cj = np.full(pad + nz + 5, -7, dtype=np.int32)            # entries no row names hold an invalid column: never looked at
    cv = np.full(pad + nz + 5, np.nan)
    for i in range(n):
        cj[rp[i]:rp[i] + rl[i]] = cols[i]
    cv[pad:pad + nz] = rng.standard_normal(nz)
    lev = _tri_levels(n, rp, rl, cj)
    nlev = int(lev.max()) + 1
    dinv = rng.standard_normal(n) if kind != "lower" else None
    rsc = rng.standard_normal(n) if kind == "upper_scaled" else None
    plans = {}
    for route in ("host", "device"):
        monkeypatch.setenv("MI355X_TRISOLVE_BUILD", route)
        monkeypatch.setenv("MI355X_TRISOLVE_SPLIT", "0")
        pl = C.c_void_p()
        if rsc is not None:
            dev.chk(k.mi355x_trisolve_plan_create_scaled(dev.h, n, nlev, lev.ctypes.data, rp.ctypes.data, rl.ctypes.data, cj.ctypes.data, cv.ctypes.data,
                                                         dinv.ctypes.data, rsc.ctypes.data, C.byref(pl)))
        else:
            dev.chk(k.mi355x_trisolve_plan_create(dev.h, n, nlev, lev.ctypes.data, rp.ctypes.data, rl.ctypes.data, cj.ctypes.data, cv.ctypes.data,
                                                  dinv.ctypes.data if dinv is not None else None, C.byref(pl)))
        plans[route] = _tri_plan_arrays(dev, pl)
        k.mi355x_trisolve_plan_destroy(pl)
    names = ["ptr", "info", "row", "col", "val", "dinv", "rscale", "nsub", "pos", "levpos"]
    for name, a, b in zip(names, plans["host"], plans["device"]):
        assert a.shape == b.shape and np.array_equal(a, b), name
    assert plans["host"][3].size > 0 or shape == "tiny" or nz == 0


def test_trisolve_device_built_plan_rejects_what_the_host_route_rejects(dev, monkeypatch):
    """a column index outside the matrix, a dependency that does not come earlier, a level without rows: an error code from either
    route, no fault"""
    k = dev.k
    n = 300
    rl = np.ones(n, dtype=np.int32); rl[0] = 0
    rp = np.arange(n, dtype=np.int32) - 1; rp[0] = 0
    cj = np.arange(n - 1, dtype=np.int32)                     # row i depends on row i - 1
    cv = np.ones(n - 1)
    lev = np.arange(n, dtype=np.int32)
    monkeypatch.setenv("MI355X_TRISOLVE_SPLIT", "0")
    for route in ("host", "device"):
        monkeypatch.setenv("MI355X_TRISOLVE_BUILD", route)
        pl = C.c_void_p()
        # a chain of 300 levels in slices of 64: a slice spans more than 255 levels?  no -- 64 rows per slice, 64 sub-steps: accepted
        assert k.mi355x_trisolve_plan_create(dev.h, n, n, lev.ctypes.data, rp.ctypes.data, rl.ctypes.data, cj.ctypes.data, cv.ctypes.data, None, C.byref(pl)) == 0
        k.mi355x_trisolve_plan_destroy(pl)
        bad_lev = lev.copy(); bad_lev[10] = 5                 # row 10 would run before row 9, which it depends on (and level 10 is empty)
        assert k.mi355x_trisolve_plan_create(dev.h, n, n, bad_lev.ctypes.data, rp.ctypes.data, rl.ctypes.data, cj.ctypes.data, cv.ctypes.data, None, C.byref(pl)) != 0
    monkeypatch.setenv("MI355X_TRISOLVE_BUILD", "device")
    pl = C.c_void_p()
    bad_cj = cj.copy(); bad_cj[50] = n + 5
    assert k.mi355x_trisolve_plan_create(dev.h, n, n, lev.ctypes.data, rp.ctypes.data, rl.ctypes.data, bad_cj.ctypes.data, cv.ctypes.data, None, C.byref(pl)) != 0
    bad_cj = cj.copy(); bad_cj[50] = -3
    assert k.mi355x_trisolve_plan_create(dev.h, n, n, lev.ctypes.data, rp.ctypes.data, rl.ctypes.data, bad_cj.ctypes.data, cv.ctypes.data, None, C.byref(pl)) != 0
    dev.sync()
