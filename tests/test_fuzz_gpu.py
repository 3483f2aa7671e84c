"""Seeded random shapes through the SpMV kernels (plain CSR, index-compressed, add form, x'y by-product, BCSR) against
the oracle: row-length mixes that put row-block boundaries at odd/even offsets, empty rows and empty row blocks,
single-row and single-entry matrices, rows that span the LDS stage, banded patterns that compress and random ones that
do not.  Short rows (<= 16 per row on average in a block) must match bit for bit, the others to 1e-12 * sum|a x|."""
import ctypes as C
import os

import numpy as np
import pytest

import orc
from gpu import ksp_type_for

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(built):
    from gpu import Dev
    d = Dev()
    yield d
    d.free_all()


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


FUZZ_OFFSET = int(os.environ.get("FUZZ_OFFSET", "0"))   # extended runs: FUZZ_OFFSET=100000 python -m pytest tests/test_fuzz_gpu.py


def make_case(seed):
    rng = np.random.default_rng(1000 + seed + FUZZ_OFFSET)
    kind = seed % 8
    m = int(rng.integers(1, 6000))
    n = m if kind in (0, 1, 2, 3) else int(rng.integers(1, 6000))
    if kind == 0:      # banded, few offsets: compresses
        offs = np.unique(rng.integers(-40, 41, int(rng.integers(1, 12))))
        rows = [np.array([c for c in r + offs if 0 <= c < n and rng.random() < 0.9], dtype=np.int64) for r in range(m)]
    elif kind == 1:    # stencil-like with empty stretches
        offs = np.array([-37, -1, 0, 1, 37])
        rows = [np.array([c for c in r + offs if 0 <= c < n]) if (r // 300) % 3 else np.zeros(0, np.int64) for r in range(m)]
    elif kind == 2:    # long banded rows (several lanes per row)
        w = int(rng.integers(30, 120))
        rows = [np.arange(max(0, r - w), min(n, r + w + 1)) for r in range(m)]
    elif kind == 3:    # a few rows longer than the LDS stage inside short ones
        big = set(rng.integers(0, m, 3).tolist())
        rows = [np.sort(rng.choice(n, size=min(n, 2500 if r in big else int(rng.integers(0, 9))), replace=False)) for r in range(m)]
    elif kind == 4:    # short random rows, rectangular
        rows = [np.sort(rng.choice(n, size=min(n, int(rng.integers(0, 17))), replace=False)) for r in range(m)]
    elif kind == 5:    # log-normal row lengths
        lens = np.clip(np.exp(rng.normal(2.5, 1.0, m)).astype(int), 0, min(n, 400))
        rows = [np.sort(rng.choice(n, size=int(l), replace=False)) for l in lens]
    elif kind == 6:    # one entry per row / one row
        if rng.random() < 0.5:
            rows = [np.array([int(rng.integers(0, n))]) for _ in range(m)]
        else:
            m = 1
            rows = [np.sort(rng.choice(n, size=min(n, int(rng.integers(1, 3000))), replace=False))]
    else:              # exactly-full row blocks: row lengths that tile the 2046-entry stage
        L = int(rng.choice([2, 3, 6, 11, 31, 62, 93, 186]))
        rows = [np.sort(rng.choice(n, size=min(n, L), replace=False)) for _ in range(m)]
    lens = np.array([r.size for r in rows])
    ai = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    aj = (np.concatenate(rows) if lens.sum() else np.zeros(0)).astype(np.int32)
    aa = rng.standard_normal(aj.size)
    x = rng.standard_normal(n)
    return ai, aj, aa, x, m, n


def check(got, ref, ai, aj, aa, x, what):
    m = ai.size - 1
    lens = np.diff(ai)
    scale = np.zeros(m)
    if aj.size:
        np.add.at(scale, np.repeat(np.arange(m), lens), np.abs(aa * x[aj]))
    assert np.all(np.abs(got - ref) <= 1e-12 * scale + 0.0), what
    # rows in blocks of short rows are summed by one lane in column order: most of them bit for bit
    short = lens <= 8
    if short.all():
        assert np.array_equal(bits(got), bits(ref)), what + " (short rows: bit-exact)"


@pytest.mark.parametrize("seed", range(48))
def test_spmv_random_shapes(dev, seed):
    k = dev.k
    ai, aj, aa, x, m, n = make_case(seed)
    dai = dev.put(ai); daj = dev.put(aj if aj.size else np.zeros(2, np.int32)); daa = dev.put(aa if aa.size else np.zeros(2))
    dx = dev.put(x)
    y0 = np.random.default_rng(seed).standard_normal(m)
    # the reference's dispatch: inode routines (pair summation) when Mat_CheckInode keeps them, plain loops otherwise;
    # the plan gets the same decision and, where it pays, the grouped-row form
    ref, nodes = orc.matmult(ai, aj, aa, x)
    refadd = orc.matmult(ai, aj, aa, x, y0)[0]
    ns = np.ascontiguousarray(orc.check_inode(ai, aj)[1], dtype=np.int32)
    for compress in (False, True):
        plan = C.c_void_p()
        dev.chk(k.mi355x_spmv_plan_create(dev.h, m, ai.ctypes.data, None, C.byref(plan)))
        dev.chk(k.mi355x_spmv_plan_set_pairsum(plan, 1 if nodes else 0))
        if nodes and not compress and aj.size:
            dev.chk(k.mi355x_spmv_plan_group_rows(dev.h, plan, ai.ctypes.data, aj.ctypes.data, nodes, ns.ctypes.data))
        nt = C.c_int(0)
        if compress:
            dev.chk(k.mi355x_spmv_plan_compress_indices(dev.h, plan, ai.ctypes.data, (aj if aj.size else np.zeros(2, np.int32)).ctypes.data))
            k.mi355x_spmv_plan_is_compressed(plan, C.byref(nt))
        dy = dev.put(np.full(m, 7.0))
        dev.chk(k.mi355x_spmv_csr(dev.h, plan, dai, daj, daa, dx, dy))
        got = dev.get(dy, m)
        check(got, ref, ai, aj, aa, x, "seed %d compress %d ntab %d" % (seed, compress, nt.value))
        dz = dev.put(y0)
        dev.chk(k.mi355x_spmv_csr_add(dev.h, plan, dai, daj, daa, dx, dz, dz))
        check(dev.get(dz, m), refadd, ai, aj, aa, x, "add seed %d compress %d" % (seed, compress))
        if compress and nt.value and m == n:
            dev.chk(k.mi355x_vec_set(dev.h, m, 7.0, dy))
            dout = dev.alloc(64)
            dev.chk(k.mi355x_spmv_csr_dot(dev.h, plan, dai, daj, daa, dx, dy))
            dev.chk(k.mi355x_spmv_dot_finish(dev.h, plan, dout))
            y2 = dev.get(dy, m)
            assert np.array_equal(bits(y2), bits(got))
            d = dev.get(dout, 1)[0]
            assert abs(d - float(np.dot(x[:m], y2))) <= 1e-12 * float(np.sum(np.abs(x[:m] * y2))) + 1e-300
            dev.free(dout)
        dev.chk(k.mi355x_spmv_plan_destroy(plan))
        dev.free(dy); dev.free(dz)
    for q in (dai, daj, daa, dx):
        dev.free(q)


@pytest.mark.parametrize("seed", range(16))
def test_bsr_random_shapes(dev, seed):
    k = dev.k
    rng = np.random.default_rng(3000 + seed + FUZZ_OFFSET)
    bs = int(rng.integers(2, 9))
    mbs = int(rng.integers(1, 1200)); nbs = int(rng.integers(1, 1200))
    maxb = [1, 3, 10, 40, 300][seed % 5]
    cnt = np.minimum(rng.integers(0, maxb + 1, mbs), nbs)
    if seed % 5 == 4:
        cnt[:] = np.minimum(rng.integers(0, 4, mbs), nbs); cnt[int(rng.integers(0, mbs))] = min(nbs, 300)   # one block row wider than the stage
    ai = np.concatenate(([0], np.cumsum(cnt))).astype(np.int32)
    aj = (np.concatenate([np.sort(rng.choice(nbs, int(c), replace=False)) for c in cnt]) if cnt.sum() else np.zeros(0)).astype(np.int32)
    aa = rng.standard_normal(aj.size * bs * bs)
    x = rng.standard_normal(nbs * bs)
    dai = dev.put(ai); daj = dev.put(aj if aj.size else np.zeros(2, np.int32)); daa = dev.put(aa if aa.size else np.zeros(2))
    dx = dev.put(x); dy = dev.put(np.full(mbs * bs, 7.0))
    plan = C.c_void_p()
    sc = (ai.astype(np.int64) * bs * bs).astype(np.int32)
    dev.chk(k.mi355x_spmv_plan_create(dev.h, mbs, sc.ctypes.data, None, C.byref(plan)))
    dev.chk(k.mi355x_spmv_bsr_planned(dev.h, plan, bs, dai, daj, daa, dx, dy))
    got = dev.get(dy, mbs * bs)
    ref = orc.spmv_bsr(bs, ai, aj, aa, x)
    scale = np.zeros(mbs * bs)
    if aj.size:
        blocks = aa.reshape(-1, bs, bs)                       # column-major blocks: [blk][c][r]
        xb = x.reshape(nbs, bs)[aj]                           # [blk][c]
        contrib = np.abs(blocks * xb[:, :, None]).sum(1)      # [blk][r]
        np.add.at(scale.reshape(mbs, bs), np.repeat(np.arange(mbs), cnt), contrib)
    assert np.all(np.abs(got - ref) <= 1e-12 * scale), "bsr seed %d bs %d" % (seed, bs)
    dev.chk(k.mi355x_spmv_plan_destroy(plan))
    for q in (dai, daj, daa, dx, dy):
        dev.free(q)


@pytest.fixture(scope="module")
def P(built):
    import importlib
    return importlib.import_module("petsc-dev_amd.petsc")


@pytest.mark.parametrize("seed", range(30))
def test_ksp_random_small_systems(P, seed):
    """every solver x preconditioner of the path on small random diagonally dominant systems (sizes 1..700, also
    n = 1 and n = 2, symmetric for CG) against the oracle: same convergence reason, iteration count within +-1
    (BiCGStab: +-max(2, 20 %)), histories to 1e-6 while the residual is above 1e-4 of its start, same solution."""
    rng = np.random.default_rng(5000 + seed + FUZZ_OFFSET)
    ksp = ["cg", "gmres", "bcgs"][seed % 3]
    pc = ["none", "jacobi", "bjacobi", "ilu"][(seed // 3) % 4]
    n = [1, 2, 3, 17, 64, 257, 700][seed % 7]
    dens = min(n, int(rng.integers(1, 9)))
    rows = [np.unique(np.concatenate(([r], rng.choice(n, size=dens, replace=False)))) for r in range(n)]
    if ksp == "cg":     # symmetric pattern and values
        import scipy.sparse as sp
        A = sp.lil_matrix((n, n))
        for r in range(n):
            for c in rows[r]:
                v = rng.standard_normal()
                A[r, c] = v; A[c, r] = v
        A = sp.csr_matrix(A)
        A.setdiag(np.abs(A).sum(1).A1 + 1.0 + rng.random(n))
        A.sort_indices()
        ai, aj, aa = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    else:
        lens = np.array([r.size for r in rows])
        ai = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
        aj = np.concatenate(rows).astype(np.int32)
        aa = rng.standard_normal(aj.size)
        rr = np.repeat(np.arange(n), lens)
        rowabs = np.zeros(n); np.add.at(rowabs, rr, np.abs(aa))
        aa[aj == rr] = rowabs + 1.0 + rng.random(n)
    xs = rng.standard_normal(n)
    b = orc.spmv(ai, aj, aa, xs)
    L = P.lib()
    Am = P.Mat.from_csr(ai, aj, aa)
    vb = P.Vec.from_array(b, comm=L.COMM_SELF); vx = P.Vec.from_array(np.zeros(n), comm=L.COMM_SELF)
    k = P.KSP(comm=L.COMM_SELF)
    k.set_operators(Am)
    L.PetscOptionsClear()
    L.PetscOptionsInsertString(("-ksp_type %s -pc_type %s -ksp_gmres_restart 12" % (ksp_type_for(ksp), pc)).encode())
    k.set_tolerances(rtol=1e-9, max_it=400)
    k.set_from_options()
    k.record_history()
    k.solve(vb, vx)
    L.PetscOptionsClear()
    x, h, its, reason = vx.array(), k.history(), k.its, k.reason
    okw = dict(blocks=[0, n], sub_ksp="preonly", sub_pc="ilu") if pc == "bjacobi" else {}
    xr, hr, itsr, rr_ = orc.ksp_solve(ai, aj, aa, b, ksp=ksp, pc=pc, rtol=1e-9, max_it=400, restart=12, **okw)
    assert reason == rr_, (ksp, pc, n, reason, rr_)
    tol_its = max(2, itsr // 5) if ksp == "bcgs" else 1
    assert abs(its - itsr) <= tol_its, (ksp, pc, n, its, itsr)
    kk = min(len(h), len(hr))
    early = hr[:kk] > 1e-4 * hr[0]
    assert np.allclose(h[:kk][early], hr[:kk][early], rtol=1e-6, atol=0)
    assert np.linalg.norm(x - xr) <= 1e-6 * max(np.linalg.norm(xr), 1e-300)
    assert np.linalg.norm(x - xs) <= 1e-6 * max(np.linalg.norm(xs), 1e-300)


@pytest.mark.parametrize("ksp,pc", [("cg", "jacobi"), ("gmres", "bjacobi"), ("bcgs", "jacobi"), ("gmres", "ilu")])
def test_repeated_solves_are_bitwise_reproducible(P, ksp, pc):
    """fixed reduction trees, ticketed hand-off, no floating-point atomics: five solves of the same system (new KSP and
    vectors each time, 0.6 M unknowns so that every kernel runs many workgroups per CU) give the same history bits and
    the same solution bits"""
    L = P.lib()
    ai, aj, aa = P.gen_poisson7(96, 80, 80)
    n = ai.size - 1
    aa = aa * (1.0 + 0.05 * np.cos(np.arange(aa.size)))
    b = np.cos(0.01 * np.arange(n))
    A = P.Mat.from_csr(ai, aj, aa)
    ref = None
    for rep in range(5):
        vb = P.Vec.from_array(b, comm=L.COMM_SELF); vx = P.Vec.from_array(np.zeros(n), comm=L.COMM_SELF)
        k = P.KSP(comm=L.COMM_SELF)
        k.set_operators(A)
        L.PetscOptionsClear()
        L.PetscOptionsInsertString(("-ksp_type %s -pc_type %s -ksp_gmres_restart 10" % (ksp_type_for(ksp), pc)).encode())
        k.set_tolerances(rtol=1e-6, max_it=60)
        k.set_from_options(); k.record_history()
        k.solve(vb, vx)
        L.PetscOptionsClear()
        got = (k.its, k.reason, bits(k.history()).copy(), bits(vx.array()).copy())
        if ref is None:
            ref = got
        else:
            assert got[:2] == ref[:2] and np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3]), "run %d differs" % rep


@pytest.mark.parametrize("seed", range(24))
def test_spmv_fuzz_value_patterns(dev, seed):
    """constant-coefficient operators of random shape: a handful of row kinds {offsets, values} -- some with more than 8 entries,
    empty ones, stored zeros of either sign, rectangular shapes -- dealt to the rows in runs or at random; the product from the row
    dictionary (the value array on the device is NaN) carries the oracle's bits for y = Ax, y += Ax, y = d .* Ax and the inode
    summation order; one more kind than the table holds and the analysis declines"""
    k = dev.k
    rng = np.random.default_rng(7000 + seed + FUZZ_OFFSET)
    m = int(rng.integers(1, 5000))
    n = m if seed % 3 else int(rng.integers(max(1, m // 2), 2 * m + 2))
    nk = int(rng.integers(1, 30))
    kinds = []
    for _ in range(nk):
        ln = int(rng.choice([0, 1, 2, 3, 5, 7, 8, 9, 13, 27]))
        offs = np.unique(rng.integers(-min(n, 300), min(n, 300) + 1, ln))
        vals = rng.standard_normal(offs.size)
        if offs.size and rng.random() < 0.3:
            vals[int(rng.integers(0, offs.size))] = 0.0 if rng.random() < 0.5 else -0.0
        kinds.append((offs, vals))
    if seed % 2:
        which = rng.integers(0, nk, m)
    else:                                   # runs of one kind, as the planes of a stencil give them
        which = np.repeat(rng.integers(0, nk, m // 50 + 1), 50)[:m]
    rows, vals = [], []
    for r in range(m):
        offs, v = kinds[int(which[r])]
        c = r + offs
        keep = (c >= 0) & (c < n)           # clipping at the boundary makes further kinds: the dictionary holds them too
        rows.append(c[keep]); vals.append(v[keep])
    lens = np.array([c.size for c in rows])
    ai = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    aj = (np.concatenate(rows) if lens.sum() else np.zeros(0)).astype(np.int32)
    aa = (np.concatenate(vals) if lens.sum() else np.zeros(0)).astype(np.float64)
    x = rng.standard_normal(n); y0 = rng.standard_normal(m); d = rng.standard_normal(m)
    distinct = {(tuple((aj[ai[r]:ai[r + 1]] - r).tolist()), aa[ai[r]:ai[r + 1]].tobytes()) for r in range(m)}
    slots = sum(1 + (len(o) + 7) // 8 * 8 for o, _ in distinct)
    dai, daj = dev.put(ai), dev.put(aj if aj.size else np.zeros(1, np.int32))
    daa = dev.put(np.full(max(aa.size, 1) + 2, np.nan))
    dx, dy = dev.put(x), dev.put(y0)
    dd = dev.put(d)
    plan = C.c_void_p()
    dev.chk(k.mi355x_spmv_plan_create(dev.h, m, ai.ctypes.data, None, C.byref(plan)))
    nv = C.c_int()
    aa_h = np.ascontiguousarray(aa if aa.size else np.zeros(1))
    dev.chk(k.mi355x_spmv_plan_value_patterns(dev.h, plan, ai.ctypes.data, aj.ctypes.data, aa_h.ctypes.data, C.byref(nv)))
    if slots > 512:
        assert nv.value == 0                # more table than there is: declined, the value array would be streamed
    else:
        assert nv.value == len(distinct)
        dz = dev.alloc(8 * m)
        dev.chk(k.mi355x_spmv_csr(dev.h, plan, dai, daj, daa, dx, dz))
        assert np.array_equal(bits(dev.get(dz, m)), bits(orc.spmv(ai, aj, aa, x)))
        dev.chk(k.mi355x_spmv_csr_add(dev.h, plan, dai, daj, daa, dx, dy, dz))
        assert np.array_equal(bits(dev.get(dz, m)), bits(orc.spmv_add(ai, aj, aa, x, y0)))
        dev.chk(k.mi355x_spmv_csr_scaled(dev.h, plan, dai, daj, daa, dx, dd, dz))
        assert np.array_equal(bits(dev.get(dz, m)), bits(orc.spmv(ai, aj, aa, x) * d))
        dev.chk(k.mi355x_spmv_plan_set_pairsum(plan, 1))
        dev.chk(k.mi355x_spmv_csr(dev.h, plan, dai, daj, daa, dx, dz))
        assert np.array_equal(bits(dev.get(dz, m)), bits(orc.spmv_inode(ai, aj, aa, x)))
        dev.free(dz)
    dev.chk(k.mi355x_spmv_plan_destroy(plan))
    for p in (dai, daj, daa, dx, dy, dd):
        dev.free(p)
