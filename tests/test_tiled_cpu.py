"""The column-tiled SpMV layout (csrc/spmv_tiled.hip, mi355x_spmv_tiled_build: host code, no device) read back and checked on the
CPU: every nonzero is in exactly one place (a staged chunk or the CSR remainder), a row's staged products are met in column order,
and the product the kernel's order gives agrees with the oracle (MatMult_SeqAIJ, aij.c:1225-1285)."""
import numpy as np
import pytest

import orc
import tiled


@pytest.fixture(scope="module")
def k(built):
    return built.load_kernels()


def random_csr(rng, m, n, lens, band, far_frac):
    rows, cols = [], []
    for r in range(m):
        ln = int(lens[r])
        c = np.where(rng.random(ln) < far_frac, rng.integers(0, n, ln), np.clip(r * n // max(m, 1) + rng.integers(-band, band + 1, ln), 0, n - 1))
        c = np.unique(c)
        rows.append(np.full(c.size, r)); cols.append(c)
    ai = np.concatenate(([0], np.cumsum([c.size for c in cols]))).astype(np.int32)
    aj = np.concatenate(cols).astype(np.int32) if cols else np.zeros(0, np.int32)
    aa = rng.standard_normal(aj.size)
    return ai, aj, aa


def check(k, ai, aj, aa, n, stage_min, expect_all_staged=False, expect_none_staged=False):
    m = ai.size - 1
    plan = tiled.build(k, ai, aj, n, stage_min)
    try:
        inf = tiled.info(k, plan)
        assert inf["staged"] + inf["remainder"] == aj.size
        (rows, cols, pos), far = tiled.walk(k, plan, m)
        fr, fj, fp = tiled.remainder(far, m)
        assert rows.size == inf["staged"] and fj.size == inf["remainder"]
        if expect_all_staged:
            assert inf["remainder"] == 0
        if expect_none_staged:
            assert inf["staged"] == 0
        # every CSR position exactly once, with its own row and column
        seen = np.zeros(aj.size, dtype=np.int64)
        np.add.at(seen, pos, 1)
        np.add.at(seen, fp, 1)
        assert np.all(seen == 1)
        rowof = np.repeat(np.arange(m), np.diff(ai))
        assert np.array_equal(rowof[pos], rows) and np.array_equal(aj[pos], cols)
        assert np.array_equal(aj[fp], fj) and np.array_equal(rowof[fp], fr)
        # a row's staged entries are met in ascending column order (= ascending CSR position)
        order = np.lexsort((np.arange(rows.size), rows))
        pr, pp = rows[order], pos[order]
        same = pr[1:] == pr[:-1]
        assert np.all(pp[1:][same] > pp[:-1][same])
        # the product in the kernel's order against the oracle: <= 1e-12 * sum |a x| per row; bit for bit when nothing is out of column order
        x = np.sin(0.37 * np.arange(n)) + 1.0
        y = tiled.apply(k, plan, m, aa, x)
        ref = orc.spmv(ai, aj, aa, x) if m else np.zeros(0)
        scale = np.zeros(m)
        np.add.at(scale, rowof, np.abs(aa * x[aj]))
        assert np.all(np.abs(y - ref) <= 1e-12 * scale + 1e-300)
        if inf["remainder"] == 0 or inf["staged"] == 0:
            assert np.array_equal(y.view(np.uint64), ref.view(np.uint64))
        return inf
    finally:
        k.mi355x_spmv_tiled_destroy(plan)


def test_tiled_layout_holds_every_nonzero_once_and_in_column_order(k):
    rng = np.random.default_rng(5)
    g = tiled.geometry(k)
    assert g["tw"] % 128 == 0 and g["block"] == 128 and g["panel"] < 65536
    m = g["panel"] + 700                               # a full panel and a ragged one
    n = 3 * g["tw"] + 1234                             # ragged last tile
    lens = np.clip(np.exp(rng.normal(3.0, 0.7, m)), 0, 200).astype(int)
    lens[::17] = 0                                     # empty rows
    ai, aj, aa = random_csr(rng, m, n, lens, band=3000, far_frac=0.2)
    inf = check(k, ai, aj, aa, n, stage_min=4000)
    assert inf["staged"] > 0 and inf["remainder"] > 0 and inf["panels"] == 2
    check(k, ai, aj, aa, n, stage_min=1, expect_all_staged=True)        # every pair staged: one stream, the reference's order
    check(k, ai, aj, aa, n, stage_min=10 ** 9, expect_none_staged=True)  # nothing worth staging: all of it remainder


def test_tiled_layout_rows_longer_than_a_chunk_and_degenerate_shapes(k):
    rng = np.random.default_rng(6)
    g = tiled.geometry(k)
    # rows with many entries in one tile beside short ones: a wavefront's range of the panel may be one row, or none
    m, n = 300, g["tw"] + 500
    lens = np.full(m, 5); lens[7] = 1800; lens[8] = 700; lens[130] = 513; lens[299] = 3000
    ai, aj, aa = random_csr(rng, m, n, lens, band=n, far_frac=0.0)
    inf = check(k, ai, aj, aa, n, stage_min=1, expect_all_staged=True)
    assert inf["blocks"] * 128 >= inf["staged"] > 3000
    # one row, one column; no rows at all; a matrix without entries
    check(k, np.array([0, 1], np.int32), np.array([0], np.int32), np.array([2.5]), 1, stage_min=1, expect_all_staged=True)
    check(k, np.array([0], np.int32), np.zeros(0, np.int32), np.zeros(0), 10, stage_min=1)
    check(k, np.zeros(50, np.int32), np.zeros(0, np.int32), np.zeros(0), 10, stage_min=1)
    # a column count that is an exact multiple of the tile width, rows an exact multiple of the panel
    m, n = g["panel"], 2 * g["tw"]
    ai, aj, aa = random_csr(rng, m, n, np.full(m, 6), band=2000, far_frac=0.1)
    check(k, ai, aj, aa, n, stage_min=32)


def test_tiled_probe_tells_scattered_gathers_from_shared_ones(k):
    import ctypes as C
    rng = np.random.default_rng(8)
    m = n = 40000
    ai, aj, aa = random_csr(rng, m, n, np.full(m, 40), band=15000, far_frac=0.2)
    v = C.c_double()
    k.mi355x_spmv_tiled_probe(m, ai.ctypes.data, aj.ctypes.data, C.byref(v))
    assert v.value > 0.6                                # nearly every gather its own line of x
    # a 7-point stencil: neighbouring rows share their lines
    import petsc_dev_amd.petsc as P
    ai, aj, aa = P.gen_poisson7(32, 32, 32)
    k.mi355x_spmv_tiled_probe(ai.size - 1, ai.ctypes.data, aj.ctypes.data, C.byref(v))
    assert v.value < 0.2


def test_tiled_panels_hold_equal_shares_of_the_nonzeros(k, monkeypatch):
    """mi355x_spmv_tiled_build's row panels: never more rows than the geometry allows, the smallest bound on a panel's nonzeros that fits
    the panel count (greedy cuts, bisection), a multiple of 256 panels when the rows do not fit 256, the count forced by
    MI355X_TILED_PANELS for experiments; the wavefronts' row ranges inside a panel are equal shares too."""
    rng = np.random.default_rng(21)
    g = tiled.geometry(k)
    m = 3 * g["panel"] + 1000
    lens = rng.integers(3, 9, m)
    lens[: m // 5] = 1                                   # a sparse head: its panels hit the row cap, the others make up for it
    ai = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    n = 4 * g["tw"]
    off = np.arange(ai[-1]) - np.repeat(ai[:-1], lens)
    aj = ((np.repeat(np.arange(m), lens) // 7 + off * 5) % n).astype(np.int32)
    aj = np.concatenate([np.sort(aj[ai[r]:ai[r + 1]]) for r in range(m)]).astype(np.int32)
    for r in range(m):                                   # distinct columns inside a row (offsets of 5 are distinct below n)
        assert np.all(np.diff(aj[ai[r]:ai[r + 1]]) > 0)
    for forced in (None, 9, 40):
        if forced:
            monkeypatch.setenv("MI355X_TILED_PANELS", str(forced))
        plan = tiled.build(k, ai, aj, n, 10 ** 9)
        try:
            prow = tiled.get(k, plan, 11, np.int32)
            wrow = tiled.get(k, plan, 10, np.int32).reshape(-1, g["waves"] + 1)
            np_ = prow.size - 1
            assert prow[0] == 0 and prow[-1] == m and np.all(np.diff(prow) >= 1) and np.all(np.diff(prow) <= g["panel"])
            if forced:
                assert np_ == forced
            else:
                assert np_ == 4                            # ceil(m / panel): fewer than 256, nothing to round
            nz = np.diff(ai[prow]).astype(np.int64)
            capped = np.diff(prow) == g["panel"]
            # no panel holds more than the bound; a smaller bound would need more panels: check by re-cutting greedily with max(nz) - 1
            bound = nz.max() - 1
            cnt, r = 0, 0
            while r < m:
                e = int(np.searchsorted(ai, ai[r] + bound, side="right")) - 1
                e = min(max(e, r + 1), r + g["panel"], m)
                r = e; cnt += 1
            assert cnt > np_ or np.all(capped | (nz <= bound)), "the bound is the smallest that fits"
            assert np.all(wrow[:, 0] == prow[:-1]) and np.all(wrow[:, -1] == prow[1:]) and np.all(np.diff(wrow, axis=1) >= 0)
            wnz = np.diff(ai[wrow], axis=1)
            assert np.all(wnz.max(axis=1) <= nz / g["waves"] + lens.max() + 1), "a wavefront's share exceeds the mean by less than one row"
        finally:
            k.mi355x_spmv_tiled_destroy(plan)
