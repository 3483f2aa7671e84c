"""Reading the column-tiled SpMV layout (csrc/spmv_tiled.hip) back on the host: the tests' restatement of what the kernel does with
the arrays the builder leaves -- which entries a (wavefront, tile) stream holds, which lane meets them at which step, which row they
are added into and in which order.  Test infrastructure: numpy / Python loops over small matrices."""
import ctypes as C

import numpy as np


def geometry(k):
    g = [C.c_int() for _ in range(4)]
    k.mi355x_spmv_tiled_geometry(*[C.byref(v) for v in g])
    return dict(zip(("panel", "tw", "waves", "block"), [v.value for v in g]))


def build(k, ai, aj, n, stage_min):
    ai = np.ascontiguousarray(ai, dtype=np.int32)
    aj = np.ascontiguousarray(aj, dtype=np.int32)
    plan = C.c_void_p()
    rc = k.mi355x_spmv_tiled_build(ai.size - 1, n, ai.ctypes.data, aj.ctypes.data, stage_min, C.byref(plan))
    assert rc == 0, rc
    return plan


def get(k, plan, which, dtype):
    nb = C.c_size_t()
    assert k.mi355x_spmv_tiled_debug_get(plan, which, None, 0, C.byref(nb)) == 0
    out = np.zeros(nb.value // np.dtype(dtype).itemsize, dtype=dtype)
    if nb.value:
        assert k.mi355x_spmv_tiled_debug_get(plan, which, out.ctypes.data, nb.value, C.byref(nb)) == 0
    return out


def info(k, plan):
    a, b, s = C.c_long(), C.c_long(), C.c_long()
    p, q = C.c_int(), C.c_int()
    k.mi355x_spmv_tiled_info(plan, C.byref(a), C.byref(b), C.byref(p), C.byref(q), C.byref(s))
    return {"staged": a.value, "remainder": b.value, "panels": p.value, "pairs": q.value, "blocks": s.value}


NEWTILE = 0x8000


def walk(k, plan, m):
    """Every staged entry as (row, column, position in the CSR value array), panel after panel, inside a panel tile after tile, inside
    a tile wavefront after wavefront, inside a wavefront's blocks in the order the kernel adds them (entry q of a block is stored at
    2 (q mod 64) + q div 64: the first ds_add_f64 of a block takes entries 0..63 in lane order, the second 64..127) -- for one row
    that is the order its products are added in -- plus the remainder's CSR (far_i, far_j, far_perm).  Checks on the way: a wavefront's
    stream holds one flagged run of blocks per staged tile of its panel, its rows stay inside its range of the panel, a (wavefront,
    tile)'s entries are in CSR order and only its last block is padded (value position -1, the spare accumulator's row)."""
    g = geometry(k)
    W, B = g["waves"], g["block"]
    pt_ptr, pt_tile, pw_e0 = get(k, plan, 0, np.int32), get(k, plan, 1, np.int32), get(k, plan, 2, np.int32)
    word, perm = get(k, plan, 3, np.uint32), get(k, plan, 4, np.int32)
    wrow, prow = get(k, plan, 10, np.int32).reshape(-1, W + 1), get(k, plan, 11, np.int32)
    far = get(k, plan, 7, np.int32), get(k, plan, 8, np.int32), get(k, plan, 9, np.int32)
    assert prow[0] == 0 and prow[-1] == m and prow.size == pt_ptr.size and np.all(np.diff(prow) > 0) and np.all(np.diff(prow) <= g["panel"])
    assert pw_e0.size == (prow.size - 1) * W + 1 and wrow.shape[0] == prow.size - 1 and word.size == perm.size == pw_e0[-1]
    assert np.all(pw_e0 % B == 0)
    slot = np.array([2 * (q % 64) + q // 64 for q in range(B)])
    rows, cols, pos = [], [], []
    nblocks = 0
    for p in range(pt_ptr.size - 1):
        ntp = int(pt_ptr[p + 1] - pt_ptr[p])
        assert wrow[p, 0] == prow[p] and wrow[p, W] == prow[p + 1] and np.all(np.diff(wrow[p]) >= 0)
        runs = []                                                   # per wavefront: its blocks' starts, one run per tile
        for w in range(W):
            starts = np.arange(pw_e0[p * W + w], pw_e0[p * W + w + 1], B)
            flagged = [int(b) for b in starts if word[b] & NEWTILE]
            assert len(flagged) == ntp and (ntp == 0 or flagged[0] == starts[0]), "one flagged block per staged tile, the stream opens with one"
            ends = flagged[1:] + [int(pw_e0[p * W + w + 1])]
            runs.append([np.arange(f, e, B) for f, e in zip(flagged, ends)])
            nblocks += starts.size
        last_tile = -1
        for i in range(ntp):
            t = int(pt_tile[pt_ptr[p] + i])
            assert t > last_tile, "a panel's staged tiles ascend"
            last_tile = t
            for w in range(W):
                prev = (-1, -1)
                done = False
                for b0 in runs[w][i]:
                    for q in range(B):
                        s = int(b0 + slot[q])
                        wd = int(word[s])
                        if q:
                            assert not wd & NEWTILE, "only a block's first stored word carries the flag"
                        if perm[s] < 0:
                            assert (wd >> 16) == g["panel"] and (wd & 0x7fff) == 0, "padding goes to the spare accumulator"
                            done = True
                            continue
                        assert not done and b0 == runs[w][i][-1] or not done, "padding only at the end of a (wavefront, tile)'s last block"
                        r, c = int(prow[p]) + (wd >> 16), t * g["tw"] + (wd & 0x7fff)
                        assert wrow[p, w] <= r < wrow[p, w + 1] and (wd & 0x7fff) < g["tw"]
                        assert (r, c) > prev, "CSR order inside a (wavefront, tile)"
                        prev = (r, c)
                        rows.append(r); cols.append(c); pos.append(int(perm[s]))
    assert nblocks == info(k, plan)["blocks"]
    return (np.array(rows, dtype=np.int64), np.array(cols, dtype=np.int64), np.array(pos, dtype=np.int64)), far


def remainder(far, m):
    """the remainder's entries as (row, column, position in the CSR value array), pass after pass (the remainder is cut into column
    ranges applied one after the other), rows in order inside a pass: the order its products reach a row's sum"""
    fi, fj, fp = far
    npass = fi.size // (m + 1) if m + 1 else 0
    rows, cols, pos = [], [], []
    lastcol = {}
    for q in range(npass):
        base = q * (m + 1)
        assert fi[base] % 2 == 0, "a pass starts on an even entry"
        for r in range(m):
            for kk in range(fi[base + r], fi[base + r + 1]):
                assert fp[kk] >= 0
                assert lastcol.get(r, -1) < fj[kk], "a row's remainder entries ascend in column, pass after pass"
                lastcol[r] = int(fj[kk])
                rows.append(r); cols.append(int(fj[kk])); pos.append(int(fp[kk]))
    return np.array(rows, dtype=np.int64), np.array(cols, dtype=np.int64), np.array(pos, dtype=np.int64)


def apply(k, plan, m, aa, x, yin=None):
    """y = (yin or 0) + A x from the layout, in the kernel's order: a row's staged products in stream order, then its remainder pass after pass"""
    (rows, cols, pos), far = walk(k, plan, m)
    y = np.zeros(m) if yin is None else yin.astype(np.float64).copy()
    for r, c, q in zip(rows, cols, pos):
        y[r] = y[r] + aa[q] * x[c]
    fr, fc, fq = remainder(far, m)
    for r, c, q in zip(fr, fc, fq):
        y[r] = y[r] + aa[q] * x[c]
    return y
