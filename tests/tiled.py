"""Reading the column-tiled SpMV layout (csrc/spmv_tiled.hip) back on the host: the tests' restatement of what the kernel does with
the arrays the builder leaves -- which entries a chunk holds, in which lane's registers, which row they are added into and in which
order.  Test infrastructure: numpy / Python loops over small matrices."""
import ctypes as C

import numpy as np


def geometry(k):
    g = [C.c_int() for _ in range(5)]
    k.mi355x_spmv_tiled_geometry(*[C.byref(v) for v in g])
    return dict(zip(("panel", "tw", "waves", "rpl", "ch"), [v.value for v in g]))


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
    a, b = C.c_long(), C.c_long()
    p, q, c = C.c_int(), C.c_int(), C.c_int()
    k.mi355x_spmv_tiled_info(plan, C.byref(a), C.byref(b), C.byref(p), C.byref(q), C.byref(c))
    return {"staged": a.value, "remainder": b.value, "panels": p.value, "pairs": q.value, "chunks": c.value}


def walk(k, plan, m):
    """Every staged entry as (row, column, position in the CSR value array), in the order the kernel adds a row's products, plus the
    remainder's CSR (far_i, far_j, far_perm)."""
    g = geometry(k)
    sub = 64 * g["rpl"]
    pt_ptr, pt_tile, pt_chunk0 = get(k, plan, 0, np.int32), get(k, plan, 1, np.int32), get(k, plan, 2, np.int32)
    chunk_e0, perm = get(k, plan, 3, np.int32), get(k, plan, 4, np.int32)
    lcol = get(k, plan, 5, np.uint16).reshape(-1, 64, 8)
    cend = get(k, plan, 6, np.uint16).reshape(-1, 64, g["rpl"])
    far = get(k, plan, 7, np.int32), get(k, plan, 8, np.int32), get(k, plan, 9, np.int32)
    rows, cols, pos = [], [], []
    npanels = pt_ptr.size - 1
    for p in range(npanels):
        last_tile = -1
        for pt in range(pt_ptr[p], pt_ptr[p + 1]):
            t = int(pt_tile[pt])
            assert t > last_tile, "a panel's staged tiles ascend"
            last_tile = t
            for w in range(g["waves"]):
                for c in range(pt_chunk0[pt * g["waves"] + w], pt_chunk0[pt * g["waves"] + w + 1]):
                    e0 = int(chunk_e0[c])
                    assert e0 % 2 == 0
                    start = 0
                    for rl in range(sub):
                        end = int(cend[c, rl & 63, rl >> 6])
                        assert start <= end <= g["ch"], (start, end)
                        for kk in range(start, end):
                            pi = kk >> 1
                            lc = int(lcol[c, pi & 63, 2 * (pi >> 6) + (kk & 1)])
                            rows.append(p * g["panel"] + w * sub + rl)
                            cols.append(t * g["tw"] + lc)
                            pos.append(int(perm[e0 + kk]))
                        start = end
                    ne = start
                    assert ne > 0 and chunk_e0[c + 1] - e0 == ne + (ne & 1)
                    if ne & 1:
                        assert perm[e0 + ne] == -1                      # padding to an even count
    return (np.array(rows, dtype=np.int64), np.array(cols, dtype=np.int64), np.array(pos, dtype=np.int64)), far


def apply(k, plan, m, aa, x, yin=None):
    """y = (yin or 0) + A x from the layout, in the kernel's order: a row's staged products in stream order, then its remainder in CSR order"""
    (rows, cols, pos), (fi, fj, fp) = walk(k, plan, m)
    y = np.zeros(m) if yin is None else yin.astype(np.float64).copy()
    for r, c, q in zip(rows, cols, pos):
        y[r] = y[r] + aa[q] * x[c]
    for r in range(m):
        for kk in range(fi[r], fi[r + 1]):
            y[r] = y[r] + aa[fp[kk]] * x[fj[kk]]
    return y
