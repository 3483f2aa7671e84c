"""Reading the column-tiled SpMV layout (csrc/spmv_tiled.hip) back on the host: the tests' restatement of what the kernel does with
the arrays the builder leaves -- which entries a (wavefront, tile) stream holds, which lane meets them at which step, which row they
are added into and in which order.  Test infrastructure: numpy / Python loops over small matrices."""
import ctypes as C

import numpy as np


def geometry(k):
    g = [C.c_int() for _ in range(6)]
    k.mi355x_spmv_tiled_geometry(*[C.byref(v) for v in g])
    return dict(zip(("panel", "tw", "waves", "rounds", "group", "trip"), [v.value for v in g]))


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
    return {"staged": a.value, "remainder": b.value, "panels": p.value, "pairs": q.value, "steps": s.value}


NEWTILE, ROUNDEND = 0x8000, 0x4000


def walk(k, plan, m):
    """Every staged entry as (row, column, position in the CSR value array), panel after panel, inside a panel tile after tile, inside
    a tile in storage order -- for one row that is the order its lane adds the products in -- plus the remainder's CSR (far_i, far_j,
    far_perm).  A wavefront's stream runs through all staged tiles of its panel: its step words and entries are contiguous, a tile's
    steps padded to whole groups (at least one), the first word of a tile's first group flagged.  Checks the jagged-diagonal
    invariants on the way: per (panel, tile) the rows are sorted by count over ALL rounds (round g = a * waves + w is wavefront w's
    round a), so a step's active lanes are 0 .. n - 1, and every row of the panel appears at most once per tile."""
    g = geometry(k)
    W, R, U = g["waves"], g["rounds"], g["group"]
    pt_ptr, pt_tile, pw_e0 = get(k, plan, 0, np.int32), get(k, plan, 1, np.int32), get(k, plan, 2, np.int32)
    desc = get(k, plan, 3, np.uint32).reshape(-1, 64, R)
    perm, lcol = get(k, plan, 4, np.int32), get(k, plan, 5, np.uint16)
    stepw, pw_s0, prow = get(k, plan, 6, np.uint16), get(k, plan, 10, np.int32), get(k, plan, 11, np.int32)
    far = get(k, plan, 7, np.int32), get(k, plan, 8, np.int32), get(k, plan, 9, np.int32)
    assert prow[0] == 0 and prow[-1] == m and prow.size == pt_ptr.size and np.all(np.diff(prow) > 0) and np.all(np.diff(prow) <= g["panel"])
    assert pw_e0.size == (prow.size - 1) * W + 1 and pw_s0.size == pw_e0.size
    rows, cols, pos = [], [], []
    steps = 0
    for p in range(pt_ptr.size - 1):
        ntp = int(pt_ptr[p + 1] - pt_ptr[p])
        nrow = int(prow[p + 1] - prow[p])
        off = [int(pw_e0[p * W + w]) for w in range(W)]           # each wavefront's cursor in its own stream
        sw = [int(pw_s0[p * W + w]) for w in range(W)]
        assert all(v % g["trip"] == 0 for v in sw), "a wavefront's step words start on a trip boundary"
        last_tile = -1
        for i in range(ntp):
            t = int(pt_tile[pt_ptr[p] + i])
            assert t > last_tile, "a panel's staged tiles ascend"
            last_tile = t
            seen_rows = set()
            ranked = np.zeros(W * R * 64, dtype=np.int64)
            for w in range(W):
                d = desc[pt_ptr[p] * W + w * ntp + i]
                first = True
                for a in range(R):
                    cnt = (d[:, a] & 0xffff).astype(np.int64)
                    rl = (d[:, a] >> 16).astype(np.int64)
                    ranked[(a * W + w) * 64:(a * W + w) * 64 + 64] = cnt
                    assert a == 0 or cnt[0] == 0 or (d[0, a - 1] & 0xffff) > 0, "a wavefront's empty rounds come last"
                    for l in range(64):
                        if cnt[l]:
                            assert int(rl[l]) not in seen_rows and rl[l] < nrow
                            seen_rows.add(int(rl[l]))
                    for j in range(int(cnt[0])):
                        nact = int(np.sum(cnt > j))
                        assert stepw[sw[w]] == (nact | (ROUNDEND if j + 1 == int(cnt[0]) else 0) | (NEWTILE if first else 0)), \
                            "the step word the kernel reads: active lanes, last step of the round, tile switch"
                        first = False
                        sw[w] += 1
                        steps += 1
                        for l in range(nact):
                            rows.append(int(prow[p]) + int(rl[l]))
                            cols.append(t * g["tw"] + int(lcol[off[w] + l]))
                            pos.append(int(perm[off[w] + l]))
                        off[w] += nact
                if first:                                         # nothing of this tile for this wavefront: one group of padding, flagged
                    assert stepw[sw[w]] == NEWTILE
                    sw[w] += 1
                npad = (-sw[w]) % U
                assert np.all(stepw[sw[w]:sw[w] + npad] == 0), "padding words to the next group boundary"
                sw[w] += npad
            assert np.all(ranked[:-1] >= ranked[1:]), "the panel's rows sorted by count across the rounds"
        for w in range(W):
            end = int(pw_s0[p * W + w + 1])
            assert off[w] == pw_e0[p * W + w + 1] and end % g["trip"] == 0 and 0 <= end - sw[w] < g["trip"] and np.all(stepw[sw[w]:end] == 0), \
                "a wavefront's stream ends on a whole trip of the kernel's loop, padded with empty steps"
    assert steps == info(k, plan)["steps"]
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
