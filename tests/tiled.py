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


NEWTILE, NEWWIN, FW_SHIFT = 0x8000, 0x80000000, 18


def walk(k, plan, m):
    """Every stored entry as (row, column, position in the CSR value array) in two lists: the STAGED ones -- panel after panel, inside a
    panel tile after tile, inside a tile wavefront after wavefront, inside a wavefront's blocks in the order the kernel adds them
    (entry q of a block is stored at 2 (q mod 64) + q div 64: the first ds_add_f64 of a block takes entries 0..63 in lane order, the
    second 64..127) -- and the REMAINDER -- panel after panel, wavefront after wavefront, its windows of 2^18 columns ascending, block
    order inside.  For one row the staged list followed by the remainder list is the order its products are added in.  Checks on the
    way: a wavefront's stream holds one flagged run of blocks per staged tile of its panel, then one flagged run per window it has
    remainder entries in; its rows stay inside its range of the panel; a run's entries are in CSR order and only its last block is
    padded (value position -1, the spare accumulator's row)."""
    g = geometry(k)
    W, B = g["waves"], g["block"]
    pt_ptr, pt_tile, pw_e0 = get(k, plan, 0, np.int32), get(k, plan, 1, np.int32), get(k, plan, 2, np.int32)
    word, perm = get(k, plan, 3, np.uint32), get(k, plan, 4, np.int32)
    pw_f0, fw_ptr, fw_win = get(k, plan, 7, np.int32), get(k, plan, 8, np.int32), get(k, plan, 9, np.int32)
    wrow, prow = get(k, plan, 10, np.int32).reshape(-1, W + 1), get(k, plan, 11, np.int32)
    assert prow[0] == 0 and prow[-1] == m and prow.size == pt_ptr.size and np.all(np.diff(prow) > 0) and np.all(np.diff(prow) <= g["panel"])
    npw = (prow.size - 1) * W
    assert pw_e0.size == npw + 1 and pw_f0.size == npw and fw_ptr.size == npw + 1 and wrow.shape[0] == prow.size - 1 and word.size == perm.size == pw_e0[-1]
    assert np.all(pw_e0 % B == 0) and np.all(pw_f0 % B == 0) and np.all(pw_e0[:-1] <= pw_f0) and np.all(pw_f0 <= pw_e0[1:])
    slot = np.array([2 * (q % 64) + q // 64 for q in range(B)])
    near, far = ([], [], []), ([], [], [])
    nblocks = 0

    def run_entries(p, w, blocks, col0, shift, colmask, flag, out):
        prev = (-1, -1)
        done = False
        for b0 in blocks:
            for q in range(B):
                s = int(b0 + slot[q])
                wd = int(word[s])
                if q:
                    assert not wd & flag, "only a block's first stored word carries the flag"
                wd &= ~flag
                if perm[s] < 0:
                    assert (wd >> shift) == g["panel"] and (wd & colmask) == 0, "padding goes to the spare accumulator"
                    done = True
                    continue
                assert not done, "padding only at the end of a run's last block"
                r, c = int(prow[p]) + (wd >> shift), col0 + (wd & colmask)
                assert wrow[p, w] <= r < wrow[p, w + 1]
                assert (r, c) > prev, "CSR order inside a run"
                prev = (r, c)
                out[0].append(r); out[1].append(c); out[2].append(int(perm[s]))

    for p in range(pt_ptr.size - 1):
        ntp = int(pt_ptr[p + 1] - pt_ptr[p])
        assert wrow[p, 0] == prow[p] and wrow[p, W] == prow[p + 1] and np.all(np.diff(wrow[p]) >= 0)
        runs = []                                                   # per wavefront: its staged blocks' starts, one run per tile
        for w in range(W):
            ipw = p * W + w
            starts = np.arange(pw_e0[ipw], pw_f0[ipw], B)
            flagged = [int(b) for b in starts if word[b] & NEWTILE]
            assert len(flagged) == ntp and (ntp == 0 or flagged[0] == starts[0]), "one flagged block per staged tile, the stream opens with one"
            ends = flagged[1:] + [int(pw_f0[ipw])]
            runs.append([np.arange(f, e, B) for f, e in zip(flagged, ends)])
            nblocks += starts.size
        last_tile = -1
        for i in range(ntp):
            t = int(pt_tile[pt_ptr[p] + i])
            assert t > last_tile, "a panel's staged tiles ascend"
            last_tile = t
            for w in range(W):
                run_entries(p, w, runs[w][i], t * g["tw"], 16, 0x7fff, NEWTILE, near)
        for w in range(W):
            ipw = p * W + w
            starts = np.arange(pw_f0[ipw], pw_e0[ipw + 1], B)
            nblocks += starts.size
            wins = fw_win[fw_ptr[ipw]:fw_ptr[ipw + 1]]
            assert np.all(np.diff(wins) > 0), "a wavefront's windows ascend"
            flagged = [int(b) for b in starts if word[b] & NEWWIN]
            assert len(flagged) == wins.size and (wins.size == 0) == (starts.size == 0) and (wins.size == 0 or flagged[0] == starts[0])
            ends = flagged[1:] + [int(pw_e0[ipw + 1])]
            for f, e, win in zip(flagged, ends, wins):
                n0 = len(far[0])
                run_entries(p, w, np.arange(f, e, B), int(win) << FW_SHIFT, FW_SHIFT, (1 << FW_SHIFT) - 1, NEWWIN, far)
                assert len(far[0]) > n0 and all((c >> FW_SHIFT) == win for c in far[1][n0:]), "a listed window holds entries, all of them its own"
    assert nblocks == info(k, plan)["blocks"]
    arr = lambda t: tuple(np.array(v, dtype=np.int64) for v in t)
    return arr(near), arr(far)


def remainder(far, m):
    """the remainder's entries as (row, column, position in the CSR value array) in the order their products reach the rows' sums"""
    fr, fc, fp = far
    lastcol = {}
    for r, c in zip(fr, fc):
        assert lastcol.get(int(r), -1) < c, "a row's remainder entries ascend in column"
        lastcol[int(r)] = int(c)
    return fr, fc, fp


def apply(k, plan, m, aa, x, yin=None):
    """y = (yin or 0) + A x from the layout, in the kernel's order: a row's staged products in stream order, then its remainder's,
    window after window -- every product added to the row's sum one after the other (the kernel's ds_add_f64 sequence)"""
    (rows, cols, pos), far = walk(k, plan, m)
    y = np.zeros(m) if yin is None else yin.astype(np.float64).copy()
    for r, c, q in zip(rows, cols, pos):
        y[r] = y[r] + aa[q] * x[c]
    fr, fc, fq = remainder(far, m)
    for r, c, q in zip(fr, fc, fq):
        y[r] = y[r] + aa[q] * x[c]
    return y
