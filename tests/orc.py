"""ctypes/numpy front end of oracle/liboracle.so (the CPU restatement; test infrastructure only)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_path = os.path.join(ROOT, "oracle", "liboracle.so")
_lib = None

pd = C.POINTER(C.c_double)
pi = C.POINTER(C.c_int)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_path):
            import subprocess
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
        _lib = C.CDLL(_path)
        _lib.orc_vec_dot.restype = C.c_double
        _lib.orc_gen_p7.restype = C.c_long
        _lib.orc_gen_p7.argtypes = [C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


def D(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(pd)


def I(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(pi)


def ptrs(vecs):
    arr = (pd * len(vecs))(*[D(v) for v in vecs])
    return arr


def f64(x):
    return C.c_double(float(x))


def n_(a):
    return C.c_size_t(a.size)


# ---- Vec ----
def vec_set(x, alpha): lib().orc_vec_set(n_(x), f64(alpha), D(x))
def vec_copy(x, y): lib().orc_vec_copy(n_(x), D(x), D(y))
def vec_scale(x, alpha): lib().orc_vec_scale(n_(x), f64(alpha), D(x))
def vec_swap(x, y): lib().orc_vec_swap(n_(x), D(x), D(y))
def vec_axpy(y, alpha, x): lib().orc_vec_axpy(n_(x), f64(alpha), D(x), D(y))
def vec_aypx(y, alpha, x): lib().orc_vec_aypx(n_(x), f64(alpha), D(x), D(y))
def vec_axpby(y, alpha, beta, x): lib().orc_vec_axpby(n_(x), f64(alpha), f64(beta), D(x), D(y))
def vec_waxpy(w, alpha, x, y): lib().orc_vec_waxpy(n_(x), f64(alpha), D(x), D(y), D(w))
def vec_axpbypcz(z, a, b, g, x, y): lib().orc_vec_axpbypcz(n_(x), f64(a), f64(b), f64(g), D(x), D(y), D(z))
def vec_pointwise_mult(w, x, y): lib().orc_vec_pointwise_mult(n_(x), D(x), D(y), D(w))
def vec_pointwise_divide(w, x, y): lib().orc_vec_pointwise_divide(n_(x), D(x), D(y), D(w))
def vec_reciprocal(x): lib().orc_vec_reciprocal(n_(x), D(x))


def vec_maxpy(x, alpha, ys):
    alpha = np.ascontiguousarray(alpha, dtype=np.float64)
    lib().orc_vec_maxpy(n_(x), C.c_int(len(ys)), D(alpha), ptrs(ys), D(x))


def vec_dot(x, y):
    return lib().orc_vec_dot(n_(x), D(x), D(y))


def vec_mdot(x, ys):
    z = np.zeros(len(ys))
    lib().orc_vec_mdot(n_(x), C.c_int(len(ys)), D(x), ptrs(ys), D(z))
    return z


def vec_norm(x, ntype):
    out = np.zeros(2)
    lib().orc_vec_norm(n_(x), C.c_int(ntype), D(x), D(out))
    return out if ntype == 4 else out[0]


def vec_dotnorm2(s, t):
    dp = C.c_double()
    nm = C.c_double()
    lib().orc_vec_dotnorm2(n_(s), D(s), D(t), C.byref(dp), C.byref(nm))
    return dp.value, nm.value


# ---- Mat ----
def spmv(ai, aj, aa, x):
    m = ai.size - 1
    y = np.zeros(m)
    lib().orc_spmv_csr(C.c_int(m), I(ai), I(aj), D(aa), D(x), D(y))
    return y


def check_inode(ai, aj, limit=5):
    m = ai.size - 1
    ns = np.zeros(m + 1, dtype=np.int32)
    nc = lib().orc_check_inode(C.c_int(m), I(ai), I(aj), C.c_int(limit), I(ns))
    return nc, ns[:nc].copy()


def spmv_inode(ai, aj, aa, x):
    m = ai.size - 1
    y = np.zeros(m)
    lib().orc_spmv_csr_inode(C.c_int(m), I(ai), I(aj), D(aa), D(x), D(y))
    return y


def matmult(ai, aj, aa, x, z=None):
    """MatMult (z None) / MatMultAdd as the reference DISPATCHES them for a seqaij matrix: the inode routines when
    Mat_CheckInode keeps them, the plain loops otherwise.  Returns (y, number of nodes or 0)."""
    m = ai.size - 1
    y = np.zeros(m)
    ns = np.zeros(m + 1, dtype=np.int32)
    lib().orc_matmult_seqaij.restype = C.c_int
    nodes = lib().orc_matmult_seqaij(C.c_int(m), I(ai), I(aj), D(aa), D(x), D(z) if z is not None else None, D(y), I(ns))
    return y, nodes


class device_reduction_order:
    """with orc.device_reduction_order(): the oracle's dots / norms use the HIP reduction tree (same terms, same order), so a
    whole solve can be compared with the HIP path bit for bit"""
    def __enter__(self):
        lib().orc_set_device_reduction_order(C.c_int(1))

    def __exit__(self, *a):
        lib().orc_set_device_reduction_order(C.c_int(0))


def pbjacobi_setup(bs, bi, bj, ba):
    """inverted diagonal blocks (MatInvertBlockDiagonal_SeqBAIJ), column-major, mbs x bs x bs"""
    mbs = bi.size - 1
    idiag = np.zeros(mbs * bs * bs)
    lib().orc_bsr_invert_block_diagonal.restype = C.c_int
    rc = lib().orc_bsr_invert_block_diagonal(C.c_int(mbs), C.c_int(bs), I(bi), I(bj), D(ba), D(idiag))
    assert rc == 0, rc
    return idiag


def pbjacobi_apply(bs, idiag, x):
    mbs = idiag.size // (bs * bs)
    y = np.zeros(mbs * bs)
    lib().orc_pbjacobi_apply(C.c_int(mbs), C.c_int(bs), D(idiag), D(x), D(y))
    return y


def spmv_add(ai, aj, aa, x, y):
    m = ai.size - 1
    z = np.zeros(m)
    lib().orc_spmv_csr_add(C.c_int(m), I(ai), I(aj), D(aa), D(x), D(y), D(z))
    return z


def spmv_t(ai, aj, aa, x, n):
    m = ai.size - 1
    y = np.zeros(n)
    lib().orc_spmv_csr_transpose(C.c_int(m), C.c_int(n), I(ai), I(aj), D(aa), D(x), D(y))
    return y


def spmv_t_add(ai, aj, aa, x, z, n):
    m = ai.size - 1
    y = np.zeros(n)
    lib().orc_spmv_csr_transpose_add(C.c_int(m), C.c_int(n), I(ai), I(aj), D(aa), D(x), D(z), D(y))
    return y


def diagonal_scale(ai, aj, aa, l=None, r=None):
    """returns the values of diag(l) A diag(r)"""
    out = np.array(aa, dtype=np.float64)
    lib().orc_csr_diagonal_scale(C.c_int(ai.size - 1), I(ai), I(aj), D(out), D(l) if l is not None else None, D(r) if r is not None else None)
    return out


def get_diagonal(ai, aj, aa):
    m = ai.size - 1
    d = np.zeros(m)
    lib().orc_csr_get_diagonal(C.c_int(m), I(ai), I(aj), D(aa), D(d))
    return d


def csr_transpose(ai, aj, aa, n):
    m = ai.size - 1
    ti = np.zeros(n + 1, dtype=np.int32)
    tj = np.zeros(max(aj.size, 1), dtype=np.int32)
    ta = np.zeros(max(aj.size, 1))
    lib().orc_csr_transpose(C.c_int(m), C.c_int(n), I(ai), I(aj), D(aa), I(ti), I(tj), D(ta))
    return ti, tj[:aj.size], ta[:aj.size]


def spmv_bsr(bs, ai, aj, aa, x):
    mbs = ai.size - 1
    y = np.zeros(mbs * bs)
    lib().orc_spmv_bsr(C.c_int(mbs), C.c_int(bs), I(ai), I(aj), D(aa), D(x), D(y))
    return y


def gen_p7(nx, ny, nz, rstart=0, rend=None):
    if rend is None:
        rend = nx * ny * nz
    m = rend - rstart
    nnz = lib().orc_gen_p7(nx, ny, nz, rstart, rend, None, None, None)
    ai = np.zeros(m + 1, dtype=np.int32)
    aj = np.zeros(nnz, dtype=np.int32)
    aa = np.zeros(nnz)
    lib().orc_gen_p7(nx, ny, nz, rstart, rend, ai.ctypes.data, aj.ctypes.data, aa.ctypes.data)
    return ai, aj, aa


# ---- MPIAIJ set-up ----
def mpiaij_split(rstart, rend, cstart, cend, ai, aj, aa):
    mloc = rend - rstart
    nz = int(ai[rend] - ai[rstart])
    ad_i = np.zeros(mloc + 1, dtype=np.int32); bo_i = np.zeros(mloc + 1, dtype=np.int32)
    ad_j = np.zeros(max(nz, 1), dtype=np.int32); bo_j = np.zeros(max(nz, 1), dtype=np.int32)
    ad_a = np.zeros(max(nz, 1)); bo_a = np.zeros(max(nz, 1))
    garray = np.zeros(max(nz, 1), dtype=np.int32)
    ec = lib().orc_mpiaij_split(rstart, rend, cstart, cend, I(ai), I(aj), D(aa), I(ad_i), I(ad_j), D(ad_a),
                                I(bo_i), I(bo_j), D(bo_a), I(garray))
    na, nb = int(ad_i[-1]), int(bo_i[-1])
    return dict(ad_i=ad_i, ad_j=ad_j[:na].copy(), ad_a=ad_a[:na].copy(), bo_i=bo_i, bo_j=bo_j[:nb].copy(),
                bo_a=bo_a[:nb].copy(), garray=garray[:ec].copy())


def scatter_create(size, rank, ranges, garrays):
    ranges = np.ascontiguousarray(ranges, dtype=np.int32)
    ecs = np.array([g.size for g in garrays], dtype=np.int32)
    gs = [np.ascontiguousarray(g, dtype=np.int32) if g.size else np.zeros(1, dtype=np.int32) for g in garrays]
    gp = (pi * size)(*[I(g) for g in gs])
    tot = int(ecs.sum()) + 1
    rprocs = np.zeros(size + 1, dtype=np.int32); rstarts = np.zeros(size + 2, dtype=np.int32)
    rind = np.zeros(tot, dtype=np.int32)
    sprocs = np.zeros(size + 1, dtype=np.int32); sstarts = np.zeros(size + 2, dtype=np.int32)
    sind = np.zeros(tot, dtype=np.int32)
    lto = np.zeros(tot, dtype=np.int32); lfrom = np.zeros(tot, dtype=np.int32)
    nr = C.c_int(); ns = C.c_int(); nl = C.c_int()
    lib().orc_scatter_create(size, rank, I(ranges), gp, I(ecs), C.byref(nr), I(rprocs), I(rstarts), I(rind),
                             C.byref(ns), I(sprocs), I(sstarts), I(sind), C.byref(nl), I(lto), I(lfrom))
    nr, ns, nl = nr.value, ns.value, nl.value
    return dict(rprocs=rprocs[:nr].copy(), rstarts=rstarts[:nr + 1].copy(), rindices=rind[:rstarts[nr]].copy(),
                sprocs=sprocs[:ns].copy(), sstarts=sstarts[:ns + 1].copy(), sindices=sind[:sstarts[ns]].copy(),
                lto=lto[:nl].copy(), lfrom=lfrom[:nl].copy())


def ilu0_factor(ai, aj, aa):
    n = ai.size - 1
    nz = int(ai[-1])
    bi = np.zeros(n + 1, dtype=np.int32); bj = np.zeros(nz + 1, dtype=np.int32)
    bd = np.zeros(n + 1, dtype=np.int32); ba = np.zeros(nz + 1)
    rc = lib().orc_ilu0_factor(C.c_int(n), I(ai), I(aj), D(aa), I(bi), I(bj), I(bd), D(ba))
    assert rc == 0
    return bi, bj, bd, ba


def ilu0_factor_shift(ai, aj, aa):
    """(factor, restarts of the factorisation MatPivotCheck_nz asked for)"""
    n = ai.size - 1
    nz = int(ai[-1])
    bi = np.zeros(n + 1, dtype=np.int32); bj = np.zeros(nz + 1, dtype=np.int32)
    bd = np.zeros(n + 1, dtype=np.int32); ba = np.zeros(nz + 1)
    ns = C.c_int()
    rc = lib().orc_ilu0_factor_shift(C.c_int(n), I(ai), I(aj), D(aa), I(bi), I(bj), I(bd), D(ba), C.byref(ns))
    assert rc == 0
    return (bi, bj, bd, ba), ns.value


def ilu0_solve(f, b):
    bi, bj, bd, ba = f
    x = np.zeros(b.size)
    lib().orc_ilu0_solve(C.c_int(b.size), I(bi), I(bj), I(bd), D(ba), D(b), D(x))
    return x


def ilu0_solve_inode(f, ns, b):
    """MatSolve_SeqAIJ_Inode: the triangular solves the reference runs on the factor of a matrix whose rows form nodes `ns`"""
    bi, bj, bd, ba = f
    ns = np.ascontiguousarray(ns, dtype=np.int32)
    assert int(ns.sum()) == b.size
    x = np.zeros(b.size)
    lib().orc_ilu0_solve_inode(C.c_int(b.size), C.c_int(ns.size), I(ns), I(bi), I(bj), I(bd), D(ba), D(b), D(x))
    return x


def icc0_factor(ai, aj, aa):
    """(ui, uj, ua), number of positive-definite shifts the factorisation needed"""
    n = ai.size - 1
    nz = lib().orc_icc0_count(C.c_int(n), I(ai), I(aj))
    ui = np.zeros(n + 1, dtype=np.int32); uj = np.zeros(nz + 1, dtype=np.int32); ua = np.zeros(nz + 1)
    rc = lib().orc_icc0_factor(C.c_int(n), I(ai), I(aj), D(aa), I(ui), I(uj), D(ua))
    assert rc >= 0
    return (ui, uj, ua), rc


def icc0_solve(f, b):
    ui, uj, ua = f
    x = np.zeros(b.size)
    lib().orc_icc0_solve(C.c_int(b.size), I(ui), I(uj), D(ua), D(b), D(x))
    return x


# ---- KSP ----
class KspOpts(C.Structure):
    _fields_ = [("ksp_type", C.c_int), ("pc_type", C.c_int), ("rtol", C.c_double), ("abstol", C.c_double),
                ("dtol", C.c_double), ("max_it", C.c_int), ("restart", C.c_int), ("refine_always", C.c_int),
                ("guess_nonzero", C.c_int), ("nblocks", C.c_int), ("blk", pi), ("sub_ksp_type", C.c_int),
                ("sub_pc_type", C.c_int), ("sub_rtol", C.c_double), ("sub_abstol", C.c_double),
                ("sub_dtol", C.c_double), ("sub_max_it", C.c_int), ("cg_single", C.c_int), ("norm_type", C.c_int), ("pb_bs", C.c_int), ("pc_right", C.c_int),
                ("blk_ksp_type", pi), ("blk_pc_type", pi), ("blk_rtol", C.POINTER(C.c_double))]


KSP = dict(cg=0, gmres=1, bcgs=2, preonly=3, groppcg=4, pipecg=5)
PC = dict(none=0, jacobi=1, bjacobi=2, ilu=3, pbjacobi=4, icc=5)


def ksp_solve(ai, aj, aa, b, ksp="gmres", pc="none", x0=None, blocks=None, sub_ksp="preonly", sub_pc="jacobi", block_solvers=None, **kw):
    """block_solvers: [(ksp, pc, rtol), ...] one per block -- sub-solvers set block by block (PCBJacobiGetSubKSP)"""
    o = KspOpts()
    lib().orc_ksp_default_opts(C.byref(o))
    if block_solvers is not None:
        bk = np.array([KSP[t[0]] for t in block_solvers], dtype=np.int32)
        bp = np.array([PC[t[1]] for t in block_solvers], dtype=np.int32)
        br = np.array([t[2] for t in block_solvers], dtype=np.float64)
        o.blk_ksp_type = I(bk); o.blk_pc_type = I(bp); o.blk_rtol = br.ctypes.data_as(C.POINTER(C.c_double))
    o.ksp_type = KSP[ksp]; o.pc_type = PC[pc]
    o.sub_ksp_type = KSP[sub_ksp]; o.sub_pc_type = PC[sub_pc]
    for k, v in kw.items():
        setattr(o, k, v)
    n = ai.size - 1
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    if x0 is not None:
        o.guess_nonzero = 1
    blk = None
    if blocks is not None:
        blk = np.ascontiguousarray(blocks, dtype=np.int32)
        o.nblocks = blk.size - 1
        o.blk = I(blk)
    cap = 20000
    hist = np.zeros(cap)
    nh = C.c_int(); its = C.c_int(); reason = C.c_int()
    rc = lib().orc_ksp_solve(C.byref(o), C.c_int(n), I(ai), I(aj), D(aa), D(b), D(x), D(hist), C.c_int(cap),
                             C.byref(nh), C.byref(its), C.byref(reason))
    assert rc == 0
    return x, hist[:min(nh.value, cap)].copy(), its.value, reason.value


def cg_jacobi_mt(ai, aj, aa, b, its, nthreads):
    """CG + Jacobi, `its` iterations from x = 0, one thread per block of rows (oracle/cpu_baseline_mt.c): the CPU
    baseline's timed loop.  Returns (seconds, x, last preconditioned residual norm)."""
    L = lib()
    L.orc_cg_jacobi_mt.restype = C.c_double
    n = ai.size - 1
    x = np.zeros(n)
    rn = C.c_double()
    ai = np.ascontiguousarray(ai, dtype=np.int32); aj = np.ascontiguousarray(aj, dtype=np.int32)
    aa = np.ascontiguousarray(aa, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
    t = L.orc_cg_jacobi_mt(C.c_int(n), ai.ctypes.data_as(C.c_void_p), aj.ctypes.data_as(C.c_void_p), aa.ctypes.data_as(C.c_void_p),
                           b.ctypes.data_as(C.c_void_p), C.c_int(its), C.c_int(nthreads), x.ctypes.data_as(C.c_void_p), C.byref(rn))
    return t, x, rn.value
