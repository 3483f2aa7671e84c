"""GPU parity of the host library (Vec/Mat/KSP/PC objects over the HIP kernels) against the oracle and the
reference's golden outputs.  These read like the reference's own example tests: build the objects through
the PETSc-named API, run the op, compare.

Tolerances (fp64): element-wise ops, MAXPY, SpMV/SpMV-add/transpose: bit-exact.  Reductions:
|err| <= 1e-13 * sum|terms|.  Krylov residual histories: equal to 6 significant digits -- the text
-ksp_monitor_short prints -- with identical iteration counts for the golden cases, and relative 1e-6 /
iteration count +-1 for the long CG case (BASELINE.md tolerances)."""
import ctypes as C
import os

import numpy as np
import pytest

import orc
import problems as pb
from gpu import ksp_type_for

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def P(built):
    from petsc_dev_amd import petsc as P
    P.lib()
    return P


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


def rnd(n, seed):
    return np.random.default_rng(seed).standard_normal(n)


def V(P, a):
    return P.Vec.from_array(a, comm=P.lib().COMM_SELF)


def test_vec_ops_through_function_table(P):
    L = P.lib()
    n = 100003
    x, y, z = rnd(n, 1), rnd(n, 2), rnd(n, 3)
    vx, vy, vz = V(P, x), V(P, y), V(P, z)
    t = C.c_char_p(); L.VecGetType(vx.h, C.byref(t)); assert t.value == b"seqhipmi355x"
    L.VecAXPY(vy.h, 0.3, vx.h); r = y.copy(); orc.vec_axpy(r, 0.3, x); assert np.array_equal(bits(vy.array()), bits(r)); y = r
    L.VecAYPX(vy.h, -1.7, vx.h); orc.vec_aypx(r, -1.7, x); assert np.array_equal(bits(vy.array()), bits(r))
    L.VecWAXPY(vz.h, 2.5, vx.h, vy.h); w = np.zeros(n); orc.vec_waxpy(w, 2.5, x, r); assert np.array_equal(bits(vz.array()), bits(w))
    L.VecAXPBYPCZ(vz.h, 0.5, -0.25, 1.0, vx.h, vy.h); orc.vec_axpbypcz(w, 0.5, -0.25, 1.0, x, r); assert np.array_equal(bits(vz.array()), bits(w))
    L.VecPointwiseMult(vz.h, vx.h, vy.h); orc.vec_pointwise_mult(w, x, r); assert np.array_equal(bits(vz.array()), bits(w))
    # reductions + norm cache (rvector.c:205-224): second call is served from the cache, a write invalidates it
    n2 = vx.norm(P.NORM_2)
    assert abs(n2 - orc.vec_norm(x, 1)) <= 1e-13 * n2 * 4
    assert vx.norm(P.NORM_2) == n2
    L.VecScale(vx.h, 2.0)
    assert vx.norm(P.NORM_2) == 2.0 * n2            # VecScale rescales the cached norm (rvector.c:476)
    L.VecAXPY(vx.h, 1.0, vy.h)
    x2 = 2.0 * x; orc.vec_axpy(x2, 1.0, r)
    assert abs(vx.norm(P.NORM_2) - orc.vec_norm(x2, 1)) <= 1e-12 * orc.vec_norm(x2, 1)
    assert abs(vx.dot(vy) - orc.vec_dot(x2, r)) <= 1e-13 * np.sum(np.abs(x2 * r))
    assert vx.norm(P.NORM_INFINITY) == np.max(np.abs(x2))
    n1, n2b = vx.norm(P.NORM_1_AND_2)
    assert abs(n1 - np.sum(np.abs(x2))) <= 1e-13 * n1 and abs(n2b - np.linalg.norm(x2)) <= 1e-12 * n2b
    dp, nm = C.c_double(), C.c_double()
    L.VecDotNorm2(vx.h, vy.h, C.byref(dp), C.byref(nm))
    rdp, rnm = orc.vec_dotnorm2(x2, r)
    assert abs(dp.value - rdp) <= 1e-13 * np.sum(np.abs(x2 * r)) and abs(nm.value - rnm) <= 1e-13 * rnm
    # VecSet caches the norms of a constant vector
    L.VecSet(vz.h, -3.0)
    assert vz.norm(P.NORM_INFINITY) == 3.0 and vz.norm(P.NORM_1) == 3.0 * n
    # error behaviour of the wrappers
    with pytest.raises(P.PetscError) as e:
        L.VecAXPY(vx.h, 1.0, vx.h)
    assert e.value.code == 61
    small = V(P, np.ones(5))
    with pytest.raises(P.PetscError) as e:
        L.VecAXPY(vx.h, 1.0, small.h)
    assert e.value.code == 75


def test_vec_tests_ex18_dot_known_answer(P):
    """src/vec/vec/examples/tests/ex18.c vs output/ex18_1.out on HIPMI355X vectors: VecDot (and VecTDot, VecMDot's first entry) of
    the example's two 15-entry vectors, printed with %16.12e"""
    import ctypes as C
    from test_oracle_golden import ex18_vectors
    L = P.lib()
    xa, ya = ex18_vectors()
    x, y = V(P, xa), V(P, ya)
    want = open(os.path.join(G, "vec_tests", "ex18_1.out")).read().strip()
    assert want == "Vector inner product %16.12e" % x.dot(y)
    t = C.c_double(); L.VecTDot(x.h, y.h, C.byref(t))
    tab = (C.c_void_p * 1)(y.h.value); d1 = (C.c_double * 1)()
    L.VecMDot(x.h, 1, tab, d1)
    assert want == "Vector inner product %16.12e" % t.value == "Vector inner product %16.12e" % d1[0]


def test_vec_tutorial_ex1_golden(P):
    """src/vec/vec/examples/tutorials/ex1.c replayed on HIPMI355X vectors (n = 20): VecSet, Dot, MDot, Scale, Copy, AXPY,
    AYPX, Swap, WAXPY, PointwiseMult, PointwiseDivide, MAXPY and the norms in between, printed as the example prints
    them and compared with the reference's output/ex1_1.out (the VecMax/VecMin lines are not on the ported path)."""
    import ctypes as C
    L = P.lib()
    n = 20
    small = 1e-10                                      # PETSC_SMALL
    x = P.Vec.create(n, comm=L.COMM_SELF)
    y, w = x.duplicate(), x.duplicate()
    z = [x.duplicate() for _ in range(3)]
    L.VecSet(x.h, 1.0); L.VecSet(y.h, 2.0)
    for k, v in enumerate((1.0, 2.0, 3.0)):
        L.VecSet(z[k].h, v)
    assert x.dot(y) == 2.0 * n
    tab = (C.c_void_p * 3)(*[v.h.value for v in z])
    dots = (C.c_double * 3)()
    L.VecMDot(x.h, 3, tab, dots)
    assert list(dots) == [1.0 * n, 2.0 * n, 3.0 * n]
    out = ["Vector length %d" % n, "All other values should be near zero"]

    def chk(name, vec, expect):
        v = vec.norm() - expect * np.sqrt(float(n))
        if -small < v < small:
            v = 0.0
        return "%s %g" % (name, v)
    L.VecScale(x.h, 2.0); out.append(chk("VecScale", x, 2.0))
    L.VecCopy(x.h, w.h); out.append(chk("VecCopy ", w, 2.0))
    L.VecAXPY(y.h, 3.0, x.h); out.append(chk("VecAXPY", y, 8.0))
    L.VecAYPX(y.h, 2.0, x.h); out.append(chk("VecAYPX", y, 18.0))
    L.VecSwap(x.h, y.h); out.append(chk("VecSwap ", y, 2.0)); out.append(chk("VecSwap ", x, 18.0))
    L.VecWAXPY(w.h, 2.0, x.h, y.h); out.append(chk("VecWAXPY", w, 38.0))
    L.VecPointwiseMult(w.h, y.h, x.h); out.append(chk("VecPointwiseMult", w, 36.0))
    L.VecPointwiseDivide(w.h, x.h, y.h); out.append(chk("VecPointwiseDivide", w, 9.0))
    al = (C.c_double * 3)(1.0, 3.0, 2.0)
    L.VecSet(x.h, 1.0)
    L.VecMAXPY(x.h, 3, al, tab)
    vs = []
    for k, e in enumerate((1.0, 2.0, 3.0)):
        v = z[k].norm() - e * np.sqrt(float(n))
        vs.append(0.0 if -small < v < small else v)
    out.append("VecMAXPY %g %g %g " % tuple(vs))
    assert np.array_equal(x.array(), np.full(n, 1.0 + 1.0 + 6.0 + 6.0))
    gold = [l.rstrip("\n") for l in open(os.path.join(G, "vec_tutorials", "ex1_1.out")) if not l.startswith(("VecMax", "VecMin"))]
    assert [l.rstrip() for l in out] == [l.rstrip() for l in gold]


def test_vec_mdot_maxpy_and_host_access(P):
    L = P.lib()
    n, nv = 50001, 13
    x = rnd(n, 5)
    ys = [rnd(n, 10 + j) for j in range(nv)]
    vx = V(P, x)
    vys = [V(P, a) for a in ys]
    tab = P.vec_table(vys)
    out = np.zeros(nv)
    L.VecMDot(vx.h, nv, tab, out.ctypes.data_as(C.c_void_p))
    ref = orc.vec_mdot(x, ys)
    for j in range(nv):
        assert abs(out[j] - ref[j]) <= 1e-13 * np.sum(np.abs(x * ys[j]))
    al = rnd(nv, 6)
    L.VecMAXPY(vx.h, nv, al.ctypes.data_as(C.c_void_p), tab)
    r = x.copy(); orc.vec_maxpy(r, al, ys)
    assert np.array_equal(bits(vx.array()), bits(r))
    # host write access through VecGetArray/VecRestoreArray marks the device copy stale (coherence flags)
    p = C.c_void_p()
    L.VecGetArray(vx.h, C.byref(p))
    arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), (n,))
    arr[:] = 2.0
    L.VecRestoreArray(vx.h, C.byref(p))
    assert vx.norm(P.NORM_1) == 2.0 * n
    # VecPlaceArray / VecResetArray (used by PCBJACOBI in the reference)
    other = np.full(n, 0.5)
    L.VecPlaceArray(vx.h, other.ctypes.data_as(C.c_void_p))
    assert vx.norm(P.NORM_1) == 0.5 * n
    L.VecResetArray(vx.h)
    assert vx.norm(P.NORM_1) == 2.0 * n


@pytest.mark.parametrize("name,rect,mtype", [("ex5_11_A.out", 2, "seq"), ("ex5_11_B.out", -2, "seq"), ("ex5_21.out", 0, "mpi")])
def test_mat_ex5_golden(P, name, rect, mtype):
    """src/mat/examples/tests/ex5.c: MatMult, MatMultAdd, MatMultTranspose, MatMultTransposeAdd, MatGetDiagonal,
    MatScale on seqaij (-rectA) and mpiaij (np=1); printed vectors must equal the golden files"""
    L = P.lib()
    ai, aj, aa, m, n = pb.ex5_mat(8, rect=rect, alpha=1.0)
    if mtype == "seq":
        A = P.Mat.from_csr(ai, aj, aa, ncols=n)
    else:
        A = P.Mat.from_csr_mpi(ai, aj, aa, n, m, n, comm=L.COMM_SELF)
    L.MatScale(A.h, 0.1)
    gold = pb.parse_vecview(os.path.join(G, name))
    fmt = lambda v: np.array([float("%g" % t) for t in v])  # noqa: E731
    y = V(P, np.arange(n, dtype=np.float64)); x = V(P, np.zeros(m)); z = V(P, 100.0 * (np.arange(m) + 1)); w = V(P, np.zeros(m))
    u = V(P, 100.0 * np.arange(n)); s = V(P, np.zeros(n))
    L.MatMult(A.h, y.h, x.h)
    assert np.array_equal(fmt(x.array()), gold[0])
    L.MatMultAdd(A.h, y.h, z.h, w.h)
    L.VecAXPY(x.h, 1.0, z.h); L.VecAXPY(x.h, -1.0, w.h)
    assert x.norm() <= 1e-8                         # the example's own check
    x.set_array(np.arange(m, dtype=np.float64))
    L.MatMultTranspose(A.h, x.h, y.h)
    assert np.array_equal(fmt(y.array()), gold[1])
    L.MatMultTransposeAdd(A.h, x.h, u.h, s.h)
    L.VecAXPY(y.h, 1.0, u.h); L.VecAXPY(y.h, -1.0, s.h)
    assert y.norm() <= 1e-8
    if rect == 0:
        L.MatGetDiagonal(A.h, x.h)
        assert np.array_equal(fmt(x.array()), gold[-1])
    # bit-exact against the oracle too
    ai2, aj2, aa2, _, _ = pb.ex5_mat(8, rect=rect, alpha=1.0)
    aa2 = 0.1 * aa2
    yy = np.arange(n, dtype=np.float64)
    y.set_array(yy)
    L.MatMult(A.h, y.h, w.h)
    assert np.array_equal(bits(w.array()), bits(orc.matmult(ai2, aj2, aa2, yy)[0]))


@pytest.mark.parametrize("name,mtype", [("ex5_31.out", "aij"), ("ex5_32.out", "baij")])
def test_mat_ex5_diagonalscale_golden_aij_and_baij(P, name, mtype):
    """src/mat/examples/tests/ex5.c -test_diagonalscale with -mat_type mpiaij / mpibaij on one rank (makefile:796-804) vs
    output/ex5_31.out, ex5_32.out: MatMult, MatMultTranspose, MatGetDiagonal and the matrix after MatDiagonalScale(C, x, y),
    every printed number.  ex5_32.out is the one output the reference holds for the BAIJ type on this path (block size 1:
    MatMult_SeqBAIJ_1); it equals the AIJ output, as the BAIJ type here equals the AIJ type at block size 1."""
    import re
    L = P.lib()
    ai, aj, aa, m, n = pb.ex5_mat(8, rect=0, alpha=1.0)
    A = P.Mat.from_csr(ai, aj, aa, ncols=n) if mtype == "aij" else P.Mat.from_bsr(1, ai, aj, aa)
    L.MatScale(A.h, 0.1)
    text = open(os.path.join(G, name)).read()
    gold = pb.parse_vecview(os.path.join(G, name))
    fmt = lambda v: np.array([float("%g" % t) for t in v])  # noqa: E731
    y = V(P, np.arange(n, dtype=np.float64)); x = V(P, np.zeros(m))
    L.MatMult(A.h, y.h, x.h)
    assert np.array_equal(fmt(x.array()), gold[0])
    x.set_array(np.arange(m, dtype=np.float64))
    L.MatMultTranspose(A.h, x.h, y.h)
    assert np.array_equal(fmt(y.array()), gold[1])
    L.VecSet(x.h, 1.0)
    L.MatGetDiagonal(A.h, x.h)
    assert np.array_equal(fmt(x.array()), gold[2])
    y.set_array(np.arange(1, n + 1, dtype=np.float64))
    L.MatDiagonalScale(A.h, x.h, y.h)
    # the two MatView blocks of the golden: before and after the scaling
    views = text.split("Matrix Object:")[1:]
    assert len(views) == 2
    want = np.array([[float(v) for _, v in re.findall(r"\((\d+), ([-0-9.e+]+)\)", line)] for line in views[1].splitlines() if line.startswith("row ")])
    m_, i_, j_, a_ = C.c_int(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    L.MatSeqAIJGetArrays(A.h, C.byref(m_), C.byref(i_), C.byref(j_), C.byref(a_))
    vals = np.ctypeslib.as_array(C.cast(a_, C.POINTER(C.c_double)), (m * n,)).reshape(m, n)
    assert np.array_equal(fmt(vals.ravel()).reshape(m, n), want)
    # and the device copy was scaled side by side: the product with ones equals the row sums of the printed matrix
    ones = V(P, np.ones(n)); r = V(P, np.zeros(m))
    L.MatMult(A.h, ones.h, r.h)
    assert np.allclose(r.array(), vals.sum(1), rtol=1e-14)


def test_mat_p7_and_transpose_bitexact(P):
    L = P.lib()
    ai, aj, aa = P.gen_poisson7(20, 17, 13)
    n = ai.size - 1
    aa = aa * (1.0 + 0.01 * np.sin(np.arange(aa.size)))     # make the transpose non-trivial
    A = P.Mat.from_csr(ai, aj, aa)
    x = np.sin(0.37 * np.arange(n)) + 1.0
    vx, vy, vz = V(P, x), V(P, np.zeros(n)), V(P, rnd(n, 3))
    L.MatMult(A.h, vx.h, vy.h)
    assert np.array_equal(bits(vy.array()), bits(orc.matmult(ai, aj, aa, x)[0]))
    L.MatMultTranspose(A.h, vx.h, vy.h)
    assert np.array_equal(bits(vy.array()), bits(orc.spmv_t(ai, aj, aa, x, n)))
    z = vz.array()
    L.MatMultTransposeAdd(A.h, vx.h, vz.h, vy.h)
    assert np.array_equal(bits(vy.array()), bits(orc.spmv_t_add(ai, aj, aa, x, z, n)))
    L.MatMultAdd(A.h, vx.h, vz.h, vz.h)            # in place
    assert np.array_equal(bits(vz.array()), bits(orc.matmult(ai, aj, aa, x, z)[0]))
    # value-only update: MatScale bumps the state, only `a` is re-sent
    L.MatScale(A.h, -2.0)
    L.MatMult(A.h, vx.h, vy.h)
    assert np.array_equal(bits(vy.array()), bits(orc.matmult(ai, aj, -2.0 * aa, x)[0]))
    fl = C.c_double(); L.PetscGetFlops(C.byref(fl)); assert fl.value > 0


@pytest.mark.parametrize("bs,opt", [(3, ""), (4, ""), (4, "-mat_hipmi355x_baij4 fma"), (2, ""), (5, "")])
def test_baij_matmult(P, bs, opt):
    """MatMult_SeqBAIJ_N / _3 / _4 (baij2.c:331-436,981); bs = 4: the matrix cores by default, -mat_hipmi355x_baij4 fma
    selects the row-block FMA kernel; BASELINE.md tolerance against the oracle's restatement"""
    L = P.lib()
    rng = np.random.default_rng(4)
    mbs = 300
    import scipy.sparse as sp
    S = sp.random(mbs, mbs, density=0.05, random_state=5, format="csr") + sp.eye(mbs, format="csr")
    S = sp.csr_matrix(S); S.sort_indices()
    bi, bj = S.indptr.astype(np.int32), S.indices.astype(np.int32)
    ba = rng.standard_normal(bj.size * bs * bs)
    set_options(L, opt)
    A = P.Mat.from_bsr(bs, bi, bj, ba)
    x = rnd(mbs * bs, 6)
    vx, vy = V(P, x), V(P, np.zeros(mbs * bs))
    L.MatMult(A.h, vx.h, vy.h)
    set_options(L, "")
    ref = orc.spmv_bsr(bs, bi, bj, ba, x)
    assert np.allclose(vy.array(), ref, rtol=0, atol=1e-12 * 50)
    # MatMultAdd_SeqBAIJ_3/_4/_N (baij2.c:1168-1480): z = y + A x, separate z and in place
    y0 = rnd(mbs * bs, 7)
    vz, vw = V(P, y0), V(P, np.zeros(mbs * bs))
    L.MatMultAdd(A.h, vx.h, vz.h, vw.h)
    assert np.allclose(vw.array(), y0 + ref, rtol=0, atol=1e-12 * 50)
    L.MatMultAdd(A.h, vx.h, vz.h, vz.h)
    assert np.allclose(vz.array(), y0 + ref, rtol=0, atol=1e-12 * 50)
    # MatDiagonalScale_SeqBAIJ (baij2.c:2026-2084) and MatScale, then the product again; and a duplicate of the scaled matrix
    l, r = 1.0 + 0.1 * rnd(mbs * bs, 8), 1.0 + 0.1 * rnd(mbs * bs, 9)
    vl, vr = V(P, l), V(P, r)
    L.MatDiagonalScale(A.h, vl.h, vr.h)
    L.MatScale(A.h, -0.5)
    blk_row = np.repeat(np.arange(mbs), np.diff(bi))
    ba2 = ba.reshape(-1, bs, bs).copy()                      # [block][column][row]: column-major blocks
    for c in range(bs):
        for rr in range(bs):
            ba2[:, c, rr] = ((ba2[:, c, rr] * l[blk_row * bs + rr]) * r[bj * bs + c]) * -0.5
    ref2 = orc.spmv_bsr(bs, bi, bj, ba2.reshape(-1), x)
    L.MatMult(A.h, vx.h, vy.h)
    assert np.allclose(vy.array(), ref2, rtol=0, atol=1e-12 * 50)
    import ctypes as C
    Bh = C.c_void_p()
    L.MatDuplicate(A.h, 1, C.byref(Bh))
    B = P.Mat(Bh)
    L.MatMult(B.h, vx.h, vw.h)                           # (the copy reads the options anew: bs = 4 may take the other kernel than A did)
    assert np.allclose(vw.array(), ref2, rtol=0, atol=1e-12 * 50)
    # MatMultTranspose_SeqBAIJ / MatMultTransposeAdd_SeqBAIJ (baij2.c:1579, 1740) against the scalar matrix's transpose product
    Sp = sp.bsr_matrix((ba2.transpose(0, 2, 1), bj, bi), shape=(mbs * bs, mbs * bs)).tocsr()   # blocks [column][row] -> [row][column]
    reft = Sp.T @ x
    scale = np.abs(Sp.T) @ np.abs(x)
    L.MatMultTranspose(A.h, vx.h, vy.h)
    assert np.all(np.abs(vy.array() - reft) <= 1e-12 * scale + 1e-300)
    vz2 = V(P, y0)
    L.MatMultTransposeAdd(A.h, vx.h, vz2.h, vw.h)
    assert np.all(np.abs(vw.array() - (y0 + reft)) <= 1e-12 * (scale + np.abs(y0)) + 1e-300)
    L.MatMultTransposeAdd(A.h, vx.h, vz2.h, vz2.h)
    assert np.all(np.abs(vz2.array() - (y0 + reft)) <= 1e-12 * (scale + np.abs(y0)) + 1e-300)
    L.MatScale(A.h, 2.0)                                 # the block transpose follows the matrix
    L.MatMultTranspose(A.h, vx.h, vy.h)
    assert np.all(np.abs(vy.array() - 2.0 * reft) <= 2e-12 * scale + 1e-300)


@pytest.mark.parametrize("pc,opts", [("jacobi", ""), ("none", ""), ("ilu", ""), ("jacobi", "-ksp_gmres_restart 7"), ("jacobi", "-ksp_gmres_restart 40"),
                                     ("jacobi", "-ksp_pc_side right"), ("jacobi", "-ksp_gmres_cgs_refinement_type refine_always")])
def test_gmres_fused_equals_op_by_op_bit_for_bit(P, pc, opts):
    """-ksp_gmres_fused (default): SpMV with PCJACOBI's scaling in its epilogue, MDot -> MAXPY + norm -> scale with the
    scalars on the device.  Every residual norm and the solution carry the bits of the reference's op-by-op sequence
    (-ksp_gmres_fused 0), for restarts below / above the 32-vector sweep, other PCs, right preconditioning, refinement."""
    ai, aj, aa = orc.gen_p7(14, 12, 10)
    aa = aa * (1.0 + 0.2 * np.sin(np.arange(aa.size)))          # nonsymmetric values
    n = ai.size - 1
    b = np.cos(0.37 * np.arange(n))
    x1, h1, it1, r1 = solve(P, ai, aj, aa, b, "gmres", pc, opts=opts, rtol=1e-11)
    x0, h0, it0, r0 = solve(P, ai, aj, aa, b, "gmres", pc, opts=opts + " -ksp_gmres_fused 0", rtol=1e-11)
    assert (it1, r1) == (it0, r0) and r1 > 0 and it1 > 8
    assert np.array_equal(h1.view(np.uint64), h0.view(np.uint64))
    assert np.array_equal(x1.view(np.uint64), x0.view(np.uint64))
    if pc == "jacobi" and not opts:                                # and both walk the oracle's path (reductions: other tree, rounding apart)
        xo, ho, ito, ro = orc.ksp_solve(ai, aj, aa, b, ksp="gmres", pc="jacobi", rtol=1e-11)
        assert abs(ito - it1) <= 1 and np.allclose(ho[:50], h1[:50], rtol=1e-9, atol=0)


@pytest.mark.parametrize("ksp,pc,opts", [("cg", "jacobi", "-ksp_cg_fused 3"), ("cg", "none", "-ksp_cg_fused 3"), ("cg", "jacobi", "-ksp_cg_fused 0"), ("gmres", "jacobi", ""),
                                        ("gmres", "ilu", ""), ("gmres", "none", "-ksp_gmres_restart 9"), ("bcgs", "jacobi", ""), ("bcgs", "none", ""),
                                        ("groppcg", "jacobi", ""), ("cg", "jacobi", "-ksp_norm_type natural -ksp_cg_fused 3"),
                                        ("gmres", "jacobi", "-ksp_gmres_cgs_refinement_type refine_always"), ("pipecg", "jacobi", "-ksp_norm_type natural"),
                                        ("gmres", "jacobi", "-ksp_gmres_fused 0"), ("bcgs", "jacobi", "-ksp_bcgs_fused 0")])
def test_whole_solves_equal_the_oracle_bit_for_bit_in_the_device_summation_order(P, ksp, pc, opts):
    """The parity tolerances elsewhere in this file exist for ONE reason: the reference adds the terms of a dot product one
    after the other, the HIP reductions add the same terms in a tree.  With the oracle's reductions switched to that tree
    (orc.device_reduction_order(): same terms, restated order) every residual norm of the history, the iteration count, the
    reason and the solution of a whole Krylov solve are bit-identical between the HIP path and the oracle -- CG, GMRES,
    BiCGStab, GROPPCG, with Jacobi / ILU(0) / none, fused or op by op.  On two operators: 3-D 7-point (row-pattern SpMV
    kernel) and a nonsymmetric variable-coefficient one (BiCGStab's recurrences amplify any last-bit difference: none left)."""
    norm = 3 if "natural" in opts else 1
    for which in ("p7", "nonsym"):
        ai, aj, aa = orc.gen_p7(17, 15, 13)
        if which == "nonsym":
            aa = aa * (1.0 + 0.3 * np.sin(0.7 * np.arange(aa.size)))
            if ksp in ("cg", "groppcg", "pipecg"):
                continue
        n = ai.size - 1
        b = np.cos(0.37 * np.arange(n)) + 0.1
        kw = dict(rtol=1e-10, max_it=300)
        x, h, its, reason = solve(P, ai, aj, aa, b, ksp, pc, opts=opts, **kw)
        okw = {}
        if "-ksp_gmres_restart" in opts:
            okw["restart"] = 9
        if "refine_always" in opts:
            okw["refine_always"] = 1
        with orc.device_reduction_order():
            xo, ho, ito, ro = orc.ksp_solve(ai, aj, aa, b, ksp=ksp, pc=pc, norm_type=norm, **kw, **okw)
        assert (its, reason) == (ito, ro) and its > 5
        assert np.array_equal(h.view(np.uint64), ho.view(np.uint64)), (which, np.max(np.abs(h - ho) / ho))
        assert np.array_equal(x.view(np.uint64), xo.view(np.uint64))


@pytest.mark.parametrize("ksp,pc", [("cg", "jacobi"), ("cg", "none"), ("gmres", "jacobi"), ("bcgs", "jacobi"), ("gmres", "ilu"), ("groppcg", "jacobi")])
def test_value_patterns_leave_every_bit_of_a_solve_alone(P, ksp, pc):
    """the product run from the row dictionary (default) against the same solve with the value array streamed
    (-mat_hipmi355x_value_patterns 0): every residual norm and the solution, bit for bit -- 3-D 7-point and 2-D 5-point operators
    (CG at -ksp_cg_fused 3: its default forms p'w inside the SpMV pass, where the two kernels sum in different trees)"""
    for (ai, aj, aa) in (orc.gen_p7(19, 16, 14), pb.lap2d(37, 29)):
        n = ai.size - 1
        b = np.cos(0.21 * np.arange(n)) + 0.3
        lv = "-ksp_cg_fused 3 " if ksp == "cg" else ""
        runs = [solve(P, ai, aj, aa, b, ksp, pc, opts=lv + o, rtol=1e-10) for o in ("", "-mat_hipmi355x_value_patterns 0")]
        (x0, h0, its0, r0), (x1, h1, its1, r1) = runs
        assert (its0, r0) == (its1, r1) and its0 > 10
        assert np.array_equal(h0.view(np.uint64), h1.view(np.uint64)) and np.array_equal(x0.view(np.uint64), x1.view(np.uint64))
        if ksp == "cg":    # CG's default takes p'w out of the SpMV pass, per workgroup: the two kernels' trees differ, the products do not
            (x4, h4, its4, r4), (x5, h5, its5, r5) = [solve(P, ai, aj, aa, b, ksp, pc, opts=o, rtol=1e-10) for o in ("", "-mat_hipmi355x_value_patterns 0")]
            assert r4 == r5 == r0 and abs(its4 - its0) <= 1 and abs(its5 - its0) <= 1
            assert np.linalg.norm(x4 - x0) <= 1e-8 * np.linalg.norm(x0) and np.linalg.norm(x5 - x0) <= 1e-8 * np.linalg.norm(x0)
    # and the default really is the dictionary, the option really the streamed values
    L = P.lib(); nv = C.c_int()
    A = P.Mat.from_csr(*orc.gen_p7(6, 5, 4)); L.MatHIPMI355XGetValuePatterns(A.h, C.byref(nv)); assert nv.value > 0; A.destroy()
    L.PetscOptionsSetValue(b"-mat_hipmi355x_value_patterns", b"0")
    try:
        A = P.Mat.from_csr(*orc.gen_p7(6, 5, 4)); L.MatHIPMI355XGetValuePatterns(A.h, C.byref(nv)); assert nv.value == 0; A.destroy()
    finally:
        L.PetscOptionsClear()


def test_config1_cg_jacobi_equals_the_oracle_bit_for_bit_in_the_device_summation_order(P):
    """BASELINE.json configs[0] (ex2 -m 100 -n 100, CG + Jacobi, 160 iterations): all 161 residual norms and the solution,
    bit for bit, with the oracle's reductions in the device order"""
    ai, aj, aa = pb.lap2d(100, 100)
    b = orc.spmv(ai, aj, aa, np.ones(10000))
    rtol = 1e-2 / (101 * 101)                                     # ex2.c: KSPSetTolerances(ksp, 1.e-2/((m+1)*(n+1)), 1.e-50, ...)
    x, h, its, reason = solve(P, ai, aj, aa, b, "cg", "jacobi", rtol=rtol, abstol=1e-50, opts="-ksp_cg_fused 3")
    x4, h4, its4, reason4 = solve(P, ai, aj, aa, b, "cg", "jacobi", rtol=rtol, abstol=1e-50)      # the default: p'w out of the SpMV pass
    assert (its4, reason4) == (its, reason) and np.allclose(h4, h, rtol=1e-9, atol=0) and np.linalg.norm(x4 - x) <= 1e-9 * np.linalg.norm(x)
    with orc.device_reduction_order():
        xo, ho, ito, ro = orc.ksp_solve(ai, aj, aa, b, ksp="cg", pc="jacobi", rtol=rtol, abstol=1e-50)
    assert (its, reason) == (ito, ro) and its == 160
    assert np.array_equal(h.view(np.uint64), ho.view(np.uint64)) and np.array_equal(x.view(np.uint64), xo.view(np.uint64))


@pytest.mark.parametrize("bs", [2, 3, 4, 5])
def test_pbjacobi_on_baij(P, bs):
    """PCPBJACOBI on the BAIJ type (pbjacobi.c + MatInvertBlockDiagonal_SeqBAIJ baij.c:13): PCApply equals the
    restatement bit for bit (-ksp_type preonly: x = B b), and CG + PCPBJACOBI walks the restatement's residual history
    (the BAIJ product differs from the point-wise one in summation order: BASELINE.md tolerance)."""
    L = P.lib()
    (ai, aj, aa), (bi, bj, ba) = pb.spd_blocks(7, 6, 5, dof=bs)
    n = ai.size - 1
    b = np.cos(0.1 * np.arange(n))
    idiag = orc.pbjacobi_setup(bs, bi, bj, ba)

    def run(ksp, **tol):
        A = P.Mat.from_bsr(bs, bi, bj, ba)
        vb, vx = V(P, b), V(P, np.zeros(n))
        k = P.KSP(comm=L.COMM_SELF); k.set_operators(A)
        set_options(L, "-ksp_type %s -pc_type pbjacobi" % ksp)
        if tol:
            k.set_tolerances(**tol)
        k.set_from_options(); set_options(L, "")
        k.record_history()
        k.solve(vb, vx)
        return vx.array(), k.history(), k.its, k.reason

    x, _, _, _ = run("preonly")
    assert np.array_equal(x.view(np.uint64), orc.pbjacobi_apply(bs, idiag, b).view(np.uint64))
    x, h, its, reason = run("cg", rtol=1e-10)
    xo, ho, ito, ro = orc.ksp_solve(ai, aj, aa, b, ksp="cg", pc="pbjacobi", pb_bs=bs, rtol=1e-10)
    assert (its, reason) == (ito, ro) and reason > 0
    assert np.allclose(h, ho, rtol=1e-8, atol=0) and np.allclose(x, xo, rtol=1e-9, atol=1e-12)


def test_pbjacobi_zero_pivot_and_wrong_type_are_errors(P):
    """MatInvertBlockDiagonal's zero-pivot error reaches KSPSetUp; a missing diagonal block too"""
    L = P.lib()
    bs = 2
    bi = np.array([0, 1, 2], dtype=np.int32)
    for bj, ba in ((np.array([0, 1], dtype=np.int32), np.array([1., 1, 1, 1, 2, 0, 0, 2])),      # first block singular
                   (np.array([1, 1], dtype=np.int32), np.array([2., 0, 0, 2, 2, 0, 0, 2]))):     # block row 0 has no diagonal block
        A = P.Mat.from_bsr(bs, bi, bj, ba)
        k = P.KSP(comm=L.COMM_SELF); k.set_operators(A)
        set_options(L, "-ksp_type preonly -pc_type pbjacobi"); k.set_from_options(); set_options(L, "")
        vb, vx = V(P, np.ones(4)), V(P, np.zeros(4))
        with pytest.raises(P.PetscError):
            k.solve(vb, vx)


def test_config5_full_size_baij_equals_aij(P):
    """BASELINE configs[4] at full size: the 27-point, 3-dof elasticity shape on 128^3 nodes (6.3 M rows, 5.6e7 blocks)
    stored as BAIJ bs = 3, as BAIJ zero-padded to bs = 4 (matrix-core kernel and FMA kernel) and as point-wise AIJ (5.1e8
    nonzeros: the reference's inode matrix).  The reference's own check for BAIJ (src/mat/examples/tests/ex48.c: the
    BAIJ product equals the AIJ product of the same matrix) at BASELINE.md's tolerance, plus sampled rows against a
    host dot product."""
    L = P.lib()
    nn, bs = 128, 3
    import sys as _sys
    _sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    from bench_configs import gen_baij27
    bi, bj, ba3 = gen_baij27(nn, bs=3)
    mbs, nnzb = bi.size - 1, bj.size
    blocks_cr = ba3.reshape(nnzb, 3, 3)                      # [blk][col][row]
    x3 = np.sin(0.1 * np.arange(mbs * 3))
    vx3, vy3 = V(P, x3), V(P, np.zeros(mbs * 3))
    A3 = P.Mat.from_bsr(3, bi, bj, ba3)
    A3.mult(vx3, vy3)
    y3 = vy3.array()
    # sampled block rows against the host
    absrow = None
    for br in np.random.default_rng(8).integers(0, mbs, 300):
        blk = slice(bi[br], bi[br + 1])
        t = blocks_cr[blk] * x3.reshape(mbs, 3)[bj[blk]][:, :, None]     # [blk][col][row] products
        assert np.all(np.abs(y3[3 * br:3 * br + 3] - t.sum(axis=(0, 1))) <= 1e-12 * np.abs(t).sum(axis=(0, 1)))
    # padded to 4x4: both bs = 4 kernels
    b4 = np.zeros((nnzb, 4, 4)); b4[:, :3, :3] = blocks_cr
    x4 = np.zeros((mbs, 4)); x4[:, :3] = x3.reshape(mbs, 3)
    scale = np.abs(y3).max()
    for opt in ("", "-mat_hipmi355x_baij4 fma"):
        set_options(L, opt)
        A4 = P.Mat.from_bsr(4, bi, bj, b4.ravel())
        vx4, vy4 = V(P, x4.ravel()), V(P, np.zeros(mbs * 4))
        A4.mult(vx4, vy4)
        set_options(L, "")
        y4 = vy4.array().reshape(mbs, 4)
        assert np.all(y4[:, 3] == 0.0)
        assert np.max(np.abs(y4[:, :3].ravel() - y3)) <= 1e-12 * 81 * scale
        del A4, vx4, vy4
    del b4
    # the same operator point-wise (AIJ): 81 nonzeros per row, 135 distinct offsets, rows of a node share their pattern
    ai, aj, aa = pb.expand_blocks(bi, bj, np.ascontiguousarray(blocks_cr.transpose(0, 2, 1)))
    A = P.Mat.from_csr(ai, aj, aa)
    A.mult(vx3, vy3)
    nodes = C.c_int(); L.MatHIPMI355XGetInodeInfo(A.h, C.byref(nodes), None, None)
    assert nodes.value == mbs
    assert np.max(np.abs(vy3.array() - y3)) <= 1e-12 * 81 * scale


def test_elasticity_aij_vs_inode_and_baij(P):
    """the same 3-dof operator as point AIJ (reference: inode routines) and as BAIJ bs=3
    (MatMult_SeqBAIJ_3).  The HIP AIJ path equals the plain row loop bit for bit and the inode variant to 1e-12
    (scaled per row); the BCSR kernel agrees with MatMult_SeqBAIJ_3's order to the same tolerance."""
    L = P.lib()
    (ai, aj, aa), (bi, bj, ba) = pb.elasticity_like(9, 8, 7)
    m = ai.size - 1
    x = rnd(m, 11)
    scale = np.zeros(m); np.add.at(scale, np.repeat(np.arange(m), np.diff(ai)), np.abs(aa * x[aj]))
    A = P.Mat.from_csr(ai, aj, aa)
    vx, vy = V(P, x), V(P, np.zeros(m))
    L.MatMult(A.h, vx.h, vy.h)
    y = vy.array()
    # rows of 24..81 nonzeros use several lanes per row + a shuffle tree: tolerance, not bit-exactness
    assert np.all(np.abs(y - orc.spmv(ai, aj, aa, x)) <= 1e-12 * scale)
    assert np.all(np.abs(y - orc.spmv_inode(ai, aj, aa, x)) <= 1e-12 * scale)
    B = P.Mat.from_bsr(3, bi, bj, ba)
    L.MatMult(B.h, vx.h, vy.h)
    yb = vy.array()
    assert np.all(np.abs(yb - orc.spmv_bsr(3, bi, bj, ba, x)) <= 1e-12 * scale)
    assert np.all(np.abs(yb - y) <= 1e-12 * scale)


def solve(P, ai, aj, aa, b, ksp, pc, x0=None, opts="", comm=None, mpi=False, **tol):
    L = P.lib()
    comm = comm or L.COMM_SELF
    n_ = ai.size - 1
    A = P.Mat.from_csr_mpi(ai, aj, aa, n_, n_, n_, comm=comm) if mpi else P.Mat.from_csr(ai, aj, aa, comm=comm)
    vb = P.Vec.from_array(b, comm=comm)
    vx = P.Vec.from_array(np.zeros(b.size) if x0 is None else x0, comm=comm)
    k = P.KSP(comm=comm)
    k.set_operators(A)
    L.PetscOptionsClear()
    L.PetscOptionsInsertString(("-ksp_type %s -pc_type %s %s" % (ksp_type_for(ksp, opts), pc, opts)).encode())
    if tol:
        k.set_tolerances(**tol)
    k.set_from_options()
    if x0 is not None:
        L.KSPSetInitialGuessNonzero(k.h, 1)
    k.record_history()
    k.solve(vb, vx)
    L.PetscOptionsClear()
    return vx.array(), k.history(), k.its, k.reason


def test_ksp_config1_cg_jacobi(P):
    """BASELINE.json configs[0]: ex2 -m 100 -n 100 -ksp_type cg -pc_type jacobi -> 160 its, error 5.70785e-05"""
    ai, aj, aa = pb.lap2d(100, 100)
    u = np.ones(10000)
    b = orc.spmv(ai, aj, aa, u)
    x, h, its, reason = solve(P, ai, aj, aa, b, "cg", "jacobi", rtol=1e-2 / (101 * 101), abstol=1e-50)
    xr, hr, itsr, rr = orc.ksp_solve(ai, aj, aa, b, ksp="cg", pc="jacobi", rtol=1e-2 / (101 * 101), abstol=1e-50)
    assert abs(its - itsr) <= 1 and reason == rr == 2
    k = min(len(h), len(hr))
    assert np.allclose(h[:k], hr[:k], rtol=1e-6, atol=0)
    assert ["%g" % v for v in h[:4]] == ["5.04975", "2.54845", "1.81892", "1.67695"]
    assert "%.5g" % np.linalg.norm(x - u) == "5.7078e-05" or "%.5g" % np.linalg.norm(x - u) == "5.7079e-05"


@pytest.mark.parametrize("pc", ["jacobi", "none", "ilu"])
def test_ksp_cg_fused_forms_are_bit_identical(P, pc):
    """KSPSolve_CG's fused forms: level 1 -- with PCJACOBI one sweep does both AXPYs, the PCApply, the norm and the
    dot, otherwise norm and dot share one VecDotNorm2; level 2 (default) -- additionally p'w stays on the device and
    the sweep forms a = beta/dpi itself.  -ksp_cg_fused 0 runs the reference's op-by-op sequence (cg.c:206-232):
    iterates and the whole residual history must carry the same bits at every level."""
    ai, aj, aa = pb.lap2d(41, 37)
    n = ai.size - 1
    b = orc.spmv(ai, aj, aa, np.cos(0.3 * np.arange(n)))
    xu, hu, itsu, ru = solve(P, ai, aj, aa, b, "cg", pc, opts="-ksp_cg_fused 0", rtol=1e-9)
    assert ru == 2 and itsu > 20
    for level in ("1", "2", "3"):
        xf, hf, itsf, rf = solve(P, ai, aj, aa, b, "cg", pc, opts="-ksp_cg_fused " + level, rtol=1e-9)
        assert itsf == itsu and rf == ru
        assert np.array_equal(bits(hf), bits(hu))
        assert np.array_equal(bits(xf), bits(xu))
    # level 4 (the default): p'w is a by-product of the SpMV pass, summed in another order -> agreement to rounding
    x4 = None
    for opts4 in ("-ksp_cg_fused 4", ""):
        xf, hf, itsf, rf = solve(P, ai, aj, aa, b, "cg", pc, opts=opts4, rtol=1e-9)
        assert x4 is None or np.array_equal(bits(xf), bits(x4))               # the default IS level 4; deterministic
        x4 = xf
        assert abs(itsf - itsu) <= 1 and rf == ru
        k_ = min(len(hf), len(hu))
        assert np.allclose(hf[:k_], hu[:k_], rtol=1e-9, atol=0)
        assert np.linalg.norm(xf - xu) <= 1e-9 * np.linalg.norm(xu)


def test_ksp_cg_indefinite_exits_match_at_every_fusion_level(P):
    """A sign change of p'Ap is KSP_DIVERGED_INDEFINITE_MAT (cg.c:198) and leaves x at the previous iterate; with the
    device-resident scalar the update kernel itself has to refuse the step.  Same iteration, same x bits at all levels."""
    ai, aj, aa = pb.lap2d(12, 11)
    n = ai.size - 1
    aa = aa.copy()
    for row in (5, 40, 77):                       # negative diagonal entries: indefinite, Jacobi PC indefinite too
        k = ai[row] + int(np.where(aj[ai[row]:ai[row + 1]] == row)[0][0])
        aa[k] = -3.0
    b = np.cos(0.7 * np.arange(n))
    res = {}
    for level in ("0", "1", "2", "3", "4"):
        for pc in ("none", "jacobi"):
            x, h, its, reason = solve(P, ai, aj, aa, b, "cg", pc, opts="-ksp_cg_fused " + level, rtol=1e-12, max_it=200)
            res[level, pc] = (bits(x).copy(), bits(h).copy(), its, reason)
    for pc in ("none", "jacobi"):
        assert res["0", pc][3] in (-8, -10)       # KSP_DIVERGED_INDEFINITE_PC / _MAT
        for level in ("1", "2", "3"):
            assert res[level, pc][2:] == res["0", pc][2:]
            assert np.array_equal(res[level, pc][0], res["0", pc][0]) and np.array_equal(res[level, pc][1], res["0", pc][1])
        assert res["4", pc][2:] == res["0", pc][2:]          # same exit at the same iteration; x to rounding
        x4, x0 = res["4", pc][0].view(np.float64), res["0", pc][0].view(np.float64)
        assert np.linalg.norm(x4 - x0) <= 1e-9 * max(np.linalg.norm(x0), 1e-300)
    assert any(res["0", pc][3] == -10 for pc in ("none", "jacobi")), "no case exercised the INDEFINITE_MAT exit"


def test_device_collectives_on_a_one_rank_rccl_communicator(P):
    """The several-GPU reduction path (result in HBM scratch -> RCCL all-reduce in place on the compute stream -> copy
    kernel to pinned memory -> one stream synchronisation; device-resident p'w for CG) run on ONE GPU by attaching a
    one-rank RCCL communicator to a one-rank PetscComm.  Everything must equal the plain one-rank path bit for bit."""
    import ctypes as C
    import importlib
    PD = importlib.import_module("petsc-dev_amd.dist")
    L = P.lib()
    k = P.load_kernels()
    comm = PD.make_comm(0, 1, lambda b_: [b_], lambda a, op: a, lambda: None)
    uid = C.create_string_buffer(128)
    assert k.mi355x_comm_get_unique_id(uid) == 0
    dcomm = C.c_void_p()
    rc = k.mi355x_comm_init_rank(C.byref(dcomm), 1, 0, uid.raw)
    assert rc == 0, k.mi355x_comm_error_string(rc).decode()
    # as the launcher arranges it: a second communicator for what is queued on the halo stream
    uid2 = C.create_string_buffer(128)
    assert k.mi355x_comm_get_unique_id(uid2) == 0
    hcomm = C.c_void_p()
    rc = k.mi355x_comm_init_rank(C.byref(hcomm), 1, 0, uid2.raw)
    assert rc == 0, k.mi355x_comm_error_string(rc).decode()
    assert PD.transport_report(comm) == {"transport": "single", "rccl_ranks": 0, "rccl_communicators": 0}
    L.PetscCommSetDeviceComms(comm, dcomm, hcomm)
    assert PD.transport_report(comm) == {"transport": "rccl", "rccl_ranks": 1, "rccl_communicators": 2}
    try:
        # the launcher's self test: one all-reduce and one grouped ncclSend/ncclRecv (to itself on one rank) through the
        # C wrappers of include/mi355x_comm.h
        assert PD._rccl_self_test(k, dcomm, 0, 1) == "" and PD._rccl_self_test(k, hcomm, 0, 1) == ""
        ai, aj, aa = pb.lap2d(33, 29)
        n = ai.size - 1
        b = orc.spmv(ai, aj, aa, np.cos(0.3 * np.arange(n)))
        x = V(P, rnd(n, 5)); y = V(P, rnd(n, 6))
        xc = P.Vec.from_array(x.array(), comm=comm); yc = P.Vec.from_array(y.array(), comm=comm)
        for nt in (0, 1, 2, 3):
            assert x.norm(nt) == xc.norm(nt)
        assert x.dot(y) == xc.dot(yc)
        # groppcg: its split-phase all-reduce travels on the HALO stream and is published from the halo handle
        for ksp, pc in (("cg", "jacobi"), ("cg", "none"), ("gmres", "jacobi"), ("bcgs", "jacobi"), ("groppcg", "jacobi"), ("groppcg", "bjacobi")):
            o = "-ksp_cg_fused 3" if ksp == "cg" else ""      # (level 4 takes p'w out of the sequential SpMV pass; a parallel matrix runs level 3)
            ref = solve(P, ai, aj, aa, b, ksp, pc, rtol=1e-9, opts=o)
            got = solve(P, ai, aj, aa, b, ksp, pc, rtol=1e-9, comm=comm, opts=o)
            assert got[2:] == ref[2:]
            assert np.array_equal(bits(got[1]), bits(ref[1])) and np.array_equal(bits(got[0]), bits(ref[0]))
            # the same through a MATMPIAIJ on that communicator (MatMult_MPIAIJ's stream choreography, parallel Vec
            # type, the queued-ahead CG front half over it); everything is in the diagonal block, so same bits again
            got = solve(P, ai, aj, aa, b, ksp, pc, rtol=1e-9, comm=comm, mpi=True, opts=o)
            assert got[2:] == ref[2:]
            assert np.array_equal(bits(got[1]), bits(ref[1])) and np.array_equal(bits(got[0]), bits(ref[0]))
    finally:
        L.PetscCommSetDeviceComm(comm, None)
        k.mi355x_comm_destroy(dcomm)
        k.mi355x_comm_destroy(hcomm)


@pytest.mark.parametrize("pc", ["jacobi", "none", "ilu"])
def test_ksp_bcgs_fused_forms_are_bit_identical(P, pc):
    """KSPSolve_BCGS with the PCApply fused into the dots that follow it and the x/r update fused with the norm and the
    next rho (default) against the reference's op-by-op sequence (-ksp_bcgs_fused 0, bcgs.c:98-150): same iteration
    count, same history bits, same x bits (nonsymmetric operator so that the two-sided recurrences matter)."""
    ai, aj, aa = pb.lap2d(37, 33)
    n = ai.size - 1
    aa = aa * (1.0 + 0.3 * np.sin(np.arange(aa.size)))        # nonsymmetric values on the symmetric pattern
    rows = np.repeat(np.arange(n), np.diff(ai))
    aa[aj == rows] = 5.0                                      # keep it diagonally dominant
    b = np.cos(0.3 * np.arange(n))
    xu, hu, itsu, ru = solve(P, ai, aj, aa, b, "bcgs", pc, opts="-ksp_bcgs_fused 0", rtol=1e-9)
    xf, hf, itsf, rf = solve(P, ai, aj, aa, b, "bcgs", pc, rtol=1e-9)
    assert ru == 2 and itsu >= 4 and itsf == itsu and rf == ru
    assert np.array_equal(bits(hf), bits(hu))
    assert np.array_equal(bits(xf), bits(xu))


@pytest.mark.parametrize("norm", ["preconditioned", "unpreconditioned", "natural", "none"])
@pytest.mark.parametrize("pc", ["jacobi", "ilu"])
def test_ksp_cg_norm_types(P, norm, pc):
    """KSPSolve_CG's four norm types (cg.c:136-161,233-260; -ksp_norm_type): the history against the oracle's
    restatement, and the fused sweeps (level 3) against the op-by-op sequence bit for bit (the fused sweep also returns r'r,
    so every norm comes out of the same reduction).  KSP_NORM_NONE runs to max_it and ends KSP_CONVERGED_ITS (4)."""
    ai, aj, aa = pb.lap2d(33, 29)
    n = ai.size - 1
    d = 1.0 + 0.5 * np.sin(np.arange(n))
    rows = np.repeat(np.arange(n), np.diff(ai))
    aa = aa * d[rows] * d[aj]
    b = orc.spmv(ai, aj, aa, np.cos(0.3 * np.arange(n)))
    kw = dict(rtol=1e-8, max_it=37 if norm == "none" else 500)
    o = "-ksp_norm_type " + norm
    xf, hf, itsf, rf = solve(P, ai, aj, aa, b, "cg", pc, opts=o + " -ksp_cg_fused 3", **kw)
    xu, hu, itsu, ru = solve(P, ai, aj, aa, b, "cg", pc, opts=o + " -ksp_cg_fused 0", **kw)
    xd, hd, itsd, rd = solve(P, ai, aj, aa, b, "cg", pc, opts=o, **kw)          # the default (level 4: p'w out of the SpMV pass): to rounding
    assert rd == ru and abs(itsd - itsu) <= 1 and np.linalg.norm(xd - xu) <= 1e-7 * np.linalg.norm(xu)
    nt = dict(none=0, preconditioned=1, unpreconditioned=2, natural=3)[norm]
    xr, hr, itsr, rr = orc.ksp_solve(ai, aj, aa, b, ksp="cg", pc=pc, norm_type=nt, **kw)
    assert itsf == itsu and rf == ru and np.array_equal(bits(hf), bits(hu)) and np.array_equal(bits(xf), bits(xu))
    assert rf == rr and abs(itsf - itsr) <= 1
    if norm == "none":
        assert rf == 4 and itsf == 37 and np.all(hf[:38] == 0.0)
    else:
        assert rf == 2
        k = min(len(hf), len(hr))
        assert np.allclose(hf[:k], hr[:k], rtol=1e-6, atol=1e-14 * hr[0])
    assert np.linalg.norm(xf - xr) <= 1e-6 * np.linalg.norm(xr)
    if norm != "none":
        # nonzero initial guess: the relative tolerance is then measured against the norm of the right-hand side in
        # the selected norm (KSPDefaultConverged, iterativ.c:718-737)
        x0 = 0.3 * np.sin(np.arange(n))
        xg, hg, itsg, rg = solve(P, ai, aj, aa, b, "cg", pc, x0=x0, opts=o, **kw)
        xo, ho, itso, ro = orc.ksp_solve(ai, aj, aa, b, ksp="cg", pc=pc, x0=x0, norm_type=nt, **kw)
        assert rg == ro == 2 and abs(itsg - itso) <= 1
        k = min(len(hg), len(ho))
        assert np.allclose(hg[:k], ho[:k], rtol=1e-6, atol=1e-14 * ho[0])


def test_ksp_bcgs_as_smoother_without_norms(P):
    """-ksp_norm_type none with BiCGStab (bcgs.c:76,131: no VecNorm, KSPSkipConverged): exactly max_it iterations,
    KSP_CONVERGED_ITS, zero history, and the iterate of the normal run after the same number of iterations"""
    ai, aj, aa = pb.lap2d(21, 19)
    n = ai.size - 1
    b = np.cos(0.3 * np.arange(n))
    x0, h0, its0, r0 = solve(P, ai, aj, aa, b, "bcgs", "jacobi", opts="-ksp_norm_type none", max_it=6)
    xr, hr, itsr, rr = orc.ksp_solve(ai, aj, aa, b, ksp="bcgs", pc="jacobi", norm_type=0, max_it=6)
    x1, h1, its1, r1 = solve(P, ai, aj, aa, b, "bcgs", "jacobi", rtol=1e-30, max_it=6)
    assert (its0, r0) == (6, 4) and (itsr, rr) == (6, 4) and np.all(h0[:7] == 0.0)
    assert np.array_equal(bits(x0), bits(x1))
    assert np.linalg.norm(x0 - xr) <= 1e-10 * np.linalg.norm(xr)


@pytest.mark.parametrize("pc", ["jacobi", "ilu", "bjacobi"])
def test_ksp_gmres_right_preconditioning(P, pc):
    """-ksp_pc_side right with KSPGMRES (KSPInitialResidual itres.c:55-64, PCApplyBAorAB precon.c:617, the unwinding in
    KSPGMRESBuildSoln gmres.c:343-346): the monitored norm is the TRUE residual norm; history and solution against the
    oracle's restatement, with a restart inside the solve and a nonzero initial guess."""
    ai, aj, aa = pb.lap2d(23, 19)
    n = ai.size - 1
    aa = aa * (1.0 + 0.3 * np.sin(np.arange(aa.size)))
    rows = np.repeat(np.arange(n), np.diff(ai))
    aa[aj == rows] = 5.0 + np.cos(np.arange(n))
    xs = np.cos(0.3 * np.arange(n))
    b = orc.spmv(ai, aj, aa, xs)
    x0 = 0.1 * np.sin(np.arange(n))
    x, h, its, reason = solve(P, ai, aj, aa, b, "gmres", pc, x0=x0, opts="-ksp_pc_side right -ksp_gmres_restart 7", rtol=1e-9)
    okw = dict(blocks=[0, n], sub_ksp="preonly", sub_pc="ilu") if pc == "bjacobi" else {}
    xr, hr, itsr, rr = orc.ksp_solve(ai, aj, aa, b, ksp="gmres", pc=pc, x0=x0, pc_right=1, restart=7, rtol=1e-9, **okw)
    assert reason == rr == 2 and abs(its - itsr) <= 1 and its > 7
    k = min(len(h), len(hr))
    assert np.allclose(h[:k], hr[:k], rtol=1e-6, atol=1e-14 * hr[0])
    assert np.linalg.norm(x - xr) <= 1e-7 * np.linalg.norm(xr)
    # the first monitored value is the true residual norm of the initial guess
    assert abs(h[0] - np.linalg.norm(b - orc.spmv(ai, aj, aa, x0))) <= 1e-12 * np.linalg.norm(b)
    # right preconditioning is refused where it is not ported
    with pytest.raises(P.PetscError):
        solve(P, ai, aj, aa, b, "cg", pc, opts="-ksp_pc_side right", rtol=1e-9)
    P.lib().PetscOptionsClear()


def test_monitor_text_is_the_reference_output(P, capfd):
    """-ksp_monitor_short / -ksp_converged_reason print the reference's own text (KSPMonitorDefaultShort iterativ.c:484,
    itfunc.c:662): the stdout of `ex2 -m 5 -n 5 -ksp_monitor_short -ksp_gmres_cgs_refinement_type refine_always`
    replayed here must equal the monitor lines of output/ex2_1.out character for character, and CG + Jacobi on ex2's
    100 x 100 grid must print config 1's first residuals."""
    ai, aj, aa = pb.lap2d(5, 5)
    u = np.ones(25)
    b = orc.spmv(ai, aj, aa, u)
    capfd.readouterr()
    x, h, its, reason = solve(P, ai, aj, aa, b, "gmres", "ilu", opts="-ksp_monitor_short -ksp_converged_reason -ksp_gmres_cgs_refinement_type refine_always",
                              rtol=1e-2 / 36, abstol=1e-50)
    out = capfd.readouterr().out.splitlines()
    gold = open(os.path.join(G, "ksp_tutorials", "ex2_1.out")).read().splitlines()
    assert out[:5] == gold[:5]
    assert out[5] == "Linear solve converged due to CONVERGED_RTOL iterations 4"
    assert "Norm of error %g iterations %d" % (np.linalg.norm(x - u), its) == gold[5]
    ai, aj, aa = pb.lap2d(100, 100)
    b = orc.spmv(ai, aj, aa, np.ones(10000))
    solve(P, ai, aj, aa, b, "cg", "jacobi", opts="-ksp_monitor_short", rtol=1e-2 / (101 * 101), abstol=1e-50)
    out = capfd.readouterr().out.splitlines()
    assert out[:4] == ["  0 KSP Residual norm 5.04975 ", "  1 KSP Residual norm 2.54845 ", "  2 KSP Residual norm 1.81892 ", "  3 KSP Residual norm 1.67695 "]
    assert out[160] == "160 KSP Residual norm 4.47805e-06 " and len(out) == 161
    solve(P, ai, aj, aa, b, "cg", "jacobi", opts="-ksp_monitor", rtol=1e-2 / (101 * 101), abstol=1e-50, max_it=1)
    out = capfd.readouterr().out.splitlines()
    assert out[0].startswith("  0 KSP Residual norm 5.04975") and out[0].endswith("e+00 ") and len(out[0]) == len("  0 KSP Residual norm 5.049752469181e+00 ")


@pytest.mark.parametrize("norm", ["natural", "unpreconditioned"])
def test_ksp_cg_single_reduction_with_other_norms(P, norm):
    """-ksp_cg_single_reduction combined with the natural / unpreconditioned norm (cg.c:141-147,247-270): against the
    oracle's restatement of the same branches"""
    ai, aj, aa = pb.lap2d(31, 27)
    n = ai.size - 1
    b = orc.spmv(ai, aj, aa, np.cos(0.2 * np.arange(n)))
    nt = dict(unpreconditioned=2, natural=3)[norm]
    x, h, its, reason = solve(P, ai, aj, aa, b, "cg", "jacobi", opts="-ksp_cg_single_reduction 1 -ksp_norm_type " + norm, rtol=1e-8)
    xr, hr, itsr, rr = orc.ksp_solve(ai, aj, aa, b, ksp="cg", pc="jacobi", rtol=1e-8, cg_single=1, norm_type=nt)
    assert reason == rr == 2 and abs(its - itsr) <= 1
    k = min(len(h), len(hr))
    assert np.allclose(h[:k], hr[:k], rtol=1e-6, atol=1e-14 * hr[0])
    assert np.linalg.norm(x - xr) <= 1e-6 * np.linalg.norm(xr)


@pytest.mark.parametrize("norm", ["preconditioned", "unpreconditioned", "natural", "none"])
def test_ksp_groppcg(P, norm):
    """KSPGROPPCG (groppcg.c:40-175) on split-phase reductions (VecDotBegin/End, VecNormBegin/End,
    PetscCommSplitReductionBegin): history against the oracle's restatement for every norm type, same solution as
    KSPCG, and the split-reduction calls refuse a wrong Begin/End order as comb.c does."""
    ai, aj, aa = pb.lap2d(37, 31)
    n = ai.size - 1
    d = 1.0 + 0.5 * np.sin(np.arange(n))
    rows = np.repeat(np.arange(n), np.diff(ai))
    aa = aa * d[rows] * d[aj]
    b = orc.spmv(ai, aj, aa, np.cos(0.3 * np.arange(n)))
    nt = dict(none=0, preconditioned=1, unpreconditioned=2, natural=3)[norm]
    kw = dict(rtol=1e-8, max_it=41 if norm == "none" else 500)
    for pc in ("jacobi", "bjacobi"):
        x, h, its, reason = solve(P, ai, aj, aa, b, "groppcg", pc, opts="-ksp_norm_type " + norm, **kw)
        okw = dict(blocks=[0, n], sub_ksp="preonly", sub_pc="ilu") if pc == "bjacobi" else {}
        xr, hr, itsr, rr = orc.ksp_solve(ai, aj, aa, b, ksp="groppcg", pc=pc, norm_type=nt, **kw, **okw)
        assert reason == rr and abs(its - itsr) <= 1
        k = min(len(h), len(hr))
        assert np.allclose(h[:k], hr[:k], rtol=1e-6, atol=1e-14 * max(hr[0], 1e-300))
        assert np.linalg.norm(x - xr) <= 1e-6 * np.linalg.norm(xr)
        if norm == "preconditioned":
            xc, hc, itsc, rc = solve(P, ai, aj, aa, b, "cg", pc, **kw)
            assert np.linalg.norm(x - xc) <= 1e-6 * np.linalg.norm(xc) and abs(its - itsc) <= 2
    # Begin/End discipline
    L = P.lib()
    u, v = V(P, rnd(100, 1)), V(P, rnd(100, 2))
    r1, r2 = C.c_double(), C.c_double()
    L.VecDotBegin(u.h, v.h, C.byref(r1)); L.VecNormBegin(u.h, 1, C.byref(r2))
    L.PetscCommSplitReductionBegin(L.COMM_SELF)
    with pytest.raises(P.PetscError):
        L.VecNormEnd(u.h, 1, C.byref(r2))                  # the dot was begun first
    L.VecDotEnd(u.h, v.h, C.byref(r1)); L.VecNormEnd(u.h, 1, C.byref(r2))
    assert r1.value == u.dot(v) and r2.value == u.norm()


@pytest.mark.parametrize("norm", ["natural", "none", "preconditioned", "unpreconditioned"])
def test_ksp_pipecg(P, norm):
    """KSPPIPECG (pipecg.c:49-205) as this snapshot has it: history against the oracle's restatement for every norm type.
    With the natural norm (and none) it is CG -- same solution and iteration count as KSPCG; with the (un)preconditioned norm
    the reference reduces gamma = (r,u) in iteration 0 only (pipecg.c:124-131) and the recurrence runs with beta = 1: the
    port reproduces that walk too, for the first iterations (it does not converge)."""
    ai, aj, aa = pb.lap2d(37, 31)
    n = ai.size - 1
    d = 1.0 + 0.5 * np.sin(np.arange(n))
    rows = np.repeat(np.arange(n), np.diff(ai))
    aa = aa * d[rows] * d[aj]
    b = orc.spmv(ai, aj, aa, np.cos(0.3 * np.arange(n)))
    nt = dict(none=0, preconditioned=1, unpreconditioned=2, natural=3)[norm]
    works = norm in ("natural", "none")
    kw = dict(rtol=1e-8, max_it=(41 if norm == "none" else 500) if works else 12)
    for pc in ("jacobi", "none"):
        x, h, its, reason = solve(P, ai, aj, aa, b, "pipecg", pc, opts="-ksp_norm_type " + norm, **kw)
        xr, hr, itsr, rr = orc.ksp_solve(ai, aj, aa, b, ksp="pipecg", pc=pc, norm_type=nt, **kw)
        assert reason == rr and abs(its - itsr) <= (1 if works else 0)
        k = min(len(h), len(hr))
        assert np.allclose(h[:k], hr[:k], rtol=1e-6 if works else 1e-9, atol=1e-14 * max(hr[0], 1e-300))
        assert np.linalg.norm(x - xr) <= 1e-6 * np.linalg.norm(xr)
        if norm == "natural":
            xc, hc, itsc, rc = solve(P, ai, aj, aa, b, "cg", pc, opts="-ksp_norm_type natural", **kw)
            assert rc == reason and np.linalg.norm(x - xc) <= 1e-6 * np.linalg.norm(xc) and abs(its - itsc) <= 2


def test_ksp_cg_single_reduction(P):
    """-ksp_cg_single_reduction (cg.c:116-122,200-203,263-270): two reductions per iteration instead of
    three (VecMDot(2) for delta and beta), A*p by recurrence; same op sequence as the oracle's restatement"""
    ai, aj, aa = pb.lap2d(40, 37)
    n = ai.size - 1
    b = orc.spmv(ai, aj, aa, np.ones(n))
    x, h, its, reason = solve(P, ai, aj, aa, b, "cg", "jacobi", opts="-ksp_cg_single_reduction 1", rtol=1e-8)
    xr, hr, itsr, rr = orc.ksp_solve(ai, aj, aa, b, ksp="cg", pc="jacobi", rtol=1e-8, cg_single=1)
    x0, h0, its0, _ = orc.ksp_solve(ai, aj, aa, b, ksp="cg", pc="jacobi", rtol=1e-8)
    assert reason == rr == 2 and abs(its - itsr) <= 1 and abs(itsr - its0) <= 2
    k = min(len(h), len(hr))
    assert np.allclose(h[:k], hr[:k], rtol=1e-6, atol=0)
    assert np.linalg.norm(x - 1.0) < 1e-6


@pytest.mark.parametrize("fused", [0, 1, 3, 4])
def test_ksp_cg_pc_tests_ex2_golden(P, fused):
    """src/ksp/pc/examples/tests/ex2.c -ksp_type cg -ksp_monitor_short vs output/ex2_1.out (the reference's own CG
    golden: tridiagonal n = 10, PCNONE, 5 iterations) on the HIP path, op-by-op and fused forms"""
    ai, aj, aa = pb.tridiag(10)
    b = orc.spmv(ai, aj, aa, np.ones(10))
    gold = pb.parse_monitor(os.path.join(G, "pc_tests", "ex2_1.out"))[0]
    x, h, its, reason = solve(P, ai, aj, aa, b, "cg", "none", opts="-ksp_cg_fused %d" % fused)
    pb.check_monitor(h, gold)
    assert its == 5 and reason == 2 and np.linalg.norm(x - 1.0) <= 1e-14


def test_ksp_golden_ex4_and_ex5_two_systems(P):
    """ksp/examples/tests/ex4.c (ex4_1.out) and tutorials/ex5.c (ex5_1.out: two solves with ONE KSP, the second after
    MatZeroEntries + re-assembly of new values into the same pattern -- the device copy is refreshed, not rebuilt)"""
    L = P.lib()
    (ai, aj, aa), b, u0, ustar = pb.ex3_fem(5)
    gold = pb.parse_monitor(os.path.join(G, "ksp_tests", "ex4_1.out"))[0]
    x, h, its, reason = solve(P, ai, aj, aa, b, "gmres", "jacobi", x0=u0, opts="-ksp_gmres_cgs_refinement_type refine_always")
    pb.check_monitor(h, gold)
    solves = pb.parse_monitor(os.path.join(G, "ksp_tutorials", "ex5_1.out"))
    (ai, aj, aa), u = pb.ex5_tutorial(1, False)
    A = P.Mat.from_csr(ai, aj, aa)
    vu = V(P, u); vb = vu.duplicate(); vx = vu.duplicate()
    k = P.KSP(comm=L.COMM_SELF)
    k.set_operators(A)
    L.PetscOptionsClear()
    L.PetscOptionsInsertString(b"-ksp_type gmres -pc_type jacobi -ksp_gmres_cgs_refinement_type refine_always")
    k.set_from_options()
    L.PetscOptionsClear()
    for second in (False, True):
        if second:
            (ai2, aj2, aa2), _ = pb.ex5_tutorial(1, True)
            L.MatZeroEntries(A.h)
            rows = np.repeat(np.arange(ai2.size - 1, dtype=np.int32), np.diff(ai2))
            for r, c, v in zip(rows, aj2, aa2):     # the example's loop of MatSetValues(ADD_VALUES)
                L.MatSetValues(A.h, 1, C.byref(C.c_int(int(r))), 1, C.byref(C.c_int(int(c))), C.byref(C.c_double(float(v))), P.ADD_VALUES)
            L.MatAssemblyBegin(A.h, P.MAT_FINAL_ASSEMBLY); L.MatAssemblyEnd(A.h, P.MAT_FINAL_ASSEMBLY)
            k.set_operators(A)
        A.mult(vu, vb)
        k.record_history()
        k.solve(vb, vx)
        pb.check_monitor(k.history(), solves[1 if second else 0])
        assert np.linalg.norm(vx.array() - u) < 1e-4 * np.linalg.norm(u)


@pytest.mark.parametrize("name,rtol,fused", [("ex1_1.out", 1e-5, 1), ("ex23_1.out", 1e-7, 1), ("ex23_1.out", 1e-7, 0)])
def test_ksp_golden_ex1_ex23_happy_breakdown(P, name, rtol, fused):
    """tutorials ex1.c / ex23.c vs output/ex1_1.out, ex23_1.out on the HIP path: GMRES + PCJACOBI on the tridiagonal n = 10
    system ends in its happy-breakdown branch after 5 steps ('< 1.e-11'), fused or op by op"""
    ai, aj, aa = pb.tridiag(10)
    b = orc.spmv(ai, aj, aa, np.ones(10))
    gold = pb.parse_monitor(os.path.join(G, "ksp_tutorials", name))[0]
    x, h, its, reason = solve(P, ai, aj, aa, b, "gmres", "jacobi", opts="-ksp_gmres_cgs_refinement_type refine_always -ksp_gmres_fused %d" % fused, rtol=rtol)
    pb.check_monitor(h, gold)
    assert its == 5 and reason > 0 and np.linalg.norm(x - 1.0) <= 1e-11


def test_ksp_golden_ex3_ex2f_ex9(P):
    (ai, aj, aa), b, u0, ustar = pb.ex3_fem(5)
    gold = pb.parse_monitor(os.path.join(G, "ksp_tests", "ex3_1.out"))[0]
    x, h, its, reason = solve(P, ai, aj, aa, b, "gmres", "jacobi", x0=u0, opts="-ksp_gmres_cgs_refinement_type refine_always")
    pb.check_monitor(h, gold)
    assert np.linalg.norm(x - ustar) <= 1e-12
    ai, aj, aa = pb.lap2d(3, 3)
    b = orc.spmv(ai, aj, aa, np.ones(9))
    gold = pb.parse_monitor(os.path.join(G, "ksp_tutorials", "ex2f_1.out"))[0]
    x, h, its, reason = solve(P, ai, aj, aa, b, "gmres", "jacobi", opts="-ksp_gmres_cgs_refinement_type refine_always")
    pb.check_monitor(h, gold)
    solves = pb.parse_monitor(os.path.join(G, "ksp_tutorials", "ex9_1.out"))
    (ai, aj, aa), u = pb.ex9_system(1, 0)
    x, h, its, reason = solve(P, ai, aj, aa, orc.spmv(ai, aj, aa, u), "gmres", "jacobi", opts="-ksp_gmres_cgs_refinement_type refine_always")
    pb.check_monitor(h, solves[0])
    for t, g in ((0, solves[1]), (1, solves[3])):
        (ai, aj, aa), u = pb.ex9_system(2, t)
        x, h, its, reason = solve(P, ai, aj, aa, orc.spmv(ai, aj, aa, u), "bcgs", "jacobi")
        pb.check_monitor(h, g)
        assert np.linalg.norm(x - u) < 1e-4


def test_icc0_apply_bitexact_and_golden(P):
    """SURVEY 8f.1 names ILU(0)/ICC(0): PCICC (zero fill, natural ordering).  The device solve reproduces
    MatSolve_SeqSBAIJ_1_NaturalOrdering bit for bit (U^T sweep as the row form that adds in the same order, D^-1 in between, U sweep
    with the rows read last entry first), replays included; a matrix that needs MatPivotCheck_pd's diagonal shift takes the same
    number of shifts as the oracle; ksp/tests/ex32.c -ksp_type cg -pc_type icc (natural ordering, levels 0) prints output/ex32_5.out's
    first block; and CG + block Jacobi(ICC) converges like the oracle's"""
    L = P.lib()
    cases = [pb.lap2d(9, 7), P.gen_poisson7(7, 6, 5), P.gen_poisson7(12, 11, 10), pb.ex32()[0]]
    # symmetric, varying coefficients (scaled symmetrically), and one that is NOT diagonally dominant enough: shifts
    ai, aj, aa = P.gen_poisson7(8, 7, 6)
    s_ = 1.0 + 0.3 * np.sin(np.arange(ai.size - 1))
    rows = np.repeat(np.arange(ai.size - 1), np.diff(ai))
    cases.append((ai, aj, aa * s_[rows] * s_[aj]))
    ai2, aj2, aa2 = pb.lap2d(8, 8)
    aa2 = aa2.copy(); aa2[aj2 == np.repeat(np.arange(ai2.size - 1), np.diff(ai2))] = 1.5      # diagonal 1.5 against four -1: indefinite
    cases.append((ai2, aj2, aa2))
    for ci, (ai, aj, aa) in enumerate(cases):
        n = ai.size - 1
        A = P.Mat.from_csr(ai, aj, aa)
        pc = C.c_void_p()
        k = P.KSP(comm=L.COMM_SELF); k.set_operators(A); L.KSPGetPC(k.h, C.byref(pc)); L.PCSetType(pc, b"icc")
        L.raw("PCSetUp")(pc)
        f, nshift = orc.icc0_factor(ai, aj, aa)
        nl, nu, ns = C.c_int(), C.c_int(), C.c_int()
        L.PCICCGetInfo_HIPMI355X(pc, C.byref(nl), C.byref(nu), C.byref(ns))
        assert ns.value == nshift and (nshift > 0) == (ci == len(cases) - 1) and nl.value == nu.value and nl.value > 1
        vb, vx = V(P, np.zeros(n)), V(P, np.zeros(n))
        for rep in range(3):                            # replays: the sentinel of the hand-off buffers is restored every time
            b2 = rnd(n, 90 + rep); vb.set_array(b2)
            L.raw("PCApply")(pc, vb.h, vx.h)
            assert np.array_equal(bits(vx.array()), bits(orc.icc0_solve(f, b2)))
    (ai, aj, aa), b = pb.ex32()
    gold = pb.parse_monitor(os.path.join(G, "ksp_tests", "ex32_5.out"))[0]
    x, h, its, reason = solve(P, ai, aj, aa, b, "cg", "icc")
    pb.check_monitor(h, gold)
    xo, ho, itso, ro = orc.ksp_solve(ai, aj, aa, b, ksp="cg", pc="icc")
    assert (its, reason) == (itso, ro) and np.allclose(x, xo, rtol=1e-12, atol=1e-15)
    # the SPD default under block Jacobi: several ICC(0) blocks per process
    ai, aj, aa = P.gen_poisson7(10, 9, 8)
    n = ai.size - 1
    b = orc.spmv(ai, aj, aa, np.ones(n))
    x, h, its, reason = solve(P, ai, aj, aa, b, "cg", "bjacobi", opts="-pc_bjacobi_blocks 4 -sub_pc_type icc", rtol=1e-8)
    blocks = [0, n // 4, n // 2, 3 * n // 4, n]
    blocks = [0] + [sum([(n // 4) + (1 if i < n % 4 else 0) for i in range(j + 1)]) for j in range(4)]
    xo, ho, itso, ro = orc.ksp_solve(ai, aj, aa, b, ksp="cg", pc="bjacobi", blocks=blocks, sub_pc="icc", rtol=1e-8)
    assert reason == ro == 2 and abs(its - itso) <= 1 and np.allclose(h[:min(len(h), len(ho))], ho[:min(len(h), len(ho))], rtol=1e-8)
    assert np.linalg.norm(x - 1.0) < 1e-6 * np.sqrt(n)
    # the blocks solved as ONE block-diagonal system (default) against block after block (-pc_bjacobi_merge_blocks 0): same bits,
    # also when one block needs the positive-definite shift and the others do not (each block keeps its own shift loop)
    ai, aj, aa = pb.lap2d(16, 8)
    n = ai.size - 1
    aa = aa.copy()
    rows = np.repeat(np.arange(n), np.diff(ai))
    aa[(aj == rows) & (rows >= 3 * n // 4)] = 1.5          # the last quarter of the rows: not diagonally dominant
    b = np.cos(0.3 * np.arange(n)) + 0.2
    runs = [solve(P, ai, aj, aa, b, "gmres", "bjacobi", opts="-pc_bjacobi_blocks 4 -sub_pc_type icc " + o, rtol=1e-9) for o in ("", "-pc_bjacobi_merge_blocks 0")]
    assert runs[0][2] == runs[1][2] and runs[0][2] > 3 and runs[0][3] == runs[1][3] == 2
    assert np.array_equal(bits(runs[0][1]), bits(runs[1][1])) and np.array_equal(bits(runs[0][0]), bits(runs[1][0]))
    blocks = [0, n // 4, n // 2, 3 * n // 4, n]
    xo, ho, itso, ro = orc.ksp_solve(ai, aj, aa, b, ksp="gmres", pc="bjacobi", blocks=blocks, sub_pc="icc", rtol=1e-9)
    assert itso == runs[0][2] and np.allclose(runs[0][1], ho, rtol=1e-7) and np.allclose(runs[0][0], xo, rtol=1e-9, atol=1e-12)
    # the same for ILU(0) blocks with the nonzero shift: one block has a zero pivot (tridiag(1,1,1) rows), the others do not
    import scipy.sparse as sp
    T = sp.block_diag([sp.diags([-np.ones(19), 4.0 * np.ones(20), -np.ones(19)], [-1, 0, 1]), sp.diags([np.ones(19), np.ones(20), np.ones(19)], [-1, 0, 1]),
                       sp.diags([-np.ones(19), 3.0 * np.ones(20), -np.ones(19)], [-1, 0, 1])]).tolil()
    T[19, 20] = T[20, 19] = 0.25; T[39, 40] = T[40, 39] = 0.25      # couplings between the blocks (dropped by block Jacobi)
    T = T.tocsr(); T.sort_indices()
    ai, aj, aa = T.indptr.astype(np.int32), T.indices.astype(np.int32), T.data.copy()
    b = np.cos(0.3 * np.arange(60)) + 0.2
    runs = [solve(P, ai, aj, aa, b, "gmres", "bjacobi", opts="-pc_bjacobi_blocks 3 " + o, rtol=1e-9, max_it=40) for o in ("", "-pc_bjacobi_merge_blocks 0", "-sub_pc_type ilu")]
    for r in runs[1:]:
        assert r[2] == runs[0][2] and np.array_equal(bits(r[1]), bits(runs[0][1])) and np.array_equal(bits(r[0]), bits(runs[0][0]))


@pytest.mark.parametrize("kernels", ["split", "onewave", "onewave_hostbuild"])
@pytest.mark.parametrize("seed", range(12))
def test_factor_fuzz_ilu0_icc0(P, seed, kernels, monkeypatch):
    """random symmetric sparsity patterns (random density, a band plus scattered entries, sometimes empty off-diagonal rows) with diagonals
    from strongly dominant to not dominant at all: ILU(0) and ICC(0) on the device take the oracle's number of shifts and apply
    with its bits -- through both families of sync-free kernels: narrow dependency levels run the split-role kernels with every row a
    node of its own (the default here), MI355X_TRISOLVE_SPLIT=0 keeps the one-wavefront-per-slice kernels that wide levels use,
    their plans laid out on the device (csrc/trisolve_build.hip: radix sort + fill kernels) or, MI355X_TRISOLVE_BUILD=host, by the
    host threads: the same plan, the same bits"""
    import scipy.sparse as sp
    L = P.lib()
    monkeypatch.setenv("MI355X_TRISOLVE_SPLIT", "1" if kernels == "split" else "0")
    monkeypatch.setenv("MI355X_TRISOLVE_BUILD", "host" if kernels == "onewave_hostbuild" else "device")
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.integers(2, 2500))
    R = sp.random(n, n, density=min(1.0, float(rng.uniform(1.0, 6.0)) / n), random_state=int(rng.integers(1 << 30)), data_rvs=rng.standard_normal)
    B = sp.diags([rng.standard_normal(n - 1)], [1]) if seed % 2 else sp.csr_matrix((n, n))
    S = (R + R.T + B + B.T).tolil()
    S.setdiag(0.0)
    S = S.tocsr(); S.eliminate_zeros()
    rowsum = np.asarray(abs(S).sum(axis=1)).ravel()
    dom = [2.0, 1.05, 0.7, 0.3][seed % 4]                      # < 1: not diagonally dominant -> pivot shifts
    A = (S + sp.diags(dom * rowsum + (0.5 if seed % 3 else 0.0) + 1e-3)).tocsr(); A.sort_indices()
    ai, aj, aa = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy()
    bvec = rng.standard_normal(n)
    for pct in ("icc", "ilu"):
        Am = P.Mat.from_csr(ai, aj, aa)
        pc = C.c_void_p()
        k = P.KSP(comm=L.COMM_SELF); k.set_operators(Am); L.KSPGetPC(k.h, C.byref(pc)); L.PCSetType(pc, pct.encode())
        set_options(L, "-pc_factor_hipmi355x_trisolve syncfree")
        L.raw("PCSetUp")(pc)
        set_options(L, "")
        vb, vx = V(P, bvec), V(P, np.zeros(n))
        L.raw("PCApply")(pc, vb.h, vx.h)
        ns = C.c_int()
        if pct == "icc":
            f, nso = orc.icc0_factor(ai, aj, aa)
            ref = orc.icc0_solve(f, bvec)
            L.PCICCGetInfo_HIPMI355X(pc, None, None, C.byref(ns))
        else:
            f, nso = orc.ilu0_factor_shift(ai, aj, aa)
            ref = orc.ilu0_solve(f, bvec)
            L.PCILUGetShiftCount_HIPMI355X(pc, C.byref(ns))
        assert ns.value == nso
        assert np.array_equal(bits(vx.array()), bits(ref)), (pct, seed, n)
        Am.destroy()


@pytest.mark.parametrize("pct", ["ilu", "icc"])
def test_syncfree_solve_that_gave_up_is_noticed_at_the_next_host_wait(P, pct):
    """A dependency wait of the sync-free triangular solves is bounded; when one gives up the application's result is unusable.
    The abort flag (pinned host memory) is looked at by every host wait the solvers perform: the first reduction / VecGetArray
    after it raises PETSC_ERR_LIB instead of returning numbers computed from a poisoned vector, and the factor then runs the
    same plans one dependency level per launch -- same bits, ILU(0) and ICC(0) alike.  An application queued while the flag is
    up and nobody has waited yet falls back by itself."""
    L = P.lib()
    ai, aj, aa = P.gen_poisson7(12, 11, 10)
    n = ai.size - 1
    aa = aa * (1.0 + 0.05 * np.sin(np.arange(aa.size))) if pct == "ilu" else aa
    A = P.Mat.from_csr(ai, aj, aa)
    pc = C.c_void_p()
    k = P.KSP(comm=L.COMM_SELF); k.set_operators(A); L.KSPGetPC(k.h, C.byref(pc)); L.PCSetType(pc, pct.encode())
    set_options(L, "-pc_factor_hipmi355x_trisolve syncfree")
    L.raw("PCSetUp")(pc)
    set_options(L, "")
    bvec = rnd(n, 5); vb, vx = V(P, bvec), V(P, np.zeros(n))
    L.raw("PCApply")(pc, vb.h, vx.h)
    good = bits(vx.array()).copy()
    ref = orc.ilu0_solve(orc.ilu0_factor(ai, aj, aa), bvec) if pct == "ilu" else orc.icc0_solve(orc.icc0_factor(ai, aj, aa)[0], bvec)
    assert np.array_equal(good, bits(ref))
    # (1) the flag goes up after an application was queued: the next host wait reports it
    L.raw("PCApply")(pc, vb.h, vx.h)
    L.PCFactorDebugSetAborted_HIPMI355X(pc)
    with pytest.raises(P.PetscError) as e:
        vx.norm()
    assert e.value.code == 76 and "gave up" in str(e.value)
    # (2) from now on: level by level over the same plans, same bits; the PC says so
    for rep in range(2):
        L.raw("PCApply")(pc, vb.h, vx.h)
        assert np.array_equal(bits(vx.array()), good)
    if pct == "ilu":
        sf, ab = C.c_int(), C.c_int()
        L.PCILUGetSolver_HIPMI355X(pc, C.byref(sf), C.byref(ab))
        assert (sf.value, ab.value) == (0, 1)
    # (3) a whole solve on the fallen-back factor converges to the oracle's iterates
    x, h, its, reason = (None,) * 4
    k.set_type("gmreshipmi355x" if pct == "ilu" else "cghipmi355x"); k.set_tolerances(rtol=1e-9); k.record_history()
    vx2 = V(P, np.zeros(n)); k.solve(vb, vx2)
    xo, ho, ito, ro = orc.ksp_solve(ai, aj, aa, bvec, ksp="gmres" if pct == "ilu" else "cg", pc=pct, rtol=1e-9)
    assert (k.its, k.reason) == (ito, ro) and np.allclose(k.history(), ho, rtol=1e-8, atol=0)
    # (4) a fresh factor whose flag is raised BEFORE anything waited: the application itself takes the level path
    A2 = P.Mat.from_csr(ai, aj, aa)
    pc2 = C.c_void_p()
    k2 = P.KSP(comm=L.COMM_SELF); k2.set_operators(A2); L.KSPGetPC(k2.h, C.byref(pc2)); L.PCSetType(pc2, pct.encode())
    set_options(L, "-pc_factor_hipmi355x_trisolve syncfree")
    L.raw("PCSetUp")(pc2)
    set_options(L, "")
    L.PCFactorDebugSetAborted_HIPMI355X(pc2)
    L.raw("PCApply")(pc2, vb.h, vx.h)
    assert np.array_equal(bits(vx.array()), good)


@pytest.mark.parametrize("nth", [1, 3, 8])
def test_ilu0_threaded_factorisation_carries_the_same_bits(P, nth):
    """The host ILU(0) factorisation deals the rows of a dependency level of L to several threads (-mat_factor_hipmi355x_threads;
    by default for 200 000 rows and more).  Every row's arithmetic is the sequential loop's, so the factor -- seen through one
    application -- carries the oracle's bits for any thread count, also when MatPivotCheck_nz makes the passes restart with a
    shifted diagonal (as many restarts as the oracle takes), and also for independent blocks with different shifts."""
    import scipy.sparse as sp
    L = P.lib()
    cases = [P.gen_poisson7(23, 17, 11), pb.gen_fem3(7, 6, 5)]
    T = sp.diags([np.ones(299), np.ones(300), np.ones(299)], [-1, 0, 1]).tocsr(); T.sort_indices()      # zero pivots: shifts
    cases.append((T.indptr.astype(np.int32), T.indices.astype(np.int32), T.data.copy()))
    for ai, aj, aa in cases:
        n = ai.size - 1
        aa = aa * (1.0 + 0.03 * np.cos(np.arange(aa.size))) if n != 300 else aa
        A = P.Mat.from_csr(ai, aj, aa)
        pc = C.c_void_p()
        k = P.KSP(comm=L.COMM_SELF); k.set_operators(A); L.KSPGetPC(k.h, C.byref(pc)); L.PCSetType(pc, b"ilu")
        set_options(L, "-mat_factor_hipmi355x_threads %d -pc_factor_hipmi355x_trisolve_nodes 0 -pc_factor_hipmi355x_trisolve_order column" % nth)
        L.raw("PCSetUp")(pc)
        set_options(L, "")
        f, ns_o = orc.ilu0_factor_shift(ai, aj, aa)
        ns = C.c_int(); L.PCILUGetShiftCount_HIPMI355X(pc, C.byref(ns))
        assert ns.value == ns_o and (ns_o >= 1) == (n == 300)
        b = rnd(n, 11); vb, vx = V(P, b), V(P, np.zeros(n))
        L.raw("PCApply")(pc, vb.h, vx.h)
        assert np.array_equal(bits(vx.array()), bits(orc.ilu0_solve(f, b))), (n, nth)


def test_ilu0_apply_bitexact_and_golden(P):
    """PCILU (ILU(0), natural ordering).  The level-scheduled device solve reproduces
    MatSolve_SeqAIJ_NaturalOrdering bit for bit (one lane per row, products subtracted in column order), and with NO
    -pc_type option a one-rank solve picks ILU like the reference: ex2 -m 5 -n 5 refine_always == ex2_1.out."""
    L = P.lib()
    # the third case has 31 + 31 levels: above 16 the solves run sync-free (two launches), or -- with
    # -pc_factor_hipmi355x_trisolve level -- as one launch per level replayed from a captured hipGraph: same bits
    for mode in ("", "-pc_factor_hipmi355x_trisolve level", "-pc_factor_hipmi355x_trisolve syncfree"):
        for ai, aj, aa in (pb.lap2d(9, 7), P.gen_poisson7(7, 6, 5), P.gen_poisson7(12, 11, 10)):
            n = ai.size - 1
            aa = aa * (1.0 + 0.05 * np.sin(np.arange(aa.size)))
            A = P.Mat.from_csr(ai, aj, aa)
            pc = C.c_void_p()
            k = P.KSP(comm=L.COMM_SELF); k.set_operators(A); L.KSPGetPC(k.h, C.byref(pc)); L.PCSetType(pc, b"ilu")
            bvec = rnd(n, 77)
            vb, vx = V(P, bvec), V(P, np.zeros(n))
            set_options(L, mode)
            L.raw("PCSetUp")(pc)
            set_options(L, "")
            L.raw("PCApply")(pc, vb.h, vx.h)
            ref = orc.ilu0_solve(orc.ilu0_factor(ai, aj, aa), bvec)
            assert np.array_equal(bits(vx.array()), bits(ref))
            for rep in range(3):                        # replays: the sentinel of the hand-off buffers is restored every time
                b2 = rnd(n, 78 + rep); vb.set_array(b2)
                L.raw("PCApply")(pc, vb.h, vx.h)
                assert np.array_equal(bits(vx.array()), bits(orc.ilu0_solve(orc.ilu0_factor(ai, aj, aa), b2)))
            nl, nu = C.c_int(), C.c_int()
            L.PCILUGetLevels_HIPMI355X(pc, C.byref(nl), C.byref(nu))
            assert nl.value == nu.value and nl.value > 1
            sf, ab = C.c_int(), C.c_int()
            L.PCILUGetSolver_HIPMI355X(pc, C.byref(sf), C.byref(ab))
            assert ab.value == 0
            assert sf.value == (1 if (mode.endswith("syncfree") or (mode == "" and nl.value + nu.value > 16)) else 0)
    # a zero pivot: PCILU's default MAT_SHIFT_NONZERO restarts the factorisation with a shifted diagonal, as many times as the oracle's
    import scipy.sparse as sp
    T = sp.diags([np.ones(39), np.ones(40), np.ones(39)], [-1, 0, 1]).tocsr(); T.sort_indices()
    ai, aj, aa = T.indptr.astype(np.int32), T.indices.astype(np.int32), T.data.copy()
    A = P.Mat.from_csr(ai, aj, aa)
    pc = C.c_void_p()
    k = P.KSP(comm=L.COMM_SELF); k.set_operators(A); L.KSPGetPC(k.h, C.byref(pc)); L.PCSetType(pc, b"ilu")
    L.raw("PCSetUp")(pc)
    f, ns_o = orc.ilu0_factor_shift(ai, aj, aa)
    ns = C.c_int(); L.PCILUGetShiftCount_HIPMI355X(pc, C.byref(ns))
    assert ns.value == ns_o and ns_o >= 1
    bvec = rnd(40, 81); vb, vx = V(P, bvec), V(P, np.zeros(40))
    L.raw("PCApply")(pc, vb.h, vx.h)
    assert np.array_equal(bits(vx.array()), bits(orc.ilu0_solve(f, bvec)))
    ai, aj, aa = pb.lap2d(5, 5)
    u = np.ones(25)
    b = orc.spmv(ai, aj, aa, u)
    gold = pb.parse_monitor(os.path.join(G, "ksp_tutorials", "ex2_1.out"))[0]
    A = P.Mat.from_csr(ai, aj, aa)
    k = P.KSP(comm=L.COMM_SELF); k.set_operators(A)
    L.PetscOptionsClear(); L.PetscOptionsInsertString(b"-ksp_gmres_cgs_refinement_type refine_always")
    k.set_tolerances(rtol=1e-2 / 36, abstol=1e-50)
    k.set_from_options(); k.record_history()
    vb, vx = V(P, b), V(P, np.zeros(25))
    k.solve(vb, vx)
    L.PetscOptionsClear()
    pb.check_monitor(k.history(), gold)
    assert k.its == 4 and "%.5g" % np.linalg.norm(vx.array() - u) in ("0.0003927", "0.00039270")


def test_tutorial_ex7_block_jacobi_with_a_different_solver_on_every_block_golden(P):
    """tutorials/ex7 vs output/ex7_1.out on the HIP path, on ONE rank: eight blocks of ten rows (-pc_bjacobi_blocks 8, kept as eight
    solvers: -pc_bjacobi_merge_blocks 0), every block's KSP customised after KSPSetUp through PCBJacobiGetSubKSP as the example does --
    blocks 0..3 (the example's rank 0) alternately BiCGStab + PCNONE (rtol 1e-6) and the default ILU(0), blocks 4..7 (its rank 1)
    GMRES + Jacobi (rtol 1e-7).  Fourteen monitor lines, 'Norm of error 1.09983e-05 iterations 13'."""
    L = P.lib()
    ai, aj, aa = pb.lap2d(8, 10)
    u = np.ones(80)
    A = P.Mat.from_csr(ai, aj, aa)
    vb, vx = V(P, orc.spmv(ai, aj, aa, u)), V(P, np.zeros(80))
    k = P.KSP(comm=L.COMM_SELF); k.set_operators(A)
    L.PetscOptionsClear()
    L.PetscOptionsInsertString(b"-pc_type bjacobi -pc_bjacobi_blocks 8 -pc_bjacobi_merge_blocks 0 -ksp_gmres_cgs_refinement_type refine_always")
    k.set_from_options(); k.record_history()
    L.KSPSetUp(k.h)
    pc, nloc, first, sub = C.c_void_p(), C.c_int(), C.c_int(), C.c_void_p()
    L.KSPGetPC(k.h, C.byref(pc))
    L.PCBJacobiGetSubKSP(pc, C.byref(nloc), C.byref(first), C.byref(sub))
    assert (nloc.value, first.value) == (8, 0)
    subs = C.cast(sub, C.POINTER(C.c_void_p))
    for i in range(8):
        spc = C.c_void_p(); L.KSPGetPC(subs[i], C.byref(spc))
        if i < 4 and i % 2:
            L.PCSetType(spc, b"ilu")
        elif i < 4:
            L.PCSetType(spc, b"none"); L.KSPSetType(subs[i], b"bcgs"); L.KSPSetTolerances(subs[i], 1e-6, P.PETSC_DEFAULT, P.PETSC_DEFAULT, int(P.PETSC_DEFAULT))
        else:
            L.PCSetType(spc, b"jacobi"); L.KSPSetType(subs[i], b"gmres"); L.KSPSetTolerances(subs[i], 1e-7, P.PETSC_DEFAULT, P.PETSC_DEFAULT, int(P.PETSC_DEFAULT))
    k.solve(vb, vx)
    L.PetscOptionsClear()
    pb.check_monitor(k.history(), pb.parse_monitor(os.path.join(G, "ksp_tutorials", "ex7_1.out"))[0])
    assert open(os.path.join(G, "ksp_tutorials", "ex7_1.out")).read().splitlines()[-1] == "Norm of error %g iterations %d" % (np.linalg.norm(vx.array() - u), k.its)


@pytest.mark.parametrize("opts", ["-ksp_gmres_fused 0", ""])
def test_ksp_tests_ex40_default_gmres_without_a_preconditioner_golden(P, opts):
    """ksp/examples/tests/ex40 -pc_type none vs output/ex40.out on the HIP path: default GMRES (no refinement step), PCNONE, ex2's 8 x 7
    operator: 'Norm of error 1.68964e-05 iterations 13' -- plain call sequence and the plug-in's own GMRES."""
    ai, aj, aa = pb.lap2d(8, 7)
    u = np.ones(56)
    x, h, its, reason = solve(P, ai, aj, aa, orc.spmv(ai, aj, aa, u), "gmres", "none", opts=opts, rtol=1e-2 / 72, abstol=1e-50)
    assert open(os.path.join(G, "ksp_tests", "ex40.out")).read().strip() == "Norm of error %g iterations %d" % (np.linalg.norm(x - u), its)


@pytest.mark.parametrize("opts", ["-ksp_cg_fused 0", ""])
def test_ksp_tests_ex10_cg_ilu0_on_a_matrix_with_inodes_golden(P, opts):
    """The reference's ksp/examples/tests/ex10 (one 20-node elasticity brick, AIJ with 21 inodes of 1, 2 and 3 rows, KSPCG + the
    default PCILU) on the HIP path vs output/ex10_1.out: the inode count MatView reports, the ten monitor lines, 9 iterations --
    through the plain CG call sequence and through the plug-in's own CG."""
    L = P.lib()
    (ai, aj, aa), b, u = pb.ex10_elasticity()
    text = open(os.path.join(G, "ksp_tests", "ex10_1.out")).read()
    gold = pb.parse_monitor(os.path.join(G, "ksp_tests", "ex10_1.out"))[0]
    A = P.Mat.from_csr(ai, aj, aa)
    vu, vb = V(P, u), V(P, np.zeros(u.size))
    A.mult(vu, vb)                                                   # b = A u as the example forms it (MatMult_SeqAIJ_Inode)
    nodes, groups, shared = C.c_int(), C.c_int(), C.c_int()
    L.MatHIPMI355XGetInodeInfo(A.h, C.byref(nodes), C.byref(groups), C.byref(shared))
    assert "found %d nodes, limit used is 5" % nodes.value in text
    assert np.all(np.abs(vb.array() - b) <= 1e-13 * np.abs(b).max())
    x, h, its, reason = solve(P, ai, aj, aa, vb.array(), "cg", "ilu", opts=opts, rtol=1e-10)
    pb.check_monitor(h, gold)
    assert its == 9 and reason == 2 and np.linalg.norm(x - u) < 1e-11


@pytest.mark.parametrize("shape", ["fem3", "irr", "deep", "p7"])
def test_ilu0_syncfree_solves_on_irregular_factors(P, shape):
    """the sync-free triangular solves on factors whose levels are ragged: slices that span many levels (in-wavefront
    sub-steps), rows of very different lengths in one slice, hundreds of dependencies per row, more levels than slices;
    bit-exact against MatSolve_SeqAIJ_NaturalOrdering's restatement, applied several times"""
    L = P.lib()
    if shape == "fem3":
        ai, aj, aa = pb.gen_fem3(10, 10, 6)
    elif shape == "irr":
        ai, aj, aa = pb.gen_irr(n=6000, mean=40.0, seed=4)
    elif shape == "deep":      # tridiagonal + a few long-range entries: n levels of one row each
        n = 3000
        import scipy.sparse as sp
        S = sp.diags([-np.ones(n - 1), 4.0 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1]).tolil()
        for r in range(50, n, 97):
            S[r, r - 50] = -0.5; S[r - 50, r] = -0.25
        ai, aj, aa = pb.csr(S)
    else:
        ai, aj, aa = P.gen_poisson7(40, 33, 21)
    n = ai.size - 1
    A = P.Mat.from_csr(ai, aj, aa)
    pc = C.c_void_p()
    k = P.KSP(comm=L.COMM_SELF); k.set_operators(A); L.KSPGetPC(k.h, C.byref(pc)); L.PCSetType(pc, b"ilu")
    set_options(L, "-pc_factor_hipmi355x_trisolve syncfree")
    L.raw("PCSetUp")(pc)
    set_options(L, "")
    f = orc.ilu0_factor(ai, aj, aa)
    vb, vx = V(P, np.zeros(n)), V(P, np.zeros(n))
    nodes, ns = orc.check_inode(ai, aj)
    for rep in range(3):
        b = rnd(n, 90 + rep)
        vb.set_array(b)
        L.raw("PCApply")(pc, vb.h, vx.h)
        # a factor with inodes: the reference solves it with MatSolve_SeqAIJ_Inode (node by node, two columns at a time), and so
        # does the device (node-blocked plans) -- by default with every node's columns in dependency-level order (rounding apart),
        # in the routine's own order on request (below); otherwise the natural-ordering loop, bit for bit
        if nodes:
            ref = orc.ilu0_solve_inode(f, ns, b)
            assert np.linalg.norm(vx.array() - ref) <= 1e-13 * np.linalg.norm(ref)
        else:
            assert np.array_equal(bits(vx.array()), bits(orc.ilu0_solve(f, b)))
    sf, ab = C.c_int(), C.c_int()
    L.PCILUGetSolver_HIPMI355X(pc, C.byref(sf), C.byref(ab))
    assert sf.value == 1 and ab.value == 0
    nn, nlL, nlU = C.c_int(), C.c_int(), C.c_int()
    L.PCILUGetNodeInfo_HIPMI355X(pc, C.byref(nn), C.byref(nlL), C.byref(nlU))
    assert abs(nn.value) == nodes                      # (negative: the node plans whose columns are whole dependency nodes)
    if nodes:
        rl, ru = C.c_int(), C.c_int(); L.PCILUGetLevels_HIPMI355X(pc, C.byref(rl), C.byref(ru))
        assert 0 < nlL.value < rl.value and 0 < nlU.value < ru.value          # levels over nodes: fewer than over rows
        # row-granular plans on request: column order = the natural-ordering loop's bits; level order agrees to rounding
        for opts, exact in (("-pc_factor_hipmi355x_trisolve_nodes 0 -pc_factor_hipmi355x_trisolve_order column", "natural"),
                            ("-pc_factor_hipmi355x_trisolve_order column", "inode"),
                            ("-pc_factor_hipmi355x_trisolve_order column -pc_factor_hipmi355x_trisolve_block_columns 1", "inode"),
                            ("-pc_factor_hipmi355x_trisolve_nodes 0", False), ("-pc_factor_hipmi355x_trisolve_order level", False)):
            k2 = P.KSP(comm=L.COMM_SELF); k2.set_operators(A); pc2 = C.c_void_p(); L.KSPGetPC(k2.h, C.byref(pc2)); L.PCSetType(pc2, b"ilu")
            set_options(L, "-pc_factor_hipmi355x_trisolve syncfree " + opts)
            L.raw("PCSetUp")(pc2)
            set_options(L, "")
            b = rnd(n, 92); vb.set_array(b)
            L.raw("PCApply")(pc2, vb.h, vx.h)
            ref = orc.ilu0_solve_inode(f, ns, b) if exact == "inode" else orc.ilu0_solve(f, b)
            if exact:
                assert np.array_equal(bits(vx.array()), bits(ref))
            else:
                assert np.linalg.norm(vx.array() - ref) <= 1e-13 * np.linalg.norm(ref)
                first = vx.array().copy()
                L.raw("PCApply")(pc2, vb.h, vx.h)                                # deterministic
                assert np.array_equal(bits(vx.array()), bits(first))


@pytest.mark.parametrize("form", ["split", "onewave", "blockcols"])
@pytest.mark.parametrize("seed", range(6))
def test_ilu0_node_blocked_solves_carry_the_bits_of_the_inode_routine(P, seed, form, monkeypatch):
    """MatSolve_SeqAIJ_Inode on the device: factors of matrices whose rows form nodes of 1..5 rows (a random node graph, every node a
    dense block of dofs: what a FEM matrix with a varying number of dofs per node looks like).  The node-blocked sync-free plans walk
    a node's shared column list once, two columns at a time, then solve the node's triangle -- the reference routine's order (inode.c:
    2327-2760), restated in the oracle: same bits, several applications; the level-by-level fall-back over the same plans too."""
    import scipy.sparse as sp
    L = P.lib()
    rng = np.random.default_rng(300 + seed)
    nn = int(rng.integers(40, 900))
    maxdof = [3, 5, 2, 4, 5, 3][seed]
    dof = rng.integers(1, maxdof + 1, nn) if seed % 2 else np.full(nn, maxdof)
    G = sp.random(nn, nn, density=min(1.0, float(rng.uniform(2.0, 7.0)) / nn), random_state=int(rng.integers(1 << 30)), format="csr")
    G = ((G + G.T + sp.diags([np.ones(nn - 1)], [1]) + sp.diags([np.ones(nn - 1)], [-1]) + sp.eye(nn)) != 0).tocsr()      # symmetric pattern, a path so that levels are deep
    start = np.concatenate([[0], np.cumsum(dof)])
    n = int(start[-1])
    rows, cols = [], []
    for u in range(nn):
        nb = G.indices[G.indptr[u]:G.indptr[u + 1]]
        cc = np.concatenate([np.arange(start[v], start[v + 1]) for v in sorted(nb)])
        for r in range(start[u], start[u + 1]):
            rows.append(np.full(cc.size, r)); cols.append(cc)
    rows = np.concatenate(rows); cols = np.concatenate(cols)
    vals = -rng.random(rows.size)
    A_ = sp.csr_matrix((vals, (rows, cols)), shape=(n, n)); A_.sort_indices()
    rs = np.asarray(abs(A_).sum(axis=1)).ravel()
    A_ = (A_ + sp.diags(rs + 1.0)).tocsr(); A_.sort_indices()
    ai, aj, aa = A_.indptr.astype(np.int32), A_.indices.astype(np.int32), A_.data.copy()
    nodes, ns = orc.check_inode(ai, aj)
    assert nodes > 0 and ns.max() <= 5
    A = P.Mat.from_csr(ai, aj, aa)
    pc = C.c_void_p()
    k = P.KSP(comm=L.COMM_SELF); k.set_operators(A); L.KSPGetPC(k.h, C.byref(pc)); L.PCSetType(pc, b"ilu")
    # the three kernel forms over the same plans: loader + solver wavefront per workgroup (the default), one wavefront doing both
    # (MI355X_TRISOLVE_SPLIT=0, read when the plan is made), whole dependency nodes as columns (on request)
    blockcols = 1 if form == "blockcols" else 0
    monkeypatch.setenv("MI355X_TRISOLVE_SPLIT", "0" if form == "onewave" else "1")
    set_options(L, "-pc_factor_hipmi355x_trisolve syncfree -pc_factor_hipmi355x_trisolve_order column -pc_factor_hipmi355x_trisolve_block_columns %d" % blockcols)
    L.raw("PCSetUp")(pc)
    set_options(L, "")
    got = C.c_int(); L.PCILUGetNodeInfo_HIPMI355X(pc, C.byref(got), None, None)
    assert abs(got.value) == nodes
    uniform = bool(np.all(dof == dof[0])) and dof[0] <= 4 and bool(np.all(ns == dof[0]))
    assert (got.value < 0) == (uniform and blockcols == 1)      # on request, with a fixed number of dofs per node: whole dependency nodes as columns
    f = orc.ilu0_factor(ai, aj, aa)
    vb, vx = V(P, np.zeros(n)), V(P, np.zeros(n))
    for rep in range(3):
        b = rnd(n, 500 + rep); vb.set_array(b)
        L.raw("PCApply")(pc, vb.h, vx.h)
        assert np.array_equal(bits(vx.array()), bits(orc.ilu0_solve_inode(f, ns, b))), (seed, rep)
    L.PCFactorDebugSetAborted_HIPMI355X(pc)          # the same plans, one launch per node level
    L.raw("PCApply")(pc, vb.h, vx.h)
    assert np.array_equal(bits(vx.array()), bits(orc.ilu0_solve_inode(f, ns, b)))


@pytest.mark.parametrize("dof,order", [(3, "column"), (3, "level"), (2, "column"), (4, "column"), (5, "column"), (5, "level")])
def test_ilu0_split_role_solves_on_fem_factors(P, dof, order):
    """The batched path of the split-role node kernels (a loader and a solver wavefront per workgroup, the slice's indices and
    values through a ring in LDS, batches aligned to the END of every lane's column list): FEM-like factors with `dof` rows per
    node, wide dependency levels (slices without dependent sub-steps), five and more batches per slice.  Columns in column order =
    MatSolve_SeqAIJ_Inode's bits (inode.c:2327-2760); in dependency-level order to rounding, and deterministic."""
    L = P.lib()
    ai, aj, aa = pb.gen_fem3(ex=60, ey=50, ez=10, dof=dof, seed=40 + dof) if dof == 3 else pb.gen_fem3(ex=40, ey=36, ez=12, dof=dof, seed=40 + dof)
    n = ai.size - 1
    nodes, ns = orc.check_inode(ai, aj)
    assert nodes > 0 and ns.max() == dof
    A = P.Mat.from_csr(ai, aj, aa)
    pc = C.c_void_p()
    k = P.KSP(comm=L.COMM_SELF); k.set_operators(A); L.KSPGetPC(k.h, C.byref(pc)); L.PCSetType(pc, b"ilu")
    set_options(L, "-pc_factor_hipmi355x_trisolve syncfree -pc_factor_hipmi355x_trisolve_order %s" % order)
    L.raw("PCSetUp")(pc)
    set_options(L, "")
    got, nlL, nlU = C.c_int(), C.c_int(), C.c_int(); L.PCILUGetNodeInfo_HIPMI355X(pc, C.byref(got), C.byref(nlL), C.byref(nlU))
    assert got.value == nodes and nodes / nlL.value > 20          # wide levels: about half of them fill whole slices
    f = orc.ilu0_factor(ai, aj, aa)
    vb, vx = V(P, np.zeros(n)), V(P, np.zeros(n))
    for rep in range(3):
        b = rnd(n, 700 + rep); vb.set_array(b)
        L.raw("PCApply")(pc, vb.h, vx.h)
        ref = orc.ilu0_solve_inode(f, ns, b)
        if order == "column":
            assert np.array_equal(bits(vx.array()), bits(ref)), (dof, rep)
        else:
            assert np.linalg.norm(vx.array() - ref) <= 1e-13 * np.linalg.norm(ref)
            first = vx.array().copy()
            L.raw("PCApply")(pc, vb.h, vx.h)
            assert np.array_equal(bits(vx.array()), bits(first))
    sf, ab = C.c_int(), C.c_int()
    L.PCILUGetSolver_HIPMI355X(pc, C.byref(sf), C.byref(ab))
    assert sf.value == 1 and ab.value == 0


@pytest.mark.parametrize("name,nblocks,err,nits", [("ex2_bjacobi_2.out", 2, "0.000496964", 4), ("ex2_bjacobi_3.out", 4, "0.000404746", 7)])
def test_ksp_bjacobi_several_local_blocks_golden(P, name, nblocks, err, nits):
    """ex2_bjacobi_2.out / ex2_bjacobi_3.out (tutorials makefile:355,360: 4 ranks, 2 and 4 blocks of 28 / 14 rows, inexact
    GMRES + Jacobi sub-solves) reproduced on ONE rank with -pc_bjacobi_blocks 2 / 4: PCSetUp_BJacobi_Multiblock's even split
    gives the same blocks; sub-vectors alias slices of the device vectors ("VecShareSubArrayBegin_C")"""
    ai, aj, aa = pb.lap2d(8, 7)
    u = np.ones(56)
    b = orc.spmv(ai, aj, aa, u)
    gold = pb.parse_monitor(os.path.join(G, "ksp_tutorials", name))[0]
    x, h, its, reason = solve(P, ai, aj, aa, b, "gmres", "bjacobi", opts="-pc_bjacobi_blocks %d -sub_pc_type jacobi -sub_ksp_type gmres" % nblocks,
                              rtol=1e-2 / 72, abstol=1e-50)
    pb.check_monitor(h, gold)
    assert its == nits and "%g" % np.linalg.norm(x - u) == err


@pytest.mark.parametrize("nblocks,sub", [(4, "-sub_pc_type ilu"), (4, "-sub_pc_type ilu -pc_bjacobi_merge_blocks 0"), (5, ""), (7, "-sub_pc_type jacobi"), (3, "-sub_ksp_type gmres -sub_pc_type jacobi -sub_ksp_rtol 1e-3")])
def test_ksp_bjacobi_several_local_blocks_equal_the_oracle_bit_for_bit(P, nblocks, sub):
    """GMRES + block Jacobi with several blocks on one rank (uneven split: 2210 rows), ILU(0) / Jacobi / inexact GMRES sub-solves:
    history and solution equal the oracle's block-Jacobi restatement bit for bit under the device summation order -- also when
    the ILU(0) blocks are solved as ONE triangular solve over the block-diagonal matrix (the default for preonly + ILU
    sub-solvers; -pc_bjacobi_merge_blocks 0: block after block)"""
    ai, aj, aa = orc.gen_p7(17, 13, 10)
    aa = aa * (1.0 + 0.3 * np.sin(0.7 * np.arange(aa.size)))
    n = ai.size - 1
    b = np.cos(0.37 * np.arange(n)) + 0.1
    x, h, its, reason = solve(P, ai, aj, aa, b, "gmres", "bjacobi", opts="-pc_bjacobi_blocks %d %s" % (nblocks, sub), rtol=1e-10, max_it=60)
    starts = [0]
    for i in range(nblocks):
        starts.append(starts[-1] + n // nblocks + (1 if (n % nblocks) > i else 0))
    okw = dict(sub_ksp="gmres", sub_pc="jacobi", sub_rtol=1e-3) if "gmres" in sub else dict(sub_ksp="preonly", sub_pc="jacobi" if "-sub_pc_type jacobi" in sub else "ilu")
    with orc.device_reduction_order():
        xo, ho, ito, ro = orc.ksp_solve(ai, aj, aa, b, ksp="gmres", pc="bjacobi", blocks=starts, rtol=1e-10, max_it=60, **okw)
    assert (its, reason) == (ito, ro) and its > 5
    assert np.array_equal(h.view(np.uint64), ho.view(np.uint64)) and np.array_equal(x.view(np.uint64), xo.view(np.uint64))


def test_ksp_bjacobi_single_block_golden(P):
    """ex2_bjacobi.out: one block, sub-KSP GMRES + Jacobi (-sub_ksp_type gmres -sub_pc_type jacobi); device-side aliasing
    of the work vectors instead of the reference's host VecPlaceArray round trip"""
    ai, aj, aa = pb.lap2d(8, 7)
    u = np.ones(56)
    b = orc.spmv(ai, aj, aa, u)
    gold = pb.parse_monitor(os.path.join(G, "ksp_tutorials", "ex2_bjacobi.out"))[0]
    x, h, its, reason = solve(P, ai, aj, aa, b, "gmres", "bjacobi", opts="-pc_bjacobi_blocks 1 -sub_pc_type jacobi -sub_ksp_type gmres",
                              rtol=1e-2 / 72, abstol=1e-50)
    pb.check_monitor(h, gold)
    assert its == 1 and "%.5g" % np.linalg.norm(x - u) in ("2.1014e-06", "2.1015e-06")


@pytest.mark.parametrize("ksp", ["cg", "gmres", "bcgs"])
@pytest.mark.parametrize("pc", ["none", "jacobi", "bjacobi", "ilu"])
def test_ksp_vs_oracle_p7(P, ksp, pc):
    """every solver x preconditioner of the north star on a 3-D 7-pt operator with varying coefficients"""
    ai, aj, aa = P.gen_poisson7(12, 11, 10)
    n = ai.size - 1
    d = 1.0 + 0.5 * np.sin(np.arange(n))           # symmetric diagonal scaling keeps it SPD
    rows = np.repeat(np.arange(n), np.diff(ai))
    aa = aa * d[rows] * d[aj]
    b = orc.spmv(ai, aj, aa, np.cos(0.1 * np.arange(n)))
    kw = dict(rtol=1e-8)
    x, h, its, reason = solve(P, ai, aj, aa, b, ksp, pc, opts="-ksp_gmres_restart 20", **kw)
    okw = dict(blocks=[0, n], sub_ksp="preonly", sub_pc="ilu") if pc == "bjacobi" else {}   # default sub-PC = ILU(0)
    xr, hr, itsr, rr = orc.ksp_solve(ai, aj, aa, b, ksp=ksp, pc=pc, rtol=1e-8, restart=20, **okw)
    k = min(len(h), len(hr))
    if ksp == "bcgs":
        # BiCGStab's recurrences amplify the last-bit differences of the dot products: histories agree tightly
        # while the residual is still large, later only in order of magnitude; the solutions agree
        early = hr[:k] > 1e-3 * hr[0]
        assert np.allclose(h[:k][early], hr[:k][early], rtol=1e-6, atol=0)
        assert reason == rr and abs(its - itsr) <= max(3, itsr // 5)
    else:
        assert reason == rr and abs(its - itsr) <= 1
        assert np.allclose(h[:k], hr[:k], rtol=1e-6, atol=1e-14 * hr[0])
    assert np.linalg.norm(x - xr) <= 1e-6 * np.linalg.norm(xr)


def _which_spmv(P, A):
    """(row-pattern dictionary size, value-pattern dictionary size) of an uploaded sequential matrix: which SpMV kernel it runs"""
    import ctypes as C
    npat, nvpat = C.c_int(), C.c_int()
    P.lib().MatHIPMI355XGetRowPatterns(A.h, C.byref(npat))
    P.lib().MatHIPMI355XGetValuePatterns(A.h, C.byref(nvpat))
    return npat.value, nvpat.value


@pytest.mark.parametrize("value_patterns", ["default", "0"])
def test_p7_full_size_properties(P, value_patterns):
    """BASELINE.json configs[1] at full size (P7(256): 16.8M rows) through size-independent properties:
    A*1 is the boundary indicator pattern (row sums), linearity A(ax+by) = aAx + bAy, symmetry <Ax,y> = <x,Ay>.
    Both SpMV kernels this operator can get: the library default (value patterns: spmv_csr_valpat_kernel) and, with
    -mat_hipmi355x_value_patterns 0, the kernel bench.py's headline and `roofline` quote (spmv_csr_rowblock_pat_kernel, the
    value array streamed) -- and the two products agree bit for bit."""
    L = P.lib()
    n = 256
    ai, aj, aa = P.gen_poisson7(n, n, n)
    L.PetscOptionsClear()
    if value_patterns == "0":
        L.PetscOptionsInsertString(b"-mat_hipmi355x_value_patterns 0")
    A = P.Mat.from_csr(ai, aj, aa)
    N = n ** 3
    one = P.Vec.create(N, comm=L.COMM_SELF); L.VecSet(one.h, 1.0)
    r = one.duplicate()
    A.mult(one, r)
    L.PetscOptionsClear()
    npat, nvpat = _which_spmv(P, A)
    assert (nvpat == 0 and npat > 0) if value_patterns == "0" else nvpat > 0, (npat, nvpat)
    rs = r.array()
    expect = 6.0 - np.diff(ai) + 1.0                  # 6 - (#neighbours) ; row has 1 + #neighbours entries
    assert np.array_equal(rs, expect)
    del rs, expect
    x = P.Vec.from_array(np.sin(0.37 * np.arange(N)) + 1.0, comm=L.COMM_SELF)
    y = P.Vec.from_array(np.cos(0.11 * np.arange(N)), comm=L.COMM_SELF)
    ax, ay, w, aw = x.duplicate(), x.duplicate(), x.duplicate(), x.duplicate()
    A.mult(x, ax); A.mult(y, ay)
    L.VecWAXPY(w.h, 2.5, x.h, y.h)                    # w = y + 2.5 x
    A.mult(w, aw)
    L.VecAXPY(aw.h, -2.5, ax.h); L.VecAXPY(aw.h, -1.0, ay.h)
    assert aw.norm() <= 1e-12 * ax.norm()
    assert abs(ax.dot(y) - x.dot(ay)) <= 1e-12 * ax.norm() * y.norm()
    if value_patterns == "0":
        # the same matrix object switched to the other kernel: the product carries the same bits
        axv = ax.array().copy()
        L.MatHIPMI355XSetValuePatterns(A.h, 1)
        A.mult(x, ax)
        assert _which_spmv(P, A)[1] > 0
        assert np.array_equal(bits(ax.array()), bits(axv))


@pytest.mark.parametrize("value_patterns", ["default", "0"])
def test_cg_jacobi_full_size_properties(P, value_patterns):
    """KSPCG + PCJACOBI at BASELINE.json's full size (P7(256)) through size-independent properties: the solve of
    A x = A*1 converges to the vector of ones, the reported (preconditioned) residual norm equals the recomputed
    ||D^-1 (b - A x)|| to 1e-6, and the fused sweeps (level 3) and the op-by-op sequence (-ksp_cg_fused 0) produce the same
    iteration count, the same history bits and the same x bits at this size too; the default (level 4) agrees to rounding.
    value_patterns "0": the value array streamed -- bench.py's headline configuration, whose level-4 solve runs
    spmv_csr_rowblock_pat_kernel<0, true> (the product that also leaves p'w, the kernel `roofline` quotes)."""
    L = P.lib()
    n = 256
    ai, aj, aa = P.gen_poisson7(n, n, n)
    vp_opt = "-mat_hipmi355x_value_patterns 0" if value_patterns == "0" else ""
    L.PetscOptionsClear(); L.PetscOptionsInsertString(vp_opt.encode())
    A = P.Mat.from_csr(ai, aj, aa)
    N = n ** 3
    one = P.Vec.create(N, comm=L.COMM_SELF); L.VecSet(one.h, 1.0)
    b = one.duplicate(); A.mult(one, b)
    npat, nvpat = _which_spmv(P, A)
    assert (nvpat == 0 and npat > 0) if value_patterns == "0" else nvpat > 0, (npat, nvpat)
    outs = []
    for opts in ("-ksp_cg_fused 3", "-ksp_cg_fused 0", ""):
        x = one.duplicate(); L.VecSet(x.h, 0.0)
        k = P.KSP(comm=L.COMM_SELF)
        k.set_operators(A)
        L.PetscOptionsClear(); L.PetscOptionsInsertString(("-ksp_type %s -pc_type jacobi %s %s" % (ksp_type_for("cg", opts), opts, vp_opt)).encode())
        k.set_tolerances(rtol=1e-8, max_it=5000)
        k.set_from_options()
        k.record_history()
        k.solve(b, x)
        L.PetscOptionsClear()
        outs.append((k.its, k.reason, k.history().copy(), x))
    (its, reason, h, x), (its0, reason0, h0, x0), (its4, reason4, h4, x4) = outs
    assert _which_spmv(P, A) == (npat, nvpat)                                                      # no re-upload switched the kernel under the solves
    assert reason4 == 2 and abs(its4 - its0) <= 2 and np.max(np.abs(x4.array() - 1.0)) < 1e-4     # the default (level 4): to rounding
    assert np.allclose(h4[:50], h0[:50], rtol=1e-9, atol=0)                                        # and so does the head of its residual history
    del x4
    assert reason == reason0 == 2 and its == its0 and 200 < its < 2000
    assert np.array_equal(bits(h), bits(h0))
    xa = x.array()
    assert np.array_equal(bits(xa), bits(x0.array()))
    assert np.max(np.abs(xa - 1.0)) < 1e-4
    del xa
    # recomputed preconditioned residual: diag(A) = 6 everywhere
    r = one.duplicate(); A.mult(x, r)
    L.VecAYPX(r.h, -1.0, b.h)                          # r = b - A x
    assert abs(r.norm() / 6.0 - h[-1]) <= 1e-6 * h[0]


def test_no_device_memory_leak_over_object_lifetimes(P):
    """create / solve / destroy cycles of every object kind on the path (Seq and MPI matrices, BAIJ, KSP with each PC
    incl. ILU's hipGraph and the transpose cache) must give the device memory back: hipMemGetInfo after 30 cycles
    equals the value after the warm-up cycles to within 16 MiB (runtime pools may still settle; a leaked matrix, vector,
    plan, factor or graph of this size would lose several hundred MiB over the loop)."""
    import ctypes as C
    import gc
    L = P.lib()
    k = P.load_kernels()

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert k.mi355x_mem_info(C.byref(f), C.byref(t)) == 0
        return f.value

    ai, aj, aa = P.gen_poisson7(48, 40, 32)              # 61 440 rows: ~5 MB of matrix, 0.5 MB per vector
    n = ai.size - 1
    b = np.cos(0.3 * np.arange(n))

    def cycle(i):
        for ksp, pc in (("cg", "jacobi"), ("gmres", "ilu"), ("bcgs", "bjacobi"), ("cg", "none")):
            solve(P, ai, aj, aa, b, ksp, pc, rtol=1e-6, mpi=(i % 2 == 1 and pc != "ilu"))   # PCILU wants a sequential matrix, as in the reference
        A = P.Mat.from_csr(ai, aj, aa)
        x = P.Vec.from_array(b, comm=L.COMM_SELF); y = x.duplicate()
        L.MatMultTranspose(A.h, x.h, y.h)
        bi = np.arange(0, 201, dtype=np.int32); bj = np.arange(200, dtype=np.int32)
        Ab = P.Mat.from_bsr(3, bi, bj, np.ones(200 * 9))
        xb = P.Vec.from_array(np.ones(600), comm=L.COMM_SELF); yb = xb.duplicate()
        Ab.mult(xb, yb)
        for o in (A, x, y, Ab, xb, yb):
            o.destroy()
        gc.collect()
        k.mi355x_device_synchronize()

    import psutil
    proc = psutil.Process()
    for i in range(8):
        cycle(i)
    base = free_bytes(); rss0 = proc.memory_info().rss
    for i in range(30):
        cycle(i)
    after = free_bytes(); rss1 = proc.memory_info().rss
    # host side: every cycle allocates ~25 MB of CSR copies, factors and work arrays; all of it has to come back
    assert rss1 - rss0 < (24 << 20), "host RSS grew by %d MB over 30 object life cycles" % ((rss1 - rss0) >> 20)
    assert after >= base - (16 << 20), "device memory shrank by %d bytes over 30 object life cycles" % (base - after)


def test_standalone_c_program_reproduces_the_tutorial_outputs(built):
    """examples/poisson2d.c -- plain C on the PETSc-named API, MatSetValues assembly, options from the command line, no
    Python in the loop -- run as the reference's makefile runs its ex2 tutorial: the complete stdout must equal
    output/ex2_1.out (default GMRES + ILU(0) on one rank), and BASELINE.json's configs[0] must give 160 iterations."""
    import subprocess
    exe = os.path.join(os.path.dirname(G), "..", "examples", "poisson2d")
    exe = os.path.abspath(exe)
    assert os.path.exists(exe), "examples/poisson2d was not built (python -c 'import __graft_entry__ as g; g.build()')"
    r = subprocess.run([exe, "-m", "5", "-n", "5", "-ksp_monitor_short", "-ksp_gmres_cgs_refinement_type", "refine_always"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout == open(os.path.join(G, "ksp_tutorials", "ex2_1.out")).read()
    # ex2_bjacobi.out (makefile:350: 4 ranks, ONE block spanning them = the whole matrix; same arithmetic on one rank)
    r = subprocess.run([exe, "-pc_type", "bjacobi", "-pc_bjacobi_blocks", "1", "-ksp_monitor_short", "-sub_pc_type", "jacobi", "-sub_ksp_type", "gmres"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout == open(os.path.join(G, "ksp_tutorials", "ex2_bjacobi.out")).read()
    r = subprocess.run([exe, "-m", "100", "-n", "100", "-ksp_type", "cg", "-pc_type", "jacobi", "-ksp_converged_reason"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.splitlines()
    assert lines[0] == "Linear solve converged due to CONVERGED_RTOL iterations 160"
    assert lines[1].startswith("Norm of error 5.7078") and lines[1].endswith("iterations 160")
    # an unknown type is an error message and a non-zero exit, not a crash
    r = subprocess.run([exe, "-ksp_type", "nosuchmethod"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "nosuchmethod" in r.stderr


def test_value_updates_on_the_device_copy(P):
    """MatScale / MatDiagonalScale / MatZeroEntries after the matrix has been used on the device: host and
    device copies are updated side by side (no re-upload), results bit-identical to the oracle's loops (aij.c:2055:
    left pass then right pass) and to the route through a fresh upload; the cached transpose follows."""
    L = P.lib()
    ai, aj, aa = pb.lap2d(23, 19)
    n = ai.size - 1
    aa = aa * (1.0 + 0.3 * np.sin(np.arange(aa.size)))
    l, r = 1.0 + 0.5 * np.cos(np.arange(n)), 2.0 + np.sin(0.7 * np.arange(n))
    x = np.cos(0.3 * np.arange(n))
    vx = V(P, x); vy = V(P, np.zeros(n)); vl, vr = V(P, l), V(P, r)
    for used_first in (True, False):                      # device-side update / host update + upload
        A = P.Mat.from_csr(ai, aj, aa)
        if used_first:
            A.mult(vx, vy); L.MatMultTranspose(A.h, vx.h, vy.h)       # uploads A and builds the transpose cache
        L.MatDiagonalScale(A.h, vl.h, vr.h)
        ref = orc.diagonal_scale(ai, aj, aa, l, r)
        A.mult(vx, vy)
        assert np.array_equal(bits(vy.array()), bits(orc.matmult(ai, aj, ref, x)[0]))
        L.MatMultTranspose(A.h, vx.h, vy.h)
        assert np.allclose(vy.array(), orc.spmv_t(ai, aj, ref, x, n), rtol=0, atol=1e-12)
        L.MatDiagonalScale(A.h, None, vr.h)
        ref = orc.diagonal_scale(ai, aj, ref, None, r)
        L.MatDiagonalScale(A.h, vl.h, None)
        ref = orc.diagonal_scale(ai, aj, ref, l, None)
        L.MatScale(A.h, -0.37)
        ref = -0.37 * ref
        A.mult(vx, vy)
        assert np.array_equal(bits(vy.array()), bits(orc.matmult(ai, aj, ref, x)[0]))
        d = vx.duplicate(); L.MatGetDiagonal(A.h, d.h)
        assert np.array_equal(bits(d.array()), bits(orc.get_diagonal(ai, aj, ref)))
        L.MatZeroEntries(A.h)
        A.mult(vx, vy)
        assert np.all(vy.array() == 0.0)
        nup = C.c_int()
        L.MatHIPMI355XGetUploadCount(A.h, C.byref(nup))
        assert nup.value == 1, "values crossed PCIe %d times" % nup.value   # used_first: only the first use; else: one upload after the first update
        A.destroy()


def test_value_patterns_follow_the_values(P):
    """-mat_hipmi355x_value_patterns (default on): a constant-coefficient operator runs from the row dictionary (27 kinds
    of rows for the 7-point box); every device-side change of the values (MatScale, MatDiagonalScale, MatZeroEntries) drops
    it, the products stay bit-identical to the oracle's throughout, and the next upload derives it again.  With the option
    off the dictionary is never built."""
    L = P.lib()
    ai, aj, aa = orc.gen_p7(12, 11, 10)
    n = ai.size - 1
    x = np.cos(0.3 * np.arange(n)); l = 1.0 + 0.5 * np.cos(np.arange(n))
    vx = V(P, x); vy = V(P, np.zeros(n)); vl = V(P, l)
    nv = C.c_int()

    def count(A):
        L.MatHIPMI355XGetValuePatterns(A.h, C.byref(nv)); return nv.value

    def check(A, vals):
        A.mult(vx, vy)
        assert np.array_equal(bits(vy.array()), bits(orc.matmult(ai, aj, vals, x)[0]))
        L.MatMultAdd(A.h, vx.h, vl.h, vy.h)
        assert np.array_equal(bits(vy.array()), bits(orc.spmv_add(ai, aj, vals, x, l)))

    A = P.Mat.from_csr(ai, aj, aa)
    check(A, aa)
    assert count(A) == 27
    L.MatScale(A.h, 0.5)                                   # on the device: the dictionary describes the old values
    assert count(A) == 0
    check(A, 0.5 * aa)
    L.MatDiagonalScale(A.h, vl.h, None)
    ref = orc.diagonal_scale(ai, aj, 0.5 * aa, l, None)
    assert count(A) == 0
    check(A, ref)
    # new values into the same pattern from the host: one upload, dictionary derived again
    L.MatZeroEntries(A.h)
    assert count(A) == 0
    for r in range(n):
        cols = np.ascontiguousarray(aj[ai[r]:ai[r + 1]]); v = np.ascontiguousarray(3.0 * aa[ai[r]:ai[r + 1]])
        rr = np.array([r], np.int32)
        L.MatSetValues(A.h, 1, rr.ctypes.data_as(C.c_void_p), cols.size, cols.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), 1)
    L.MatAssemblyBegin(A.h, 0); L.MatAssemblyEnd(A.h, 0)
    check(A, 3.0 * aa)
    assert count(A) == 27
    # varying coefficients into the same pattern: refused, values streamed
    L.MatDiagonalScale(A.h, vl.h, None)                    # device side
    A2 = P.Mat.from_csr(ai, aj, orc.diagonal_scale(ai, aj, 3.0 * aa, l, None))      # the same values through an upload
    check(A2, orc.diagonal_scale(ai, aj, 3.0 * aa, l, None))
    assert count(A2) == 0 and count(A) == 0
    A.destroy(); A2.destroy()
    # switched off
    L.PetscOptionsSetValue(b"-mat_hipmi355x_value_patterns", b"0")
    try:
        B = P.Mat.from_csr(ai, aj, aa)
        check(B, aa)
        assert count(B) == 0
        B.destroy()
    finally:
        L.PetscOptionsClear()


def test_setvaluesbatch_device_assembly(P):
    """MatSetValuesBatch (matrix.c:1698): first assembly goes through the reference's loop of MatSetValues; every later
    one with the same connectivity is a device-side value assembly through the cached map -- bit-identical to the
    loop (contributions added per nonzero in call order, also with repeated and negative indices inside a block),
    host mirror refreshed, and no upload of the matrix in between."""
    L = P.lib()
    rng = np.random.default_rng(42)
    nn, nb, bs = 900, 2500, 4
    rows = rng.integers(0, nn, (nb, bs)).astype(np.int32)
    rows[::7, 1] = rows[::7, 0]                            # a repeated index inside a block
    rows[::11, 2] = -1                                     # a negative index: ignored
    v1 = rng.standard_normal((nb, bs, bs)); v2 = rng.standard_normal((nb, bs, bs))

    def new_mat():
        A = P.Mat(); L.MatCreate(L.COMM_SELF, C.byref(A.h))
        L.MatSetSizes(A.h, nn, nn, nn, nn); L.MatSetType(A.h, b"seqaijhipmi355x"); L.MatSetUp(A.h)
        return A

    def batch(A, v):
        L.MatSetValuesBatch(A.h, nb, bs, rows.ctypes.data_as(C.c_void_p), np.ascontiguousarray(v).ctypes.data_as(C.c_void_p))
        L.MatAssemblyBegin(A.h, 0); L.MatAssemblyEnd(A.h, 0)

    def values(A):
        m = C.c_int(); pi, pj, pa = C.c_void_p(), C.c_void_p(), C.c_void_p()
        L.MatSeqAIJGetArrays(A.h, C.byref(m), C.byref(pi), C.byref(pj), C.byref(pa))
        ai = np.ctypeslib.as_array((C.c_int * (m.value + 1)).from_address(pi.value)).copy()
        return ai, np.ctypeslib.as_array((C.c_double * int(ai[-1])).from_address(pa.value)).copy()

    A = new_mat(); batch(A, v1)                            # loop route (nothing assembled yet)
    x = P.Vec.from_array(np.cos(0.3 * np.arange(nn)), comm=L.COMM_SELF); y = x.duplicate()
    A.mult(x, y)                                           # first (and only) upload
    batch(A, v2)                                           # device route: A += second set
    ai, a_dev = values(A)
    A.mult(x, y); y_dev = y.array()
    nup = C.c_int(); L.MatHIPMI355XGetUploadCount(A.h, C.byref(nup))
    assert nup.value == 1
    L.PetscOptionsClear()
    B = new_mat(); batch(B, v1)
    for b in range(nb):                                    # the reference's default: one MatSetValues per block
        r = rows[b].ctypes.data_as(C.c_void_p)
        L.MatSetValues(B.h, bs, r, bs, r, np.ascontiguousarray(v2[b]).ctypes.data_as(C.c_void_p), 2)
    L.MatAssemblyBegin(B.h, 0); L.MatAssemblyEnd(B.h, 0)
    bi, b_ref = values(B)
    assert np.array_equal(ai, bi) and np.array_equal(bits(a_dev), bits(b_ref))
    B.mult(x, y)
    assert np.array_equal(bits(y_dev), bits(y.array()))
    # zero and re-assemble on the device: the map is reused
    L.MatZeroEntries(A.h); batch(A, v2)
    C_ = new_mat(); batch(C_, v2)
    assert np.array_equal(bits(values(A)[1]), bits(values(C_)[1]))
    L.MatHIPMI355XGetUploadCount(A.h, C.byref(nup))
    assert nup.value == 1


def test_standalone_load_and_solve(built):
    """examples/loadsolve.c (the flow of the reference's tutorial ex10: MatSetType before the sizes are known, MatLoad,
    VecLoad, KSPSetFromOptions, KSPSolve) on the reference's own data files: GMRES(30) + block Jacobi converges in 4
    iterations on spd-real-int32-float64 (as the compiled reference did), BiCGStab + Jacobi solves the
    nonsymmetric one."""
    import subprocess
    exe = os.path.abspath(os.path.join(os.path.dirname(G), "..", "examples", "loadsolve"))
    assert os.path.exists(exe)
    r = subprocess.run([exe, "-f", os.path.join(G, "matrices", "spd-real-int32-float64"), "-ksp_type", "gmres", "-pc_type", "bjacobi"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.splitlines()[0] == "Number of iterations =   4"
    r = subprocess.run([exe, "-f", os.path.join(G, "matrices", "ns-real-int32-float64"), "-ksp_type", "bcgs", "-pc_type", "jacobi", "-ksp_rtol", "1e-10"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Residual norm < 1.e-6 |b| (reason 2)" in r.stdout


@pytest.mark.parametrize("pc", ["jacobi", "ilu", "bjacobi"])
def test_time_step_loop_with_device_side_value_updates(P, pc):
    """A time-step / Newton-style loop: the same Mat and KSP are re-used while the values change on the device
    (MatZeroEntries + MatSetValuesBatch re-assembly, MatDiagonalScale, MatScale), KSPSetOperators announces each change
    and the preconditioner is rebuilt from the updated values (Jacobi from the device copy, ILU(0) from the refreshed
    host mirror), with ordinary host-side MatSetValues mixed in between (they must invalidate what the device-side
    updates stamped).  Every step must give the bits of a solve with a freshly built matrix and solver."""
    L = P.lib()
    rng = np.random.default_rng(9)
    nn, ne, bs = 400, 1400, 3
    rows = np.stack([rng.choice(nn, bs, replace=False) for _ in range(ne)]).astype(np.int32)
    def elems(seed):
        r = np.random.default_rng(seed)
        k = r.standard_normal((ne, bs, bs)); k = 0.2 * (k + k.transpose(0, 2, 1))
        k[:, np.arange(bs), np.arange(bs)] = 2.0 + r.random((ne, bs))          # dominant diagonals
        return np.ascontiguousarray(k)

    def build(v, l=None, r=None, scale=None):
        A = P.Mat(); L.MatCreate(L.COMM_SELF, C.byref(A.h))
        L.MatSetSizes(A.h, nn, nn, nn, nn); L.MatSetType(A.h, b"seqaijhipmi355x"); L.MatSetUp(A.h)
        L.MatSetValuesBatch(A.h, ne, bs, rows.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p))
        ident = np.arange(nn, dtype=np.int32); ones = np.ones(nn)
        for i in range(nn):                                                          # every row gets a diagonal entry
            L.MatSetValues(A.h, 1, ident[i:i + 1].ctypes.data_as(C.c_void_p), 1, ident[i:i + 1].ctypes.data_as(C.c_void_p), ones[i:i + 1].ctypes.data_as(C.c_void_p), 2)
        L.MatAssemblyBegin(A.h, 0); L.MatAssemblyEnd(A.h, 0)
        if l is not None:
            L.MatDiagonalScale(A.h, l.h, r.h)
        if scale is not None:
            L.MatScale(A.h, scale)
        return A

    def solve_with(k, A, b):
        x = b.duplicate(); L.VecSet(x.h, 0.0)
        k.set_operators(A)
        k.solve(b, x)
        return bits(x.array()).copy(), k.its, k.reason

    def new_ksp():
        k = P.KSP(comm=L.COMM_SELF)
        L.PetscOptionsClear(); L.PetscOptionsInsertString(("-ksp_type gmres -pc_type %s" % pc).encode())
        k.set_tolerances(rtol=1e-10); k.set_from_options(); L.PetscOptionsClear()
        return k

    b = P.Vec.from_array(np.cos(0.3 * np.arange(nn)), comm=L.COMM_SELF)
    dl = 1.0 + 0.3 * np.cos(np.arange(nn)); vl = P.Vec.from_array(dl, comm=L.COMM_SELF)
    A = build(elems(0))
    k = new_ksp()
    got = [solve_with(k, A, b)]
    ident = np.arange(nn, dtype=np.int32); ones = np.ones(nn)
    for step in (1, 2):
        L.MatZeroEntries(A.h)                                                        # device + host, no upload
        v = elems(step)
        L.MatSetValuesBatch(A.h, ne, bs, rows.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p))   # device-side assembly
        for i in range(nn):
            L.MatSetValues(A.h, 1, ident[i:i + 1].ctypes.data_as(C.c_void_p), 1, ident[i:i + 1].ctypes.data_as(C.c_void_p), ones[i:i + 1].ctypes.data_as(C.c_void_p), 2)
        L.MatAssemblyBegin(A.h, 0); L.MatAssemblyEnd(A.h, 0)
        L.MatDiagonalScale(A.h, vl.h, vl.h)
        L.MatScale(A.h, 1.5)
        got.append(solve_with(k, A, b))
    ref = [solve_with(new_ksp(), build(elems(0)), b)]
    for step in (1, 2):
        ref.append(solve_with(new_ksp(), build(elems(step), vl, vl, 1.5), b))
    for g, r_ in zip(got, ref):
        assert g[1:] == r_[1:] and g[2] == 2 and np.array_equal(g[0], r_[0])


def test_time_step_loop_with_a_transpose_product_stays_on_the_device(P):
    """MatMultTranspose in a time-step loop (aij.c:1078-1135): the explicit A^T behind it is built ONCE on the host; when only the
    values change -- on the device (MatScale, MatDiagonalScale, MatZeroEntries + MatSetValuesBatch) or by a same-pattern
    re-assembly from the host -- its values are refreshed by one gather over the device copy through the permutation kept
    from the build.  Every step: the bits of the oracle's transpose product on the current values; host builds stay at 1, and the
    device-side updates cause no upload of the matrix either."""
    L = P.lib()
    ai, aj, aa = orc.gen_p7(11, 9, 7)
    aa = aa * (1.0 + 0.3 * np.sin(np.arange(aa.size)))                 # nonsymmetric values: A^T x != A x
    n = ai.size - 1
    A = P.Mat.from_csr(ai, aj, aa)
    x = np.cos(0.21 * np.arange(n)); vx, vy, vz = V(P, x), V(P, np.zeros(n)), V(P, np.sin(0.1 * np.arange(n)))
    cur = aa.copy()

    def check():
        L.MatMultTranspose(A.h, vx.h, vy.h)
        assert np.array_equal(bits(vy.array()), bits(orc.spmv_t(ai, aj, cur, x, n)))
        L.MatMultTransposeAdd(A.h, vx.h, vz.h, vy.h)
        assert np.array_equal(bits(vy.array()), bits(orc.spmv_t_add(ai, aj, cur, x, vz.array(), n)))

    def counts():
        b_, r_, u_ = C.c_int(), C.c_int(), C.c_int()
        L.MatHIPMI355XGetTransposeCounts(A.h, C.byref(b_), C.byref(r_)); L.MatHIPMI355XGetUploadCount(A.h, C.byref(u_))
        return b_.value, r_.value, u_.value

    check()
    assert counts()[:2] == (1, 0)
    up0 = counts()[2]
    rows = np.repeat(np.arange(n), np.diff(ai))
    dl = 1.0 + 0.4 * np.cos(np.arange(n)); dr = 2.0 - 0.3 * np.sin(0.5 * np.arange(n))
    vl, vr = V(P, dl), V(P, dr)
    for step in range(4):
        if step % 2 == 0:
            L.MatScale(A.h, 1.25 + step); cur = (1.25 + step) * cur                                   # device copy scaled in place
        else:
            L.MatDiagonalScale(A.h, vl.h, vr.h); cur = (cur * dl[rows]) * dr[aj]                       # (a l) r, aij.c:2055-2092
        check()
        assert counts() == (1, step + 1, up0)                          # no host rebuild of A^T, no upload of the matrix
    # a same-pattern re-assembly from the host: one upload of the VALUES, A^T follows on the device
    new = aa * (2.0 + np.cos(np.arange(aa.size)))
    L.MatZeroEntries(A.h)
    for r in range(n):
        cols = aj[ai[r]:ai[r + 1]].copy(); vals = new[ai[r]:ai[r + 1]].copy(); rr = np.array([r], dtype=np.int32)
        L.MatSetValues(A.h, 1, rr.ctypes.data_as(C.c_void_p), cols.size, cols.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p), 2)
    L.MatAssemblyBegin(A.h, 0); L.MatAssemblyEnd(A.h, 0)
    cur = new.copy()
    check()
    b_, r_, u_ = counts()
    assert (b_, r_) == (1, 5) and u_ == up0 + 1


def test_pattern_change_after_device_use(P):
    """new nonzeros inserted after the matrix has been used on the device (and after a device-side batch assembly has
    cached its map): plan, index dictionary, transpose cache and batch map are rebuilt; MatMult / MatMultTranspose equal
    the oracle on the new matrix, and a batch assembly on the new pattern equals the loop of MatSetValues."""
    L = P.lib()
    ai, aj, aa = pb.lap2d(17, 13)
    n = ai.size - 1
    A = P.Mat(); L.MatCreate(L.COMM_SELF, C.byref(A.h))
    L.MatSetSizes(A.h, n, n, n, n); L.MatSetType(A.h, b"seqaijhipmi355x"); L.MatSetUp(A.h)
    for r in range(n):
        cols = aj[ai[r]:ai[r + 1]].copy(); vals = aa[ai[r]:ai[r + 1]].copy()
        rr = np.array([r], dtype=np.int32)
        L.MatSetValues(A.h, 1, rr.ctypes.data_as(C.c_void_p), cols.size, cols.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p), 1)
    L.MatAssemblyBegin(A.h, 0); L.MatAssemblyEnd(A.h, 0)
    x = np.cos(0.3 * np.arange(n)); vx = V(P, x); vy = V(P, np.zeros(n))
    A.mult(vx, vy); L.MatMultTranspose(A.h, vx.h, vy.h)
    rows = np.array([[0, 1], [5, 6]], dtype=np.int32); v = np.arange(8.0).reshape(2, 2, 2)
    L.MatSetValuesBatch(A.h, 2, 2, rows.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p))      # builds and caches a map
    L.MatAssemblyBegin(A.h, 0); L.MatAssemblyEnd(A.h, 0)
    import scipy.sparse as sp
    S = sp.csr_matrix((aa, aj, ai), shape=(n, n)).tolil()
    for b_ in range(2):
        for i in range(2):
            for j in range(2):
                S[rows[b_, i], rows[b_, j]] += v[b_, i, j]
    # three entries that are not in the pattern yet, far from the diagonal
    new = [(3, n - 2, 2.5), (n - 1, 0, -1.25), (40, 100, 0.5)]
    for r, c, val in new:
        rr, cc, vv = np.array([r], np.int32), np.array([c], np.int32), np.array([val])
        L.MatSetValues(A.h, 1, rr.ctypes.data_as(C.c_void_p), 1, cc.ctypes.data_as(C.c_void_p), vv.ctypes.data_as(C.c_void_p), 2)
        S[r, c] += val
    L.MatAssemblyBegin(A.h, 0); L.MatAssemblyEnd(A.h, 0)
    S = sp.csr_matrix(S); S.sort_indices()
    si, sj, sa = S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data.astype(np.float64)
    A.mult(vx, vy)
    assert np.array_equal(bits(vy.array()), bits(orc.matmult(si, sj, sa, x)[0]))
    L.MatMultTranspose(A.h, vx.h, vy.h)
    assert np.allclose(vy.array(), orc.spmv_t(si, sj, sa, x, n), rtol=0, atol=1e-12)
    # batch assembly on the NEW pattern, touching one of the new entries
    rows2 = np.array([[3, n - 2], [5, 6]], dtype=np.int32)
    L.MatSetValuesBatch(A.h, 2, 2, rows2.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p))
    L.MatAssemblyBegin(A.h, 0); L.MatAssemblyEnd(A.h, 0)
    S = S.tolil()
    for b_ in range(2):
        for i in range(2):
            for j in range(2):
                S[rows2[b_, i], rows2[b_, j]] += v[b_, i, j]
    S = sp.csr_matrix(S); S.sort_indices()
    A.mult(vx, vy)
    assert np.allclose(vy.array(), orc.spmv(S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data.astype(np.float64), x), rtol=0, atol=1e-13)


def multidof_stencil(m, n, dof, seed):
    """2-D 5-point node stencil with dof x dof coupling, point-wise AIJ: rows of 3*dof..5*dof entries (<= 16 for dof <= 3)"""
    bi, bj, _ = pb.lap2d(m, n)
    rng = np.random.default_rng(seed)
    return pb.expand_blocks(bi, bj, rng.standard_normal((bj.size, dof, dof)))


def set_options(L, s):
    L.PetscOptionsClear()
    if s:
        L.PetscOptionsInsertString(s.encode())


def test_inode_matrices_carry_the_reference_dispatch(P):
    """SURVEY 8(a5).  A seqaij matrix whose rows repeat their pattern (several dof per node) is run by the reference
    through MatMult_SeqAIJ_Inode, which sums two products at a time.  The HIP type makes the same decision with the same
    loop (Mat_CheckInode, node sizes identical to the oracle's), groups the rows on the device, and for short rows
    reproduces THAT routine bit for bit; -mat_no_inode restores MatMult_SeqAIJ's order, also bit for bit."""
    L = P.lib()
    for dof in (2, 3):
        ai, aj, aa = multidof_stencil(23, 19, dof, 5 + dof)
        m = ai.size - 1
        x = rnd(m, 21); z = rnd(m, 22)
        nodes_ref, ns_ref = orc.check_inode(ai, aj)
        assert nodes_ref == m // dof
        for opt, inode in (("", True), ("-mat_no_inode", False), ("-mat_inode_limit 1", False), ("-mat_hipmi355x_index_compression 0", True)):
            set_options(L, opt)
            A = P.Mat.from_csr(ai, aj, aa)
            vx, vy, vz = V(P, x), V(P, np.zeros(m)), V(P, z)
            A.mult(vx, vy)
            nodes, groups, shared = C.c_int(), C.c_int(), C.c_int()
            L.MatHIPMI355XGetInodeInfo(A.h, C.byref(nodes), C.byref(groups), C.byref(shared))
            set_options(L, "")
            ref = orc.spmv_inode(ai, aj, aa, x) if inode else orc.spmv(ai, aj, aa, x)
            assert nodes.value == (nodes_ref if inode else 0)
            if opt == "-mat_hipmi355x_index_compression 0":     # no offset dictionary: the rows of a node share one column list
                assert groups.value == nodes_ref and shared.value * dof == aj.size
            assert np.array_equal(bits(vy.array()), bits(ref))
            L.MatMultAdd(A.h, vx.h, vz.h, vy.h)
            refadd = orc.matmult(ai, aj, aa, x, z)[0] if inode else orc.spmv_add(ai, aj, aa, x, z)
            assert np.array_equal(bits(vy.array()), bits(refadd))
        assert not np.array_equal(bits(orc.spmv_inode(ai, aj, aa, x)), bits(orc.spmv(ai, aj, aa, x)))   # the two orders do differ


def test_fem_like_config4_shape_small(P):
    """BASELINE configs[3] stand-in at test size (tests/problems.py gen_fem3: 3 dof per node, hexahedral flange mesh,
    RCM-ordered): grouped-row kernel vs the reference's dispatch (rows of ~60-81 entries: several lanes per row,
    BASELINE.md tolerance), GMRES(30) + block Jacobi histories vs the oracle"""
    L = P.lib()
    ai, aj, aa = pb.gen_fem3(14, 14, 8)
    m = ai.size - 1
    x = rnd(m, 31)
    scale = np.zeros(m); np.add.at(scale, np.repeat(np.arange(m), np.diff(ai)), np.abs(aa * x[aj]))
    A = P.Mat.from_csr(ai, aj, aa)
    vx, vy = V(P, x), V(P, np.zeros(m))
    A.mult(vx, vy)
    nodes, groups, shared = C.c_int(), C.c_int(), C.c_int()
    L.MatHIPMI355XGetInodeInfo(A.h, C.byref(nodes), C.byref(groups), C.byref(shared))
    nodes_ref, _ = orc.check_inode(ai, aj)
    assert nodes.value == nodes_ref == m // 3 and groups.value == nodes_ref and shared.value * 3 == aj.size
    ref, used = orc.matmult(ai, aj, aa, x)
    assert used == nodes_ref
    assert np.all(np.abs(vy.array() - ref) <= 1e-12 * scale)
    b = orc.spmv(ai, aj, aa, np.ones(m))
    xh, h, its, reason = solve(P, ai, aj, aa, b, "gmres", "bjacobi", opts="-sub_pc_type jacobi", rtol=1e-8)
    xr, hr, itsr, rr = orc.ksp_solve(ai, aj, aa, b, ksp="gmres", pc="bjacobi", blocks=[0, m], sub_ksp="preonly", sub_pc="jacobi", rtol=1e-8)
    assert its == itsr and reason == rr
    assert np.allclose(h, hr, rtol=1e-8, atol=0)
    assert np.linalg.norm(xh - 1.0) < 1e-6 * np.sqrt(m)


@pytest.mark.parametrize("bs,nbr,nbc,seed", [(2, 300, 300, 1), (3, 257, 400, 2), (4, 190, 150, 3), (5, 120, 121, 4), (2, 1, 7, 5)])
def test_blocked_companion_over_block_sizes_and_shapes(P, bs, nbr, nbc, seed):
    """The blocked companion asked for (-mat_hipmi355x_blocked 1) on random block-structured AIJ matrices, bs = 2..5, square and
    rectangular, block rows of uneven length incl. empty ones: MatMult, MatMultAdd, MatMultTranspose and MatMultTransposeAdd against
    the exact sums within 1e-12 sum |a x|, and the companion follows MatScale on the device."""
    import scipy.sparse as sp
    L = P.lib()
    rng = np.random.default_rng(seed)
    pat = sp.random(nbr, nbc, density=min(1.0, 9.0 / nbc), random_state=seed, format="csr")
    pat.data[:] = 1.0
    S = sp.kron(pat, np.ones((bs, bs)), format="csr"); S.sort_indices()
    ai, aj = S.indptr.astype(np.int32), S.indices.astype(np.int32)
    aa = rng.standard_normal(aj.size)
    m, n = nbr * bs, nbc * bs
    x, z, xt, zt = rnd(n, 51), rnd(m, 52), rnd(m, 53), rnd(n, 54)
    Sv = sp.csr_matrix((aa, aj, ai), shape=(m, n))
    Sa = abs(Sv)
    set_options(L, "-mat_hipmi355x_blocked 0")
    A0 = P.Mat.from_csr(ai, aj, aa, ncols=n)
    vx, vy0 = V(P, x), V(P, np.zeros(m))
    A0.mult(vx, vy0)
    set_options(L, "-mat_hipmi355x_blocked 1")
    A = P.Mat.from_csr(ai, aj, aa, ncols=n)
    vy = V(P, np.zeros(m))
    A.mult(vx, vy)
    set_options(L, "")
    bs_, nb_ = C.c_int(), C.c_int()
    L.MatHIPMI355XGetBlockedInfo(A.h, C.byref(bs_), C.byref(nb_))
    assert bs_.value == bs and nb_.value == pat.nnz
    # (rows this short are summed by one lane in the inode routine's pair order by the grouped-row kernel -- the reference's bits, which is
    # why the companion is not chosen for them unasked; the block kernel's order differs in the last place)
    tol = 1e-12 * (Sa @ np.abs(x)) + 1e-300
    assert np.all(np.abs(vy.array() - Sv @ x) <= tol) and np.all(np.abs(vy.array() - vy0.array()) <= tol)
    vz, vw = V(P, z), V(P, np.zeros(m))
    L.MatMultAdd(A.h, vx.h, vz.h, vw.h)
    assert np.all(np.abs(vw.array() - (z + Sv @ x)) <= tol + 1e-12 * np.abs(z))
    vxt, vyt, vzt = V(P, xt), V(P, np.zeros(n)), V(P, zt)
    L.MatMultTranspose(A.h, vxt.h, vyt.h)
    tolT = 1e-12 * (Sa.T @ np.abs(xt)) + 1e-300
    assert np.all(np.abs(vyt.array() - Sv.T @ xt) <= tolT)
    L.MatMultTransposeAdd(A.h, vxt.h, vzt.h, vyt.h)
    assert np.all(np.abs(vyt.array() - (zt + Sv.T @ xt)) <= tolT + 1e-12 * np.abs(zt))
    L.MatScale(A.h, -0.5)
    A.mult(vx, vy)
    assert np.all(np.abs(vy.array() + 0.5 * (Sv @ x)) <= 1e-12 * (Sa @ np.abs(x)) + 1e-300)
    L.MatMultTranspose(A.h, vxt.h, vyt.h)
    assert np.all(np.abs(vyt.array() + 0.5 * (Sv.T @ xt)) <= tolT)


def test_blocked_companion_of_a_three_dof_matrix(P):
    """An AIJ matrix whose nodes are complete 3 x 3 blocks (gen_fem3: 77 nonzeros per row) is multiplied by the BAIJ row-block kernel
    over BCSR arrays laid out beside the CSR ones (-mat_hipmi355x_blocked, decided for rows of more than 16 nonzeros): the same sums as
    the grouped-row kernel bit for bit, MatMultAdd too, device-side value changes reach it by one gather; a matrix with an incomplete
    block, or with short rows, keeps the grouped-row kernel."""
    L = P.lib()
    ai, aj, aa = pb.gen_fem3(14, 14, 8)
    m = ai.size - 1
    x = rnd(m, 41); z = rnd(m, 42)
    rowof = np.repeat(np.arange(m), np.diff(ai))
    scale = np.zeros(m); np.add.at(scale, rowof, np.abs(aa * x[aj]))
    bs_, nb_ = C.c_int(), C.c_int()
    set_options(L, "-mat_hipmi355x_blocked 0")
    A0 = P.Mat.from_csr(ai, aj, aa)
    vx, vy0 = V(P, x), V(P, np.zeros(m))
    A0.mult(vx, vy0)
    L.MatHIPMI355XGetBlockedInfo(A0.h, C.byref(bs_), C.byref(nb_))
    assert bs_.value == 0 and nb_.value == 0
    set_options(L, "")
    A = P.Mat.from_csr(ai, aj, aa)
    vy, vz, vw = V(P, np.zeros(m)), V(P, z), V(P, np.zeros(m))
    A.mult(vx, vy)
    L.MatHIPMI355XGetBlockedInfo(A.h, C.byref(bs_), C.byref(nb_))
    assert bs_.value == 3 and nb_.value * 9 == aj.size
    ref, _ = orc.matmult(ai, aj, aa, x)
    assert np.all(np.abs(vy.array() - ref) <= 1e-12 * scale)
    assert np.array_equal(bits(vy.array()), bits(vy0.array()))          # the grouped-row kernel's sums
    L.MatMultAdd(A.h, vx.h, vz.h, vw.h)
    L.MatMultAdd(A0.h, vx.h, vz.h, vy0.h)
    assert np.array_equal(bits(vw.array()), bits(vy0.array()))
    # values changed on the device: MatScale, MatDiagonalScale
    dl = V(P, 1.0 + 0.1 * rnd(m, 43))
    L.MatScale(A.h, -1.5); L.MatDiagonalScale(A.h, dl.h, dl.h)
    A.mult(vx, vy)
    aa2 = ((aa * -1.5) * dl.array()[rowof]) * dl.array()[aj]
    ref2, _ = orc.matmult(ai, aj, aa2, x)
    assert np.all(np.abs(vy.array() - ref2) <= 1e-12 * 1.5 * 1.3 * scale)
    n_up = C.c_int()
    L.MatHIPMI355XGetUploadCount(A.h, C.byref(n_up))
    assert n_up.value == 1
    # the transpose products take the companion's block transpose (no scalar transpose is built), and follow the device values
    import scipy.sparse as sp
    S2 = sp.csr_matrix((aa2, aj, ai), shape=(m, m))
    tolT = 1e-12 * (abs(S2).T @ np.abs(x)) + 1e-300
    L.MatMultTranspose(A.h, vx.h, vy.h)
    assert np.all(np.abs(vy.array() - S2.T @ x) <= tolT)
    L.MatMultTransposeAdd(A.h, vx.h, vz.h, vw.h)
    assert np.all(np.abs(vw.array() - (z + S2.T @ x)) <= tolT + 1e-12 * np.abs(z))
    tb_, tr_ = C.c_int(), C.c_int()
    L.MatHIPMI355XGetTransposeCounts(A.h, C.byref(tb_), C.byref(tr_))
    assert tb_.value == 0
    L.MatScale(A.h, 2.0)
    L.MatMultTranspose(A.h, vx.h, vy.h)
    assert np.all(np.abs(vy.array() - 2.0 * (S2.T @ x)) <= 2.0 * tolT)
    L.MatScale(A.h, 0.5)
    # one entry of one block missing: not a BAIJ matrix any more
    keep = np.ones(aj.size, bool); keep[ai[5] + 1] = False
    ai2 = np.concatenate(([0], np.cumsum(np.bincount(rowof[keep], minlength=m)))).astype(np.int32)
    B = P.Mat.from_csr(ai2, aj[keep].astype(np.int32), aa[keep])
    vy2 = V(P, np.zeros(m))
    B.mult(vx, vy2)
    L.MatHIPMI355XGetBlockedInfo(B.h, C.byref(bs_), C.byref(nb_))
    assert bs_.value == 0
    # short rows (2 dof, 5 + 2 neighbours): decided against (the grouped-row kernel carries the reference's bits there); forced: taken
    ai3, aj3, aa3 = multidof_stencil(23, 19, 2, 7)
    m3 = ai3.size - 1
    x3 = rnd(m3, 44)
    for opt, want in (("", 0), ("-mat_hipmi355x_blocked 1 -mat_hipmi355x_index_compression 0", 2)):
        set_options(L, opt)
        A3 = P.Mat.from_csr(ai3, aj3, aa3)
        v3, w3 = V(P, x3), V(P, np.zeros(m3))
        A3.mult(v3, w3)
        set_options(L, "")
        L.MatHIPMI355XGetBlockedInfo(A3.h, C.byref(bs_), C.byref(nb_))
        assert bs_.value == want, (opt, bs_.value)
        s3 = np.zeros(m3); np.add.at(s3, np.repeat(np.arange(m3), np.diff(ai3)), np.abs(aa3 * x3[aj3]))
        assert np.all(np.abs(w3.array() - orc.spmv(ai3, aj3, aa3, x3)) <= 1e-12 * s3)


def test_same_nonzero_count_different_pattern(P):
    """a matrix that has been used on the device is re-preallocated and re-filled with ANOTHER pattern of the same
    nonzero count (and: MatLoad-style adoption of new arrays into a used Mat): nothing of the old mirror may survive"""
    L = P.lib()
    n = 300
    rng = np.random.default_rng(9)

    def pattern(seed):
        r = np.random.default_rng(seed)
        cols = [np.sort(r.choice(n, size=5, replace=False)).astype(np.int32) for _ in range(n)]
        ai = np.arange(0, 5 * n + 1, 5, dtype=np.int32)
        return ai, np.concatenate(cols), r.standard_normal(5 * n)

    def fill(A, ai, aj, aa):
        for r in range(n):
            rr = np.array([r], np.int32); cols = aj[ai[r]:ai[r + 1]].copy(); vals = aa[ai[r]:ai[r + 1]].copy()
            L.MatSetValues(A.h, 1, rr.ctypes.data_as(C.c_void_p), cols.size, cols.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p), 1)
        L.MatAssemblyBegin(A.h, 0); L.MatAssemblyEnd(A.h, 0)

    x = rng.standard_normal(n); vx = V(P, x); vy = V(P, np.zeros(n))
    A = P.Mat(); L.MatCreate(L.COMM_SELF, C.byref(A.h))
    L.MatSetSizes(A.h, n, n, n, n); L.MatSetType(A.h, b"seqaijhipmi355x")
    L.MatSeqAIJSetPreallocation(A.h, 5, None)
    p1 = pattern(1); fill(A, *p1)
    A.mult(vx, vy); L.MatMultTranspose(A.h, vx.h, vy.h)
    assert np.allclose(vy.array(), orc.spmv_t(*p1, x, n), rtol=0, atol=1e-12)
    L.MatSeqAIJSetPreallocation(A.h, 5, None)          # same nz per row, then other columns
    p2 = pattern(2); fill(A, *p2)
    assert not np.array_equal(p1[1], p2[1])
    A.mult(vx, vy)
    assert np.array_equal(bits(vy.array()), bits(orc.matmult(*p2, x)[0]))
    L.MatMultTranspose(A.h, vx.h, vy.h)
    assert np.allclose(vy.array(), orc.spmv_t(*p2, x, n), rtol=0, atol=1e-12)


@pytest.mark.parametrize("which", ["fem", "irr"])
def test_config4_full_size_properties(P, which):
    """BASELINE configs[3] at full size (~1.5 M rows, ~1.15e8 nonzeros: the FEM-like and the IRR stand-ins for Flan_1565,
    tests/problems.py): (1) sampled rows of y = A x against a host dot product of the same row, BASELINE.md tolerance;
    (2) the grouped-row plan and the plain plan (-mat_no_inode) are two different kernels over the same matrix and must
    agree to that tolerance everywhere; (3) linearity A(2x) == 2 A x bit for bit (scaling by 2 is exact)."""
    L = P.lib()
    ai, aj, aa = pb.gen_fem3() if which == "fem" else pb.gen_irr()
    m = ai.size - 1
    x = np.sin(0.37 * np.arange(m)) + 1.0
    vx, vy = V(P, x), V(P, np.zeros(m))
    A = P.Mat.from_csr(ai, aj, aa)
    A.mult(vx, vy)
    y = vy.array()
    nodes = C.c_int(); L.MatHIPMI355XGetInodeInfo(A.h, C.byref(nodes), None, None)
    assert (nodes.value == m // 3) if which == "fem" else (nodes.value == 0)
    rows = np.random.default_rng(3).integers(0, m, 4000)
    for r in rows:
        k0, k1 = ai[r], ai[r + 1]
        t = aa[k0:k1] * x[aj[k0:k1]]
        assert abs(y[r] - t.sum()) <= 1e-12 * np.abs(t).sum()
    L.VecScale(vx.h, 2.0)
    A.mult(vx, vy)
    assert np.array_equal(bits(vy.array()), bits(2.0 * y))
    if which == "fem":
        set_options(L, "-mat_no_inode")
        B = P.Mat.from_csr(ai, aj, aa)
        B.mult(vx, vy)
        set_options(L, "")
        absrow = np.zeros(m); np.add.at(absrow, np.repeat(np.arange(m), np.diff(ai)), np.abs(aa * x[aj]))
        assert np.all(np.abs(vy.array() - 2.0 * y) <= 2e-12 * absrow)


def test_config4_full_size_ilu0_application_carries_the_reference_bits(P):
    """The ILU(0) application of BASELINE configs[3] at FULL size (FEM-like stand-in for Flan_1565: 1.53 M rows, 509 728 nodes of 3
    rows, 2 x 2144 node levels) against the oracle's restatements of the two reference routines, bit for bit: node plans with
    the columns in column order = MatSolve_SeqAIJ_Inode (inode.c:2327-2760; what the reference runs on this factor), row-granular
    plans in column order = MatSolve_SeqAIJ_NaturalOrdering (aijfact.c:3126); the defaults (columns in dependency-level order, node
    plans and row plans) to 1e-13 and deterministic.  All through the split-role sync-free kernels; no application gives up."""
    L = P.lib()
    ai, aj, aa = pb.gen_fem3()
    n = ai.size - 1
    nodes, ns = orc.check_inode(ai, aj)
    f = orc.ilu0_factor(ai, aj, aa)
    bvec = np.cos(0.1 * np.arange(n)) + 0.01 * np.sin(np.arange(n))
    ref_inode = orc.ilu0_solve_inode(f, ns, bvec)
    ref_nat = orc.ilu0_solve(f, bvec)
    A = P.Mat.from_csr(ai, aj, aa)
    vb, vx = V(P, bvec), V(P, np.zeros(n))
    for opts, ref, exact in (("-pc_factor_hipmi355x_trisolve_order column", ref_inode, True), ("-pc_factor_hipmi355x_trisolve_nodes 0 -pc_factor_hipmi355x_trisolve_order column", ref_nat, True),
                             ("", ref_inode, False), ("-pc_factor_hipmi355x_trisolve_nodes 0", ref_nat, False)):
        pc = C.c_void_p()
        k = P.KSP(comm=L.COMM_SELF); k.set_operators(A); L.KSPGetPC(k.h, C.byref(pc)); L.PCSetType(pc, b"ilu")
        set_options(L, opts)
        L.raw("PCSetUp")(pc)
        set_options(L, "")
        got = C.c_int(); L.PCILUGetNodeInfo_HIPMI355X(pc, C.byref(got), None, None)
        assert got.value == (0 if "nodes 0" in opts else nodes)
        L.raw("PCApply")(pc, vb.h, vx.h)
        xa = vx.array().copy()
        if exact:
            assert np.array_equal(bits(xa), bits(ref)), opts
        else:
            assert np.linalg.norm(xa - ref) <= 1e-13 * np.linalg.norm(ref)
        L.raw("PCApply")(pc, vb.h, vx.h)
        assert np.array_equal(bits(vx.array()), bits(xa))              # deterministic
        sf, ab = C.c_int(), C.c_int()
        L.PCILUGetSolver_HIPMI355X(pc, C.byref(sf), C.byref(ab))
        assert sf.value == 1 and ab.value == 0
        del k


@pytest.mark.parametrize("pct", ["ilu", "icc"])
def test_p7_full_size_factor_plans_built_on_the_device(P, pct, monkeypatch):
    """ILU(0) / ICC(0) on P7(256) (BASELINE configs[1]'s operator, 16.7 M rows; wide dependency levels: the one-wavefront-per-slice
    kernels): the sync-free plans laid out on the device (csrc/trisolve_build.hip, the default) and by the host threads
    (MI355X_TRISOLVE_BUILD=host) give the same application bit for bit, and that is the oracle's MatSolve restatement's."""
    L = P.lib()
    m = 256
    ai, aj, aa = P.gen_poisson7(m, m, m)
    n = ai.size - 1
    bvec = np.cos(0.1 * np.arange(n)) + 0.01 * np.sin(np.arange(n))
    A = P.Mat.from_csr(ai, aj, aa)
    vb, vx = V(P, bvec), V(P, np.zeros(n))
    got = {}
    for route in ("device", "host"):
        monkeypatch.setenv("MI355X_TRISOLVE_BUILD", route)
        pc = C.c_void_p()
        k = P.KSP(comm=L.COMM_SELF); k.set_operators(A); L.KSPGetPC(k.h, C.byref(pc)); L.PCSetType(pc, pct.encode())
        L.raw("PCSetUp")(pc)
        L.raw("PCApply")(pc, vb.h, vx.h)
        got[route] = vx.array().copy()
        L.raw("PCApply")(pc, vb.h, vx.h)
        assert np.array_equal(bits(vx.array()), bits(got[route]))      # deterministic, and the plans re-arm themselves
        if pct == "ilu":
            sf, ab = C.c_int(), C.c_int()
            L.PCILUGetSolver_HIPMI355X(pc, C.byref(sf), C.byref(ab))
            assert sf.value == 1 and ab.value == 0
        del k
    assert np.array_equal(bits(got["device"]), bits(got["host"]))
    ref = orc.ilu0_solve(orc.ilu0_factor(ai, aj, aa), bvec) if pct == "ilu" else orc.icc0_solve(orc.icc0_factor(ai, aj, aa)[0], bvec)
    assert np.array_equal(bits(got["device"]), bits(ref))


@pytest.mark.parametrize("sub", ["jacobi", "ilu"])
def test_config4_full_size_gmres_bjacobi_solve(P, sub):
    """BASELINE configs[3] END TO END at full size on the FEM-like stand-in for Flan_1565 (1.53 M rows, 1.18e8 nonzeros, 3 dof per
    node: tests/problems.py): KSPGMRES(30) + PCBJACOBI with one block per rank and (a) the all-device Jacobi sub-PC, (b) the
    reference's default sub-solver, preonly + ILU(0) (bjacobi.c:738-923 -> PCILU -> MatGetFactor -> the device triangular solves).
    Properties that do not need an oracle run of this size: the solve converges to rtol 1e-8 with KSP_CONVERGED_RTOL; the residual
    norm GMRES reports from its recurrence equals the recomputed ||B (b - A x)||; the error against the known solution is at the
    level the tolerance implies; and the plug-in's fused solver (-ksp_type gmreshipmi355x) walks the bits of the plain restatement
    of KSPSolve_GMRES (-ksp_type gmres) over the same operators -- every residual norm and every entry of x."""
    L = P.lib()
    ai, aj, aa = pb.gen_fem3()
    n = ai.size - 1
    A = P.Mat.from_csr(ai, aj, aa)
    u = P.Vec.create(n, comm=L.COMM_SELF); L.VecSet(u.h, 1.0)
    b = u.duplicate(); A.mult(u, b)
    runs = []
    for ktype in ("gmreshipmi355x", "gmres"):
        x = u.duplicate(); L.VecSet(x.h, 0.0)
        k = P.KSP(comm=L.COMM_SELF)
        k.set_operators(A)
        set_options(L, "-ksp_type %s -pc_type bjacobi -sub_pc_type %s" % (ktype, sub))
        k.set_tolerances(rtol=1e-8, abstol=1e-50, dtol=1e5, max_it=500)
        k.set_from_options(); k.record_history()
        set_options(L, "")
        k.solve(b, x)
        runs.append((k.its, k.reason, k.history().copy(), x, k))
    (its, reason, h, x, k), (its0, reason0, h0, x0, _) = runs
    assert reason == reason0 == 2 and its == its0 and 5 < its < 200
    assert h[-1] <= 1e-8 * h[0] and np.all(np.diff(h) <= 1e-6 * h[:-1])         # GMRES: the residual never grows (a restart re-measures it)
    assert np.array_equal(bits(h), bits(h0))
    xa = x.array()
    assert np.array_equal(bits(xa), bits(x0.array()))
    assert np.linalg.norm(xa - 1.0) / np.sqrt(n) < 1e-7
    # true preconditioned residual B (b - A x) against the recurrence's last value (left preconditioning, preconditioned norm)
    r = u.duplicate(); z = u.duplicate()
    A.mult(x, r); L.VecAYPX(r.h, -1.0, b.h)
    pc = C.c_void_p(); L.KSPGetPC(k.h, C.byref(pc))
    L.raw("PCApply")(pc, r.h, z.h)
    assert abs(z.norm() - h[-1]) <= 1e-3 * h[-1] + 1e-12 * h[0]


def _deferral(P, on):
    P.lib().raw("VecHIPMI355XSetDeferral")(on)


@pytest.mark.parametrize("norm", ["preconditioned", "natural", "unpreconditioned"])
@pytest.mark.parametrize("pc", ["jacobi", "none", "ilu"])
def test_unchanged_ksp_cg_runs_the_fused_sweep_with_the_same_bits(P, pc, norm):
    """The plain KSPSolve_CG (what an unchanged PETSc program drives: VecAXPY, VecAXPY, PCApply, VecNorm, VecTDot one after the other)
    over vectors that note element-wise operations instead of launching them (host/vechip.c): with PCJACOBI the five calls become ONE
    fused sweep per iteration -- counted -- and the VecTDot(Z,R) behind it is answered without a kernel; iterates, residual history,
    iteration count and reason are bit for bit those of the same solve with the noting switched off, whatever the PC or norm type."""
    L = P.lib()
    ai, aj, aa = pb.lap2d(37, 29)
    n = ai.size - 1
    b = orc.spmv(ai, aj, aa, np.cos(0.3 * np.arange(n)))
    opts = "-ksp_cg_fused 0 -ksp_norm_type " + norm
    runs = {}
    for on in (0, 1):
        _deferral(P, on)
        L.VecHIPMI355XSetCGUpdateTiming(1)
        runs[on] = solve(P, ai, aj, aa, b, "cg", pc, opts=opts, rtol=1e-9)
        nl = C.c_int(); L.VecHIPMI355XGetCGUpdateTiming(C.byref(nl), None)
        L.VecHIPMI355XSetCGUpdateTiming(0)
        runs[on] = runs[on] + (nl.value,)
    _deferral(P, -1)
    (x0, h0, its0, r0, n0), (x1, h1, its1, r1, n1) = runs[0], runs[1]
    assert r0 == r1 == 2 and its0 == its1 > 20
    assert np.array_equal(bits(h0), bits(h1)) and np.array_equal(bits(x0), bits(x1))
    assert n0 == 0                                              # the plain type by itself never calls the fused sweep
    if pc in ("jacobi", "none") and norm != "unpreconditioned":  # (PCNONE's VecCopy is the third operation then; unpreconditioned: VecNorm(R) sits between the AXPYs and the PCApply)
        assert its1 - 1 <= n1 <= its1 + 1
    else:
        assert n1 == 0


@pytest.mark.parametrize("norm", ["preconditioned", "none"])
@pytest.mark.parametrize("ksp", ["bcgs", "gmres"])
@pytest.mark.parametrize("pc", ["jacobi", "none", "ilu"])
def test_unchanged_bcgs_and_gmres_keep_their_bits_under_the_noted_operations(P, ksp, pc, norm):
    """KSPSolve_BCGS's calls as an unchanged program makes them: the Jacobi products are noted and run inside the VecDot / VecDotNorm2
    that follows, VecAXPBYPCZ(X, ...) + VecWAXPY(R, ...) run inside VecNorm(R) (or, without a norm, inside VecDot(R, RP)) together with
    (r, rp); GMRES only sees its lone products and updates noted and run at the next access.  Iterates and histories are bit for bit
    those of the same solve with the noting off."""
    ai, aj, aa = pb.lap2d(31, 23)
    aa = aa * (1.0 + 0.3 * np.sin(0.7 * np.arange(aa.size)))            # non-symmetric values
    n = ai.size - 1
    b = orc.spmv(ai, aj, aa, np.cos(0.3 * np.arange(n)))
    opts = "-ksp_bcgs_fused 0 -ksp_gmres_fused 0 -ksp_norm_type " + norm + (" -ksp_max_it 40" if norm == "none" else "")
    runs = {}
    for on in (0, 1):
        _deferral(P, on)
        runs[on] = solve(P, ai, aj, aa, b, ksp, pc, opts=opts, rtol=1e-9)
    _deferral(P, -1)
    (x0, h0, its0, r0), (x1, h1, its1, r1) = runs[0], runs[1]
    assert r0 == r1 and its0 == its1 > 5
    assert np.array_equal(bits(h0), bits(h1)) and np.array_equal(bits(x0), bits(x1))


def test_deferred_vector_operations_are_transparent(P):
    """Noted-but-not-launched operations never show: every scenario (the whole CG pattern, prefixes of it read early, operand
    aliasing that must not extend the note, another norm type, swapped dot operands, a change of r between the sweep and the dot it
    would answer, a vector destroyed while named by the note, host access) gives the bits of the same calls with the noting off."""
    L = P.lib()
    n = 5000
    rng = np.random.default_rng(77)
    base = {k: rng.standard_normal(n) for k in "xprwzd"}
    a = 0.37

    def run(scn):
        v = {k: P.Vec.from_array(base[k], comm=L.COMM_SELF) for k in "xprwzd"}
        out = []
        val = C.c_double()

        def nrm(vec, t=2):
            r = (C.c_double * 2)(); L.VecNorm(vec.h, t, r); return r[0]

        def dot(u, w_):
            L.VecTDot(u.h, w_.h, C.byref(val)); return val.value
        if scn == "full":
            L.VecAXPY(v["x"].h, a, v["p"].h); L.VecAXPY(v["r"].h, -a, v["w"].h); L.VecPointwiseMult(v["z"].h, v["r"].h, v["d"].h)
            out += [nrm(v["z"]), dot(v["z"], v["r"]), dot(v["z"], v["r"])]
        elif scn == "full_swapped_operands":
            L.VecAXPY(v["x"].h, a, v["p"].h); L.VecAXPY(v["r"].h, -a, v["w"].h); L.VecPointwiseMult(v["z"].h, v["d"].h, v["r"].h)
            out += [dot(v["r"], v["z"]), nrm(v["z"])]
        elif scn == "bcgs_update_norm_first":                      # VecDot(R, RP) of the previous iteration names rp, then the update, norm, dot
            out += [dot(v["r"], v["d"])]
            L.VecAXPBYPCZ(v["x"].h, a, 0.7, 1.0, v["p"].h, v["z"].h); L.VecWAXPY(v["r"].h, -0.7, v["w"].h, v["z"].h)
            out += [nrm(v["r"]), dot(v["r"], v["d"])]
        elif scn == "bcgs_update_dot_first":
            L.VecAXPBYPCZ(v["x"].h, a, 0.7, 1.0, v["p"].h, v["z"].h); L.VecWAXPY(v["r"].h, -0.7, v["w"].h, v["z"].h)
            out += [dot(v["d"], v["r"]), nrm(v["r"])]
        elif scn == "bcgs_update_unknown_partner":
            L.VecAXPBYPCZ(v["x"].h, a, 0.7, 1.0, v["p"].h, v["z"].h); L.VecWAXPY(v["r"].h, -0.7, v["w"].h, v["z"].h)
            out += [nrm(v["r"])]
        elif scn == "bcgs_update_broken":                          # another omega, another s: not the pattern
            L.VecAXPBYPCZ(v["x"].h, a, 0.7, 1.0, v["p"].h, v["z"].h); L.VecWAXPY(v["r"].h, -0.6, v["w"].h, v["z"].h); out += [nrm(v["r"])]
            L.VecAXPBYPCZ(v["x"].h, a, 0.7, 1.0, v["p"].h, v["z"].h); L.VecWAXPY(v["r"].h, -0.7, v["w"].h, v["d"].h); out += [nrm(v["r"])]
            L.VecAXPBYPCZ(v["x"].h, a, 0.7, 0.5, v["p"].h, v["z"].h); L.VecWAXPY(v["r"].h, -0.7, v["w"].h, v["z"].h); out += [nrm(v["r"])]
        elif scn == "product_then_dot":
            L.VecPointwiseMult(v["z"].h, v["w"].h, v["d"].h); out += [dot(v["z"], v["r"])]
            L.VecPointwiseMult(v["z"].h, v["d"].h, v["p"].h); out += [dot(v["x"], v["z"])]
        elif scn == "product_then_dotnorm2":
            L.VecPointwiseMult(v["z"].h, v["w"].h, v["d"].h)
            dp, nm2 = C.c_double(), C.c_double(); L.raw("VecDotNorm2")(v["r"].h, v["z"].h, C.byref(dp), C.byref(nm2)); out += [dp.value, nm2.value]
        elif scn == "product_then_other":
            L.VecPointwiseMult(v["z"].h, v["w"].h, v["d"].h); out += [dot(v["x"], v["r"]), nrm(v["z"])]
            L.VecPointwiseMult(v["z"].h, v["z"].h, v["d"].h); out += [dot(v["z"], v["r"])]
        elif scn in ("maxpy_norm", "maxpy_then_dot", "maxpy_norm1"):  # GMRES: VecMAXPY(w, -h, V) then VecNormalize(w)
            vs = [v[k] for k in "prwd"]
            al = np.array([0.3, -1.7, 0.01, 2.5])
            tabv = (C.c_void_p * 4)(*[u.h.value for u in vs])
            L.VecMAXPY(v["z"].h, 4, al.ctypes.data_as(C.c_void_p), tabv)
            out += [nrm(v["z"])] if scn == "maxpy_norm" else ([dot(v["z"], v["x"])] if scn == "maxpy_then_dot" else [nrm(v["z"], 0)])
            L.VecScale(v["z"].h, 1.0 / 3.0)
        elif scn == "copy_as_third":                               # PCApply_None
            L.VecAXPY(v["x"].h, a, v["p"].h); L.VecAXPY(v["r"].h, -a, v["w"].h); L.VecCopy(v["r"].h, v["z"].h); out += [nrm(v["z"]), dot(v["z"], v["r"])]
        elif scn == "copy_elsewhere":
            L.VecAXPY(v["x"].h, a, v["p"].h); L.VecAXPY(v["r"].h, -a, v["w"].h); L.VecCopy(v["x"].h, v["z"].h); out += [nrm(v["z"])]
        elif scn == "one_then_read":
            L.VecAXPY(v["x"].h, a, v["p"].h); out += [nrm(v["x"])]
        elif scn == "two_then_read":
            L.VecAXPY(v["x"].h, a, v["p"].h); L.VecAXPY(v["r"].h, -a, v["w"].h); out += [dot(v["r"], v["x"])]
        elif scn == "second_reads_first":
            L.VecAXPY(v["x"].h, a, v["p"].h); L.VecAXPY(v["r"].h, -a, v["x"].h)
        elif scn == "second_writes_first":
            L.VecAXPY(v["x"].h, a, v["p"].h); L.VecAXPY(v["x"].h, -a, v["w"].h)
        elif scn == "not_negated":
            L.VecAXPY(v["x"].h, a, v["p"].h); L.VecAXPY(v["r"].h, a, v["w"].h); L.VecPointwiseMult(v["z"].h, v["r"].h, v["d"].h); out += [nrm(v["z"])]
        elif scn == "product_overwrites_p":
            L.VecAXPY(v["x"].h, a, v["p"].h); L.VecAXPY(v["r"].h, -a, v["w"].h); L.VecPointwiseMult(v["p"].h, v["r"].h, v["d"].h); out += [nrm(v["p"])]
        elif scn == "product_into_w":                              # cg.c:122 keeps A p in Z: z may be w
            L.VecAXPY(v["x"].h, a, v["p"].h); L.VecAXPY(v["r"].h, -a, v["w"].h); L.VecPointwiseMult(v["w"].h, v["r"].h, v["d"].h); out += [nrm(v["w"]), dot(v["w"], v["r"])]
        elif scn == "norm_1":
            L.VecAXPY(v["x"].h, a, v["p"].h); L.VecAXPY(v["r"].h, -a, v["w"].h); L.VecPointwiseMult(v["z"].h, v["r"].h, v["d"].h); out += [nrm(v["z"], 0), nrm(v["z"], 3)]
        elif scn == "r_changes_before_the_dot":
            L.VecAXPY(v["x"].h, a, v["p"].h); L.VecAXPY(v["r"].h, -a, v["w"].h); L.VecPointwiseMult(v["z"].h, v["r"].h, v["d"].h)
            out += [nrm(v["z"])]; L.VecScale(v["r"].h, 1.5); out += [dot(v["z"], v["r"])]
        elif scn == "destroy_while_noted":
            L.VecAXPY(v["x"].h, a, v["p"].h); L.VecAXPY(v["r"].h, -a, v["w"].h); L.VecPointwiseMult(v["z"].h, v["r"].h, v["d"].h)
            v["d"].destroy(); v["d"].h = C.c_void_p(); v["p"].destroy(); v["p"].h = C.c_void_p()
        elif scn == "host_access":
            L.VecAXPY(v["x"].h, a, v["p"].h); L.VecAXPY(v["r"].h, -a, v["w"].h)
            v["w"].set_array(base["z"]); L.VecAXPY(v["r"].h, 2.0, v["w"].h)
        elif scn == "zero_alpha":
            L.VecAXPY(v["x"].h, 0.0, v["p"].h); L.VecAXPY(v["r"].h, -0.0, v["w"].h); L.VecAXPY(v["x"].h, a, v["p"].h)
        arrays = [v[k].array() for k in "xprwzd" if v[k].h]
        return out, arrays

    scenarios = ["full", "full_swapped_operands", "bcgs_update_norm_first", "bcgs_update_dot_first", "bcgs_update_unknown_partner", "bcgs_update_broken",
                 "product_then_dot", "product_then_dotnorm2", "product_then_other", "maxpy_norm", "maxpy_then_dot", "maxpy_norm1", "copy_as_third", "copy_elsewhere", "one_then_read", "two_then_read", "second_reads_first", "second_writes_first", "not_negated",
                 "product_overwrites_p", "product_into_w", "norm_1", "r_changes_before_the_dot", "destroy_while_noted", "host_access", "zero_alpha"]
    for scn in scenarios:
        _deferral(P, 0); o0, a0 = run(scn)
        _deferral(P, 1); o1, a1 = run(scn)
        assert np.array_equal(bits(np.array(o0)), bits(np.array(o1))), scn
        assert len(a0) == len(a1) and all(np.array_equal(bits(u), bits(w_)) for u, w_ in zip(a0, a1)), scn
    _deferral(P, -1)
    # and the reference values of the whole pattern
    _deferral(P, 1); o1, a1 = run("full"); _deferral(P, -1)
    x = base["x"].copy(); orc.vec_axpy(x, a, base["p"])
    r = base["r"].copy(); orc.vec_axpy(r, -a, base["w"])
    z = np.zeros(n); orc.vec_pointwise_mult(z, r, base["d"])
    assert np.array_equal(bits(a1[0]), bits(x)) and np.array_equal(bits(a1[2]), bits(r)) and np.array_equal(bits(a1[4]), bits(z))
    assert abs(o1[0] - np.linalg.norm(z)) <= 1e-13 * np.linalg.norm(z) and o1[1] == o1[2] and abs(o1[1] - z @ r) <= 1e-13 * np.sum(np.abs(z * r))


def test_noted_products_are_transparent(P):
    """MatMult of a sequential AIJ matrix is noted, not launched; a Jacobi product right behind it makes ONE kernel of the two and
    leaves the work vector unwritten until somebody reads it.  Every way the sequence can go on -- the work vector read later, the
    source vector or the matrix changed first, the work vector overwritten or combined into, vectors destroyed, a product that is
    not followed by the pattern -- gives the bits of the same calls with the noting off."""
    L = P.lib()
    ai, aj, aa = pb.lap2d(23, 19)
    aa = aa * (1.0 + 0.2 * np.sin(np.arange(aa.size)))
    n = ai.size - 1
    rng = np.random.default_rng(5)
    base = {k: rng.standard_normal(n) for k in "xtwdzy"}

    def run(scn):
        A = P.Mat.from_csr(ai, aj, aa)
        v = {k: P.Vec.from_array(base[k], comm=L.COMM_SELF) for k in "xtwdzy"}
        out = []
        val = C.c_double()

        def dot(u, w_):
            L.VecTDot(u.h, w_.h, C.byref(val)); return val.value
        L.MatMult(A.h, v["x"].h, v["t"].h)
        if scn == "read_t_later":
            L.VecPointwiseMult(v["w"].h, v["t"].h, v["d"].h); out += [dot(v["w"], v["y"])]
        elif scn == "operands_swapped":
            L.VecPointwiseMult(v["w"].h, v["d"].h, v["t"].h)
        elif scn == "x_changes_first":
            L.VecPointwiseMult(v["w"].h, v["t"].h, v["d"].h); L.VecScale(v["x"].h, 1.7)
        elif scn == "x_set_first":
            L.VecPointwiseMult(v["w"].h, v["t"].h, v["d"].h); L.VecSet(v["x"].h, 3.0)
        elif scn == "x_host_access":
            L.VecPointwiseMult(v["w"].h, v["t"].h, v["d"].h); v["x"].set_array(base["z"])
        elif scn == "t_overwritten_by_the_next_product":
            L.VecPointwiseMult(v["w"].h, v["t"].h, v["d"].h); L.MatMult(A.h, v["z"].h, v["t"].h); L.VecPointwiseMult(v["y"].h, v["t"].h, v["d"].h)
        elif scn == "t_overwritten_by_a_copy":
            L.VecPointwiseMult(v["w"].h, v["t"].h, v["d"].h); L.VecCopy(v["z"].h, v["t"].h)
        elif scn == "t_combined_into":
            L.VecPointwiseMult(v["w"].h, v["t"].h, v["d"].h); L.VecAXPY(v["t"].h, 0.3, v["z"].h); L.VecAXPY(v["y"].h, 0.5, v["t"].h)
        elif scn == "matrix_scaled_first":
            L.VecPointwiseMult(v["w"].h, v["t"].h, v["d"].h); L.MatScale(A.h, 2.5); L.MatMult(A.h, v["x"].h, v["z"].h)
        elif scn == "matrix_scaled_before_the_product_runs":
            L.MatScale(A.h, 2.5)
        elif scn == "matrix_diagonal_scaled_first":
            L.VecPointwiseMult(v["w"].h, v["t"].h, v["d"].h); L.MatDiagonalScale(A.h, v["d"].h, v["y"].h)
        elif scn == "x_destroyed":
            L.VecPointwiseMult(v["w"].h, v["t"].h, v["d"].h); v["x"].destroy(); v["x"].h = C.c_void_p()
        elif scn == "t_destroyed":
            L.VecPointwiseMult(v["w"].h, v["t"].h, v["d"].h); v["t"].destroy(); v["t"].h = C.c_void_p()
        elif scn == "matrix_destroyed":
            L.VecPointwiseMult(v["w"].h, v["t"].h, v["d"].h); A.destroy(); A.h = C.c_void_p()
        elif scn == "no_pattern_dot":
            out += [dot(v["t"], v["y"])]
        elif scn == "product_into_the_source":
            L.VecPointwiseMult(v["x"].h, v["t"].h, v["d"].h)
        elif scn == "product_by_itself":
            L.VecPointwiseMult(v["w"].h, v["t"].h, v["t"].h)
        elif scn == "another_product_not_t":
            L.VecPointwiseMult(v["w"].h, v["z"].h, v["d"].h)
        elif scn == "transpose_next":
            L.VecPointwiseMult(v["w"].h, v["t"].h, v["d"].h); L.MatMultTranspose(A.h, v["t"].h, v["z"].h)
        arrays = [v[k].array() for k in "xtwdzy" if v[k].h]
        return out, arrays

    scenarios = ["read_t_later", "operands_swapped", "x_changes_first", "x_set_first", "x_host_access", "t_overwritten_by_the_next_product", "t_overwritten_by_a_copy",
                 "t_combined_into", "matrix_scaled_first", "matrix_scaled_before_the_product_runs", "matrix_diagonal_scaled_first", "x_destroyed", "t_destroyed",
                 "matrix_destroyed", "no_pattern_dot", "product_into_the_source", "product_by_itself", "another_product_not_t", "transpose_next"]
    for scn in scenarios:
        _deferral(P, 0); o0, a0 = run(scn)
        _deferral(P, 1); o1, a1 = run(scn)
        assert np.array_equal(bits(np.array(o0)), bits(np.array(o1))), scn
        assert len(a0) == len(a1) and all(np.array_equal(bits(u), bits(w_)) for u, w_ in zip(a0, a1)), scn
    _deferral(P, -1)
    # against the oracle: w = d .* (A x), t = A x
    _deferral(P, 1); o1, a1 = run("read_t_later"); _deferral(P, -1)
    t_ref = orc.spmv(ai, aj, aa, base["x"])
    w_ref = np.zeros(n); orc.vec_pointwise_mult(w_ref, t_ref, base["d"])
    assert np.array_equal(bits(a1[1]), bits(t_ref)) and np.array_equal(bits(a1[2]), bits(w_ref))


def test_noting_follows_the_options_database(P):
    """-vec_hipmi355x_defer <0|1> (read when the next operation asks; default 1) switches the noted operations: the plain CG calls run
    the fused sweep once per iteration with it, never without it; same bits."""
    L = P.lib()
    ai, aj, aa = pb.lap2d(19, 17)
    b = orc.spmv(ai, aj, aa, np.cos(0.3 * np.arange(ai.size - 1)))
    res = {}
    for opt in ("-vec_hipmi355x_defer 0", "-vec_hipmi355x_defer 1", ""):
        L.VecHIPMI355XSetCGUpdateTiming(1)
        L.PetscOptionsClear(); L.PetscOptionsInsertString(opt.encode())
        _deferral(P, -1)                                   # "as the options database says": asked again at the next operation, inside solve()
        x, h, its, r = solve(P, ai, aj, aa, b, "cg", "jacobi", opts="-ksp_cg_fused 0 " + opt, rtol=1e-9)
        nl = C.c_int(); L.VecHIPMI355XGetCGUpdateTiming(C.byref(nl), None)
        L.VecHIPMI355XSetCGUpdateTiming(0)
        res[opt] = (bits(x).copy(), bits(h).copy(), its, nl.value)
    _deferral(P, -1)
    assert res["-vec_hipmi355x_defer 0"][3] == 0 and res["-vec_hipmi355x_defer 1"][3] >= res["-vec_hipmi355x_defer 1"][2] - 1 and res[""][3] == res["-vec_hipmi355x_defer 1"][3]
    for k in ("-vec_hipmi355x_defer 1", ""):
        assert np.array_equal(res[k][0], res["-vec_hipmi355x_defer 0"][0]) and np.array_equal(res[k][1], res["-vec_hipmi355x_defer 0"][1])
