"""Pins the oracle (oracle/*.c) to the reference's OWN golden outputs (tests/golden/, copied data files of
src/mat/examples/tests/output and src/ksp/ksp/examples/{tests,tutorials}/output) and to the reference run
recorded in SURVEY.md 8(c)/(d).  Residual norms are compared as the 6-significant-digit text
-ksp_monitor_short prints."""
import ctypes as C
import os

import numpy as np
import pytest

import orc
import problems as pb

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fmt(v):
    return np.array([float("%g" % x) for x in v])


@pytest.mark.parametrize("name,rect", [("ex5_11_A.out", 2), ("ex5_11_B.out", -2)])
def test_ex5_seqaij_rect(name, rect):
    """ex5 -mat_type seqaij -rectA / -rectB (makefile:754,764) vs output/ex5_11_A.out, ex5_11_B.out: MatMult,
    MatMultTranspose, MatGetDiagonal on 8 x 10 and 8 x 6 matrices"""
    ai, aj, aa, m, n = pb.ex5_mat(8, rect=rect)
    gold = pb.parse_vecview(os.path.join(G, name))
    y = np.arange(n, dtype=np.float64)
    assert np.array_equal(fmt(orc.spmv(ai, aj, aa, y)), gold[0])
    x = np.arange(m, dtype=np.float64)
    assert np.array_equal(fmt(orc.spmv_t(ai, aj, aa, x, n)), gold[1])
    assert np.array_equal(fmt(orc.get_diagonal(ai, aj, aa)), gold[-1])
    # the example's self-checks: MatMultAdd == MatMult + z, MatMultTransposeAdd == MatMultTranspose + u (norm <= 1e-8)
    z = 100.0 * (np.arange(m) + 1)
    assert np.linalg.norm(orc.spmv(ai, aj, aa, y) + z - orc.spmv_add(ai, aj, aa, y, z)) <= 1e-8
    u = 100.0 * np.arange(n)
    assert np.linalg.norm(orc.spmv_t(ai, aj, aa, x, n) + u - orc.spmv_t_add(ai, aj, aa, x, u, n)) <= 1e-8


def test_ex5_mpiaij():
    """ex5 -mat_type mpiaij on 1 and 3 ranks (makefile:775, runex5_2) vs ex5_21.out / ex5_23.out: the distributed
    product assembled from diagonal + off-diagonal blocks through garray and the scatter lists"""
    ai, aj, aa, m, n = pb.ex5_mat(8)
    y = np.arange(n, dtype=np.float64)
    x = np.arange(m, dtype=np.float64)
    for size, name in ((1, "ex5_21.out"), (3, "ex5_23.out")):
        gold = pb.parse_vecview(os.path.join(G, name))
        ranges = np.array([0] + list(np.cumsum([m // size + (m % size > r) for r in range(size)])), dtype=np.int32)
        pieces = [orc.mpiaij_split(int(ranges[r]), int(ranges[r + 1]), int(ranges[r]), int(ranges[r + 1]), ai, aj, aa) for r in range(size)]
        garrays = [p["garray"] for p in pieces]
        out = np.zeros(m); outT = np.zeros(n)
        for r, p in enumerate(pieces):
            sc = orc.scatter_create(size, r, ranges, garrays)
            lvec = np.zeros(p["garray"].size)
            # forward scatter: what each owner sends lands in the listed lvec slots (MatMult_MPIAIJ, mpiaij.c:1111-1114)
            for q, proc in enumerate(sc["rprocs"]):
                slots = sc["rindices"][sc["rstarts"][q]:sc["rstarts"][q + 1]]
                owner = orc.scatter_create(size, int(proc), ranges, garrays)
                k = list(owner["sprocs"]).index(r)
                src = owner["sindices"][owner["sstarts"][k]:owner["sstarts"][k + 1]] + ranges[proc]
                lvec[slots] = y[src]
            yl = y[ranges[r]:ranges[r + 1]]
            d = orc.spmv(p["ad_i"], p["ad_j"], p["ad_a"], yl)
            out[ranges[r]:ranges[r + 1]] = orc.spmv_add(p["bo_i"], p["bo_j"], p["bo_a"], lvec, d) if lvec.size else d
            # transpose: local part + reverse scatter ADD of B^T x (MatMultTranspose_MPIAIJ, mpiaij.c:1147-1174)
            xl = x[ranges[r]:ranges[r + 1]]
            outT[ranges[r]:ranges[r + 1]] += orc.spmv_t(p["ad_i"], p["ad_j"], p["ad_a"], xl, int(ranges[r + 1] - ranges[r]))
            if lvec.size:
                outT[p["garray"]] += orc.spmv_t(p["bo_i"], p["bo_j"], p["bo_a"], xl, lvec.size)
        assert np.array_equal(fmt(out), gold[0])
        assert np.array_equal(fmt(outT), gold[1])
        assert np.array_equal(fmt(orc.get_diagonal(ai, aj, aa)), gold[-1])
    # ex5_33.out (makefile:806: -n 3 -mat_type mpiaij -test_diagonalscale): the same products, then MatGetDiagonal and
    # MatDiagonalScale(C, x = the diagonal, y = 1..n) through the pieces -- MatDiagonalScale_MPIAIJ (mpiaij.c:2012-2046) scales the
    # diagonal block by (l, r local) and the off-diagonal block by (l, r scattered to the ghost columns)
    import re
    text = open(os.path.join(G, "ex5_33.out")).read()
    gold = pb.parse_vecview(os.path.join(G, "ex5_33.out"))
    views = [np.array([[float(v) for _, v in re.findall(r"\((\d+), ([-0-9.e+]+)\)", line)] for line in blk.splitlines() if line.startswith("row ")])
             for blk in text.split("Matrix Object:")[1:]]
    assert len(views) == 2 and np.array_equal(fmt(out), gold[0]) and np.array_equal(fmt(outT), gold[1])
    diag = orc.get_diagonal(ai, aj, aa)
    assert np.array_equal(fmt(diag), gold[2])
    rvec = np.arange(1, n + 1, dtype=np.float64)
    before, after = np.zeros((m, n)), np.zeros((m, n))
    for r, p in enumerate(pieces):
        lo, hi = int(ranges[r]), int(ranges[r + 1])
        ad = orc.diagonal_scale(p["ad_i"], p["ad_j"], p["ad_a"], diag[lo:hi].copy(), rvec[lo:hi].copy())
        bo = orc.diagonal_scale(p["bo_i"], p["bo_j"], p["bo_a"], diag[lo:hi].copy(), rvec[p["garray"]].copy())
        for i in range(hi - lo):
            sd, so = slice(p["ad_i"][i], p["ad_i"][i + 1]), slice(p["bo_i"][i], p["bo_i"][i + 1])
            before[lo + i, lo + p["ad_j"][sd]] = p["ad_a"][sd]; before[lo + i, p["garray"][p["bo_j"][so]]] = p["bo_a"][so]
            after[lo + i, lo + p["ad_j"][sd]] = ad[sd]; after[lo + i, p["garray"][p["bo_j"][so]]] = bo[so]
    assert np.array_equal(fmt(before.ravel()).reshape(m, n), views[0]) and np.array_equal(fmt(after.ravel()).reshape(m, n), views[1])


def test_pc_tests_ex2_cg_golden():
    """THE pin of the CG restatement: src/ksp/pc/examples/tests/ex2.c -ksp_type cg -ksp_monitor_short (makefile:53)
    vs output/ex2_1.out -- KSPSolve_CG (cg.c:92-286), PCNONE, tridiagonal n = 10, b = A*1, five iterations,
    default rtol 1e-5, preconditioned norm"""
    ai, aj, aa = pb.tridiag(10)
    b = orc.spmv(ai, aj, aa, np.ones(10))
    gold = pb.parse_monitor(os.path.join(G, "pc_tests", "ex2_1.out"))[0]
    x, h, its, reason = orc.ksp_solve(ai, aj, aa, b, ksp="cg", pc="none")
    pb.check_monitor(h, gold)
    assert its == 5 and reason == 2 and np.linalg.norm(x - 1.0) <= 1e-14   # the example prints the error only above 1e-14


@pytest.mark.parametrize("name,rtol", [("ex1_1.out", 1e-5), ("ex23_1.out", 1e-7), ("ex23_2.out", 1e-7)])
def test_tutorial_ex1_ex23_gmres_jacobi_happy_breakdown(name, rtol):
    """src/ksp/ksp/examples/tutorials/ex1.c and ex23.c (1 and 3 ranks), -ksp_monitor_short refine_always (makefile:295,862,868)
    vs output/ex1_1.out, ex23_1.out, ex23_2.out: tridiagonal n = 10, GMRES(30) + PCJACOBI, b = A*1.  The Krylov space is
    exhausted after n/2 = 5 steps: the last line is '< 1.e-11' and the solve ends through GMRES's happy-breakdown branch
    (gmres.c:171-178,200-203).  The examples print the error only when it exceeds 1e-11: it must not."""
    ai, aj, aa = pb.tridiag(10)
    b = orc.spmv(ai, aj, aa, np.ones(10))
    gold = pb.parse_monitor(os.path.join(G, "ksp_tutorials", name))[0]
    x, h, its, reason = orc.ksp_solve(ai, aj, aa, b, ksp="gmres", pc="jacobi", refine_always=1, rtol=rtol)
    pb.check_monitor(h, gold)
    assert its == 5 and reason > 0 and np.linalg.norm(x - 1.0) <= 1e-11


def test_ksp_tests_ex4_golden():
    """src/ksp/ksp/examples/tests/ex4.c -m 5 -pc_type jacobi refine_always (makefile:197) vs output/ex4_1.out: the
    Q1 Laplacian of ex3.c assembled sequentially (MatZeroRows boundary rows, non-zero initial guess)"""
    (ai, aj, aa), b, u0, ustar = pb.ex3_fem(5)
    gold = pb.parse_monitor(os.path.join(G, "ksp_tests", "ex4_1.out"))[0]
    x, h, its, reason = orc.ksp_solve(ai, aj, aa, b, ksp="gmres", pc="jacobi", x0=u0, refine_always=1)
    pb.check_monitor(h, gold)
    assert np.linalg.norm(x - ustar) * 0.2 <= 1e-14


@pytest.mark.parametrize("size,name,rtol", [(1, "ex5_1.out", 1e-5), (2, "ex5_2.out", 1e-6)])
def test_tutorial_ex5_two_systems_golden(size, name, rtol):
    """src/ksp/ksp/examples/tutorials/ex5.c -pc_type jacobi refine_always on 1 and 2 ranks (makefile:411,416) vs
    ex5_1.out, ex5_2.out: two solves with one KSP, the second after MatZeroEntries + re-assembly into the same
    pattern (GMRES + Jacobi does not depend on the row distribution beyond the rounding of its dot products)"""
    solves = pb.parse_monitor(os.path.join(G, "ksp_tutorials", name))
    for second in (False, True):
        (ai, aj, aa), u = pb.ex5_tutorial(size, second)
        b = orc.spmv(ai, aj, aa, u)
        x, h, its, reason = orc.ksp_solve(ai, aj, aa, b, ksp="gmres", pc="jacobi", refine_always=1, rtol=rtol)
        pb.check_monitor(h, solves[1 if second else 0])
        assert np.linalg.norm(x - u) < 1e-4 * np.linalg.norm(u)


def test_tutorial_ex5_5_and_ex2_5_default_pc_on_two_ranks_golden():
    """tutorials/makefile:340 (runex2_5: -n 2 ./ex2 -m 5 -n 5 refine_always, default PC) and :423 (runex5_5: -n 2 ./ex5 refine_always,
    default PC, default tolerances) vs output/ex2_5.out, ex5_5.out: the reference's DEFAULT preconditioner on two ranks -- block
    Jacobi with ILU(0) in each rank's block -- on two more systems (ex5: two solves, the second after re-assembly with other
    values).  ex5_5.out prints the error norm and the iteration count of each solve; ex2_5.out is ex2_2.out's run with a viewer
    option (the makefile diffs it against ex2_2.out)."""
    assert open(os.path.join(G, "ksp_tutorials", "ex2_5.out")).read() == open(os.path.join(G, "ksp_tutorials", "ex2_2.out")).read()
    want = [l.split() for l in open(os.path.join(G, "ksp_tutorials", "ex5_5.out")).read().splitlines() if l.startswith("Norm of error")]
    want = [(w[3].rstrip(","), int(w[5])) for w in want]
    assert want == [("0.00121238", 7), ("0.000322889", 6)]
    for second in (False, True):
        (ai, aj, aa), u = pb.ex5_tutorial(2, second)
        n = ai.size - 1
        b = orc.spmv(ai, aj, aa, u)
        x, h, its, reason = orc.ksp_solve(ai, aj, aa, b, ksp="gmres", pc="bjacobi", blocks=[0, n // 2, n], sub_ksp="preonly", sub_pc="ilu", refine_always=1)
        assert ("%g" % np.linalg.norm(x - u), its) == want[1 if second else 0]


def test_config1_cg_jacobi_regression_record():
    """NOT a pin (the numbers come from SURVEY.md 8(c)'s probe build, which used a hand-written petscconf.h): a
    regression record of BASELINE configs[0], ex2 -m 100 -n 100 -ksp_type cg -pc_type jacobi -> 160 iterations,
    'Norm of error 5.70785e-05'.  CG itself is pinned by test_pc_tests_ex2_cg_golden above."""
    ai, aj, aa = pb.lap2d(100, 100)
    u = np.ones(10000)
    b = orc.spmv(ai, aj, aa, u)
    x, h, its, reason = orc.ksp_solve(ai, aj, aa, b, ksp="cg", pc="jacobi", rtol=1e-2 / (101 * 101), abstol=1e-50)
    assert its == 160 and reason == 2
    assert ["%g" % v for v in h[:4]] == ["5.04975", "2.54845", "1.81892", "1.67695"] and "%g" % h[-1] == "4.47805e-06"
    assert "%g" % np.linalg.norm(x - u) == "5.70785e-05"


def test_ksp_tests_ex3_gmres_jacobi_nonzero_guess():
    """src/ksp/ksp/examples/tests/ex3.c -pc_type jacobi -m 5 -ksp_gmres_cgs_refinement_type refine_always
    (makefile:187,192) vs output/ex3_1.out, ex3_2.out: GMRES(30), Jacobi, non-zero initial guess"""
    (ai, aj, aa), b, u0, ustar = pb.ex3_fem(5)
    for name in ("ex3_1.out", "ex3_2.out"):
        gold = pb.parse_monitor(os.path.join(G, "ksp_tests", name))[0]
        x, h, its, reason = orc.ksp_solve(ai, aj, aa, b, ksp="gmres", pc="jacobi", x0=u0, refine_always=1)
        pb.check_monitor(h, gold)
        assert np.linalg.norm(x - ustar) <= 1e-12   # the example prints nothing when the error is below 1e-14/h


def test_tutorial_ex2f_gmres_jacobi():
    """src/ksp/ksp/examples/tutorials/ex2f.F (m = n = 3) -pc_type jacobi refine_always (makefile:406) vs ex2f_1.out"""
    ai, aj, aa = pb.lap2d(3, 3)
    b = orc.spmv(ai, aj, aa, np.ones(9))
    gold = pb.parse_monitor(os.path.join(G, "ksp_tutorials", "ex2f_1.out"))[0]
    x, h, its, reason = orc.ksp_solve(ai, aj, aa, b, ksp="gmres", pc="jacobi", refine_always=1)
    pb.check_monitor(h, gold)
    assert its == 3 and np.linalg.norm(x - 1.0) < 1e-12


def test_tutorial_ex2_block_jacobi():
    """ex2 (m=8, n=7) on 4 ranks, -pc_type bjacobi -sub_pc_type jacobi -sub_ksp_type gmres (makefile:350,360):
    ex2_bjacobi.out (one block spanning the ranks), ex2_bjacobi_2.out (two blocks of two ranks each, makefile:355) and
    ex2_bjacobi_3.out (one 14-row block per rank)"""
    ai, aj, aa = pb.lap2d(8, 7)
    u = np.ones(56)
    b = orc.spmv(ai, aj, aa, u)
    rtol = 1e-2 / (9 * 8)
    for name, blocks, err, nits in (("ex2_bjacobi.out", [0, 56], "2.10144e-06", 1), ("ex2_bjacobi_2.out", [0, 28, 56], "0.000496964", 4),
                                    ("ex2_bjacobi_3.out", [0, 14, 28, 42, 56], "0.000404746", 7)):
        gold = pb.parse_monitor(os.path.join(G, "ksp_tutorials", name))[0]
        x, h, its, reason = orc.ksp_solve(ai, aj, aa, b, ksp="gmres", pc="bjacobi", blocks=blocks, sub_ksp="gmres", sub_pc="jacobi",
                                          rtol=rtol, abstol=1e-50)
        pb.check_monitor(h, gold)
        assert its == nits and (err is None or "%g" % np.linalg.norm(x - u) == err)


def test_tutorial_ex9_gmres_and_bcgs_jacobi():
    """src/ksp/ksp/examples/tutorials/ex9.c -t 2 -pc_type jacobi -ksp_type gmres refine_always -s2_ksp_type bcgs
    -s2_pc_type jacobi (makefile:438-439) vs ex9_1.out: system 1 at t=0 (GMRES) and system 2 at t=0,1 (BiCGStab)"""
    solves = pb.parse_monitor(os.path.join(G, "ksp_tutorials", "ex9_1.out"))
    (ai, aj, aa), u = pb.ex9_system(1, 0)
    b = orc.spmv(ai, aj, aa, u)
    x, h, its, reason = orc.ksp_solve(ai, aj, aa, b, ksp="gmres", pc="jacobi", refine_always=1)
    pb.check_monitor(h, solves[0])
    for t, gold in ((0, solves[1]), (1, solves[3])):
        (ai, aj, aa), u = pb.ex9_system(2, t)
        b = orc.spmv(ai, aj, aa, u)
        x, h, its, reason = orc.ksp_solve(ai, aj, aa, b, ksp="bcgs", pc="jacobi")
        pb.check_monitor(h, gold)
        assert np.linalg.norm(x - u) < 1e-4   # CheckError tolerance of the example


def test_inode_detection_and_variant():
    """a 3-dof operator stored as AIJ makes the reference switch to MatMult_SeqAIJ_Inode
    ('found m/3 nodes, limit used is 5'); the scalar 7-point operator does not ('not using I-node routines')."""
    (ai, aj, aa), _ = pb.elasticity_like(4, 3, 3)
    m = ai.size - 1
    nc, ns = orc.check_inode(ai, aj)
    assert nc == m // 3 and np.all(ns == 3)
    pi_, pj, pa = orc.gen_p7(8, 8, 8)
    assert orc.check_inode(pi_, pj)[0] == 0
    x = np.random.default_rng(1).standard_normal(m)
    y0, y1 = orc.spmv(ai, aj, aa, x), orc.spmv_inode(ai, aj, aa, x)
    scale = np.zeros(m); np.add.at(scale, np.repeat(np.arange(m), np.diff(ai)), np.abs(aa * x[aj]))
    assert np.all(np.abs(y0 - y1) <= 1e-13 * scale) and not np.array_equal(y0, y1)   # same product, different rounding


def test_tutorial_ex2_gmres_ilu0():
    """ex2 -m 5 -n 5 refine_always with the reference's DEFAULT preconditioners (makefile:317-326):
    ex2_1.out = 1 rank, GMRES + ILU(0); ex2_2.out = 2 ranks, GMRES + block Jacobi with ILU(0) on each
    diagonal block (13 + 12 rows).  Pins MatILUFactorSymbolic_ilu0 / MatLUFactorNumeric / MatSolve."""
    ai, aj, aa = pb.lap2d(5, 5)
    u = np.ones(25)
    b = orc.spmv(ai, aj, aa, u)
    os.makedirs(os.path.join(G, "ksp_tutorials"), exist_ok=True)
    rtol = 1e-2 / 36
    gold = pb.parse_monitor(os.path.join(G, "ksp_tutorials", "ex2_1.out"))[0]
    x, h, its, reason = orc.ksp_solve(ai, aj, aa, b, ksp="gmres", pc="ilu", refine_always=1, rtol=rtol, abstol=1e-50)
    pb.check_monitor(h, gold)
    assert its == 4 and "%g" % np.linalg.norm(x - u) == "0.000392701"
    gold = pb.parse_monitor(os.path.join(G, "ksp_tutorials", "ex2_2.out"))[0]
    x, h, its, reason = orc.ksp_solve(ai, aj, aa, b, ksp="gmres", pc="bjacobi", blocks=[0, 13, 25], sub_ksp="preonly", sub_pc="ilu",
                                      refine_always=1, rtol=rtol, abstol=1e-50)
    pb.check_monitor(h, gold)
    assert its == 7 and "%g" % np.linalg.norm(x - u) == "0.000292349"


def test_ksp_tests_ex10_cg_ilu0_on_a_matrix_with_inodes_golden():
    """src/ksp/ksp/examples/tests/ex10.c -matconvert_type seqaij -ksp_monitor_short (makefile:272) vs output/ex10_1.out: one 20-node
    serendipity brick of linear elasticity, 36 x 36 with 1200 nonzeros stored as AIJ -- 'using I-node routines: found 21 nodes,
    limit used is 5', nodes of 1, 2 and 3 rows -- 'matrix 1 norm = 34.0627', then KSPCG (rtol 1e-10) with the one-rank default
    PCILU: ten monitor lines, 9 iterations.  The reference's own golden for MatMult_SeqAIJ_Inode + the ILU(0) routines on a matrix
    with inodes of mixed sizes under CG."""
    import scipy.sparse as sp
    (ai, aj, aa), b, u = pb.ex10_elasticity()
    m = ai.size - 1
    text = open(os.path.join(G, "ksp_tests", "ex10_1.out")).read()
    assert "rows=%d, cols=%d" % (m, m) in text and "total: nonzeros=%d" % aj.size in text
    nc, ns = orc.check_inode(ai, aj)
    assert "using I-node routines: found %d nodes, limit used is 5" % nc in text and sorted(set(ns[:nc].tolist())) == [1, 2, 3]
    assert "matrix 1 norm = %g" % abs(sp.csr_matrix((aa, aj, ai), shape=(m, m))).sum(axis=0).max() in text
    gold = pb.parse_monitor(os.path.join(G, "ksp_tests", "ex10_1.out"))[0]
    x, h, its, reason = orc.ksp_solve(ai, aj, aa, b, ksp="cg", pc="ilu", rtol=1e-10)
    pb.check_monitor(h, gold)
    assert its == 9 and reason == 2 and "Number of iterations %d" % its in text
    assert np.linalg.norm(x - u) < 1e-11                    # 'Norm of error 5.85503e-13': rounding, not reproducible digit by digit


def test_tutorial_ex16_four_right_hand_sides_on_two_ranks_golden():
    """src/ksp/ksp/examples/tutorials/ex16.c -ntimes 4 refine_always on 2 ranks (makefile:839) vs output/ex16_1.out: ONE KSP, the
    default PC of a 2-rank run (block Jacobi + ILU(0), set up once: SAME_PRECONDITIONER), four systems A x = A (k 1), k = 1..4,
    on ex2's 8 x 7 grid; per system 'Norm of error <%G> System k: iterations 9'."""
    ai, aj, aa = pb.lap2d(8, 7)
    lines = [l.split() for l in open(os.path.join(G, "ksp_tutorials", "ex16_1.out")).read().splitlines()]
    assert len(lines) == 4
    for k, want in enumerate(lines, start=1):
        u = np.full(56, float(k))
        x, h, its, reason = orc.ksp_solve(ai, aj, aa, orc.spmv(ai, aj, aa, u), ksp="gmres", pc="bjacobi", blocks=[0, 28, 56],
                                          sub_ksp="preonly", sub_pc="ilu", refine_always=1)
        assert "%g" % np.linalg.norm(x - u) == want[3] and int(want[5].rstrip(":")) == k and its == int(want[7]) == 9


EX7_BLOCK_SOLVERS = [("bcgs", "none", 1e-6) if k % 2 == 0 else ("preonly", "ilu", 1e-5) for k in range(4)] + [("gmres", "jacobi", 1e-7)] * 4


def test_tutorial_ex7_block_jacobi_with_a_different_solver_on_every_block_golden():
    """src/ksp/ksp/examples/tutorials/ex7.c -ksp_monitor_short refine_always on 2 ranks (makefile:433) vs output/ex7_1.out: 8 x 10
    grid, GMRES + block Jacobi with EIGHT blocks of ten rows (PCBJacobiSetTotalBlocks), the sub-solvers set block by block through
    PCBJacobiGetSubKSP (ex7.c:173-195) -- rank 0's blocks alternately BiCGStab without a preconditioner (rtol 1e-6) and the default
    ILU(0) application, rank 1's GMRES + Jacobi (rtol 1e-7).  Fourteen monitor lines and 'Norm of error 1.09983e-05 iterations 13':
    the reference's golden for BiCGStab, GMRES and ILU(0) as INNER solvers."""
    ai, aj, aa = pb.lap2d(8, 10)
    u = np.ones(80)
    x, h, its, reason = orc.ksp_solve(ai, aj, aa, orc.spmv(ai, aj, aa, u), ksp="gmres", pc="bjacobi", blocks=list(range(0, 81, 10)),
                                      block_solvers=EX7_BLOCK_SOLVERS, refine_always=1)
    pb.check_monitor(h, pb.parse_monitor(os.path.join(G, "ksp_tutorials", "ex7_1.out"))[0])
    assert open(os.path.join(G, "ksp_tutorials", "ex7_1.out")).read().splitlines()[-1] == "Norm of error %g iterations %d" % (np.linalg.norm(x - u), its)


def test_ksp_tests_ex40_gmres_without_refinement_and_without_a_preconditioner_golden():
    """src/ksp/ksp/examples/tests/ex40.c -pc_type none on 6 ranks (makefile:819) vs output/ex40.out: ex2's 8 x 7 operator, u = 1,
    rtol 1e-2 / 72, the DEFAULT GMRES -- classical Gram-Schmidt with no refinement step, the one form the other goldens (all
    refine_always) do not reach -- and PCNONE: 'Norm of error 1.68964e-05 iterations 13'.  (The example stores the operator as
    MATELEMENTAL, a dense distributed type outside this path; the Krylov sequence over the same operator is what the line pins, and
    without a preconditioner the rank count enters only through the order of the reductions.)"""
    ai, aj, aa = pb.lap2d(8, 7)
    u = np.ones(56)
    x, h, its, reason = orc.ksp_solve(ai, aj, aa, orc.spmv(ai, aj, aa, u), ksp="gmres", pc="none", rtol=1e-2 / 72, abstol=1e-50)
    assert open(os.path.join(G, "ksp_tests", "ex40.out")).read().strip() == "Norm of error %g iterations %d" % (np.linalg.norm(x - u), its)


def ex18_vectors(n=15):
    """src/vec/vec/examples/tests/ex18.c:20-27: x_i = i + 1 / (i + .35), y_i = x_i + 1.375547826473644376"""
    i = np.arange(n, dtype=np.float64)
    x = i + 1.0 / (i + .35)
    return x, x + 1.375547826473644376


def test_vec_tests_ex18_dot_known_answer():
    """src/vec/vec/examples/tests/ex18.c ('Compares BLAS dots on different machines') vs output/ex18_1.out: VecDot of two 15-entry
    vectors printed with %16.12e"""
    x, y = ex18_vectors()
    assert open(os.path.join(G, "vec_tests", "ex18_1.out")).read().strip() == "Vector inner product %16.12e" % orc.vec_dot(x, y)


def test_threaded_cpu_baseline_matches_the_sequential_oracle():
    """bench.py's cpu_baseline loop (one thread per block of rows, partial sums added in rank order -- the reference's
    MPI arrangement inside one process) computes the same CG + Jacobi iterates as the sequential oracle, to the
    rounding of the re-associated dot products"""
    import problems as pb
    ai, aj, aa = pb.lap2d(30, 23)
    n = ai.size - 1
    b = orc.spmv(ai, aj, aa, np.ones(n))
    xs, hs, its, _ = orc.ksp_solve(ai, aj, aa, b, ksp="cg", pc="jacobi", rtol=0.0, abstol=1e-300, dtol=1e300, max_it=25)
    assert its == 25
    for nt in (1, 3, 8):
        t, x, rn = orc.cg_jacobi_mt(ai, aj, aa, b, 25, nt)
        assert t >= 0.0
        assert np.linalg.norm(x - xs) <= 1e-12 * np.linalg.norm(xs)
        assert abs(rn - hs[25]) <= 1e-10 * hs[0]


@pytest.mark.parametrize("bs", [1, 2, 3, 4, 5, 6, 7, 9])
def test_block_inverse_and_pbjacobi_restatement(bs):
    """PetscKernel_A_gets_inverse_A_N (dgefa*.c / dgedi.c: LINPACK, partial pivoting) and PCApply_PBJacobi_N (pbjacobi.c):
    the inverse agrees with LAPACK's to rounding (also when the leading entry is zero and rows must be interchanged), a
    singular block reports its zero pivot, and the apply is the column-major block times the vector."""
    rng = np.random.default_rng(bs)
    mbs = 40
    blocks = rng.standard_normal((mbs, bs, bs)) + 3.0 * np.eye(bs)
    if bs > 1:
        blocks[0, 0, 0] = 0.0                                   # forces an interchange in the first column
        blocks[1] = blocks[1][::-1].copy()                      # anti-diagonal dominant: interchanges in every column
    bi = np.arange(mbs + 1, dtype=np.int32); bj = np.arange(mbs, dtype=np.int32)
    ba = np.ascontiguousarray(blocks.transpose(0, 2, 1)).reshape(-1)   # column-major blocks
    idiag = orc.pbjacobi_setup(bs, bi, bj, ba)
    inv = idiag.reshape(mbs, bs, bs).transpose(0, 2, 1)
    for b in range(mbs):
        assert np.allclose(inv[b] @ blocks[b], np.eye(bs), rtol=0, atol=1e-11 * np.linalg.cond(blocks[b]))
    x = rng.standard_normal(mbs * bs)
    y = orc.pbjacobi_apply(bs, idiag, x)
    assert np.allclose(y, np.einsum("brc,bc->br", inv, x.reshape(mbs, bs)).reshape(-1), rtol=1e-13, atol=1e-13)
    if bs > 1:
        sing = np.ones((bs, bs)).reshape(-1)                    # rank one: the second pivot is exactly zero
        assert orc.lib().orc_block_inverse(C.c_int(bs), orc.D(sing)) == 2


def test_pbjacobi_with_block_size_one_is_jacobi():
    """bs = 1: dgedi's 1.0 / a and the single product d * x are PCJacobi's VecReciprocal and VecPointwiseMult: same bits."""
    ai, aj, aa = pb.lap2d(12, 9)
    aa = aa * (1.0 + 0.1 * np.cos(np.arange(aa.size)))
    b = np.sin(np.arange(ai.size - 1.0))
    xj, hj, itj, rj = orc.ksp_solve(ai, aj, aa, b, ksp="cg", pc="jacobi", rtol=1e-9)
    xp, hp, itp, rp = orc.ksp_solve(ai, aj, aa, b, ksp="cg", pc="pbjacobi", pb_bs=1, rtol=1e-9)
    assert itj == itp and rj == rp and np.array_equal(xj.view(np.uint64), xp.view(np.uint64)) and np.array_equal(hj, hp)


def test_pbjacobi_preconditions_a_block_system():
    """CG + PCPBJACOBI on a 3-dof SPD operator converges in fewer iterations than CG + PCJACOBI and to the same solution."""
    (ai, aj, aa), _ = pb.spd_blocks(6, 5, 4)
    n = ai.size - 1
    b = np.cos(0.1 * np.arange(n))
    xj, _, itj, rj = orc.ksp_solve(ai, aj, aa, b, ksp="cg", pc="jacobi", rtol=1e-10)
    xp, _, itp, rp = orc.ksp_solve(ai, aj, aa, b, ksp="cg", pc="pbjacobi", pb_bs=3, rtol=1e-10)
    assert rj > 0 and rp > 0 and itp < itj, (itp, itj)
    assert np.allclose(xj, xp, rtol=1e-7, atol=1e-9)


def test_pipecg_restatement_is_cg_with_the_natural_norm():
    """KSPSolve_PIPECG (pipecg.c:49-205): with the natural norm (gamma refreshed every iteration) it spans CG's Krylov space --
    same iteration count +-1 and solution as the CG restatement pinned by pc/tests ex2_1.out; on that golden's own problem
    (tridiagonal n = 10, PCNONE) it prints CG's natural-norm residuals to 6 digits"""
    ai, aj, aa = pb.tridiag(10)
    b = orc.spmv(ai, aj, aa, np.ones(10))
    xc, hc, itc, rc = orc.ksp_solve(ai, aj, aa, b, ksp="cg", pc="none", norm_type=3)
    xp, hp, itp, rp = orc.ksp_solve(ai, aj, aa, b, ksp="pipecg", pc="none", norm_type=3)
    assert rc == rp and itc == itp and np.allclose(hp, hc, rtol=1e-6) and np.linalg.norm(xp - 1.0) < 1e-12
    ai, aj, aa = pb.lap2d(23, 19)
    b = np.cos(0.2 * np.arange(ai.size - 1))
    xc, hc, itc, rc = orc.ksp_solve(ai, aj, aa, b, ksp="cg", pc="jacobi", norm_type=3, rtol=1e-9)
    xp, hp, itp, rp = orc.ksp_solve(ai, aj, aa, b, ksp="pipecg", pc="jacobi", norm_type=3, rtol=1e-9)
    assert rc == rp and abs(itc - itp) <= 1 and np.allclose(xp, xc, rtol=1e-7, atol=1e-10)
    # the snapshot's quirk: with the preconditioned norm gamma is reduced once; the walk does not converge
    xq, hq, itq, rq = orc.ksp_solve(ai, aj, aa, b, ksp="pipecg", pc="jacobi", norm_type=1, rtol=1e-9, max_it=200)
    assert rq < 0


def test_ex32_cg_icc0_golden_and_factor_properties():
    """ksp/tests/ex32.c -dof 1 -ksp_type cg -pc_type icc -pc_factor_mat_ordering_type natural -mat_type aij -pc_factor_levels 0
    (makefile runex32_testset5, first case) vs output/ex32_5.out: pins the oracle's ICC(0) -- MatICCFactorSymbolic_SeqAIJ /
    MatCholeskyFactorNumeric_SeqAIJ / MatSolve_SeqSBAIJ_1_NaturalOrdering.  Plus what the factor has to satisfy: U^T D U agrees with A on
    A's pattern (zero fill), the preconditioner is symmetric, and a matrix that is not diagonally dominant takes the reference's
    positive-definite shifts and still yields a positive diagonal"""
    (ai, aj, aa), b = pb.ex32()
    gold = pb.parse_monitor(os.path.join(G, "ksp_tests", "ex32_5.out"))[0]
    x, h, its, reason = orc.ksp_solve(ai, aj, aa, b, ksp="cg", pc="icc")
    pb.check_monitor(h, gold)
    import scipy.sparse as sp
    for ai, aj, aa in (pb.lap2d(7, 6), orc.gen_p7(5, 4, 3)):
        n = ai.size - 1
        (ui, uj, ua), ns = orc.icc0_factor(ai, aj, aa)
        assert ns == 0
        # rebuild U (unit diagonal) and D from the stored multipliers: stored off-diagonal = -U(i,j), stored diagonal = 1/D(i)
        U = sp.lil_matrix((n, n)); D = np.zeros(n)
        for i in range(n):
            D[i] = 1.0 / ua[ui[i + 1] - 1]
            U[i, i] = 1.0
            for q in range(ui[i], ui[i + 1] - 1):
                U[i, uj[q]] = -ua[q]
        U = U.tocsr()
        R = (U.T @ sp.diags(D) @ U).tocsr()
        A = sp.csr_matrix((aa, aj, ai), shape=(n, n))
        mask = A.copy(); mask.data[:] = 1.0
        assert abs((R.multiply(mask) - A)).max() < 1e-13          # exact on the pattern of A, fill dropped
        e = np.eye(n)
        M = np.column_stack([orc.icc0_solve((ui, uj, ua), e[:, j].copy()) for j in range(n)])
        assert np.allclose(M, M.T, rtol=0, atol=1e-14) and np.allclose(M, np.linalg.inv(R.toarray()), rtol=1e-10, atol=1e-12)
    ai, aj, aa = pb.lap2d(8, 8)
    aa = aa.copy(); aa[aj == np.repeat(np.arange(ai.size - 1), np.diff(ai))] = 1.5
    (ui, uj, ua), ns = orc.icc0_factor(ai, aj, aa)
    assert 1 <= ns <= 6 and np.all(ua[ui[1:] - 1] > 0)


def test_ilu0_nonzero_shift_restarts_like_the_reference():
    """PCILU's default MAT_SHIFT_NONZERO (ilu.c:387-389, MatPivotCheck_nz): tridiag(1, 1, 1) has a zero second pivot; the
    factorisation restarts with the diagonal shifted by 100 eps, 200 eps, ... until every pivot passes |pivot| > 100 eps * (row sum of the
    factor's off-diagonal entries); a matrix whose pivots pass is factored exactly as before (no shift, same bits)"""
    import scipy.sparse as sp
    n = 12
    A = sp.diags([np.ones(n - 1), np.ones(n), np.ones(n - 1)], [-1, 0, 1]).tocsr(); A.sort_indices()
    ai, aj, aa = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy()
    f, ns = orc.ilu0_factor_shift(ai, aj, aa)
    assert ns >= 1
    # ILU(0) of a tridiagonal matrix is its LU without pivoting: an independent dense restatement of the rule gives the same number
    # of restarts and the same pivots
    zp = 100.0 * 2.220446049250313e-16
    shift, tries = 0.0, 0
    while True:
        M = A.toarray() + shift * np.eye(n)
        ok = True
        for i in range(n):
            for kk in range(i):
                if M[i, kk] != 0.0:
                    m = M[i, kk] / M[kk, kk]
                    M[i, kk] = m
                    M[i, kk + 1:] -= m * M[kk, kk + 1:]
            rs = np.abs(M[i, :i]).sum() + np.abs(M[i, i + 1:]).sum()
            if abs(M[i, i]) <= zp * rs:
                shift = zp if tries == 0 else 2.0 * shift
                tries += 1
                ok = False
                break
        if ok:
            break
    bi, bj, bd, ba = f
    assert tries == ns and np.allclose(1.0 / ba[bd[:n]], np.diag(M), rtol=1e-12, atol=0)
    ai, aj, aa = pb.lap2d(6, 5)
    f0, ns0 = orc.ilu0_factor_shift(ai, aj, aa)
    assert ns0 == 0 and all(np.array_equal(u, v) for u, v in zip(f0, orc.ilu0_factor(ai, aj, aa)))
