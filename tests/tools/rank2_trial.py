"""Development aid: run the N>1 path (MatMPIAIJ + RCCL halo + RCCL all-reduce) with world ranks that may share
one GPU, and compare every rank's result with the sequential oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch  # noqa: F401
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import petsc_dev_amd as pda  # noqa: F401
    from petsc_dev_amd import petsc as P
    from petsc_dev_amd import dist as PD
    import orc
    L = P.lib()
    staged = os.environ.get("MI355X_STAGED", "0") == "1"   # several ranks on one GPU: RCCL refuses, use the host-staged transport
    comm = PD.torch_comm(device_comm=not staged)
    tr = PD.transport_report(comm)
    print("rank %d/%d: transport=%s rccl_ranks=%d rccl_communicators=%d" % (rank, world, tr["transport"], tr["rccl_ranks"], tr["rccl_communicators"]), flush=True)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    nx, ny, nz = n, n, n * world
    if len(sys.argv) > 2 and sys.argv[2] == "cfg3":
        # the slab shape of BASELINE.json configs[2] (P7(512) on 8 ranks = 64 planes of 512^2 rows per rank) at reduced size:
        # planes twice as wide, a quarter as many per rank -- P7(2n, 2n, n/4 * world); interior ranks have two z-neighbours
        nx, ny, nz = 2 * n, 2 * n, (n // 4) * world
    mloc = nx * ny * (nz // world)
    ai, aj, aa = P.gen_poisson7(nx, ny, nz, rank * mloc, (rank + 1) * mloc)
    A = P.Mat.from_csr_mpi(ai, aj, aa, mloc, mloc * world, mloc * world, comm=comm)
    gi, gj, ga = orc.gen_p7(nx, ny, nz)
    N = mloc * world
    xg = np.sin(0.37 * np.arange(N)) + 1.0
    x = P.Vec.from_array(xg[rank * mloc:(rank + 1) * mloc], comm=comm, N=N)
    y = x.duplicate()
    A.mult(x, y)
    # MatMult_MPIAIJ order (mpiaij.c:1111-1114): diagonal block first, then += off-diagonal block * ghost values
    pc = orc.mpiaij_split(rank * mloc, (rank + 1) * mloc, rank * mloc, (rank + 1) * mloc, gi, gj, ga)
    ref = orc.matmult(pc["ad_i"], pc["ad_j"], pc["ad_a"], xg[rank * mloc:(rank + 1) * mloc])[0]
    ref = orc.matmult(pc["bo_i"], pc["bo_j"], pc["bo_a"], xg[pc["garray"]].copy(), ref)[0]
    ok1 = np.array_equal(y.array().view(np.uint64), ref.view(np.uint64))
    L.MatMultTranspose(A.h, x.h, y.h)
    reft = orc.spmv_t(gi, gj, ga, xg, N)[rank * mloc:(rank + 1) * mloc]
    ok2 = np.allclose(y.array(), reft, rtol=1e-13, atol=1e-13)
    # ... and bit for bit against MatMultTranspose_MPIAIJ's own order (mpiaij.c:1147-1174): A_d^T x, then the neighbours' B_o^T x
    # pieces added through the reverse scatter, owner after owner in rank order
    pcs_ = [orc.mpiaij_split(q * mloc, (q + 1) * mloc, q * mloc, (q + 1) * mloc, gi, gj, ga) for q in range(world)]
    rt = orc.spmv_t(pcs_[rank]["ad_i"], pcs_[rank]["ad_j"], pcs_[rank]["ad_a"], xg[rank * mloc:(rank + 1) * mloc].copy(), mloc)
    for q in range(world):
        if q == rank or not pcs_[q]["garray"].size:
            continue
        lv = orc.spmv_t(pcs_[q]["bo_i"], pcs_[q]["bo_j"], pcs_[q]["bo_a"], xg[q * mloc:(q + 1) * mloc].copy(), pcs_[q]["garray"].size)
        mine = (pcs_[q]["garray"] >= rank * mloc) & (pcs_[q]["garray"] < (rank + 1) * mloc)
        for i_ in np.nonzero(mine)[0]:
            g = int(pcs_[q]["garray"][i_]) - rank * mloc
            rt[g] = rt[g] + lv[i_]
    ok2 = ok2 and np.array_equal(y.array().view(np.uint64), rt.view(np.uint64))
    nrm = x.norm()
    ok3 = abs(nrm - np.linalg.norm(xg)) <= 1e-12 * nrm
    b = x.duplicate(); u = x.duplicate(); L.VecSet(u.h, 1.0); A.mult(u, b)
    k = P.KSP(comm=comm); k.set_operators(A); k.set_type("cghipmi355x"); k.set_pc_type("jacobi"); k.set_tolerances(rtol=1e-8); k.record_history()
    sol = x.duplicate()
    k.solve(b, sol)
    xr, hr, itsr, rr = orc.ksp_solve(gi, gj, ga, orc.spmv(gi, gj, ga, np.ones(N)), ksp="cg", pc="jacobi", rtol=1e-8)
    h = k.history()
    ok4 = abs(k.its - itsr) <= 1 and np.allclose(h[:min(len(h), len(hr))], hr[:min(len(h), len(hr))], rtol=1e-6)
    # the plain KSPSolve_CG sequence of an unchanged program over PARALLEL vectors: its update calls noted and run as one fused sweep
    # (one reduction of three values across the ranks) == every call a kernel and a reduction of its own, bit for bit
    hist = {}
    for on in (0, 1):
        L.raw("VecHIPMI355XSetDeferral")(on)
        kp = P.KSP(comm=comm); kp.set_operators(A); kp.set_type("cg"); kp.set_pc_type("jacobi"); kp.set_tolerances(rtol=1e-8); kp.record_history()
        L.VecSet(sol.h, 0.0)
        kp.solve(b, sol)
        hist[on] = (kp.history().copy(), kp.its, sol.array().copy())
    L.raw("VecHIPMI355XSetDeferral")(-1)
    ok9 = hist[0][1] == hist[1][1] and np.array_equal(hist[0][0].view(np.uint64), hist[1][0].view(np.uint64)) and np.array_equal(hist[0][2].view(np.uint64), hist[1][2].view(np.uint64))
    print("rank %d/%d: plain KSPSolve_CG with the update calls deferred == launched one by one: %s (its %d)" % (rank, world, ok9, hist[1][1]), flush=True)
    # Gropp's CG over the same operator: split-phase reductions across the ranks (staged: host all-reduce at the End)
    kg = P.KSP(comm=comm); kg.set_operators(A); kg.set_type("groppcg"); kg.set_pc_type("jacobi"); kg.set_tolerances(rtol=1e-8); kg.record_history()
    L.VecSet(sol.h, 0.0)
    kg.solve(b, sol)
    xg_, hg, itsg, rg = orc.ksp_solve(gi, gj, ga, orc.spmv(gi, gj, ga, np.ones(N)), ksp="groppcg", pc="jacobi", rtol=1e-8)
    hh = kg.history()
    ok4 = ok4 and abs(kg.its - itsg) <= 1 and np.allclose(hh[:min(len(hh), len(hg))], hg[:min(len(hh), len(hg))], rtol=1e-6)
    # ---- irregular pattern, uneven ownership: non-contiguous halo indices (device pack / unpack kernels), several
    # neighbours, reverse-mode additions in rank order -- bit-exact against the MPIAIJ-ordered oracle emulation
    import scipy.sparse as sp
    NI = 2003
    # (rows stay short -- at most ~13 entries in either block -- so that every row is summed by ONE lane in column order and the
    # comparison can be bit for bit: blocks averaging more than 16 entries per row use several lanes and a tree, BASELINE.md's 1e-12)
    S = sp.random(NI, NI, density=0.01 if world <= 3 else 0.006, random_state=11, format="csr") + sp.eye(NI, format="csr")
    S = sp.csr_matrix(S); S.sort_indices()
    si, sj, sa = S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data.astype(np.float64)
    rng_ = np.array([0] + list(np.cumsum([NI // world + (NI % world > r) for r in range(world)])), dtype=np.int32)
    rs, re_ = int(rng_[rank]), int(rng_[rank + 1])
    li = (si[rs:re_ + 1] - si[rs]).astype(np.int32)
    B = P.Mat.from_csr_mpi(li, sj[si[rs]:si[re_]].copy(), sa[si[rs]:si[re_]].copy(), re_ - rs, NI, NI, comm=comm)
    xi = np.cos(0.3 * np.arange(NI)) + 0.1
    vx = P.Vec.from_array(xi[rs:re_], comm=comm, N=NI)
    vy = vx.duplicate()
    B.mult(vx, vy)
    pcs = [orc.mpiaij_split(int(rng_[q]), int(rng_[q + 1]), int(rng_[q]), int(rng_[q + 1]), si, sj, sa) for q in range(world)]
    me = pcs[rank]
    ref = orc.matmult(me["ad_i"], me["ad_j"], me["ad_a"], xi[rs:re_].copy())[0]
    if me["garray"].size:
        ref = orc.matmult(me["bo_i"], me["bo_j"], me["bo_a"], xi[me["garray"]].copy(), ref)[0]
    ok5 = np.array_equal(vy.array().view(np.uint64), ref.view(np.uint64))
    L.MatMultTranspose(B.h, vx.h, vy.h)
    reft = orc.spmv_t(me["ad_i"], me["ad_j"], me["ad_a"], xi[rs:re_].copy(), re_ - rs)
    for q in range(world):
        if q == rank or not pcs[q]["garray"].size:
            continue
        lv = orc.spmv_t(pcs[q]["bo_i"], pcs[q]["bo_j"], pcs[q]["bo_a"], xi[rng_[q]:rng_[q + 1]].copy(), pcs[q]["garray"].size)
        for i_, g in enumerate(pcs[q]["garray"]):
            if rs <= g < re_:
                reft[g - rs] = reft[g - rs] + lv[i_]
    ok6 = np.array_equal(vy.array().view(np.uint64), reft.view(np.uint64))
    print("rank %d/%d: irregular MatMult bitexact=%s MatMultTranspose bitexact=%s" % (rank, world, ok5, ok6), flush=True)
    # the same matrix with its diagonal block's product (and the cached transpose's) run by the column-tiled kernel, every (panel, tile)
    # pair staged (-mat_hipmi355x_tiled 1 -mat_hipmi355x_tiled_stage_min 1): a row's products are then added in column order, the reference's
    L.PetscOptionsClear(); L.PetscOptionsInsertString(b"-mat_hipmi355x_tiled 1 -mat_hipmi355x_tiled_stage_min 1")
    Bt = P.Mat.from_csr_mpi(li, sj[si[rs]:si[re_]].copy(), sa[si[rs]:si[re_]].copy(), re_ - rs, NI, NI, comm=comm)
    vx.set_array(xi[rs:re_].copy())
    Bt.mult(vx, vy)
    ok5t = np.array_equal(vy.array().view(np.uint64), ref.view(np.uint64))
    L.MatMultTranspose(Bt.h, vx.h, vy.h)
    ok6t = np.array_equal(vy.array().view(np.uint64), reft.view(np.uint64))
    L.PetscOptionsClear()
    print("rank %d/%d: irregular MatMult through the column-tiled kernel bitexact=%s MatMultTranspose bitexact=%s" % (rank, world, ok5t, ok6t), flush=True)
    ok5, ok6 = ok5 and ok5t, ok6 and ok6t
    # ---- the matrix's VecScatter by itself, every InsertMode in both directions (UnPack_1's INSERT / ADD / MAX, vpscat.c:503-534):
    # forward: ghost slot i receives x[garray[i]] from its owner; reverse: owned entry g receives, neighbour after neighbour in rank
    # order, what every rank holds in its ghost slot for g.  Bit for bit (max and a single addition per step are exact operations
    # in a fixed order).
    import ctypes as C
    sc, lvh, ecn = C.c_void_p(), C.c_void_p(), C.c_int()
    L.MatMPIAIJGetScatter(B.h, C.byref(sc), C.byref(lvh), C.byref(ecn))
    lvec = P.Vec(lvh, own=False)
    ghost = lambda q: np.cos(0.1 * pcs[q]["garray"] + q) * (1.0 + q)       # what rank q holds in its ghost slots (reverse direction)
    l0 = np.sin(0.2 * np.arange(me["garray"].size)) * 0.9                   # ghost slots before a forward scatter
    y0 = np.sin(0.05 * np.arange(rs, re_)) * 0.8                            # owned entries before a reverse scatter
    ok8 = ecn.value == me["garray"].size
    INSERT, ADD, MAX = 1, 2, 3
    for mode in (INSERT, ADD, MAX):
        if me["garray"].size:
            lvec.set_array(l0)
        vx.set_array(xi[rs:re_])
        L.VecScatterBegin(sc, vx.h, lvec.h, mode, 0)
        L.VecScatterEnd(sc, vx.h, lvec.h, mode, 0)
        got = lvec.array()
        inc = xi[me["garray"]]
        want = inc if mode == INSERT else (l0 + inc if mode == ADD else np.where(l0 < inc, inc, l0))
        ok8 = ok8 and np.array_equal(got.view(np.uint64), np.ascontiguousarray(want).view(np.uint64))
        if me["garray"].size:
            lvec.set_array(ghost(rank))
        vy.set_array(y0)
        L.VecScatterBegin(sc, lvec.h, vy.h, mode, 1)
        L.VecScatterEnd(sc, lvec.h, vy.h, mode, 1)
        want = y0.copy()
        for q in range(world):
            if q == rank:
                continue
            gq, vq = pcs[q]["garray"], ghost(q)
            for i_ in np.nonzero((gq >= rs) & (gq < re_))[0]:
                g = int(gq[i_]) - rs
                want[g] = vq[i_] if mode == INSERT else (want[g] + vq[i_] if mode == ADD else (vq[i_] if want[g] < vq[i_] else want[g]))
        ok8 = ok8 and np.array_equal(vy.array().view(np.uint64), want.view(np.uint64))
    print("rank %d/%d: VecScatter INSERT / ADD / MAX, forward and reverse, bitexact=%s" % (rank, world, ok8), flush=True)
    # MatDiagonalScale_MPIAIJ (mpiaij.c:2183): left vector by rows, right vector's ghost values through the MatMult
    # scatter; the device copies of both blocks are updated in place.  Then MatScale.  Against the scaled oracle pieces.
    lg = 1.0 + 0.5 * np.cos(np.arange(NI)); rg = 2.0 + np.sin(0.7 * np.arange(NI))
    vl = P.Vec.from_array(lg[rs:re_], comm=comm, N=NI); vr = P.Vec.from_array(rg[rs:re_], comm=comm, N=NI)
    L.MatDiagonalScale(B.h, vl.h, vr.h)
    L.MatScale(B.h, -0.37)
    B.mult(vx, vy)
    ad = -0.37 * orc.diagonal_scale(me["ad_i"], me["ad_j"], me["ad_a"], lg[rs:re_].copy(), rg[rs:re_].copy())
    ref = orc.matmult(me["ad_i"], me["ad_j"], ad, xi[rs:re_].copy())[0]
    if me["garray"].size:
        bo = orc.diagonal_scale(me["bo_i"], me["bo_j"], me["bo_a"], lg[rs:re_].copy(), None)
        bo = -0.37 * orc.diagonal_scale(me["bo_i"], me["bo_j"], bo, None, rg[me["garray"]].copy())
        ref = orc.matmult(me["bo_i"], me["bo_j"], bo, xi[me["garray"]].copy(), ref)[0]
    ok7 = np.array_equal(vy.array().view(np.uint64), ref.view(np.uint64))
    print("rank %d/%d: MatDiagonalScale + MatScale then MatMult bitexact=%s" % (rank, world, ok7), flush=True)
    ok1 = ok1 and ok5 and ok6
    # ---- reference golden on 2 ranks: ex2 -m 5 -n 5 -ksp_gmres_cgs_refinement_type refine_always with NO pc option =
    # GMRES + block Jacobi + ILU(0) per rank (13 + 12 rows) == src/ksp/ksp/examples/tutorials/output/ex2_2.out
    if world == 2:
        import problems as pb
        gi2, gj2, ga2 = pb.lap2d(5, 5)
        split = [0, 13, 25]
        r0_, r1_ = split[rank], split[rank + 1]
        li2 = (gi2[r0_:r1_ + 1] - gi2[r0_]).astype(np.int32)
        E = P.Mat.from_csr_mpi(li2, gj2[gi2[r0_]:gi2[r1_]].copy(), ga2[gi2[r0_]:gi2[r1_]].copy(), r1_ - r0_, 25, 25, comm=comm)
        ue = P.Vec.create(r1_ - r0_, N=25, comm=comm); L.VecSet(ue.h, 1.0)
        be, xe = ue.duplicate(), ue.duplicate()
        E.mult(ue, be)
        ke = P.KSP(comm=comm); ke.set_operators(E)
        L.PetscOptionsClear(); L.PetscOptionsInsertString(b"-ksp_gmres_cgs_refinement_type refine_always")
        ke.set_tolerances(rtol=1e-2 / 36, abstol=1e-50); ke.set_from_options(); ke.record_history()
        ke.solve(be, xe)
        L.PetscOptionsClear()
        gold = pb.parse_monitor(os.path.join(ROOT, "tests", "golden", "ksp_tutorials", "ex2_2.out"))[0]
        pb.check_monitor(ke.history(), gold)
        L.VecAXPY(xe.h, -1.0, ue.h)
        okg = ke.its == 7 and "%.5g" % xe.norm() in ("0.00029235", "0.00029234")
        print("rank %d/%d: golden ex2_2.out (GMRES + bjacobi + ILU(0), 2 ranks) its=%d ok=%s" % (rank, world, ke.its, okg), flush=True)
        ok1 = ok1 and okg
        # ---- ex5_5.out (tutorials/makefile:423: -n 2 ./ex5 -ksp_gmres_cgs_refinement_type refine_always): two systems with ONE KSP
        # and the default PC; the second after MatZeroEntries + re-assembly into the same pattern -> ILU(0) re-factored in both
        # blocks.  The file prints "Norm of error <%G>, Iterations <its>" per solve.
        import ctypes as C
        lines5 = [l.split() for l in open(os.path.join(ROOT, "tests", "golden", "ksp_tutorials", "ex5_5.out")).read().splitlines() if l.startswith("Norm of error")]
        (g5, u5) = pb.ex5_tutorial(2, False)
        n5 = g5[0].size - 1; h5 = n5 // 2
        r0_, r1_ = (0, h5) if rank == 0 else (h5, n5)
        def local(g):
            gi_, gj_, ga_ = g
            return (gi_[r0_:r1_ + 1] - gi_[r0_]).astype(np.int32), gj_[gi_[r0_]:gi_[r1_]].copy(), ga_[gi_[r0_]:gi_[r1_]].copy()
        li5, lj5, la5 = local(g5)
        F = P.Mat.from_csr_mpi(li5, lj5, la5, r1_ - r0_, n5, n5, comm=comm)
        uf = P.Vec.from_array(u5[r0_:r1_], comm=comm, N=n5); bf, xf = uf.duplicate(), uf.duplicate()
        kf = P.KSP(comm=comm); kf.set_operators(F)
        L.PetscOptionsClear(); L.PetscOptionsInsertString(b"-ksp_gmres_cgs_refinement_type refine_always")
        kf.set_from_options()
        L.PetscOptionsClear()
        ok5_ = True
        for second in (False, True):
            if second:
                _, _, la52 = local(pb.ex5_tutorial(2, True)[0])
                L.MatZeroEntries(F.h)
                rows5 = np.repeat(np.arange(r0_, r1_, dtype=np.int32), np.diff(li5))
                for r_, c_, v_ in zip(rows5, lj5, la52):        # the example's loop of MatSetValues(ADD_VALUES)
                    L.MatSetValues(F.h, 1, C.byref(C.c_int(int(r_))), 1, C.byref(C.c_int(int(c_))), C.byref(C.c_double(float(v_))), P.ADD_VALUES)
                L.MatAssemblyBegin(F.h, P.MAT_FINAL_ASSEMBLY); L.MatAssemblyEnd(F.h, P.MAT_FINAL_ASSEMBLY)
                kf.set_operators(F)
            F.mult(uf, bf)
            kf.solve(bf, xf)
            L.VecAXPY(xf.h, -1.0, uf.h)
            want = lines5[1 if second else 0]
            ok5_ = ok5_ and ("%g" % xf.norm()) == want[3].rstrip(",") and kf.its == int(want[5])
        print("rank %d/%d: golden ex5_5.out (two systems, GMRES + bjacobi + ILU(0), 2 ranks) ok=%s" % (rank, world, ok5_), flush=True)
        ok1 = ok1 and ok5_
        # ---- ex16_1.out (tutorials/makefile:839: -n 2 ./ex16 -ntimes 4 refine_always): four right-hand sides through ONE KSP whose
        # default PC (block Jacobi + ILU(0)) is set up once; and ksp/tests ex40.out (default GMRES, no refinement, PCNONE) on the same
        # 8 x 7 operator split 28 + 28
        g16 = pb.lap2d(8, 7)
        r0_, r1_ = (0, 28) if rank == 0 else (28, 56)
        li16, lj16, la16 = (g16[0][r0_:r1_ + 1] - g16[0][r0_]).astype(np.int32), g16[1][g16[0][r0_]:g16[0][r1_]].copy(), g16[2][g16[0][r0_]:g16[0][r1_]].copy()
        H = P.Mat.from_csr_mpi(li16, lj16, la16, 28, 56, 56, comm=comm)
        uh = P.Vec.create(28, N=56, comm=comm); bh, xh = uh.duplicate(), uh.duplicate()
        kh = P.KSP(comm=comm); kh.set_operators(H)
        L.PetscOptionsClear(); L.PetscOptionsInsertString(b"-ksp_gmres_cgs_refinement_type refine_always")
        kh.set_from_options()
        L.PetscOptionsClear()
        ok16 = True
        for k16, want in enumerate([l.split() for l in open(os.path.join(ROOT, "tests", "golden", "ksp_tutorials", "ex16_1.out")).read().splitlines()], start=1):
            L.VecSet(uh.h, float(k16))
            H.mult(uh, bh)
            kh.solve(bh, xh)
            L.VecAXPY(xh.h, -1.0, uh.h)
            ok16 = ok16 and ("%g" % xh.norm()) == want[3] and kh.its == int(want[7])
        print("rank %d/%d: golden ex16_1.out (four right-hand sides, one KSP, GMRES + bjacobi + ILU(0), 2 ranks) ok=%s" % (rank, world, ok16), flush=True)
        k40 = P.KSP(comm=comm); k40.set_operators(H)
        L.PetscOptionsClear(); L.PetscOptionsInsertString(b"-pc_type none")
        k40.set_tolerances(rtol=1e-2 / 72, abstol=1e-50); k40.set_from_options()
        L.PetscOptionsClear()
        L.VecSet(uh.h, 1.0)
        H.mult(uh, bh)
        k40.solve(bh, xh)
        L.VecAXPY(xh.h, -1.0, uh.h)
        ok40 = open(os.path.join(ROOT, "tests", "golden", "ksp_tests", "ex40.out")).read().strip() == "Norm of error %g iterations %d" % (xh.norm(), k40.its)
        print("rank %d/%d: golden ex40.out (default GMRES, PCNONE, 2 ranks) ok=%s" % (rank, world, ok40), flush=True)
        ok1 = ok1 and ok16 and ok40
        # ---- ex7_1.out (tutorials/makefile:433: -n 2 ./ex7 -ksp_monitor_short refine_always) as the example runs it: 8 x 10 grid, eight
        # blocks of ten rows, four per rank, each block's solver set after KSPSetUp through PCBJacobiGetSubKSP -- rank 0 alternately
        # BiCGStab + PCNONE (rtol 1e-6) and the default ILU(0), rank 1 GMRES + Jacobi (rtol 1e-7)
        g7 = pb.lap2d(8, 10)
        r0_, r1_ = (0, 40) if rank == 0 else (40, 80)
        li7, lj7, la7 = (g7[0][r0_:r1_ + 1] - g7[0][r0_]).astype(np.int32), g7[1][g7[0][r0_]:g7[0][r1_]].copy(), g7[2][g7[0][r0_]:g7[0][r1_]].copy()
        A7 = P.Mat.from_csr_mpi(li7, lj7, la7, 40, 80, 80, comm=comm)
        u7 = P.Vec.create(40, N=80, comm=comm); L.VecSet(u7.h, 1.0)
        b7, x7 = u7.duplicate(), u7.duplicate()
        A7.mult(u7, b7)
        k7 = P.KSP(comm=comm); k7.set_operators(A7)
        L.PetscOptionsClear()
        L.PetscOptionsInsertString(b"-pc_type bjacobi -pc_bjacobi_blocks 8 -pc_bjacobi_merge_blocks 0 -ksp_gmres_cgs_refinement_type refine_always")
        k7.set_from_options(); k7.record_history()
        L.KSPSetUp(k7.h)
        pc7, nloc7, first7, sub7 = C.c_void_p(), C.c_int(), C.c_int(), C.c_void_p()
        L.KSPGetPC(k7.h, C.byref(pc7))
        L.PCBJacobiGetSubKSP(pc7, C.byref(nloc7), C.byref(first7), C.byref(sub7))
        subs7 = C.cast(sub7, C.POINTER(C.c_void_p))
        for i7 in range(nloc7.value):
            spc7 = C.c_void_p(); L.KSPGetPC(subs7[i7], C.byref(spc7))
            if rank == 0 and i7 % 2:
                L.PCSetType(spc7, b"ilu")
            elif rank == 0:
                L.PCSetType(spc7, b"none"); L.KSPSetType(subs7[i7], b"bcgs"); L.KSPSetTolerances(subs7[i7], 1e-6, P.PETSC_DEFAULT, P.PETSC_DEFAULT, int(P.PETSC_DEFAULT))
            else:
                L.PCSetType(spc7, b"jacobi"); L.KSPSetType(subs7[i7], b"gmres"); L.KSPSetTolerances(subs7[i7], 1e-7, P.PETSC_DEFAULT, P.PETSC_DEFAULT, int(P.PETSC_DEFAULT))
        k7.solve(b7, x7)
        L.PetscOptionsClear()
        gold7 = open(os.path.join(ROOT, "tests", "golden", "ksp_tutorials", "ex7_1.out")).read().splitlines()
        pb.check_monitor(k7.history(), pb.parse_monitor(os.path.join(ROOT, "tests", "golden", "ksp_tutorials", "ex7_1.out"))[0])
        L.VecAXPY(x7.h, -1.0, u7.h)
        ok7_ = (nloc7.value, first7.value) == (4, 4 * rank) and gold7[-1] == "Norm of error %g iterations %d" % (x7.norm(), k7.its)
        print("rank %d/%d: golden ex7_1.out (block Jacobi, a different solver on every block, 2 ranks) ok=%s" % (rank, world, ok7_), flush=True)
        ok1 = ok1 and ok7_
    # ---- reference golden on 3 ranks: src/mat/examples/tests/ex5.c -mat_type mpiaij -test_diagonalscale (makefile:806) == output/ex5_33.out:
    # MatMult, MatMultTranspose, MatGetDiagonal, then MatDiagonalScale(C, x = the diagonal, y = 1..n) -- every printed number; the scaled
    # matrix is read back column by column through MatMult with unit vectors (one exact product per entry)
    if world == 3:
        import re
        import problems as pb
        fmt5 = lambda v: np.array([float("%g" % t) for t in v])  # noqa: E731
        g5i, g5j, g5a, m5, n5 = pb.ex5_mat(8)
        rng5 = [0, 3, 6, 8]
        lo5, hi5 = rng5[rank], rng5[rank + 1]
        C5 = P.Mat.from_csr_mpi((g5i[lo5:hi5 + 1] - g5i[lo5]).astype(np.int32), g5j[g5i[lo5]:g5i[hi5]].copy(), g5a[g5i[lo5]:g5i[hi5]].copy(), hi5 - lo5, m5, n5, comm=comm)
        text5 = open(os.path.join(ROOT, "tests", "golden", "ex5_33.out")).read()
        gold5 = pb.parse_vecview(os.path.join(ROOT, "tests", "golden", "ex5_33.out"))
        views5 = [np.array([[float(v) for _, v in re.findall(r"\((\d+), ([-0-9.e+]+)\)", line)] for line in blk.splitlines() if line.startswith("row ")])
                  for blk in text5.split("Matrix Object:")[1:]]
        y5 = P.Vec.from_array(np.arange(lo5, hi5, dtype=np.float64), comm=comm, N=n5); x5 = y5.duplicate()
        C5.mult(y5, x5)
        ok33 = np.array_equal(fmt5(x5.array()), gold5[0][lo5:hi5])
        x5.set_array(np.arange(lo5, hi5, dtype=np.float64))
        L.MatMultTranspose(C5.h, x5.h, y5.h)
        ok33 = ok33 and np.array_equal(fmt5(y5.array()), gold5[1][lo5:hi5])
        L.MatGetDiagonal(C5.h, x5.h)
        ok33 = ok33 and np.array_equal(fmt5(x5.array()), gold5[2][lo5:hi5])
        y5.set_array(np.arange(lo5 + 1, hi5 + 1, dtype=np.float64))
        L.MatDiagonalScale(C5.h, x5.h, y5.h)
        e5, c5 = y5.duplicate(), x5.duplicate()
        for j5 in range(n5):
            unit = np.zeros(hi5 - lo5)
            if lo5 <= j5 < hi5:
                unit[j5 - lo5] = 1.0
            e5.set_array(unit)
            C5.mult(e5, c5)
            ok33 = ok33 and np.array_equal(fmt5(c5.array()), views5[1][lo5:hi5, j5])
        print("rank %d/%d: golden ex5_33.out (MatMult, MatMultTranspose, MatGetDiagonal, MatDiagonalScale of MPIAIJ, 3 ranks) ok=%s" % (rank, world, ok33), flush=True)
        ok1 = ok1 and ok33
    print("rank %d/%d: MatMult bitexact=%s MatMultTranspose=%s norm=%s CG its=%d (oracle %d) hist=%s" % (rank, world, ok1, ok2, ok3, k.its, itsr, ok4), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not (ok1 and ok2 and ok3 and ok4):
        sys.exit(1)


if __name__ == "__main__":
    main()
