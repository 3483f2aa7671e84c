"""CPU-side checks (no GPU): both C-ABI libraries load and export every symbol their headers declare;
host set-up logic (assembly, MPIAIJ split, garray, VecScatter index lists -- integer work, bit-exact
against the oracle); the product fails loudly without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", txt))
    return {n for n in names if re.match(r"(mi355x_|Petsc|Vec|Mat|KSP|PC|MPI_Comm_)", n) and not n.endswith("Fn") and n != "PetscErrorCode"}


def test_kernel_library_exports(built):
    lib = C.CDLL(built.kernels_lib_path())
    for h in ("mi355x_kernels.h", "mi355x_comm.h"):
        names = declared(h)
        assert len(names) > 10
        for n in sorted(names):
            assert hasattr(lib, n), "%s declared in %s but not exported" % (n, h)


def test_harness_and_plugin_library_exports(built):
    """the harness (stand-in PETSc object model) exports what include/petscmini.h declares and loads ON ITS OWN: it has
    no reference into the plugin or the kernel library.  The plugin exports what include/petschipmi355x.h declares."""
    import subprocess
    harness = C.CDLL(built.harness_lib_path(), mode=C.RTLD_GLOBAL)      # before anything else is loaded: must resolve alone
    names = declared("petscmini.h")
    assert len(names) > 100
    for n in sorted(names):
        assert hasattr(harness, n), "%s declared in petscmini.h but not exported by the harness" % n
    und = subprocess.run(["nm", "-D", "--undefined-only", built.harness_lib_path()], capture_output=True, text=True, check=True).stdout
    assert "mi355x_" not in und and "HIPMI355X" not in und and "hip" not in und.lower().replace("ship", ""), und
    built.load_kernels()
    plugin = C.CDLL(built.host_lib_path())
    names = declared("petschipmi355x.h")
    assert len(names) > 15
    for n in sorted(names):
        assert hasattr(plugin, n), "%s declared in petschipmi355x.h but not exported by the plugin" % n
    # what the plugin takes from the object model are PETSc's own entry points (plus the three harness spellings of
    # PetscError / PetscMalloc / communicator attributes): the list a port to a real PETSc has to satisfy
    und = subprocess.run(["nm", "-D", "--undefined-only", built.host_lib_path()], capture_output=True, text=True, check=True).stdout
    wanted = {l.split()[-1] for l in und.splitlines() if l.split() and re.match(r"(Petsc|Vec|Mat|KSP|PC|PETSC_)", l.split()[-1])}
    allowed = {"MatAssemblyBegin", "MatAssemblyEnd", "MatCreate", "MatDestroy", "MatDiagonalScale", "MatRegister", "MatScale",
               "MatSeqAIJSetPreallocation", "MatSeqAIJSetPreallocationCSR", "MatSetSizes", "MatSetType", "MatSetValues", "MatZeroEntries",
               "PCRegister", "PETSC_COMM_SELF", "PetscCommSetPluginData", "PetscCommSplitReductionBegin", "PetscError", "PetscLayoutCreateSetUp",
               "PetscLayoutDestroy", "PetscLayoutReference", "PetscLogFlops", "PetscMallocFn", "PetscObjectChangeTypeName",
               "PetscObjectComposeFunction", "PetscOptionsGetInt", "PetscOptionsGetString", "VecCreate", "VecDestroy", "VecGetArray",
               "VecGetArrayRead", "VecRegister", "VecRestoreArray", "VecRestoreArrayRead", "VecSetSizes", "VecSetType",
               # host/kspfused.c (the plug-in's KSP types): the public Vec / Mat / PC calls and the KSP implementation interface of
               # petsc-private/kspimpl.h (KSP_MatMult / KSP_PCApply / KSP_PCApplyBAorAB / KSPLogResidualHistory are macros / inlines there)
               "KSPDefaultGetWork", "KSPInitialResidual", "KSPLogResidualHistory", "KSPMonitor", "KSPRegister", "KSPSetSupportedNorm",
               "KSP_MatMult", "KSP_PCApply", "KSP_PCApplyBAorAB", "MatGetVecs", "PCApply", "PCGetOperators", "PCGetType", "PetscObjectQueryFunction",
               "PCFactorGetMatrix",   # host/ilu.c: the PC-level introspection helpers reach the factored matrix through the PC's public face
               "VecAXPBYPCZ", "VecAXPY", "VecAYPX", "VecCopy", "VecDestroyVecs", "VecDot", "VecDotNorm2", "VecDuplicate", "VecDuplicateVecs",
               "VecMAXPY", "VecMDot", "VecNorm", "VecNormalize", "VecSet", "VecTDot", "VecWAXPY"}
    assert wanted <= allowed, sorted(wanted - allowed)


def test_no_cpu_fallback(built):
    """Without a GPU every compute entry point must fail with PETSC_ERR_LIB, not silently compute."""
    k = built.load_kernels()
    n = C.c_int()
    k.mi355x_device_count(C.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is present")
    from petsc_dev_amd import petsc as P
    v = P.Vec.create(10, comm=P.lib().COMM_SELF)
    with pytest.raises(P.PetscError) as e:
        P.lib().VecSet(v.h, 1.0)
    assert e.value.code == 76 and "no gfx950 device" in str(e.value)


def lap2d(m, n):
    import scipy.sparse as sp
    N = m * n
    I = np.arange(N); i = I // n; j = I - i * n
    rows, cols, vals = [I], [I], [4.0 * np.ones(N)]
    for mask, off in ((i > 0, -n), (i < m - 1, n), (j > 0, -1), (j < n - 1, 1)):
        r = I[mask]; rows.append(r); cols.append(r + off); vals.append(-np.ones(r.size))
    A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(N, N))
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)


def seq_arrays(P, A):
    m = C.c_int(); pi_, pj, pa = C.c_void_p(), C.c_void_p(), C.c_void_p()
    P.lib().MatSeqAIJGetArrays(A, C.byref(m), C.byref(pi_), C.byref(pj), C.byref(pa))
    m = m.value
    ai = np.ctypeslib.as_array(C.cast(pi_, C.POINTER(C.c_int)), (m + 1,)).copy()
    nz = int(ai[-1])
    aj = np.ctypeslib.as_array(C.cast(pj, C.POINTER(C.c_int)), (max(nz, 1),))[:nz].copy()
    aa = np.ctypeslib.as_array(C.cast(pa, C.POINTER(C.c_double)), (max(nz, 1),))[:nz].copy()
    return ai, aj, aa


def test_matsetvalues_assembly_matches_csr(built):
    """ex2.c:96-103 style assembly through MatSetValues (unsorted insertion order, preallocation 2 then growth)"""
    from petsc_dev_amd import petsc as P
    L = P.lib()
    m, n = 9, 7
    ai, aj, aa = lap2d(m, n)
    A = C.c_void_p()
    L.MatCreate(L.COMM_SELF, C.byref(A))
    L.MatSetSizes(A, m * n, m * n, m * n, m * n)
    L.MatSetType(A, b"seqaijhipmi355x")
    L.MatSeqAIJSetPreallocation(A, 2, None)   # too small on purpose: rows must grow

    def one(v, t):
        return (t * 1)(v)
    for Ii in range(m * n):
        i, j = divmod(Ii, n)
        for cond, J in ((i > 0, Ii - n), (i < m - 1, Ii + n), (j > 0, Ii - 1), (j < n - 1, Ii + 1)):
            if cond:
                L.MatSetValues(A, 1, one(Ii, C.c_int), 1, one(J, C.c_int), one(-1.0, C.c_double), P.INSERT_VALUES)
        L.MatSetValues(A, 1, one(Ii, C.c_int), 1, one(Ii, C.c_int), one(3.0, C.c_double), P.INSERT_VALUES)
        L.MatSetValues(A, 1, one(Ii, C.c_int), 1, one(Ii, C.c_int), one(1.0, C.c_double), P.ADD_VALUES)
    L.MatAssemblyBegin(A, P.MAT_FINAL_ASSEMBLY)
    L.MatAssemblyEnd(A, P.MAT_FINAL_ASSEMBLY)
    gi, gj, ga = seq_arrays(P, A)
    assert np.array_equal(gi, ai) and np.array_equal(gj, aj) and np.array_equal(ga, aa)
    L.MatDestroy(C.byref(A))


def scatter_lists(P, ctx):
    L = P.lib()
    nr, ns, nl = C.c_int(), C.c_int(), C.c_int()
    p = [C.c_void_p() for _ in range(8)]
    L.VecScatterGetLists(ctx, C.byref(nr), C.byref(p[0]), C.byref(p[1]), C.byref(p[2]), C.byref(ns), C.byref(p[3]),
                         C.byref(p[4]), C.byref(p[5]), C.byref(nl), C.byref(p[6]), C.byref(p[7]))

    def arr(ptr, n):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int)), (max(n, 1),))[:n].copy()
    nr, ns, nl = nr.value, ns.value, nl.value
    rstarts = arr(p[1], nr + 1); sstarts = arr(p[4], ns + 1)
    return dict(rprocs=arr(p[0], nr), rstarts=rstarts, rindices=arr(p[2], int(rstarts[nr])),
                sprocs=arr(p[3], ns), sstarts=sstarts, sindices=arr(p[5], int(sstarts[ns])),
                lto=arr(p[6], nl), lfrom=arr(p[7], nl))


def mpiaij_pieces(P, A):
    L = P.lib()
    Ad, Ao, g = C.c_void_p(), C.c_void_p(), C.c_void_p()
    L.MatMPIAIJGetSeqAIJ(A, C.byref(Ad), C.byref(Ao), C.byref(g))
    ctx, lvec, ec = C.c_void_p(), C.c_void_p(), C.c_int()
    L.MatMPIAIJGetScatter(A, C.byref(ctx), C.byref(lvec), C.byref(ec))
    garray = np.ctypeslib.as_array(C.cast(g, C.POINTER(C.c_int)), (max(ec.value, 1),))[:ec.value].copy()
    return seq_arrays(P, Ad), seq_arrays(P, Ao), garray, scatter_lists(P, ctx)


def build_local(P, comm, ai, aj, aa, rs, re_, N, via_setvalues=False):
    L = P.lib()
    li = (ai[rs:re_ + 1] - ai[rs]).astype(np.int32)
    lj = aj[ai[rs]:ai[re_]].copy(); la = aa[ai[rs]:ai[re_]].copy()
    if not via_setvalues:
        return P.Mat.from_csr_mpi(li, lj, la, re_ - rs, N, N, comm=comm)
    A = P.Mat()
    L.MatCreate(comm, C.byref(A.h))
    L.MatSetSizes(A.h, re_ - rs, re_ - rs, N, N)
    L.MatSetType(A.h, b"aijhipmi355x" if True else b"")
    L.MatMPIAIJSetPreallocation(A.h, 3, None, 1, None)
    L.MatSeqAIJSetPreallocation(A.h, 3, None)
    for r in range(rs, re_):
        cols = aj[ai[r]:ai[r + 1]][::-1].copy()   # reversed insertion order
        vals = aa[ai[r]:ai[r + 1]][::-1].copy()
        L.MatSetValues(A.h, 1, (C.c_int * 1)(r), cols.size, cols.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p), P.INSERT_VALUES)
    L.MatAssemblyBegin(A.h, P.MAT_FINAL_ASSEMBLY)
    L.MatAssemblyEnd(A.h, P.MAT_FINAL_ASSEMBLY)
    return A


def check_against_oracle(size, got, ai, aj, aa, ranges):
    ref = [orc.mpiaij_split(int(ranges[r]), int(ranges[r + 1]), int(ranges[r]), int(ranges[r + 1]), ai, aj, aa) for r in range(size)]
    garrays = [r["garray"] for r in ref]
    for r in range(size):
        (di, dj, da), (oi, oj, oa), garray, lists = got[r]
        assert np.array_equal(di, ref[r]["ad_i"]) and np.array_equal(dj, ref[r]["ad_j"]) and np.array_equal(da, ref[r]["ad_a"])
        assert np.array_equal(oi, ref[r]["bo_i"]) and np.array_equal(oj, ref[r]["bo_j"]) and np.array_equal(oa, ref[r]["bo_a"])
        assert np.array_equal(garray, ref[r]["garray"])
        sc = orc.scatter_create(size, r, ranges, garrays)
        for key in sc:
            assert np.array_equal(lists[key], sc[key]), (r, key, lists[key], sc[key])


def check_mpiaij_setup(P, size, ai, aj, aa, ranges, via_setvalues=False):
    """every rank builds its MPIAIJ piece; all integer outputs must equal the oracle's, bit for bit"""
    from fakempi import FakeWorld
    N = ai.size - 1

    def work(rank, comm):
        A = build_local(P, comm, ai, aj, aa, int(ranges[rank]), int(ranges[rank + 1]), N, via_setvalues)
        out = mpiaij_pieces(P, A.h)
        A.destroy()
        return out

    got = FakeWorld(size).run(work)
    check_against_oracle(size, got, ai, aj, aa, ranges)
    return got


@pytest.mark.parametrize("size", [1, 2, 3, 8])
def test_mpiaij_setup_p7_slabs(built, size):
    """3-D 7-pt operator in z-slabs (SURVEY 8d config 3 shape): <= 2 neighbours, contiguous planes"""
    from petsc_dev_amd import petsc as P
    nx, ny, nz = 6, 5, 16
    ai, aj, aa = orc.gen_p7(nx, ny, nz)
    ranges = np.array([nx * ny * ((nz * r) // size) for r in range(size + 1)], dtype=np.int32)
    got = check_mpiaij_setup(P, size, ai, aj, aa, ranges)
    if size > 1:
        lists = got[1][3]
        assert lists["rprocs"].tolist() == ([0, 2] if size > 2 else [0])
        # halo of a rank: one plane to and from each z-neighbour
        assert all(c == nx * ny for c in np.diff(lists["sstarts"]))


@pytest.mark.parametrize("size", [2, 3, 5])
def test_mpiaij_setup_irregular(built, size):
    """random pattern, uneven PETSC_DECIDE-like ownership, via MatSetValues with reversed insertion order"""
    from petsc_dev_amd import petsc as P
    import scipy.sparse as sp
    N = 97
    A = sp.random(N, N, density=0.08, random_state=7, format="csr") + sp.eye(N, format="csr")
    A = sp.csr_matrix(A); A.sort_indices()
    ai, aj, aa = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    ranges = np.array([0] + list(np.cumsum([N // size + (N % size > r) for r in range(size)])), dtype=np.int32)
    check_mpiaij_setup(P, size, ai, aj, aa, ranges, via_setvalues=True)


@pytest.mark.parametrize("size", [2, 3])
def test_new_off_diagonal_columns_after_assembly_disassemble_and_reassemble(built, size):
    """MatDisAssemble_MPIAIJ (mmaij.c:170): an assembled MPIAIJ matrix receives entries in off-diagonal columns it does not have yet --
    on ONE rank only, some of them set by another rank (stashed).  The off-diagonal block goes back to global column numbers, the next
    final assembly rebuilds garray, the compacted block, the work vector and the scatter on EVERY rank (mpiaij.c:694-703): all
    integer pieces equal the oracle's split of the modified global matrix, values too."""
    from petsc_dev_amd import petsc as P
    from fakempi import FakeWorld
    import scipy.sparse as sp
    L = P.lib()
    N = 60
    A0 = sp.diags([np.full(N - 1, -1.0), np.full(N, 4.0), np.full(N - 1, -1.0)], [-1, 0, 1], format="csr")
    ai, aj, aa = A0.indptr.astype(np.int32), A0.indices.astype(np.int32), A0.data.astype(np.float64)
    ranges = np.array([0] + list(np.cumsum([N // size + (N % size > r) for r in range(size)])), dtype=np.int32)
    # new entries: rows of rank 0, columns owned by the LAST rank (not in rank 0's garray); one of them set by rank 1 (off-process)
    new = [(1, N - 2, 0.5, 0), (3, N - 5, 0.25, 0), (2, N - 3, -0.75, 1), (1, N - 2, 0.125, 0)]     # (row, col, value, setting rank); the last repeats a position (ADD)
    dense = A0.toarray()
    for r, c, v, _ in new:
        dense[r, c] += v
    A1 = sp.csr_matrix(dense); A1.sort_indices()
    bi, bj, ba = A1.indptr.astype(np.int32), A1.indices.astype(np.int32), A1.data.astype(np.float64)

    def work(rank, comm):
        A = build_local(P, comm, ai, aj, aa, int(ranges[rank]), int(ranges[rank + 1]), N)
        before = mpiaij_pieces(P, A.h)
        for r, c, v, who in new:
            if who == rank:
                L.MatSetValues(A.h, 1, (C.c_int * 1)(r), 1, (C.c_int * 1)(c), (C.c_double * 1)(v), P.ADD_VALUES)
        L.MatAssemblyBegin(A.h, P.MAT_FINAL_ASSEMBLY)
        L.MatAssemblyEnd(A.h, P.MAT_FINAL_ASSEMBLY)
        out = mpiaij_pieces(P, A.h)
        # and once more without anything new: nothing is rebuilt, nothing changes
        L.MatAssemblyBegin(A.h, P.MAT_FINAL_ASSEMBLY)
        L.MatAssemblyEnd(A.h, P.MAT_FINAL_ASSEMBLY)
        again = mpiaij_pieces(P, A.h)
        assert np.array_equal(again[2], out[2]) and np.array_equal(again[1][1], out[1][1])
        A.destroy()
        return before, out

    res = FakeWorld(size).run(work)
    check_against_oracle(size, [r[0] for r in res], ai, aj, aa, ranges)
    check_against_oracle(size, [r[1] for r in res], bi, bj, ba, ranges)
    assert res[0][1][2].size == res[0][0][2].size + 3          # rank 0's garray grew by the three new columns


@pytest.mark.parametrize("size", [2, 3, 4])
def test_off_process_setvalues_are_stashed_and_assembled(built, size):
    """MatSetValues / VecSetValues into rows owned by OTHER ranks (mpiaij.c:552-558, matstash.c; pdvec.c): a 1-D chain of
    2-node elements dealt round-robin to the ranks, so most contributions are off-process; after assembly every rank's
    diagonal / off-diagonal blocks, garray and scatter lists equal the oracle's split of the globally summed matrix, the
    load vector equals the sum too (both orders of contributions are rank order: compared exactly with that order), and mixing
    INSERT on one rank with ADD on another is the reference's error."""
    from petsc_dev_amd import petsc as P
    from fakempi import FakeWorld
    L = P.lib()
    N = 41
    ne = N - 1
    ke = np.array([[2.0, -1.0], [-1.0, 2.0]])
    rng = np.random.default_rng(3)
    scale = 1.0 + rng.random(ne)
    fe = rng.standard_normal((ne, 2))
    ranges = np.array([0] + list(np.cumsum([N // size + (N % size > r) for r in range(size)])), dtype=np.int32)
    owner_of_elem = [e % size for e in range(ne)]

    # expected sums, in the order the entries reach a row: the owner's own contributions as they are set, then the stashed
    # ones rank after rank, each rank's in its element order
    dense = np.zeros((N, N)); load = np.zeros(N)
    for row_owner in range(size):
        lo, hi = int(ranges[row_owner]), int(ranges[row_owner + 1])
        # local first (set order), then every rank's stash in rank order (the owner's own stash is empty for its rows)
        order = [row_owner] + [r for r in range(size) if r != row_owner]
        for src in order:
            for e in range(ne):
                if owner_of_elem[e] != src:
                    continue
                for a in range(2):
                    if lo <= e + a < hi:
                        load[e + a] += fe[e, a]
                        for b in range(2):
                            dense[e + a, e + b] += scale[e] * ke[a, b]
    import scipy.sparse as sp
    S = sp.csr_matrix(dense); S.sort_indices()
    ai, aj, aa = S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data.copy()

    def work(rank, comm):
        rs, re_ = int(ranges[rank]), int(ranges[rank + 1])
        A = P.Mat()
        L.MatCreate(comm, C.byref(A.h)); L.MatSetSizes(A.h, re_ - rs, re_ - rs, N, N); L.MatSetType(A.h, b"aijhipmi355x")
        L.MatMPIAIJSetPreallocation(A.h, 3, None, 2, None)
        v = P.Vec.create(re_ - rs, N=N, comm=comm)
        for e in range(ne):
            if owner_of_elem[e] != rank:
                continue
            idx = np.array([e, e + 1], dtype=np.int32)
            vals = (scale[e] * ke).ravel().copy()
            L.MatSetValues(A.h, 2, idx.ctypes.data_as(C.c_void_p), 2, idx.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p), P.ADD_VALUES)
            L.VecSetValues(v.h, 2, idx.ctypes.data_as(C.c_void_p), fe[e].copy().ctypes.data_as(C.c_void_p), P.ADD_VALUES)
        L.MatAssemblyBegin(A.h, P.MAT_FINAL_ASSEMBLY); L.MatAssemblyEnd(A.h, P.MAT_FINAL_ASSEMBLY)
        L.VecAssemblyBegin(v.h); L.VecAssemblyEnd(v.h)
        out = mpiaij_pieces(P, A.h), v.array().copy()
        A.destroy()
        return out

    got = FakeWorld(size).run(work)
    check_against_oracle(size, [g[0] for g in got], ai, aj, aa, ranges)
    assert np.array_equal(np.concatenate([g[1] for g in got]), load)

    def mixed(rank, comm):
        rs, re_ = int(ranges[rank]), int(ranges[rank + 1])
        v = P.Vec.create(re_ - rs, N=N, comm=comm)
        other = np.array([int(ranges[(rank + 1) % size])], dtype=np.int32)          # a row of the next rank
        one = np.ones(1)
        L.VecSetValues(v.h, 1, other.ctypes.data_as(C.c_void_p), one.ctypes.data_as(C.c_void_p), P.INSERT_VALUES if rank == 0 else P.ADD_VALUES)
        try:
            L.VecAssemblyBegin(v.h)
        except P.PetscError as e:
            return e.code
        return 0

    assert set(FakeWorld(size).run(mixed)) == {73}         # PETSC_ERR_ARG_WRONGSTATE on every rank


def test_ex5_np3_layout(built):
    """src/mat/examples/tests/ex5.c on 3 ranks (golden ex5_23.out): 8x8 dense rows, PETSC_DECIDE split 3/3/2"""
    from petsc_dev_amd import petsc as P
    m = 8
    dense = np.array([[10 * (i + 1) + j + 1 for j in range(m)] for i in range(m)], dtype=np.float64)
    ai = np.arange(0, m * m + 1, m, dtype=np.int32); aj = np.tile(np.arange(m, dtype=np.int32), m); aa = dense.ravel().copy()
    ranges = np.array([0, 3, 6, 8], dtype=np.int32)
    got = check_mpiaij_setup(P, 3, ai, aj, aa, ranges)
    assert got[0][2].tolist() == [3, 4, 5, 6, 7]


def test_options_and_types(built):
    from petsc_dev_amd import petsc as P
    L = P.lib()
    L.PetscOptionsClear()
    L.PetscOptionsInsertString(b"-ksp_type cg -pc_type jacobi -ksp_rtol 1e-7 -ksp_max_it 33 -mat_type aij -vec_type standard")
    v = C.c_void_p()
    L.VecCreate(L.COMM_SELF, C.byref(v)); L.VecSetSizes(v, 5, 5); L.VecSetFromOptions(v)
    t = C.c_char_p(); L.VecGetType(v, C.byref(t)); assert t.value == b"seqhipmi355x"
    A = C.c_void_p()
    L.MatCreate(L.COMM_SELF, C.byref(A)); L.MatSetSizes(A, 5, 5, 5, 5); L.MatSetFromOptions(A)
    L.MatGetType(A, C.byref(t)); assert t.value == b"seqaijhipmi355x"
    with pytest.raises(P.PetscError) as e:
        L.MatSetType(A, b"nosuchtype")
    assert e.value.code == 86
    with pytest.raises(P.PetscError) as e:
        L.VecSetType(v, b"nosuchtype")
    assert e.value.code == 86
    L.PetscOptionsClear()
    L.VecDestroy(C.byref(v)); L.MatDestroy(C.byref(A))


def test_matload_reference_datafiles(built, tmp_path):
    """PETSc binary format.  The reference's own 12x12 data files
    (share/petsc/datafiles/matrices/{spd,ns}-real-int32-float64: a Mat followed by Vecs) load into the AIJ type
    with exactly the arrays a direct big-endian parse gives; MatView/VecView write the same bytes back."""
    from petsc_dev_amd import petsc as P
    L = P.lib()
    for name in ("spd-real-int32-float64", "ns-real-int32-float64"):
        path = os.path.join(ROOT, "tests", "golden", "matrices", name)
        raw = open(path, "rb").read()
        hdr = np.frombuffer(raw[:16], dtype=">i4")
        assert hdr[0] == 1211216
        M, N, nz = int(hdr[1]), int(hdr[2]), int(hdr[3])
        rl = np.frombuffer(raw[16:16 + 4 * M], dtype=">i4").astype(np.int32)
        cols = np.frombuffer(raw[16 + 4 * M:16 + 4 * M + 4 * nz], dtype=">i4").astype(np.int32)
        vals = np.frombuffer(raw[16 + 4 * M + 4 * nz:16 + 4 * M + 12 * nz], dtype=">f8").astype(np.float64)
        voff = 16 + 4 * M + 12 * nz
        vh = np.frombuffer(raw[voff:voff + 8], dtype=">i4")
        assert vh[0] == 1211214 and vh[1] == M
        vec = np.frombuffer(raw[voff + 8:voff + 8 + 8 * M], dtype=">f8").astype(np.float64)
        viewer = C.c_void_p()
        L.PetscViewerBinaryOpen(L.COMM_SELF, path.encode(), 0, C.byref(viewer))
        A = C.c_void_p()
        L.MatCreate(L.COMM_SELF, C.byref(A))
        L.MatLoad(A, viewer)
        ai, aj, aa = seq_arrays(P, A)
        assert np.array_equal(np.diff(ai), rl) and np.array_equal(aj, cols) and np.array_equal(aa, vals)
        # writing it back reproduces the matrix part of the file byte for byte
        out = str(tmp_path / (name + ".out"))
        w = C.c_void_p()
        L.PetscViewerBinaryOpen(L.COMM_SELF, out.encode(), 1, C.byref(w))
        L.MatView(A, w)
        L.PetscViewerDestroy(C.byref(w))
        assert open(out, "rb").read() == raw[:voff]
        L.PetscViewerDestroy(C.byref(viewer))
        L.MatDestroy(C.byref(A))
        assert vec.size == M


def test_matload_sorts_unsorted_rows_and_matview_expands_blocks(built, tmp_path):
    """MatLoad_SeqAIJ takes a file whose rows list their columns in any order (aij.c:4093-4157): such rows are sorted, values
    following their columns, instead of being refused.  MatView of a blocked (BAIJ) matrix writes the POINT rows, as
    MatView_SeqBAIJ_Binary does (baij.c:1068-1130) -- loading the file back as AIJ gives the expanded matrix."""
    from petsc_dev_amd import petsc as P
    L = P.lib()
    path = os.path.join(ROOT, "tests", "golden", "matrices", "spd-real-int32-float64")
    raw = bytearray(open(path, "rb").read())
    M, N, nz = (int(v) for v in np.frombuffer(bytes(raw[4:16]), dtype=">i4"))
    rl = np.frombuffer(bytes(raw[16:16 + 4 * M]), dtype=">i4").astype(np.int64)
    c0, v0 = 16 + 4 * M, 16 + 4 * M + 4 * nz
    cols = np.frombuffer(bytes(raw[c0:c0 + 4 * nz]), dtype=">i4").copy()
    vals = np.frombuffer(bytes(raw[v0:v0 + 8 * nz]), dtype=">f8").copy()
    ptr = np.concatenate([[0], np.cumsum(rl)])
    rng = np.random.default_rng(5)
    sc, sv = cols.copy(), vals.copy()
    for r in range(M):                                              # every row in a random order
        perm = rng.permutation(int(rl[r])) + ptr[r]
        sc[ptr[r]:ptr[r + 1]] = cols[perm]; sv[ptr[r]:ptr[r + 1]] = vals[perm]
    shuffled = bytearray(raw)
    shuffled[c0:c0 + 4 * nz] = sc.astype(">i4").tobytes(); shuffled[v0:v0 + 8 * nz] = sv.astype(">f8").tobytes()
    f = str(tmp_path / "shuffled"); open(f, "wb").write(bytes(shuffled))
    viewer = C.c_void_p(); A = C.c_void_p()
    L.PetscViewerBinaryOpen(L.COMM_SELF, f.encode(), 0, C.byref(viewer))
    L.MatCreate(L.COMM_SELF, C.byref(A))
    L.MatLoad(A, viewer)
    ai, aj, aa = seq_arrays(P, A)
    assert np.array_equal(np.diff(ai), rl) and np.array_equal(aj, cols.astype(np.int32)) and np.array_equal(aa, vals.astype(np.float64))
    L.PetscViewerDestroy(C.byref(viewer)); L.MatDestroy(C.byref(A))
    # BAIJ: 3 block rows of 2 x 2 blocks, column-major inside a block
    bs, bi, bj = 2, np.array([0, 2, 3, 5], dtype=np.int32), np.array([0, 2, 1, 0, 2], dtype=np.int32)
    ba = np.arange(1.0, 1.0 + 5 * 4)
    Bm = P.Mat.from_bsr(bs, bi, bj, ba)
    out = str(tmp_path / "baij")
    w = C.c_void_p()
    L.PetscViewerBinaryOpen(L.COMM_SELF, out.encode(), 1, C.byref(w))
    L.MatView(Bm.h, w)
    L.PetscViewerDestroy(C.byref(w))
    dense = np.zeros((6, 6))
    for I in range(3):
        for k in range(bi[I], bi[I + 1]):
            dense[2 * I:2 * I + 2, 2 * bj[k]:2 * bj[k] + 2] = ba[4 * k:4 * k + 4].reshape(2, 2).T      # column-major blocks
    viewer = C.c_void_p(); A = C.c_void_p()
    L.PetscViewerBinaryOpen(L.COMM_SELF, out.encode(), 0, C.byref(viewer))
    L.MatCreate(L.COMM_SELF, C.byref(A))
    L.MatLoad(A, viewer)
    ai, aj, aa = seq_arrays(P, A)
    got = np.zeros((6, 6))
    for r in range(6):
        got[r, aj[ai[r]:ai[r + 1]]] = aa[ai[r]:ai[r + 1]]
    assert ai[-1] == 20 and np.array_equal(got, dense)
    L.PetscViewerDestroy(C.byref(viewer)); L.MatDestroy(C.byref(A))


def test_matload_rejects_inconsistent_files(built, tmp_path):
    """a truncated or inconsistent binary file is an error (the reference: 'Inconsistant matrix data in file',
    aij.c:4125), never a matrix whose column indices would send the SpMV gathers out of bounds"""
    from petsc_dev_amd import petsc as P
    L = P.lib()
    path = os.path.join(ROOT, "tests", "golden", "matrices", "spd-real-int32-float64")
    raw = bytearray(open(path, "rb").read())
    M, N, nz = (int(v) for v in np.frombuffer(bytes(raw[4:16]), dtype=">i4"))
    cols0 = 16 + 4 * M

    def be(v):
        return int(v).to_bytes(4, "big", signed=True)

    def attempt(data, name):
        f = str(tmp_path / name)
        open(f, "wb").write(bytes(data))
        viewer = C.c_void_p(); A = C.c_void_p()
        L.PetscViewerBinaryOpen(L.COMM_SELF, f.encode(), 0, C.byref(viewer))
        L.MatCreate(L.COMM_SELF, C.byref(A))
        try:
            with pytest.raises(P.PetscError) as e:
                L.MatLoad(A, viewer)
        finally:
            L.PetscViewerDestroy(C.byref(viewer)); L.MatDestroy(C.byref(A))
        return e.value.code

    bad = bytearray(raw); bad[12:16] = be(nz + 3)                   # header nz != sum of the row lengths
    assert attempt(bad, "nz") == 66
    bad = bytearray(raw); bad[16:20] = be(-2)                       # negative row length
    assert attempt(bad, "neg") in (66, 79)
    bad = bytearray(raw); bad[cols0:cols0 + 4] = be(N + 5)          # column index out of range
    assert attempt(bad, "col") == 66
    bad = bytearray(raw); bad[cols0 + 4:cols0 + 8] = bad[cols0:cols0 + 4]   # duplicate column in row 0
    assert attempt(bad, "dup") == 66
    assert attempt(raw[:cols0 + 4 * nz + 8], "trunc") == 66         # values cut short
    bad = bytearray(raw); bad[4:8] = be(-1)                         # negative size
    assert attempt(bad, "size") == 66


@pytest.mark.parametrize("size", [2, 3])
def test_matload_parallel(built, size):
    """parallel MatLoad: every rank reads its own rows of the reference data file; pieces equal the oracle's split"""
    from petsc_dev_amd import petsc as P
    from fakempi import FakeWorld
    L = P.lib()
    path = os.path.join(ROOT, "tests", "golden", "matrices", "ns-real-int32-float64")
    raw = open(path, "rb").read()
    M, nz = 12, int(np.frombuffer(raw[12:16], dtype=">i4")[0])
    rl = np.frombuffer(raw[16:16 + 4 * M], dtype=">i4").astype(np.int32)
    ai = np.concatenate(([0], np.cumsum(rl))).astype(np.int32)
    aj = np.frombuffer(raw[16 + 4 * M:16 + 4 * M + 4 * nz], dtype=">i4").astype(np.int32)
    aa = np.frombuffer(raw[16 + 4 * M + 4 * nz:16 + 4 * M + 12 * nz], dtype=">f8").astype(np.float64)
    ranges = np.array([0] + list(np.cumsum([M // size + (M % size > r) for r in range(size)])), dtype=np.int32)

    def work(rank, comm):
        viewer = C.c_void_p()
        L.PetscViewerBinaryOpen(comm, path.encode(), 0, C.byref(viewer))
        A = C.c_void_p()
        L.MatCreate(comm, C.byref(A))
        L.MatLoad(A, viewer)
        out = mpiaij_pieces(P, A)
        L.PetscViewerDestroy(C.byref(viewer))
        L.MatDestroy(C.byref(A))
        return out
    got = FakeWorld(size).run(work)
    check_against_oracle(size, got, ai, aj, aa, ranges)


def test_matrixmarket_converter(built, tmp_path):
    """examples/mm2petsc (the reference's converter is the example ex72.c): MatrixMarket coordinate files --
    real general, real symmetric (lower triangle stored), pattern symmetric -- become PETSc binary files that MatLoad
    reads back as exactly the matrix scipy reads from the same .mtx.  Host-only, no GPU."""
    import subprocess
    import scipy.io
    import scipy.sparse as sp
    from petsc_dev_amd import petsc as P
    L = P.lib()
    exe = os.path.join(ROOT, "examples", "mm2petsc")
    assert os.path.exists(exe), "examples/mm2petsc was not built"
    rng = np.random.default_rng(7)
    G = sp.random(37, 29, density=0.15, random_state=3, format="coo")
    S = sp.random(41, 41, density=0.1, random_state=4, format="csr"); S = sp.coo_matrix(sp.tril(S + S.T + sp.eye(41)))
    cases = {"general.mtx": (G, None), "symmetric.mtx": (S, "symmetric")}
    for name, (m, symm) in cases.items():
        scipy.io.mmwrite(str(tmp_path / name), m, symmetry=symm or "general")
    with open(tmp_path / "pattern.mtx", "w") as f:          # pattern symmetric, with comment lines
        f.write("%%MatrixMarket matrix coordinate pattern symmetric\n% a comment\n%\n5 5 6\n1 1\n2 1\n3 3\n5 2\n5 5\n4 4\n")
    for name in ("general.mtx", "symmetric.mtx", "pattern.mtx"):
        src, dst = str(tmp_path / name), str(tmp_path / (name + ".petsc"))
        r = subprocess.run([exe, "-fin", src, "-fout", dst], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stdout + r.stderr
        ref = sp.csr_matrix(scipy.io.mmread(src)); ref.sort_indices()
        viewer = C.c_void_p(); A = C.c_void_p()
        L.PetscViewerBinaryOpen(L.COMM_SELF, dst.encode(), 0, C.byref(viewer))
        L.MatCreate(L.COMM_SELF, C.byref(A)); L.MatLoad(A, viewer)
        ai, aj, aa = seq_arrays(P, A)
        assert np.array_equal(ai, ref.indptr) and np.array_equal(aj, ref.indices) and np.array_equal(aa, ref.data), name
    r = subprocess.run([exe, "-fin", str(tmp_path / "missing.mtx"), "-fout", "x"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0


def test_mat_duplicate_setfromoptions_and_vec_replacearray_slots(built):
    """Table slots SURVEY 8(b) lists that the types fill besides the products: Mat duplicate (matimpl.h:60; MatDuplicate_SeqAIJ
    aij.c:3964, _SeqBAIJ baij.c:2874), Mat setfromoptions (matimpl.h:110: the type's own options, read under the matrix's prefix),
    Vec replacearray (vecimpl.h:259; VecReplaceArray_Seq dvec2.c:1159).  Host-side behaviour only (no device needed)."""
    import ctypes as C
    from petsc_dev_amd import petsc as P
    L = P.lib()
    ai, aj, aa = P.gen_poisson7(5, 4, 3)
    A = P.Mat.from_csr(ai, aj, aa)

    def arrays(M):
        m, i_, j_, a_ = C.c_int(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        L.MatSeqAIJGetArrays(M, C.byref(m), C.byref(i_), C.byref(j_), C.byref(a_))
        ii = np.ctypeslib.as_array(C.cast(i_, C.POINTER(C.c_int)), (m.value + 1,)).copy()
        jj = np.ctypeslib.as_array(C.cast(j_, C.POINTER(C.c_int)), (ii[-1],)).copy()
        vv = np.ctypeslib.as_array(C.cast(a_, C.POINTER(C.c_double)), (ii[-1],)).copy()
        return ii, jj, vv, a_.value

    for op, expect_vals in ((1, aa), (0, np.zeros_like(aa)), (2, np.zeros_like(aa))):     # MAT_COPY_VALUES, MAT_DO_NOT_COPY_VALUES, MAT_SHARE_NONZERO_PATTERN
        Bh = C.c_void_p()
        L.MatDuplicate(A.h, op, C.byref(Bh))
        B = P.Mat(Bh)
        t = C.c_char_p()
        L.MatGetType(B.h, C.byref(t))
        assert t.value == b"seqaijhipmi355x"
        ii, jj, vv, addr = arrays(B.h)
        assert np.array_equal(ii, ai) and np.array_equal(jj, aj) and np.array_equal(vv, expect_vals)
        assert addr != arrays(A.h)[3]                       # its own storage
    # the type's options, under the matrix's prefix, through MatSetFromOptions (ops->setfromoptions)
    L.PetscOptionsClear()
    L.PetscOptionsInsertString(b"-fine_mat_hipmi355x_value_patterns 0 -mat_hipmi355x_value_patterns 1")
    L.MatSetOptionsPrefix(A.h, b"fine_")
    L.MatSetFromOptions(A.h)
    t = C.c_char_p()
    L.MatGetType(A.h, C.byref(t))
    assert t.value == b"seqaijhipmi355x"                    # no -mat_type: the type it has stays (gcreate.c:188-193)
    L.PetscOptionsClear()
    # Vec replacearray: the vector takes the array over for good
    v = P.Vec.create(6, comm=L.COMM_SELF)
    p = C.c_void_p()
    L.PetscMallocFn(6 * 8, C.byref(p))
    src = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), (6,))
    src[:] = [1.5, -2.0, 3.25, 0.0, 7.0, -1.0]
    L.VecReplaceArray(v.h, p)
    q = C.c_void_p()
    L.VecGetArray(v.h, C.byref(q))
    assert q.value == p.value
    L.VecRestoreArray(v.h, C.byref(q))
    del v                                                   # frees the adopted array with the vector
