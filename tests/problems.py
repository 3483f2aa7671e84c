"""Input builders restating the assembly loops of the reference's example programs (inputs only)."""
import numpy as np
import scipy.sparse as sp


def csr(A):
    A = sp.csr_matrix(A)
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)


def lap2d(m, n):
    """src/ksp/ksp/examples/tutorials/ex2.c:96-103: 5-point stencil, rows Ii = i*n + j, diag 4, -1 at +-n, +-1"""
    N = m * n
    I = np.arange(N); i = I // n; j = I - i * n
    rows, cols, vals = [I], [I], [4.0 * np.ones(N)]
    for mask, off in ((i > 0, -n), (i < m - 1, n), (j > 0, -1), (j < n - 1, 1)):
        r = I[mask]; rows.append(r); cols.append(r + off); vals.append(-np.ones(r.size))
    return csr(sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(N, N)))


def ex32(M=8):
    """src/ksp/ksp/examples/tests/ex32.c with -dof 1 (ComputeMatrix :133-215, the symmetrisation in main :59-63, ComputeRHS :117-129):
    M^3 grid, boundary rows hold only their diagonal, interior rows the 7-point operator scaled by the mesh widths, then
    A <- 0.5 (A + A^T) (so an interior-boundary coupling is halved and mirrored); b = 1/(M-1)^3 everywhere"""
    H = 1.0 / (M - 1)
    d = 2.0 * (H * H / H + H * H / H + H * H / H)
    nb = -(H * H / H)
    N = M ** 3
    idx = lambda i, j, k: i + M * (j + M * k)
    rows, cols, vals = [], [], []
    for k in range(M):
        for j in range(M):
            for i in range(M):
                r = idx(i, j, k)
                rows.append(r); cols.append(r); vals.append(d)
                if not (i in (0, M - 1) or j in (0, M - 1) or k in (0, M - 1)):
                    for (a, b_, c) in ((i - 1, j, k), (i + 1, j, k), (i, j - 1, k), (i, j + 1, k), (i, j, k - 1), (i, j, k + 1)):
                        rows.append(r); cols.append(idx(a, b_, c)); vals.append(nb)
    A = sp.csr_matrix((vals, (rows, cols)), shape=(N, N))
    A = ((A + A.T) * 0.5).tocsr()
    A.sort_indices()
    return csr(A), np.full(N, 1.0 / ((M - 1) ** 3))


def tridiag(n=10):
    """src/ksp/pc/examples/tests/ex2.c:33-47: tridiagonal (-1, 2, -1), n = 10, seqaij"""
    A = sp.diags([-np.ones(n - 1), 2.0 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1])
    return csr(A)


def ex5_tutorial(size=1, second=False, m=3):
    """src/ksp/ksp/examples/tutorials/ex5.c on `size` ranks: n = 2*size, 5-point operator with diagonal 4 (first
    system) or 6 (second system, re-assembled into the same pattern); exact solution u[i] = local index + 100*rank
    over PETSC_DECIDE ownership"""
    n = 2 * size
    ai, aj, aa = lap2d(m, n)
    if second:
        aa = np.where(aa == 4.0, 6.0, aa)
    N = m * n
    u = np.concatenate([np.arange(N // size + (N % size > r), dtype=np.float64) + 100.0 * r for r in range(size)])
    return (ai, aj, aa), u


def ex5_mat(m=8, rect=0, alpha=0.1):
    """src/mat/examples/tests/ex5.c: dense m x n with C[i,j] = 10(i+1)+j+1, then MatScale(alpha) (a[k] = alpha*a[k])"""
    n = m + rect
    dense = np.array([[10.0 * (i + 1) + j + 1 for j in range(n)] for i in range(m)])
    ai = np.arange(0, m * n + 1, n, dtype=np.int32)
    aj = np.tile(np.arange(n, dtype=np.int32), m)
    aa = alpha * dense.ravel()
    return ai, aj, aa, m, n


def ex3_fem(m=5):
    """src/ksp/ksp/examples/tests/ex3.c: Q1 Laplacian on the unit square, Dirichlet rows zeroed with unit diagonal
    (MatZeroRows), boundary value u = y; returns CSR, rhs b, initial guess u0 (non-zero), exact solution"""
    N = (m + 1) * (m + 1)
    h = 1.0 / m
    H = h * h
    Ke = np.array([H / 6.0, -.125 * H, H / 12.0, -.125 * H, -.125 * H, H / 6.0, -.125 * H, H / 12.0,
                   H / 12.0, -.125 * H, H / 6.0, -.125 * H, -.125 * H, H / 12.0, -.125 * H, H / 6.0]).reshape(4, 4)
    K = np.zeros((N, N))
    for e in range(m * m):
        i0 = (m + 1) * (e // m) + (e % m)
        idx = [i0, i0 + 1, i0 + 1 + m + 1, i0 + 1 + m]
        for a in range(4):
            for b_ in range(4):
                K[idx[a], idx[b_]] += Ke[a, b_]
    rows = list(range(m + 1)) + list(range(m + 1, m * (m + 1), m + 1)) + list(range(2 * m + 1, m * (m + 1), m + 1)) + [m * (m + 1) + i for i in range(m + 1)]
    u0 = np.zeros(N); b = np.zeros(N)
    for r in rows:
        y = h * (r // (m + 1))
        u0[r] = y; b[r] = y
        K[r, :] = 0.0
        K[r, r] = 1.0
    ustar = np.array([h * (i // (m + 1)) for i in range(N)])
    mask = K != 0
    return csr(sp.csr_matrix(K * mask)), b, u0, ustar


def ex9_system(which, t, m=3, n=2):
    """src/ksp/ksp/examples/tutorials/ex9.c on one rank: system 1 (diag 4) / system 2 (diag 6+t/2), both with the
    non-symmetric extra -(t+0.5) on (Ii, Ii-n); exact solution u = [0,1,..]; b = C u"""
    N = m * n
    K = np.zeros((N, N))
    for Ii in range(N):
        i, j = divmod(Ii, n)
        if i > 0: K[Ii, Ii - n] += -1.0
        if i < m - 1: K[Ii, Ii + n] += -1.0
        if j > 0: K[Ii, Ii - 1] += -1.0
        if j < n - 1: K[Ii, Ii + 1] += -1.0
        K[Ii, Ii] += 4.0 if which == 1 else 6.0 + t * 0.5
    for Ii in range(N):
        if Ii // n > 0:
            K[Ii, Ii - n] += -1.0 * (t + 0.5)
    u = np.arange(N, dtype=np.float64)
    return csr(sp.csr_matrix(K)), u


def serendipity20_stiffness(E=1.0, nu=0.3):
    """Stiffness of ONE 20-node serendipity brick as src/ksp/ksp/examples/tests/ex10.c forms it (Elastic20Stiff /
    paulsetup20 / paulintegrate20, ex10.c:230-533) -- the example's own input data: nodes on the 3 x 3 x 3 lattice of the
    cube [-1, 3]^3 without face centres and body centre, 3-point Gauss rule with the example's 15-digit abscissae and
    weights, isotropic material E = 1, nu = 0.3 with engineering shear strains (xx yy zz xy yz zx).  Returned on the
    lattice's 81 dofs, entries below 1e-8 of the largest dropped, lower triangle averaged with the upper as the example
    does (the upper one stays as integrated)."""
    lattice = [g for g in range(27) if g not in (4, 10, 12, 13, 14, 16, 22)]                # the 20 nodes, lattice index = i + 3 j + 9 k
    ref = np.array([[g % 3 - 1.0, (g // 3) % 3 - 1.0, g // 9 - 1.0] for g in lattice])     # canonical (r, s, t) of each node
    xyz = 2.0 * ref + 1.0                                                                   # physical coordinates: the cube [-1, 3]^3
    gx = np.array([-0.774596669241483, 0.0, 0.774596669241483])
    gw = np.array([0.555555555555555, 0.888888888888888, 0.555555555555555])
    lam = E / ((1.0 + nu) * (1.0 - 2.0 * nu))
    Cm = np.zeros((6, 6))
    Cm[:3, :3] = lam * nu
    Cm[np.arange(3), np.arange(3)] = lam * (1.0 - nu)
    Cm[np.arange(3, 6), np.arange(3, 6)] = lam * (0.5 - nu)
    K = np.zeros((60, 60))
    for a in range(3):
        for b_ in range(3):
            for c in range(3):
                q = np.array([gx[a], gx[b_], gx[c]])
                dN = np.zeros((3, 20))                                                      # d N_node / d (r, s, t) at the point
                for nd in range(20):
                    o = ref[nd]
                    mid = [d for d in range(3) if o[d] == 0.0]
                    if not mid:                                                             # corner: 1/8 prod(1 + q o)(sum q o - 2)
                        f = 1.0 + q * o
                        tot = float(np.dot(q, o))
                        for d in range(3):
                            e1, e2 = [e for e in range(3) if e != d]
                            dN[d, nd] = 0.125 * o[d] * f[e1] * f[e2] * (tot - 1.0 + q[d] * o[d])
                    else:                                                                   # mid-edge along direction z0: 1/4 (1 - q_z0^2) prod over the other two
                        z0 = mid[0]
                        e1, e2 = [e for e in range(3) if e != z0]
                        f1, f2 = 1.0 + q[e1] * o[e1], 1.0 + q[e2] * o[e2]
                        dN[z0, nd] = -0.5 * q[z0] * f1 * f2
                        dN[e1, nd] = 0.25 * o[e1] * f2 * (1.0 - q[z0] * q[z0])
                        dN[e2, nd] = 0.25 * o[e2] * f1 * (1.0 - q[z0] * q[z0])
                J = dN @ xyz                                                                # J[i, j] = d x_j / d r_i
                G = np.linalg.solve(J, dN)                                                  # physical gradients of the shape functions
                B = np.zeros((6, 60))
                for nd in range(20):
                    gxx, gyy, gzz = G[:, nd]
                    B[0, 3 * nd] = gxx; B[1, 3 * nd + 1] = gyy; B[2, 3 * nd + 2] = gzz
                    B[3, 3 * nd] = gyy; B[3, 3 * nd + 1] = gxx
                    B[4, 3 * nd + 1] = gzz; B[4, 3 * nd + 2] = gyy
                    B[5, 3 * nd] = gzz; B[5, 3 * nd + 2] = gxx
                K += (B.T @ (Cm @ B)) * (np.linalg.det(J) * gw[a] * gw[b_] * gw[c])
    Ke = np.zeros((81, 81))
    dofs = np.array([3 * g + d for g in lattice for d in range(3)])
    Ke[np.ix_(dofs, dofs)] = K
    Ke[np.abs(Ke) < 1.e-8 * np.abs(Ke).max()] = 0.0
    low = np.tril_indices(81, -1)
    Ke[low] = (Ke[low] + Ke.T[low]) / 2.0
    return Ke


def ex10_elasticity():
    """src/ksp/ksp/examples/tests/ex10.c as its makefile runs it (runex10: default -m 3, halved to ONE brick, 81 lattice dofs):
    the brick's stiffness added entry by entry AND transposed (AddElement, ex10.c:183-205: A = Ke + Ke^T on the entries that
    are not zero), the lattice's first plane (27 dofs, clamped) and the dofs no element touches removed (ex10.c:150-166):
    36 rows, 1200 nonzeros.  u = 0, 1, 2, ...; b = A u.  Returns CSR, b, u."""
    Ke = serendipity20_stiffness()
    A = Ke + Ke.T
    mask = (Ke != 0.0) | (Ke.T != 0.0)
    keep = [i for i in range(27, 81) if mask[i].any()]
    A = A[np.ix_(keep, keep)]; mask = mask[np.ix_(keep, keep)]
    ai = np.concatenate(([0], np.cumsum(mask.sum(axis=1)))).astype(np.int32)
    aj = np.concatenate([np.nonzero(mask[i])[0] for i in range(len(keep))]).astype(np.int32)
    aa = np.concatenate([A[i, mask[i]] for i in range(len(keep))]).astype(np.float64)
    u = np.arange(len(keep), dtype=np.float64)
    b = np.array([np.dot(aa[ai[i]:ai[i + 1]], u[aj[ai[i]:ai[i + 1]]]) for i in range(len(keep))])
    return (ai, aj, aa), b, u


def parse_monitor(path):
    """'  3 KSP Residual norm 0.146074 ' lines of -ksp_monitor_short -> list of (it, text) groups split at it == 0"""
    solves, cur = [], None
    for line in open(path):
        s = line.strip()
        if "KSP Residual norm" in s:
            it = int(s.split()[0])
            val = s.split("norm", 1)[1].strip()
            if it == 0:
                cur = []
                solves.append(cur)
            cur.append(val)
    return solves


def check_monitor(hist, golden):
    """every monitored value printed with %g (what -ksp_monitor_short does) must equal the golden text"""
    assert len(hist) == len(golden), (len(hist), len(golden), ["%g" % v for v in hist], golden)
    for v, g in zip(hist, golden):
        if g.startswith("<"):
            assert v < 1.e-11, (v, g)
        else:
            assert "%g" % v == g, ("%g" % v, g, ["%g" % x for x in hist], golden)


def parse_vecview(path):
    """numbers printed by VecView (ASCII_COMMON) after each 'type: ...' line, grouped per vector"""
    vecs, cur = [], None
    for line in open(path):
        s = line.strip()
        if s.startswith("type:") :
            cur = []
            vecs.append(cur)
        elif cur is not None:
            try:
                cur.append(float(s))
            except ValueError:
                cur = None
    return [np.array(v) for v in vecs if v]


def elasticity_like(nx, ny, nz, dof=3, seed=3):
    """A 3-dof-per-node operator stored point-wise (AIJ): 27-point node stencil, dense dof x dof coupling -- the
    shape of src/ksp/ksp/examples/tutorials/ex56.c's matrix (rows of one node share their column pattern, so the
    reference switches to its inode routines: 'found N/3 nodes, limit used is 5').  Returns point CSR and the
    equivalent block CSR (blocks column-major as in baij.h)."""
    rng = np.random.default_rng(seed)
    nn = nx * ny * nz
    bi = [0]; bj = []
    for k in range(nz):
        for j in range(ny):
            for i in range(nx):
                for dk in (-1, 0, 1):
                    for dj in (-1, 0, 1):
                        for di in (-1, 0, 1):
                            ii, jj, kk = i + di, j + dj, k + dk
                            if 0 <= ii < nx and 0 <= jj < ny and 0 <= kk < nz:
                                bj.append(ii + nx * (jj + ny * kk))
                bi.append(len(bj))
    bi = np.array(bi, dtype=np.int32); bj = np.array(bj, dtype=np.int32)
    blocks = rng.standard_normal((bj.size, dof, dof))          # blocks[b][r][c]
    ba = np.ascontiguousarray(blocks.transpose(0, 2, 1)).ravel()   # column-major per block
    ai = [0]; aj = []; aa = []
    for node in range(nn):
        for r in range(dof):
            for b in range(bi[node], bi[node + 1]):
                for c in range(dof):
                    aj.append(bj[b] * dof + c); aa.append(blocks[b, r, c])
            ai.append(len(aj))
    return (np.array(ai, dtype=np.int32), np.array(aj, dtype=np.int32), np.array(aa)), (bi, bj, ba)


def spd_blocks(nx, ny, nz, dof=3, seed=3):
    """A symmetric positive definite dof-per-node operator whose diagonal blocks are dense and badly point-scaled (the case
    point-block Jacobi is for): sym(elasticity_like) + blockdiag(c_node * K), K = I + 0.9 (J - I) (eigenvalues 0.1 .. ),
    c_node large enough for definiteness.  Returns point CSR and block CSR (blocks column-major)."""
    import scipy.sparse as sp
    (ai, aj, aa), (bi, bj, _) = elasticity_like(nx, ny, nz, dof, seed)
    n = ai.size - 1
    M = sp.csr_matrix((aa, aj, ai), shape=(n, n))
    S = sp.csr_matrix(M + M.T)
    r = np.asarray(abs(S).sum(1)).ravel().reshape(-1, dof).max(1)           # per node
    K = np.eye(dof) + 0.9 * (np.ones((dof, dof)) - np.eye(dof))
    S = sp.csr_matrix(S + sp.block_diag([10.0 * (c + 1.0) * K for c in r], format="csr"))
    S.sort_indices()
    B = sp.bsr_matrix(S, blocksize=(dof, dof)); B.sort_indices()
    ba = np.ascontiguousarray(B.data.transpose(0, 2, 1)).ravel()
    return (S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data.copy()), (B.indptr.astype(np.int32), B.indices.astype(np.int32), ba)


def gen_irr(n=1564794, mean=73.0, seed=12345):
    """SURVEY 8(d) config 4 stand-in 'IRR': log-normal row lengths clipped to [3,400], columns banded +-50 000 with
    20 % uniform long-range entries, diagonally dominant values.  (A random pattern: no FEM matrix looks like this;
    gen_fem3 below is the structured stand-in.)"""
    rng = np.random.default_rng(seed)
    lens = np.clip(np.exp(rng.normal(np.log(mean) - 0.18, 0.6, n)), 3, 400).astype(np.int64)
    tot = int(lens.sum())
    rows = np.repeat(np.arange(n, dtype=np.int64), lens)
    band = min(50000, max(1, n // 4))
    off = rng.integers(-band, band + 1, tot)
    far = rng.random(tot) < 0.2
    cols = np.where(far, rng.integers(0, n, tot), np.clip(rows + off, 0, n - 1))
    first = np.concatenate(([0], np.cumsum(lens)[:-1]))
    cols[first] = rows[first]                       # make sure the diagonal is present
    key = np.unique(rows * n + cols)                # sorts by (row, col) and removes duplicates
    rows = key // n
    cols = (key - rows * n).astype(np.int32)
    ai = np.zeros(n + 1, dtype=np.int64)
    np.add.at(ai, rows + 1, 1)
    ai = np.cumsum(ai).astype(np.int32)
    aa = -rng.random(cols.size)
    rsum = np.zeros(n)
    np.add.at(rsum, rows, -aa)
    diag = cols == rows.astype(np.int32)
    aa[diag] = rsum[rows[diag]] + 1.0               # strictly diagonally dominant
    return ai, cols, aa


def fem3_blocks(ex=130, ey=130, ez=67, seed=7, rcm=True):
    """Node graph of an unstructured-looking hexahedral mesh: an ex x ey x ez element box with a central bore and
    eight bolt holes cut out (the shape of SuiteSparse Flan_1565, a steel flange of hexahedral elements), nodes
    renumbered by reverse Cuthill-McKee.  Returns the block CSR pattern (bi, bj): node a is coupled to node b when
    they share an element (27 neighbours in the interior, fewer on the surfaces and around the holes)."""
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    cx, cy = (np.indices((ex, ey)) + 0.5)
    cx = cx / ex - 0.5; cy = cy / ey - 0.5
    rad = np.hypot(cx, cy)
    keep2 = (rad > 0.18) & (rad < 0.5)               # annulus: bore in the middle, round outer rim
    for t in range(8):
        ang = 2 * np.pi * (t + 0.5) / 8
        keep2 &= np.hypot(cx - 0.36 * np.cos(ang), cy - 0.36 * np.sin(ang)) > 0.05
    keep = np.repeat(keep2[:, :, None], ez, axis=2)
    keep[:, :, ez // 2:] &= (rad < 0.32)[:, :, None]   # the hub is taller than the flange plate
    e_i, e_j, e_k = np.nonzero(keep)
    nxn, nyn = ex + 1, ey + 1
    corner = []
    for dk in (0, 1):
        for dj in (0, 1):
            for di in (0, 1):
                corner.append((e_i + di) + nxn * ((e_j + dj) + nyn * (e_k + dk)))
    corner = np.stack(corner, 1)                     # elements x 8 grid-node ids
    used, inv = np.unique(corner.ravel(), return_inverse=True)
    nn = used.size
    ne = corner.shape[0]
    E = sp.csr_matrix((np.ones(ne * 8, dtype=np.int8), (np.repeat(np.arange(ne), 8), inv)), shape=(ne, nn))
    G = (E.T @ E).tocsr()
    G.data[:] = 1
    if rcm:
        perm = reverse_cuthill_mckee(G, symmetric_mode=True)
        G = G[perm][:, perm].tocsr()
    G.sort_indices()
    return G.indptr.astype(np.int32), G.indices.astype(np.int32)


def expand_blocks(bi, bj, blocks):
    """Point CSR (rows of a node interleaved: row 3*node + r) of a block matrix; blocks[b][r][c]."""
    dof = blocks.shape[1]
    nb = bi.size - 1
    cnt = np.diff(bi).astype(np.int64)
    ai = np.concatenate(([0], np.cumsum(np.repeat(cnt * dof, dof))))
    assert ai[-1] < 2 ** 31
    blk_row = np.repeat(np.arange(nb, dtype=np.int64), cnt)
    pos = np.arange(bj.size, dtype=np.int64) - np.repeat(bi[:-1].astype(np.int64), cnt)
    aj = np.empty(bj.size * dof * dof, dtype=np.int32)
    aa = np.empty(bj.size * dof * dof)
    for r in range(dof):
        base = ai[blk_row * dof + r] + pos * dof
        for c in range(dof):
            aj[base + c] = bj * dof + c
            aa[base + c] = blocks[:, r, c]
    return ai.astype(np.int32), aj, aa


def gen_fem3(ex=130, ey=130, ez=67, dof=3, seed=7, rcm=True):
    """Unstructured-FEM-like stand-in for BASELINE configs[3] (Flan_1565: 1 564 794 rows, 3 dof per node, hexahedra):
    the fem3_blocks mesh with a dense dof x dof coupling per node pair, stored point-wise (AIJ).  Symmetric pattern,
    strictly diagonally dominant values.  Defaults give ~1.57 M rows, ~76 nonzeros per row."""
    bi, bj = fem3_blocks(ex, ey, ez, seed, rcm)
    rng = np.random.default_rng(seed)
    blocks = -rng.random((bj.size, dof, dof))
    ai, aj, aa = expand_blocks(bi, bj, blocks)
    rows = np.repeat(np.arange(ai.size - 1, dtype=np.int64), np.diff(ai))
    rsum = np.zeros(ai.size - 1)
    np.add.at(rsum, rows, -aa)
    diag = aj == rows.astype(np.int32)
    aa[diag] = rsum[rows[diag]] + 1.0
    return ai, aj, aa
