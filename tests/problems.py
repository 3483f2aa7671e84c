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
