"""Independent dense-KKT reference (numpy) for the tree-LQR problem.

TEST INFRASTRUCTURE ONLY.  Assembles the KKT system exactly as the reference's
own dense check does (tests/lqr_test.cpp:859-929) and solves it with
numpy.linalg.solve; also evaluates the KKT residual the reference's tests use
(tests/lqr_test.cpp:152-186, 371-409, 600-639):

  node i :  Q_i x_i - y_i + sum_{e: parent(e)=i} (M_e u_e + A_e^T y_child(e)) + q_i = 0
  edge e :  M_e^T x_p + R_e u_e + B_e^T y_c + r_e = 0
  edge e :  A_e x_p + B_e u_e - x_c + c_c - delta_c o y_c = 0
  root   :  -x_root - delta_root o y_root + c_root = 0
"""
import numpy as np


def _offsets(dims):
    off = np.concatenate([[0], np.cumsum(dims)]).astype(int)
    return off[:-1], int(off[-1])


def assemble(parents, children, state_dims, control_dims, blocks, root=0):
    N, E = len(state_dims), len(control_dims)
    xo, nx = _offsets(state_dims)
    uo, nu = _offsets(control_dims)
    uo = uo + nx
    yo = xo + nx + nu
    total = 2 * nx + nu
    Kmat = np.zeros((total, total))
    rhs = np.zeros(total)
    row = 0
    for node in range(N):
        n = state_dims[node]
        Kmat[row:row + n, xo[node]:xo[node] + n] += blocks["Q"][node]
        Kmat[row:row + n, yo[node]:yo[node] + n] -= np.eye(n)
        for e in range(E):
            if parents[e] == node:
                ch, m = children[e], control_dims[e]
                Kmat[row:row + n, uo[e]:uo[e] + m] += blocks["M"][e]
                Kmat[row:row + n, yo[ch]:yo[ch] + state_dims[ch]] += blocks["A"][e].T
        rhs[row:row + n] = -np.asarray(blocks["q"][node])
        row += n
    for e in range(E):
        p, ch, m = parents[e], children[e], control_dims[e]
        Kmat[row:row + m, xo[p]:xo[p] + state_dims[p]] += blocks["M"][e].T
        Kmat[row:row + m, uo[e]:uo[e] + m] += blocks["R"][e]
        Kmat[row:row + m, yo[ch]:yo[ch] + state_dims[ch]] += blocks["B"][e].T
        rhs[row:row + m] = -np.asarray(blocks["r"][e])
        row += m
    n = state_dims[root]
    Kmat[row:row + n, xo[root]:xo[root] + n] -= np.eye(n)
    Kmat[row:row + n, yo[root]:yo[root] + n] -= np.diag(blocks["delta"][root])
    rhs[row:row + n] = -np.asarray(blocks["c"][root])
    row += n
    for e in range(E):
        p, ch = parents[e], children[e]
        nc = state_dims[ch]
        Kmat[row:row + nc, xo[p]:xo[p] + state_dims[p]] += blocks["A"][e]
        Kmat[row:row + nc, uo[e]:uo[e] + control_dims[e]] += blocks["B"][e]
        Kmat[row:row + nc, xo[ch]:xo[ch] + nc] -= np.eye(nc)
        Kmat[row:row + nc, yo[ch]:yo[ch] + nc] -= np.diag(blocks["delta"][ch])
        rhs[row:row + nc] = -np.asarray(blocks["c"][ch])
        row += nc
    assert row == total
    return Kmat, rhs, (xo, uo, yo)


def solve(parents, children, state_dims, control_dims, blocks, root=0):
    Kmat, rhs, (xo, uo, yo) = assemble(parents, children, state_dims, control_dims, blocks, root)
    z = np.linalg.solve(Kmat, rhs)
    x = [z[xo[i]:xo[i] + d] for i, d in enumerate(state_dims)]
    y = [z[yo[i]:yo[i] + d] for i, d in enumerate(state_dims)]
    u = [z[uo[e]:uo[e] + d] for e, d in enumerate(control_dims)]
    return x, u, y


class OffsetFamilies:
    """The KKT system of one problem, LU-factorised once, solved for modified offsets c[node].

    c[node] enters the right-hand side of one block row only (the initial-state row for the root, the
    dynamics row of the edge into `node` otherwise).  Used to pin the feedback gains: K_e, k_e of an
    edge e do not depend on c[j] of any node j that is not a proper descendant of parent(e) -- the
    backward sweeps (lqr.cpp:645-731, :738-796) reach c[j] only on the way up from j -- so
    u_e = K_e x_parent(e) + k_e (lqr.cpp:856-857) must hold with the SAME K_e, k_e on every solution
    of a family that varies c[parent(e)] or the c of an ancestor of it, and dim + 1 affinely
    independent x_parent(e) determine that affine map completely."""

    def __init__(self, parents, children, state_dims, control_dims, blocks, root=0):
        import scipy.linalg
        self.sd, self.cd = list(state_dims), list(control_dims)
        Kmat, self.rhs, (self.xo, self.uo, self.yo) = assemble(parents, children, state_dims, control_dims,
                                                               blocks, root)
        self.lu = scipy.linalg.lu_factor(Kmat)
        row = sum(state_dims) + sum(control_dims)  # stationarity rows come first (see assemble)
        self.c_row = {root: row}
        row += state_dims[root]
        for e in range(len(control_dims)):
            self.c_row[children[e]] = row
            row += state_dims[children[e]]

    def solve(self, node, c_values):
        """-> list of (x, u, y), one per value of c[node]."""
        import scipy.linalg
        B = np.repeat(self.rhs[:, None], len(c_values), axis=1)
        r = self.c_row[node]
        for j, c in enumerate(c_values):
            B[r:r + self.sd[node], j] = -np.asarray(c, dtype=float)
        Z = scipy.linalg.lu_solve(self.lu, B)
        out = []
        for j in range(len(c_values)):
            z = Z[:, j]
            out.append(([z[self.xo[i]:self.xo[i] + d] for i, d in enumerate(self.sd)],
                        [z[self.uo[e]:self.uo[e] + d] for e, d in enumerate(self.cd)],
                        [z[self.yo[i]:self.yo[i] + d] for i, d in enumerate(self.sd)]))
        return out


def residual_norm(parents, children, state_dims, control_dims, blocks, x, u, y, root=0):
    sq = 0.0
    N, E = len(state_dims), len(control_dims)
    sym = lambda S: np.tril(S) + np.tril(S, -1).T  # selfadjointView<Lower>
    for node in range(N):
        res = sym(np.asarray(blocks["Q"][node])) @ x[node] - y[node] + blocks["q"][node]
        for e in range(E):
            if parents[e] == node:
                res = res + blocks["M"][e] @ u[e] + blocks["A"][e].T @ y[children[e]]
        sq += float(res @ res)
    for e in range(E):
        p, ch = parents[e], children[e]
        su = blocks["M"][e].T @ x[p] + sym(np.asarray(blocks["R"][e])) @ u[e] + \
            blocks["B"][e].T @ y[ch] + blocks["r"][e]
        dy = blocks["A"][e] @ x[p] + blocks["B"][e] @ u[e] - x[ch] + blocks["c"][ch] - \
            np.asarray(blocks["delta"][ch]) * y[ch]
        sq += float(su @ su) + float(dy @ dy)
    rd = -x[root] - np.asarray(blocks["delta"][root]) * y[root] + blocks["c"][root]
    sq += float(rd @ rd)
    return float(np.sqrt(sq))


def chain_blocks_from_packed(n, m, T, mats, vecs):
    """Unpack one problem of the packed chain layout into reference-style blocks."""
    mats = np.asarray(mats, dtype=np.float64)
    vecs = np.asarray(vecs, dtype=np.float64)
    node, edge = n * n + n, n * n + 2 * n * m + m * m
    blocks = {k: [] for k in ("Q", "M", "R", "q", "r", "A", "B", "c", "delta")}
    mo = vo = 0
    for i in range(T + 1):
        blocks["Q"].append(mats[mo:mo + n * n].reshape((n, n), order="F")); mo += n * n
        blocks["delta"].append(mats[mo:mo + n].copy()); mo += n
        blocks["q"].append(vecs[vo:vo + n].copy()); vo += n
        blocks["c"].append(vecs[vo:vo + n].copy()); vo += n
        if i < T:
            blocks["A"].append(mats[mo:mo + n * n].reshape((n, n), order="F")); mo += n * n
            blocks["B"].append(mats[mo:mo + n * m].reshape((n, m), order="F")); mo += n * m
            blocks["M"].append(mats[mo:mo + n * m].reshape((n, m), order="F")); mo += n * m
            blocks["R"].append(mats[mo:mo + m * m].reshape((m, m), order="F")); mo += m * m
            blocks["r"].append(vecs[vo:vo + m].copy()); vo += m
    assert mo == len(mats) and vo == len(vecs)
    return blocks


def chain_sol_from_packed(n, m, T, sol):
    sol = np.asarray(sol, dtype=np.float64)
    x, y, u = [], [], []
    o = 0
    for i in range(T + 1):
        x.append(sol[o:o + n]); o += n
        y.append(sol[o:o + n]); o += n
        if i < T:
            u.append(sol[o:o + m]); o += m
    return x, u, y
