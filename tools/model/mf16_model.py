"""Lane-level numpy model of the 16x16x4-tile formulation of the n = 32 chain kernel (csrc/chain_mf16.hpp):
checks the operand tricks (U^T V from two C/D-layout tiles, the 4-pivot sweep with modified operands, the
row permutation of the m-row blocks) against plain linear algebra before they are written in HIP.
Not part of the product or of the tests' oracle; run by hand:  python tools/model/mf16_model.py"""
import numpy as np

LANES = 64
lane = np.arange(LANES)
J, G = lane & 15, lane >> 4


def rowmap(kind):
    """C/D row of (lane group g, register v): f32 4g + v, f64 g + 4v (cdna_hip_programming.md)."""
    return (lambda g, v: 4 * g + v) if kind == "f32" else (lambda g, v: g + 4 * v)


def mfma(a, b, c, row):
    """D = C + A B with A[i = l & 15][k = l >> 4] = a[l], B[k = l >> 4][j = l & 15] = b[l]; C/D reg v of
    lane l: [row(l >> 4, v)][l & 15]."""
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    A[J, G] = a; B[G, J] = b
    D = A @ B
    out = c.copy()
    for v in range(4):
        out[v] += D[row(G, v), J]
    return out


def to_tiles(Mx, row):
    """32 x 32 (or 16 x 32, 32 x 16, 16 x 16) matrix -> tiles[I][Jt][v][lane]."""
    R, C = Mx.shape[0] // 16, Mx.shape[1] // 16
    t = np.zeros((R, C, 4, LANES))
    for I in range(R):
        for Jt in range(C):
            for v in range(4):
                t[I, Jt, v] = Mx[16 * I + row(G, v), 16 * Jt + J]
    return t


def from_tiles(t, row):
    R, C = t.shape[:2]
    Mx = np.zeros((16 * R, 16 * C))
    for I in range(R):
        for Jt in range(C):
            for v in range(4):
                Mx[16 * I + row(G, v), 16 * Jt + J] = t[I, Jt, v]
    return Mx


def prodT(U, V, acc, row, regs=range(4)):
    """acc += U^T V, all in tile form; contraction over the rows of U and V (register index and lane
    group), `regs`: the registers that can be nonzero."""
    out = acc.copy()
    for I in range(U.shape[1]):
        for Jt in range(V.shape[1]):
            for R in range(U.shape[0]):
                for v in regs:
                    out[I, Jt] = mfma(U[R, I, v], V[R, Jt, v], out[I, Jt], row)
    return out


def inv4_ldl(P):
    """inverse of a 4 x 4 SPD block through its LDL^T (pivots d_k = those of an unblocked Cholesky,
    squared); returns (Pinv, pivots)."""
    L = np.eye(4); d = np.zeros(4)
    for k in range(4):
        d[k] = P[k, k] - sum(L[k, s] ** 2 * d[s] for s in range(k))
        for i in range(k + 1, 4):
            L[i, k] = (P[i, k] - sum(L[i, s] * L[k, s] * d[s] for s in range(k))) / d[k]
    Li = np.linalg.inv(L)
    return Li.T @ np.diag(1 / d) @ Li, d


def sweep_all(T, row, tiles=2, panels=None):
    """In-place symmetric sweep of every pivot: returns the tiles of -A^-1 and the list of pivots.
    Panel (I, v) = the rows {16 I + row(g, v), g = 0..3}: row v of every lane group, so both MFMA
    operands are registers as they stand."""
    T = T.copy()
    piv = []
    for I in range(tiles):
        for v in (panels if panels is not None else range(4)):
            # P[a][b] = A[p_a][p_b], p_g = 16 I + row(g, v): register v of lane (j = row(b, v), g = a)
            P = np.zeros((4, 4))
            for a in range(4):
                for b in range(4):
                    P[a, b] = T[I, I, v][16 * a + row(b, v)]
            Pinv, d = inv4_ldl(P)
            piv += list(d)
            diag_lane = (J == row(G, v))       # lane (j, g) holds a diagonal element of tile (I, I) in reg v
            # mix: D' = Pinv (C^T - [I at the pivot columns]); the A operand carries Pinv at rows row(kk, 0)
            # hmm: output row of (lane group kk, reg 0) so that the result is the next B operand as it stands
            a_mix = np.zeros(LANES)
            for kk in range(4):
                for kq in range(4):
                    a_mix[16 * kq + row(kk, 0)] = Pinv[kk, kq]      # A[i = row(kk, 0)][k = kq]
            Bop = []
            for Jt in range(tiles):
                src = T[I, Jt, v].copy()
                if Jt == I:
                    src = src - diag_lane * 1.0
                Dp = mfma(a_mix, src, np.zeros((4, LANES)), row)
                Bop.append(Dp[0])                                    # reg 0 of lane (j, kk) = D'[kk][j]
            Aop = []                                                 # operands are taken BEFORE the update
            for It in range(tiles):
                aop = -T[I, It, v].copy()                            # -C[16 It + i][kk] = -A[p_kk][16 It + i]
                if It == I:
                    aop = aop + diag_lane * 1.0
                Aop.append(aop)
            for It in range(tiles):
                for Jt in range(tiles):
                    T[It, Jt] = mfma(Aop[It], Bop[Jt], T[It, Jt], row)
            T[I, I, v] -= 2.0 * diag_lane
    return T, np.array(piv)


def check_sweep(kind):
    row = rowmap(kind)
    rng = np.random.default_rng(0)
    S = rng.standard_normal((32, 32))
    A = np.eye(32) + 0.1 * S @ S.T
    T = to_tiles(A, row)
    assert np.allclose(from_tiles(T, row), A)
    Tn, piv = sweep_all(T, row)
    got = from_tiles(Tn, row)
    err = np.abs(got + np.linalg.inv(A)).max()
    print(kind, "sweep: max |(-A^-1) error|", err, "min pivot", piv.min())
    assert err < 1e-12
    # an indefinite matrix shows a non-positive pivot
    A2 = A.copy(); A2[7, 7] = -3.0
    _, piv2 = sweep_all(to_tiles(A2, row), row)
    assert piv2.min() <= 0
    # products
    U, V = rng.standard_normal((32, 32)), rng.standard_normal((32, 32))
    acc = prodT(to_tiles(U, row), to_tiles(V, row), np.zeros((2, 2, 4, LANES)), row)
    assert np.allclose(from_tiles(acc, row), U.T @ V)


def check_stage(kind, M=8):
    """One backward stage in tile form against plain formulas."""
    row = rowmap(kind)
    rng = np.random.default_rng(1)
    n = 32
    # control index a  <->  (lane group a % 4, register a // 4): registers 0, 1 hold the 8 controls
    pos = np.array([row(a % 4, a // 4) for a in range(8)])           # tile row / column of control a
    S = rng.standard_normal((n, n)); W = S @ S.T / n
    A = np.eye(n) + 0.05 * rng.standard_normal((n, n)); B = 0.1 * rng.standard_normal((n, M))
    Mx = 0.01 * rng.standard_normal((n, M)); Gm = rng.standard_normal((M, M)); R = Gm.T @ Gm + 1.01 * np.eye(M)
    Q = S.T @ S / n + 1e-3 * np.eye(n)
    # reference
    F = W @ A; Z = W @ B; Gr = R + B.T @ Z; H = Mx.T + B.T @ F
    K = -np.linalg.solve(Gr, H); Vn = Q + A.T @ F + K.T @ H
    # tiles; m-row blocks padded to 16 with the permutation `pos`
    Bp = np.zeros((n, 16)); Bp[:, pos[:M]] = B
    Rp = np.eye(16); Rp[np.ix_(pos[:M], pos[:M])] = R                # identity padding: pivots 1
    Mp = np.zeros((16, n)); Mp[pos[:M], :] = Mx.T
    tW, tA, tB = to_tiles(W, row), to_tiles(A, row), to_tiles(Bp, row)
    tF = prodT(tW, tA, np.zeros((2, 2, 4, LANES)), row)
    tZ = prodT(tW, tB, np.zeros((2, 1, 4, LANES)), row)
    tG = prodT(tB, tZ, to_tiles(Rp, row), row)
    tH = prodT(tB, tF, to_tiles(Mp, row), row)
    assert np.allclose(from_tiles(tH, row)[pos[:M]], H)
    used = sorted({a // 4 for a in range(M)})                        # registers that hold controls
    tGn, piv = sweep_all(tG, row, tiles=1, panels=used)
    Gn = from_tiles(tGn, row)
    assert np.allclose(Gn[np.ix_(pos[:M], pos[:M])], -np.linalg.inv(Gr))
    # K = (-G^-1) H: contraction over the rows of both (registers `used` only)
    tK = prodT(tGn, tH, np.zeros((1, 2, 4, LANES)), row, regs=used)
    assert np.allclose(from_tiles(tK, row)[pos[:M]], K)
    tV = prodT(tA, tF, to_tiles(Q, row), row)
    tV = prodT(tK, tH, tV, row, regs=used)
    err = np.abs(from_tiles(tV, row) - Vn).max()
    print(kind, "stage: V error", err, "G pivots", np.round(piv, 3))
    assert err < 1e-10


if __name__ == "__main__":
    for kind in ("f32", "f64"):
        check_sweep(kind)
        check_stage(kind)
    print("ok")
