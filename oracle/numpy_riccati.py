"""Independent numpy Riccati for uniform chains (TEST INFRASTRUCTURE ONLY).

A second, differently-formulated derivation of the gains (K, k) and cost-to-go
(V, v) used to pin the oracle's intermediates: W = V (I + Delta V)^{-1} is formed
with numpy.linalg.solve (no Cholesky, no D^{1/2} scaling), i.e. the identity
noted in SURVEY.md 8(a) row a7, then the textbook recursion
  G = R + B^T W B,  H = M^T + B^T W A,  K = -G^{-1} H,  V = Q + A^T W A + K^T H
  g = v_c + W (c_c - delta_c o v_c),  h = r + B^T g,  k = -G^{-1} h,
  v = q + A^T g + K^T h.
"""
import numpy as np


def chain_gains(blocks, n, m, T):
    V = np.array(blocks["Q"][T], dtype=float)
    v = np.array(blocks["q"][T], dtype=float)
    Ks, ks, Vs, vs = [None] * T, [None] * T, [None] * (T + 1), [None] * (T + 1)
    Vs[T], vs[T] = V.copy(), v.copy()
    for i in range(T - 1, -1, -1):
        A, B, M, R = (np.asarray(blocks[k][i], dtype=float) for k in ("A", "B", "M", "R"))
        dc = np.asarray(blocks["delta"][i + 1], dtype=float)
        cc = np.asarray(blocks["c"][i + 1], dtype=float)
        W = np.linalg.solve(np.eye(n) + V * dc[None, :], V)  # (I + V Delta)^-1 V = V (I + Delta V)^-1
        g = v + W @ (cc - dc * v)
        G = R + B.T @ W @ B
        H = M.T + B.T @ W @ A
        h = np.asarray(blocks["r"][i], dtype=float) + B.T @ g
        K = -np.linalg.solve(G, H)
        k = -np.linalg.solve(G, h)
        V = np.asarray(blocks["Q"][i], dtype=float) + A.T @ W @ A + K.T @ H
        v = np.asarray(blocks["q"][i], dtype=float) + A.T @ g + K.T @ h
        Ks[i], ks[i], Vs[i], vs[i] = K, k, V.copy(), v.copy()
    return Ks, ks, Vs, vs
