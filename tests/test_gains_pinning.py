"""The feedback gains K, k (LQR::Workspace::{K,k}, lqr.hpp:112,118) tied to reference-pinned
quantities.

The reference holds no golden K, k; its tests pin x, u, y only (KKT residual < 1e-12, dense KKT
agreement 1e-10: tests/lqr_test.cpp:260,426,655,995-1010).  The rollout line u = k + K x
(lqr.cpp:856-857) links the two: K_e and k_e of an edge do not depend on the offset c[j] of the
parent node of e or of any of its ancestors (the backward sweeps, lqr.cpp:645-731 and :738-796,
reach c[j] only on the way up from j), so on the solutions of the SAME problem for n + 1 affinely
independent values of such a c[j] -- computed by an independent dense KKT solve, the construction
of the reference's own dense check (tests/lqr_test.cpp:859-929) --
    u_e = K_e x_parent(e) + k_e
must hold with one and the same (K_e, k_e), and n + 1 affinely independent parent states determine
the affine map, i.e. K_e and k_e, completely.  Long horizons contract the spread of the states, so
a chain is cut into segments: stage i is checked on the family that varies c of the nearest
segment start s <= i (checked: the parent states of every stage keep a singular-value ratio
> 1e-6).

Tolerance: |u - K x - k| <= 1e-9 * max|u| (fp64), the tolerance of every other fp64 parity test.
The HIP kernels get the same check in tests/test_gpu_gains_pinning.py.
"""
import glob
import os

import numpy as np
import pytest

import reference_problems as rp
from oracle import dense_kkt

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CHAINS = sorted(glob.glob(os.path.join(GOLD, "chain_*.npz")))
SEGMENT = 4  # stages per family of a chain


def offset_family(c0, seed=0):
    """n + 1 affinely independent values of an offset: the problem's own, then n perturbed ones."""
    n = len(c0)
    rng = np.random.default_rng(seed)
    out = [np.asarray(c0, dtype=float)]
    for j in range(n):
        d = 0.5 * rng.standard_normal(n)
        d[j] += 2.0
        out.append(c0 + d)
    return out


def edge_defect(sols, parent, e, K, k):
    """max over the solutions of |u_e - K x_parent - k|, max |u_e|, and the singular-value ratio of
    the centred parent states (affine independence)."""
    X = np.stack([s[0][parent] for s in sols])  # [n+1, n]
    U = np.stack([s[1][e] for s in sols])       # [n+1, m]
    margin = np.inf
    if X.shape[1] > 0:
        sv = np.linalg.svd(X[1:] - X[:1], compute_uv=False)
        margin = float(sv[-1] / sv[0])
    return float(np.abs(U - X @ K.T - k[None, :]).max()), float(np.abs(U).max()), margin


def chain_gains_from_packed(n, m, T, gains):
    Ks, ks, o = [], [], 0
    for _ in range(T):
        Ks.append(np.asarray(gains[o:o + m * n], dtype=float).reshape((m, n), order="F")); o += m * n
        ks.append(np.asarray(gains[o:o + m], dtype=float)); o += m
    return Ks, ks


class ChainFamilies:
    """Dense-KKT solutions of one packed chain problem over the offset families of its segments."""

    def __init__(self, n, m, T, mats, vecs, seed=0):
        self.n, self.m, self.T = n, m, T
        blocks = dense_kkt.chain_blocks_from_packed(n, m, T, mats, vecs)
        fam = dense_kkt.OffsetFamilies(list(range(T)), list(range(1, T + 1)), [n] * (T + 1), [m] * T, blocks)
        self.sols = {s: fam.solve(s, offset_family(blocks["c"][s], seed=seed + s)) for s in range(0, max(T, 1), SEGMENT)}

    def base_solution_packed(self):
        x, u, y = self.sols[0][0]
        return np.concatenate([np.concatenate([x[i], y[i]] + ([u[i]] if i < self.T else []))
                               for i in range(self.T + 1)])

    def defect(self, gains):
        """(worst |u - K x - k| / max |u|, smallest affine-independence margin) over all stages."""
        Ks, ks = chain_gains_from_packed(self.n, self.m, self.T, gains)
        worst, umax, margin = 0.0, 0.0, np.inf
        for i in range(self.T):
            d, u, g = edge_defect(self.sols[i - i % SEGMENT], i, i, Ks[i], ks[i])
            worst, umax, margin = max(worst, d), max(umax, u), min(margin, g)
        return worst / max(umax, 1e-300), margin


@pytest.mark.parametrize("path", CHAINS, ids=[os.path.basename(p) for p in CHAINS])
def test_oracle_gains_are_the_control_law_of_the_dense_kkt_solutions(oracle_lib, path):
    d = np.load(path)
    n, m, T = int(d["n"]), int(d["m"]), int(d["T"])
    fam = ChainFamilies(n, m, T, d["mats"][0], d["vecs"][0])
    # the first member of the first family is the golden solution itself
    assert np.abs(fam.base_solution_packed() - d["sol"][0]).max() <= 1e-9 * np.abs(d["sol"][0]).max()
    _, gains, status = oracle_lib.chain_batch(n, m, T, d["mats"][:1], d["vecs"][:1])
    assert status[0] == 0
    for name, g in (("oracle", gains[0]), ("golden (numpy Riccati)", d["gains"][0])):
        defect, margin = fam.defect(g)
        print(f"{os.path.basename(path)} {name}: control-law defect {defect:.2e}, affine-independence margin {margin:.1e}")
        assert margin > 1e-6, (name, margin)   # the n + 1 parent states do determine the affine map
        assert defect <= 1e-9, (name, defect)
    # and a wrong gain is caught: the check has teeth
    bad = gains[0].copy()
    stage = slice((T // 2) * (m * n + m), (T // 2) * (m * n + m) + m * n)
    bad[stage.start + int(np.abs(bad[stage]).argmax())] *= 1.0 + 1e-4
    assert fam.defect(bad)[0] > 1e-7


@pytest.mark.parametrize("name", ["nonuniform_diagonal_delta", "branch_tree", "variable_dimension_branch",
                                  "five_node_variable_tree_eigen"])
def test_oracle_tree_gains_are_the_control_law_of_the_dense_kkt_solutions(oracle_lib, name):
    """The reference's own tree fixtures (tests/lqr_test.cpp:229-247, 300-335, 494-532, 695-762):
    every edge checked on the family that varies c of its own parent node."""
    prob = getattr(rp, name)()
    par, ch, sd, cd = prob["parents"], prob["children"], prob["state_dims"], prob["control_dims"]
    root = prob.get("root", 0)
    fam = dense_kkt.OffsetFamilies(par, ch, sd, cd, prob["blocks"], root=root)
    lqr = oracle_lib.TreeLQR(par, ch, sd, cd, prob["blocks"], root=root)
    assert lqr.factor() == 0
    lqr.solve()
    Ks, ks = lqr.gains()
    for e in range(len(cd)):
        sols = fam.solve(par[e], offset_family(np.asarray(prob["blocks"]["c"][par[e]], dtype=float), seed=e))
        d, u, margin = edge_defect(sols, par[e], e, Ks[e], ks[e])
        assert margin > 1e-6, (e, margin)
        assert d <= 1e-9 * max(u, 1e-300), (e, d)
