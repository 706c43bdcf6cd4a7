"""General tree / variable-dimension HIP path (through the C ABI) against the
reference's own tree fixtures, the oracle and the dense-KKT golden vectors.
Tolerance: 1e-10 relative l2 per block (the reference's own dense check,
tests/lqr_test.cpp:995-1010) and KKT residual < 1e-12."""
import os

import numpy as np
import pytest

import reference_problems as rp
from oracle import dense_kkt

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _solver(prob, batch=1, **kw):
    from sip_optimal_control_amd.tree import BatchedTreeLQR
    return BatchedTreeLQR(prob["parents"], prob["children"], prob["state_dims"], prob["control_dims"],
                          batch=batch, **kw)


def _status(prob):
    s = _solver(prob)
    if s.topology_status == 0:
        s.pack([prob["blocks"]])
    st = s.factor()
    torch.cuda.synchronize()
    return int(st[0])


def test_status_kats():
    """lqr_test.cpp:188-227: SUCCESS, INVALID_DELTA, F and G factorization failures."""
    assert _status(rp.default_chain(2, 1, 2)) == 0
    p = rp.default_chain(2, 1, 2)
    p["blocks"]["delta"][2][0] = 0.0
    assert _status(p) == 1
    p = rp.default_chain(1, 1, 1)
    p["blocks"]["Q"][1][0, 0] = -2.0
    assert _status(p) == 2
    p = rp.default_chain(1, 1, 1)
    p["blocks"]["Q"][1][0, 0] = 0.0
    p["blocks"]["R"][0][0, 0] = -1.0
    assert _status(p) == 3


def test_invalid_topology_is_latched():
    """lqr_test.cpp:452-464, 955-980: INVALID_TOPOLOGY from every factor call."""
    p = rp.branch_tree()
    p["children"][1] = 1
    s = _solver(p)
    assert s.topology_status == 4
    for _ in range(2):
        st = s.factor()
        torch.cuda.synchronize()
        assert int(st[0]) == 4


@pytest.mark.parametrize("name", ["nonuniform_diagonal_delta", "branch_tree",
                                  "variable_dimension_branch", "five_node_variable_tree"])
def test_reference_fixtures(oracle_lib, name):
    builders = {"nonuniform_diagonal_delta": rp.nonuniform_diagonal_delta, "branch_tree": rp.branch_tree,
                "variable_dimension_branch": rp.variable_dimension_branch,
                "five_node_variable_tree": rp.five_node_variable_tree_eigen}
    prob = builders[name]()
    s = _solver(prob)
    s.pack([prob["blocks"]])
    assert int(s.factor()[0]) == 0
    assert int(s.factor()[0]) == 0          # factor twice, solve twice (lqr_test.cpp:431-450)
    s.solve()
    s.solve()
    torch.cuda.synchronize()
    x, u, y = s.unpack_solution()
    res = dense_kkt.residual_norm(prob["parents"], prob["children"], prob["state_dims"],
                                  prob["control_dims"], prob["blocks"], x, u, y)
    assert res < 1e-12
    d = np.load(os.path.join(GOLD, f"tree_{name}.npz"))
    for got, want in ((np.concatenate(x), d["x"]), (np.concatenate(u), d["u"]), (np.concatenate(y), d["y"])):
        assert np.linalg.norm(got - want) <= 1e-10 * np.linalg.norm(want)
    lqr = oracle_lib.TreeLQR(prob["parents"], prob["children"], prob["state_dims"],
                             prob["control_dims"], prob["blocks"])
    assert lqr.factor() == 0
    xo, uo, yo = lqr.solve()
    for a, b in list(zip(x, xo)) + list(zip(u, uo)) + list(zip(y, yo)):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-12 * max(1.0, np.abs(b).max()))
    Ko, ko = lqr.gains()
    Kg, kg = s.unpack_gains()
    for a, b in list(zip(Kg, Ko)) + list(zip(kg, ko)):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-12)
    assert s.topology_arrays() == lqr.topology_arrays()


def test_zero_dimensional_root():
    """tests/variable_dimensions_test.cpp:316-336."""
    rng = np.random.default_rng(5)
    blocks = {"Q": [np.zeros((0, 0)), np.eye(2) * 1.5], "q": [np.zeros(0), rng.normal(size=2)],
              "c": [np.zeros(0), rng.normal(size=2)], "delta": [np.zeros(0), np.array([0.5, 0.7])],
              "M": [np.zeros((0, 1))], "R": [np.array([[1.2]])], "A": [np.zeros((2, 0))],
              "B": [rng.normal(size=(2, 1))], "r": [rng.normal(size=1)]}
    prob = dict(parents=[0], children=[1], state_dims=[0, 2], control_dims=[1], blocks=blocks)
    s = _solver(prob)
    s.pack([blocks])
    assert int(s.factor()[0]) == 0
    s.solve()
    torch.cuda.synchronize()
    x, u, y = s.unpack_solution()
    assert dense_kkt.residual_norm([0], [1], [0, 2], [1], blocks, x, u, y) < 1e-12


def test_random_tree_batch_matches_oracle(oracle_lib):
    """A batch of instances of one random tree (variable dims), one of them made to fail."""
    rng = np.random.default_rng(21)
    N = 9
    parents = [int(rng.integers(0, e + 1)) for e in range(N - 1)]
    children = list(range(1, N))
    sd = [int(rng.integers(1, 7)) for _ in range(N)]
    cd = [int(rng.integers(1, 4)) for _ in range(N - 1)]
    batch = 6
    probs = []
    for b in range(batch):
        blocks = {k: [] for k in ("Q", "M", "R", "q", "r", "A", "B", "c", "delta")}
        for n in sd:
            S = rng.normal(size=(n, n))
            blocks["Q"].append(S.T @ S + 1e-3 * np.eye(n))
            blocks["q"].append(rng.normal(size=n))
            blocks["c"].append(rng.normal(size=n))
            blocks["delta"].append(1e-3 + 0.1 * rng.random(n))
        for e, m in enumerate(cd):
            np_, nc = sd[parents[e]], sd[children[e]]
            G = rng.normal(size=(m, m))
            blocks["A"].append(0.3 * rng.normal(size=(nc, np_)))
            blocks["B"].append(0.3 * rng.normal(size=(nc, m)))
            blocks["M"].append(0.05 * rng.normal(size=(np_, m)))
            blocks["R"].append(G.T @ G + 1.01 * np.eye(m))
            blocks["r"].append(rng.normal(size=m))
        probs.append(blocks)
    probs[3]["R"][2] = -np.eye(cd[2])  # G failure on instance 3
    prob = dict(parents=parents, children=children, state_dims=sd, control_dims=cd)
    s = _solver(dict(prob, blocks=None), batch=batch)
    s.pack(probs)
    st = s.factor()
    s.solve()
    torch.cuda.synchronize()
    st = st.cpu().numpy()
    for b in range(batch):
        lqr = oracle_lib.TreeLQR(parents, children, sd, cd, probs[b])
        assert lqr.factor() == st[b]
        if st[b] != 0:
            assert b == 3 and st[b] == 3
            continue
        xo, uo, yo = lqr.solve()
        x, u, y = s.unpack_solution(b)
        for a, bb in list(zip(x, xo)) + list(zip(u, uo)) + list(zip(y, yo)):
            np.testing.assert_allclose(a, bb, rtol=0, atol=1e-11 * max(1.0, np.abs(bb).max()))


@pytest.mark.parametrize("shape", [0, 1, 2])
@pytest.mark.parametrize("T,base_n", [(31, 4), (31, 8), (63, 4)])
def test_reference_variable_benchmark_shapes(oracle_lib, shape, T, base_n):
    """The shapes of BM_LQRVariableFactorSolve (benchmarks/lqr_benchmark.cpp:209-310, 547-555):
    heterogeneous chain, shallow wide tree, binary tree with per-node dimensions."""
    rng = np.random.default_rng(17 + 31 * shape)
    batch = 3
    probs = [rp.variable_benchmark_problem(shape, T, base_n, 2, rng) for _ in range(batch)]
    s = _solver(probs[0], batch=batch)
    s.pack([p["blocks"] for p in probs])
    st = s.factor()
    s.solve()
    torch.cuda.synchronize()
    assert st.cpu().tolist() == [0] * batch
    for b, prob in enumerate(probs):
        lqr = oracle_lib.TreeLQR(prob["parents"], prob["children"], prob["state_dims"], prob["control_dims"],
                                 prob["blocks"])
        assert lqr.factor() == 0
        xo, uo, yo = lqr.solve()
        x, u, y = s.unpack_solution(b)
        for a, bb in list(zip(x, xo)) + list(zip(u, uo)) + list(zip(y, yo)):
            np.testing.assert_allclose(a, bb, rtol=0, atol=1e-10 * max(1.0, np.abs(bb).max()))
        assert dense_kkt.residual_norm(prob["parents"], prob["children"], prob["state_dims"], prob["control_dims"],
                                       prob["blocks"], x, u, y) < 1e-9


# ---- the fused size-class path (csrc/tree_qw16.hpp) through sip_lqr_tree_factor_solve -------------------
def _check_against_oracle(oracle_lib, s, probs, tol=1e-10, expect=None):
    out, st = s.factor_solve()
    torch.cuda.synchronize()
    st = st.cpu().numpy()
    for b, blocks in enumerate(probs):
        lqr = oracle_lib.TreeLQR(s.parents, s.children, s.state_dims, s.control_dims, blocks)
        ref = lqr.factor()
        assert ref == st[b], (b, ref, st[b])
        if expect is not None:
            assert ref == expect[b]
        if ref != 0:
            continue
        xo, uo, yo = lqr.solve()
        x, u, y = s.unpack_solution(b)
        for a, bb in list(zip(x, xo)) + list(zip(u, uo)) + list(zip(y, yo)):
            np.testing.assert_allclose(a, bb, rtol=0, atol=tol * max(1.0, np.abs(bb).max(initial=0.0)))
        Ko, ko = lqr.gains()
        Kg, kg = s.unpack_gains(b)
        for a, bb in list(zip(Kg, Ko)) + list(zip(kg, ko)):
            np.testing.assert_allclose(a, bb, rtol=0, atol=tol * max(1.0, np.abs(bb).max(initial=0.0)))


@pytest.mark.parametrize("name", ["nonuniform_diagonal_delta", "branch_tree", "variable_dimension_branch",
                                  "five_node_variable_tree_eigen", "default_chain"])
def test_fused_tree_kernel_on_the_reference_fixtures(oracle_lib, name):
    """tests/lqr_test.cpp:229-263, 411-429, 641-659, 982-1013 through the fused sweep: residual < 1e-12,
    x, u, y and K, k against the oracle; twice on the same object (lqr_test.cpp:431-450)."""
    prob = rp.default_chain(2, 1, 2) if name == "default_chain" else getattr(rp, name)()
    s = _solver(prob)
    assert "tree_factor_solve_qw16" in s.kernel_name
    s.pack([prob["blocks"]])
    for _ in range(2):
        _check_against_oracle(oracle_lib, s, [prob["blocks"]], tol=1e-12)
    x, u, y = s.unpack_solution()
    assert dense_kkt.residual_norm(prob["parents"], prob["children"], prob["state_dims"], prob["control_dims"],
                                   prob["blocks"], x, u, y) < 1e-12


def test_fused_tree_kernel_statuses_and_precedence(oracle_lib):
    """FactorStatus through the fused sweep (lqr.hpp:68-74): the KATs of lqr_test.cpp:188-227 and, on a
    five-node tree, first failing node in postorder and G before delta before F at a node
    (lqr.cpp:696-701, 722-727)."""
    def status_of(prob):
        s = _solver(prob)
        s.pack([prob["blocks"]])
        _, st = s.factor_solve()
        torch.cuda.synchronize()
        lqr = oracle_lib.TreeLQR(prob["parents"], prob["children"], prob["state_dims"], prob["control_dims"],
                                 prob["blocks"])
        assert lqr.factor() == int(st[0])
        return int(st[0])
    assert status_of(rp.default_chain(2, 1, 2)) == 0
    p = rp.default_chain(2, 1, 2); p["blocks"]["delta"][2][0] = 0.0
    assert status_of(p) == 1
    p = rp.default_chain(1, 1, 1); p["blocks"]["Q"][1][0, 0] = -2.0
    assert status_of(p) == 2
    p = rp.default_chain(1, 1, 1); p["blocks"]["Q"][1][0, 0] = 0.0; p["blocks"]["R"][0][0, 0] = -1.0
    assert status_of(p) == 3
    # five-node tree: parents {0,0,1,1}, children {1,2,3,4}; postorder 2, 4, 3, 1, 0
    base = rp.five_node_variable_tree_eigen
    p = base(); p["blocks"]["R"][2] = -1e4 * np.eye(p["control_dims"][2]); p["blocks"]["delta"][1][0] = 0.0
    assert status_of(p) == 3      # node 1: its child edge 2 (G) before its own delta
    p = base(); p["blocks"]["delta"][1][0] = 0.0; p["blocks"]["Q"][1] = -1e6 * np.eye(p["state_dims"][1])
    assert status_of(p) == 1      # same node: delta before F
    p = base(); p["blocks"]["delta"][4][0] = -1.0; p["blocks"]["R"][0] = -1e4 * np.eye(p["control_dims"][0])
    assert status_of(p) == 1      # node 4 precedes edge 0 (processed at the root) in postorder
    p = base(); p["blocks"]["Q"][2] = -1e6 * np.eye(p["state_dims"][2]); p["blocks"]["delta"][3][0] = 0.0
    assert status_of(p) == 2      # node 2 is first in postorder


def test_fused_tree_kernel_random_batches(oracle_lib):
    """Random trees with per-node dimensions in several size classes, zero-dimensional nodes, one
    failing instance per batch, a batch that is not a multiple of the 4 problems of a wavefront."""
    rng = np.random.default_rng(77)
    for trial, (nmax, mmax, N) in enumerate([(4, 2, 7), (6, 3, 12), (9, 4, 9), (12, 4, 6), (15, 3, 5), (14, 8, 4)]):
        parents = [int(rng.integers(0, e + 1)) for e in range(N - 1)]
        children = list(range(1, N))
        sd = [int(rng.integers(0 if trial % 2 else 1, nmax + 1)) for _ in range(N)]
        sd[int(rng.integers(0, N))] = nmax
        cd = [int(rng.integers(1, mmax + 1)) for _ in range(N - 1)]
        cd[0] = mmax
        batch = 6
        probs = []
        for b in range(batch):
            blocks = {k: [] for k in ("Q", "M", "R", "q", "r", "A", "B", "c", "delta")}
            for n in sd:
                S = rng.normal(size=(n, n))
                blocks["Q"].append(S.T @ S + 1e-3 * np.eye(n))
                blocks["q"].append(rng.normal(size=n)); blocks["c"].append(rng.normal(size=n))
                blocks["delta"].append(1e-3 + 0.1 * rng.random(n))
            for e, m in enumerate(cd):
                np_, nc = sd[parents[e]], sd[children[e]]
                G = rng.normal(size=(m, m))
                blocks["A"].append(0.3 * rng.normal(size=(nc, np_))); blocks["B"].append(0.3 * rng.normal(size=(nc, m)))
                blocks["M"].append(0.05 * rng.normal(size=(np_, m))); blocks["R"].append(G.T @ G + 1.01 * np.eye(m))
                blocks["r"].append(rng.normal(size=m))
            probs.append(blocks)
        probs[3]["R"][N - 2] = -1e4 * np.eye(cd[N - 2])
        s = _solver(dict(parents=parents, children=children, state_dims=sd, control_dims=cd), batch=batch)
        assert "tree_factor_solve_qw16" in s.kernel_name, s.kernel_name
        s.pack(probs)
        _check_against_oracle(oracle_lib, s, probs, tol=1e-10, expect=[0, 0, 0, 3, 0, 0])


@pytest.mark.parametrize("shape", [0, 1, 2])
@pytest.mark.parametrize("T,base_n", [(31, 4), (63, 8)])
def test_fused_tree_kernel_on_the_variable_benchmark_family(oracle_lib, shape, T, base_n):
    """BM_LQRVariableFactorSolve (benchmarks/lqr_benchmark.cpp:209-310, 716-744): heterogeneous chain,
    shallow wide tree (63 edges out of the root), binary tree; and the general engine agrees."""
    rng = np.random.default_rng(17 + 31 * shape)
    batch = 5
    probs = [rp.variable_benchmark_problem(shape, T, base_n, 2, rng) for _ in range(batch)]
    s = _solver(probs[0], batch=batch)
    assert "tree_factor_solve_qw16" in s.kernel_name
    s.pack([p["blocks"] for p in probs])
    _check_against_oracle(oracle_lib, s, [p["blocks"] for p in probs], tol=1e-10)
    fused = s.output.clone()
    s.factor(); s.solve()
    torch.cuda.synchronize()
    assert float((s.output - fused).abs().max()) <= 1e-10 * float(fused.abs().max())


def test_trees_beyond_the_size_classes_fall_back_to_the_general_engine(oracle_lib, monkeypatch):
    prob = rp.default_chain(17, 2, 3)
    s = _solver(prob)
    assert s.kernel_name == "tree_generic/f64"
    s.pack([prob["blocks"]])
    _check_against_oracle(oracle_lib, s, [prob["blocks"]], tol=1e-10)
    monkeypatch.setenv("SIP_LQR_TREE", "general")
    assert _solver(rp.branch_tree()).kernel_name == "tree_generic/f64"


def test_fused_sweep_writes_every_workspace_field(oracle_lib):
    """sip_lqr_tree_factor_solve_workspace: the fused size-class kernel leaves W, K, G_factor, k (edges) and V,
    F_factor, sqrt_delta, sqrt_delta_inv, v (nodes) of LQR::Workspace (lqr.hpp:109-135) in the work arena, as
    sip_lqr_tree_factor + sip_lqr_tree_solve of the general engine do -- what lets the drop-in LQR class run on the
    fused kernels by default (helpers.cpp:521-665 reads those fields).  Whole arenas compared (G_factor / F_factor:
    lower triangles; above the diagonal both keep the pre-factor values of Eigen's in-place LLT), random trees in
    three size classes with zero-dimensional nodes, the reference's fixtures, a batch of 6."""
    rng = np.random.default_rng(5)
    cases = [dict(parents=p["parents"], children=p["children"], state_dims=p["state_dims"],
                  control_dims=p["control_dims"], probs=[p["blocks"]])
             for p in (rp.five_node_variable_tree_eigen(), rp.branch_tree(), rp.variable_dimension_branch(),
                       rp.nonuniform_diagonal_delta())]
    for nmax, mmax, N in [(5, 2, 8), (10, 4, 7), (15, 8, 5)]:
        parents = [int(rng.integers(0, e + 1)) for e in range(N - 1)]
        sd = [int(rng.integers(0, nmax + 1)) for _ in range(N)]
        sd[int(rng.integers(0, N))] = nmax
        cd = [int(rng.integers(1, mmax + 1)) for _ in range(N - 1)]
        cd[0] = mmax
        probs = []
        for b in range(6):
            blocks = {k: [] for k in ("Q", "M", "R", "q", "r", "A", "B", "c", "delta")}
            for n in sd:
                S = rng.normal(size=(n, n))
                blocks["Q"].append(S.T @ S + 1e-3 * np.eye(n))
                blocks["q"].append(rng.normal(size=n)); blocks["c"].append(rng.normal(size=n))
                blocks["delta"].append(1e-3 + 0.1 * rng.random(n))
            for e, m in enumerate(cd):
                np_, nc = sd[parents[e]], sd[e + 1]
                G = rng.normal(size=(m, m))
                blocks["A"].append(0.3 * rng.normal(size=(nc, np_))); blocks["B"].append(0.3 * rng.normal(size=(nc, m)))
                blocks["M"].append(0.05 * rng.normal(size=(np_, m))); blocks["R"].append(G.T @ G + 1.01 * np.eye(m))
                blocks["r"].append(rng.normal(size=m))
            probs.append(blocks)
        cases.append(dict(parents=parents, children=list(range(1, N)), state_dims=sd, control_dims=cd, probs=probs))
    for case in cases:
        probs = case.pop("probs")
        s = _solver(case, batch=len(probs))
        assert "tree_factor_solve_qw16" in s.kernel_name
        s.pack(probs)
        s.work.zero_()
        s.factor(); s.solve()
        torch.cuda.synchronize()
        want, want_out = s.work.clone().cpu().numpy(), s.output.clone()
        s.work.zero_(); s.output.zero_()
        s.factor_solve(workspace=True)
        torch.cuda.synchronize()
        got = s.work.cpu().numpy()
        assert float((s.output - want_out).abs().max()) <= 1e-10 * max(1.0, float(want_out.abs().max()))
        sd, cd, par, ch = case["state_dims"], case["control_dims"], case["parents"], case["children"]
        max_n = max(sd)

        def close(a, b, what):
            scale = max(1.0, float(np.abs(b).max(initial=0.0)))
            assert float(np.abs(a - b).max(initial=0.0)) <= 1e-10 * scale, what

        for b in range(len(probs)):
            for e in range(len(cd)):
                np_, nc, m = sd[par[e]], sd[ch[e]], cd[e]
                o = s.offset(1, 1, e)
                close(got[b, o:o + nc * nc], want[b, o:o + nc * nc], ("W", e))
                o += max_n * max_n
                close(got[b, o:o + m * np_], want[b, o:o + m * np_], ("K", e))
                o += m * np_
                tri = np.tril(np.ones((m, m), dtype=bool)).T.reshape(-1)        # column-major lower triangle
                close(got[b, o:o + m * m][tri], want[b, o:o + m * m][tri], ("G_factor", e))
                o += m * m
                close(got[b, o:o + m], want[b, o:o + m], ("k", e))
            for j, n in enumerate(sd):
                o = s.offset(1, 0, j)
                close(got[b, o:o + n * n], want[b, o:o + n * n], ("V", j))
                tri = np.tril(np.ones((n, n), dtype=bool)).T.reshape(-1)
                close(got[b, o + n * n:o + 2 * n * n][tri], want[b, o + n * n:o + 2 * n * n][tri], ("F_factor", j))
                close(got[b, o + 2 * n * n:o + 2 * n * n + 3 * n], want[b, o + 2 * n * n:o + 2 * n * n + 3 * n],
                      ("sqrt_delta | sqrt_delta_inv | v", j))
