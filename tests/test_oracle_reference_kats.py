"""Pins the CPU oracle against every known-answer / property test the reference
holds for the hot path (/root/reference/tests/lqr_test.cpp).

The reference has no golden output vectors: its tests are status KATs, an
integer topology KAT, KKT-residual bounds (< 1e-12) and agreement with a dense
KKT solve (1e-10, Eigen isApprox = relative l2).  Each test below names the
reference test it re-expresses.
"""

import numpy as np
import pytest

import reference_problems as rp
from oracle import dense_kkt

SUCCESS, INVALID_DELTA, F_FAIL, G_FAIL, INVALID_TOPOLOGY = 0, 1, 2, 3, 4


def _lqr(oracle_lib, prob, **kw):
    return oracle_lib.TreeLQR(prob["parents"], prob["children"], prob["state_dims"],
                              prob["control_dims"], prob["blocks"], **kw)


def _residual(prob, x, u, y):
    return dense_kkt.residual_norm(prob["parents"], prob["children"], prob["state_dims"],
                                   prob["control_dims"], prob["blocks"], x, u, y)


def test_factor_reports_success(oracle_lib):
    """LQRFactor.ReportsSuccess / BoolFactorWrapsStatusApi (lqr_test.cpp:188-204)."""
    lqr = _lqr(oracle_lib, rp.default_chain(2, 1, 2))
    assert lqr.topology_status == SUCCESS
    assert lqr.factor() == SUCCESS


def test_factor_reports_invalid_delta(oracle_lib):
    """LQRFactor.ReportsInvalidDelta (lqr_test.cpp:206-211): delta must be > 0."""
    prob = rp.default_chain(2, 1, 2)
    prob["blocks"]["delta"][2][0] = 0.0
    assert _lqr(oracle_lib, prob).factor() == INVALID_DELTA


def test_factor_reports_f_failure(oracle_lib):
    """LQRFactor.ReportsFFactorizationFailure (lqr_test.cpp:213-219)."""
    prob = rp.default_chain(1, 1, 1)
    prob["blocks"]["Q"][1][0, 0] = -2.0
    prob["blocks"]["delta"][1][0] = 1.0
    assert _lqr(oracle_lib, prob).factor() == F_FAIL


def test_factor_reports_g_failure(oracle_lib):
    """LQRFactor.ReportsGFactorizationFailure (lqr_test.cpp:221-227)."""
    prob = rp.default_chain(1, 1, 1)
    prob["blocks"]["Q"][1][0, 0] = 0.0
    prob["blocks"]["R"][0][0, 0] = -1.0
    assert _lqr(oracle_lib, prob).factor() == G_FAIL


def test_solves_nonuniform_diagonal_delta_problem(oracle_lib):
    """LQRSolve.SolvesNonuniformDiagonalDeltaProblem (lqr_test.cpp:229-263)."""
    prob = rp.nonuniform_diagonal_delta()
    lqr = _lqr(oracle_lib, prob)
    assert lqr.factor() == SUCCESS
    assert _residual(prob, *lqr.solve()) < 1e-12


def test_solves_branching_tree_problem(oracle_lib):
    """LQRSolve.SolvesBranchingTreeProblem (lqr_test.cpp:411-429)."""
    prob = rp.branch_tree()
    lqr = _lqr(oracle_lib, prob)
    assert lqr.factor() == SUCCESS
    assert _residual(prob, *lqr.solve()) < 1e-12


def test_reuses_compiled_topology(oracle_lib):
    """LQRTopology.ReusesCompiledTopologyAcrossFactorAndSolveCalls (:431-450)."""
    prob = rp.branch_tree()
    lqr = _lqr(oracle_lib, prob)
    assert lqr.factor() == SUCCESS
    assert lqr.factor() == SUCCESS
    first = lqr.solve()
    second = lqr.solve()
    for a, b in zip(first, second):
        for va, vb in zip(a, b):
            np.testing.assert_array_equal(va, vb)
    assert _residual(prob, *second) < 1e-12


def test_rejects_invalid_tree_topology(oracle_lib):
    """LQRFactor.RejectsInvalidTreeTopology (lqr_test.cpp:452-464): two edges into node 1."""
    prob = rp.branch_tree()
    prob["children"][1] = 1
    assert _lqr(oracle_lib, prob).factor() == INVALID_TOPOLOGY


def test_solves_variable_dimension_branch(oracle_lib):
    """LQRSolve.SolvesVariableDimensionBranchingTreeProblem (lqr_test.cpp:641-659)."""
    prob = rp.variable_dimension_branch()
    lqr = _lqr(oracle_lib, prob)
    assert lqr.factor() == SUCCESS
    assert _residual(prob, *lqr.solve()) < 1e-12


def test_compiles_multichild_preorder_postorder(oracle_lib):
    """LQRTopology.CompilesMultiChildPreorderAndPostorder (lqr_test.cpp:931-953): exact ints."""
    lqr = _lqr(oracle_lib, rp.five_node_variable_tree_eigen())
    assert lqr.factor() == SUCCESS
    topo = lqr.topology_arrays()
    assert topo["child_offsets"] == [0, 2, 4, 4, 4, 4]
    assert topo["child_edges"] == [0, 1, 2, 3]
    assert topo["preorder_nodes"] == [0, 1, 3, 4, 2]
    assert topo["postorder_nodes"] == [2, 4, 3, 1, 0]


def test_rejects_disconnected_tree(oracle_lib):
    """LQRTopology.RejectsDisconnectedTree (lqr_test.cpp:955-967)."""
    prob = rp.five_node_variable_tree_eigen()
    prob["parents"][3], prob["children"][3] = 4, 3
    # dims of the mutated edge no longer match its blocks; only the status matters
    assert _lqr(oracle_lib, prob).factor() == INVALID_TOPOLOGY


def test_rejects_cycle(oracle_lib):
    """LQRTopology.RejectsCycle (lqr_test.cpp:969-980)."""
    prob = rp.five_node_variable_tree_eigen()
    prob["parents"][0] = 4
    assert _lqr(oracle_lib, prob).factor() == INVALID_TOPOLOGY


def test_rejects_null_topology_and_bad_root(oracle_lib):
    """compile_topology_data guards (lqr.cpp:567-574): null arrays, root out of range."""
    prob = rp.branch_tree()
    assert _lqr(oracle_lib, prob, null_topology=True).factor() == INVALID_TOPOLOGY
    assert _lqr(oracle_lib, prob, root=3).factor() == INVALID_TOPOLOGY
    assert _lqr(oracle_lib, prob, root=-1).factor() == INVALID_TOPOLOGY


def test_matches_dense_kkt_on_variable_dimension_tree(oracle_lib):
    """LQRSolve.MatchesDenseKKTOnVariableDimensionTreeProblem (lqr_test.cpp:982-1013).
    Eigen isApprox(v, w, 1e-10): ||v - w|| <= 1e-10 * min(||v||, ||w||)."""
    prob = rp.five_node_variable_tree_eigen()
    lqr = _lqr(oracle_lib, prob)
    assert lqr.factor() == SUCCESS
    x, u, y = lqr.solve()
    xd, ud, yd = dense_kkt.solve(prob["parents"], prob["children"], prob["state_dims"],
                                 prob["control_dims"], prob["blocks"])
    for got, want in list(zip(x, xd)) + list(zip(y, yd)) + list(zip(u, ud)):
        assert np.linalg.norm(got - want) <= 1e-10 * min(np.linalg.norm(got), np.linalg.norm(want))


def test_zero_dimensional_root_state(oracle_lib):
    """tests/variable_dimensions_test.cpp:316-336: state_dim == 0 at the root must not crash."""
    rng = np.random.default_rng(5)
    n1, m = 2, 1
    blocks = {
        "Q": [np.zeros((0, 0)), np.eye(n1) * 1.5], "q": [np.zeros(0), rng.normal(size=n1)],
        "c": [np.zeros(0), rng.normal(size=n1)], "delta": [np.zeros(0), np.array([0.5, 0.7])],
        "M": [np.zeros((0, m))], "R": [np.array([[1.2]])], "A": [np.zeros((n1, 0))],
        "B": [rng.normal(size=(n1, m))], "r": [rng.normal(size=m)],
    }
    prob = dict(parents=[0], children=[1], state_dims=[0, n1], control_dims=[m], blocks=blocks)
    lqr = _lqr(oracle_lib, prob)
    assert lqr.factor() == SUCCESS
    assert _residual(prob, *lqr.solve()) < 1e-12


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_trees_match_dense_kkt(oracle_lib, seed):
    """Beyond the reference's fixtures: random trees with variable dims vs dense KKT."""
    rng = np.random.default_rng(seed)
    N = int(rng.integers(4, 9))
    parents = [int(rng.integers(0, e + 1)) for e in range(N - 1)]
    children = list(range(1, N))
    sd = [int(rng.integers(1, 5)) for _ in range(N)]
    cd = [int(rng.integers(1, 4)) for _ in range(N - 1)]
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
    prob = dict(parents=parents, children=children, state_dims=sd, control_dims=cd, blocks=blocks)
    lqr = _lqr(oracle_lib, prob)
    assert lqr.factor() == SUCCESS
    x, u, y = lqr.solve()
    assert _residual(prob, x, u, y) < 1e-10
    xd, ud, yd = dense_kkt.solve(parents, children, sd, cd, blocks)
    for got, want in list(zip(x, xd)) + list(zip(y, yd)) + list(zip(u, ud)):
        assert np.linalg.norm(got - want) <= 1e-8 * (1 + np.linalg.norm(want))


def test_chain_batch_layout_matches_tree_api(oracle_lib):
    """The packed-chain convenience entry runs the very same factor/solve."""
    from sip_optimal_control_amd import ChainShape, synthetic
    n, m, T = 4, 2, 6
    shape = ChainShape(n, m, T)
    mats, vecs = synthetic.make_chain_batch(shape, 3, seed=11, cross_term=0.05)
    sol, gains, status = oracle_lib.chain_batch(n, m, T, mats.numpy(), vecs.numpy())
    assert (status == 0).all()
    for p in range(3):
        blocks = dense_kkt.chain_blocks_from_packed(n, m, T, mats[p].numpy(), vecs[p].numpy())
        lqr = oracle_lib.TreeLQR(list(range(T)), list(range(1, T + 1)), [n] * (T + 1), [m] * T, blocks)
        assert lqr.factor() == SUCCESS
        x, u, y = lqr.solve()
        xs, us, ys = dense_kkt.chain_sol_from_packed(n, m, T, sol[p])
        for a, b in list(zip(x, xs)) + list(zip(u, us)) + list(zip(y, ys)):
            np.testing.assert_array_equal(a, b)
        K, k = lqr.gains()
        off = 0
        for e in range(T):
            np.testing.assert_array_equal(gains[p, off:off + m * n].reshape((m, n), order="F"), K[e])
            off += m * n
            np.testing.assert_array_equal(gains[p, off:off + m], k[e])
            off += m
