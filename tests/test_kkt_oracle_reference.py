"""Pins oracle/kkt_oracle.c (the CPU restatement of CallbackProvider::factor /
solve / add_Kx_to_y, helpers.cpp:242-370, 749-893, 953-1368) against the
reference's own tests of that path and an independent numpy dense solve.

Re-expressed reference tests (tests/variable_dimensions_test.cpp):
  expect_kkt_solve (:135-181): ||K sol - rhs|| < 1e-9 through add_Kx_to_y, for
  SolvesChainWithNodeAndEdgeConstraints (:265-288),
  SolvesIndependentConstraintsOnSiblingEdges (:290-313),
  SolvesBranchedSystemWithZeroDimensionalRoot (:315-336).
"""
import numpy as np
import pytest

from oracle.kkt import KKTDims, KKTOracle, dense_kkt_matrix
from tests import reference_kkt_problems as rk


@pytest.mark.parametrize("name", sorted(rk.REFERENCE_CASES))
def test_reference_expect_kkt_solve(name):
    dims, model, (w, r1, r2, r3, rhs) = rk.reference_case(name)
    o = KKTOracle(dims)
    assert (o.dim(0), o.dim(1), o.dim(2), o.dim(3)) == (dims.x_dim, dims.y_dim, dims.z_dim, dims.model_len)
    assert o.factor(model, w, r1, r2, r3) == 0
    sol = o.solve(model, rhs)
    product = o.add_Kx_to_y(model, w, r1, r2, r3, sol)
    assert np.linalg.norm(product - rhs) < 1e-9  # the reference's tolerance, :180
    # independent: dense assembly + LU
    K = dense_kkt_matrix(dims, model, w, r1, r2, r3)
    np.testing.assert_allclose(product, K @ sol, rtol=0, atol=1e-12)
    dense = np.linalg.solve(K, rhs)
    np.testing.assert_allclose(sol, dense, rtol=1e-9, atol=1e-11)
    # repeated solve on the same factorization, new right-hand side
    rhs2 = np.cos(np.arange(dims.kkt_dim))
    np.testing.assert_allclose(o.solve(model, rhs2), np.linalg.solve(K, rhs2), rtol=1e-9, atol=1e-11)


def test_offset_tables_match_python_restatement():
    dims, _, _ = rk.reference_case("chain_node_edge_constraints")
    o = KKTOracle(dims)
    from oracle.kkt import NODE_BLOCKS, EDGE_BLOCKS, _TABLES
    for t, name in enumerate(_TABLES):
        count = dims.E if name in ("x_control", "y_edge_c", "z_edge") else dims.N
        assert [o.vector_offset(t, i) for i in range(count)] == dims.off[name][:count]
    for b, name in enumerate(NODE_BLOCKS):
        assert [o.model_offset(b, i) for i in range(dims.N)] == dims.node_off[name]
    for b, name in enumerate(EDGE_BLOCKS):
        assert [o.model_offset(len(NODE_BLOCKS) + b, e) for e in range(dims.E)] == dims.edge_off[name]
    # flattened orderings of types.cpp:24-64 on the reference's chain case
    assert dims.off["x_state"] == [0, 3, 6] and dims.off["x_control"][:2] == [2, 4]
    assert (dims.x_dim, dims.y_dim, dims.z_dim) == (9, 12, 6)


def test_null_constraint_dims_mean_zero():
    dims = KKTDims([0, 1], [1, 2], [2, 1, 3], [1, 2])
    o = KKTOracle(dims, null_dims=True)
    assert (o.dim(1), o.dim(2)) == (6, 0)
    model = rk.initialize_model(dims)
    w, r1, r2, r3, rhs = rk.regularization(dims)
    assert o.factor(model, w, r1, r2, r3) == 0
    sol = o.solve(model, rhs)
    assert np.linalg.norm(o.add_Kx_to_y(model, w, r1, r2, r3, sol) - rhs) < 1e-9


def test_condensed_lqr_blocks_match_numpy():
    """Q_mod = tril(Q) + r1 + sum J^T D J mirrored, etc. (helpers.cpp:299-361)."""
    dims, model, (w, r1, r2, r3, _) = rk.reference_case("branch_sibling_edges")
    o = KKTOracle(dims)
    assert o.factor(model, w, r1, r2, r3) == 0
    nodes, edges = dims.unpack_model(model)
    off = dims.off
    for i in range(dims.N):
        n = dims.sd[i]
        Q = nodes[i]["d2L_dx2"] + np.diag(r1[off["x_state"][i]:off["x_state"][i] + n])
        Q = Q + nodes[i]["dc_dx"].T @ np.diag(1 / r2[off["y_node_c"][i]:off["y_node_c"][i] + dims.ncd[i]]) @ nodes[i]["dc_dx"]
        zs = slice(off["z_node"][i], off["z_node"][i] + dims.ngd[i])
        Q = Q + nodes[i]["dg_dx"].T @ np.diag(1 / (w[zs] + r3[zs])) @ nodes[i]["dg_dx"]
        for e in range(dims.E):
            if dims.parents[e] != i:
                continue
            ys = slice(off["y_edge_c"][e], off["y_edge_c"][e] + dims.ecd[e])
            ze = slice(off["z_edge"][e], off["z_edge"][e] + dims.egd[e])
            Q = Q + edges[e]["d2L_dx2"] + edges[e]["dc_dx"].T @ np.diag(1 / r2[ys]) @ edges[e]["dc_dx"] + \
                edges[e]["dg_dx"].T @ np.diag(1 / (w[ze] + r3[ze])) @ edges[e]["dg_dx"]
        np.testing.assert_allclose(o.lqr_block("Q", i, n * n).reshape((n, n), order="F"), Q, rtol=1e-13, atol=1e-14)
        np.testing.assert_array_equal(o.lqr_block("d", i, n), r2[off["y_dyn"][i]:off["y_dyn"][i] + n])
    for e in range(dims.E):
        n, m = dims.sd[dims.parents[e]], dims.cd[e]
        ys = slice(off["y_edge_c"][e], off["y_edge_c"][e] + dims.ecd[e])
        ze = slice(off["z_edge"][e], off["z_edge"][e] + dims.egd[e])
        Dc, Dg = np.diag(1 / r2[ys]), np.diag(1 / (w[ze] + r3[ze]))
        M = edges[e]["d2L_dxdu"] + edges[e]["dc_dx"].T @ Dc @ edges[e]["dc_du"] + edges[e]["dg_dx"].T @ Dg @ edges[e]["dg_du"]
        R = edges[e]["d2L_du2"] + np.diag(r1[off["x_control"][e]:off["x_control"][e] + m]) + \
            edges[e]["dc_du"].T @ Dc @ edges[e]["dc_du"] + edges[e]["dg_du"].T @ Dg @ edges[e]["dg_du"]
        np.testing.assert_allclose(o.lqr_block("M", e, n * m).reshape((n, m), order="F"), M, rtol=1e-13, atol=1e-14)
        np.testing.assert_allclose(o.lqr_block("R", e, m * m).reshape((m, m), order="F"), R, rtol=1e-13, atol=1e-14)


def test_factor_false_paths():
    dims, model, (w, r1, r2, r3, _) = rk.reference_case("chain_node_edge_constraints")
    o = KKTOracle(dims)
    bad = r2.copy()
    bad[dims.off["y_edge_c"][1]] = 0.0  # helpers.cpp:281-286
    assert o.factor(model, w, r1, bad, r3) == 5
    bad = r2.copy()
    bad[dims.off["y_dyn"][2] + 1] = -1.0  # helpers.cpp:256-261
    assert o.factor(model, w, r1, bad, r3) == 5
    badw = w.copy()
    badw[dims.off["z_node"][1]] = -r3[dims.off["z_node"][1]]  # w + r3 == 0, helpers.cpp:269-276
    assert o.factor(model, badw, r1, r2, r3) == 5
    assert o.factor(model, w, r1, r2, r3) == 0
    # InputValidation (:183-228): negative dimension -> factor() is false
    from oracle.kkt import _lib, _ints
    h = _lib().kkt_oracle_create(2, 0, _ints([0, 0]), _ints([1, 2]), _ints([2, 1, 3]), _ints([1, 2]), None, None,
                                 _ints([-1, 1]), None)
    assert _lib().kkt_oracle_factor(h, None, None, None, None, None) == 6
    _lib().kkt_oracle_destroy(h)
    # the DAG of :210-214 (node 2 has two parents) is latched by the traversal
    h = _lib().kkt_oracle_create(2, 0, _ints([0, 1]), _ints([2, 2]), _ints([2, 1, 3]), _ints([1, 2]), None, None,
                                 None, None)
    m2 = np.zeros(max(1, _lib().kkt_oracle_dim(h, 3)))
    ones = np.ones(64)
    assert _lib().kkt_oracle_factor(h, m2.ctypes.data, ones.ctypes.data, ones.ctypes.data, ones.ctypes.data,
                                    ones.ctypes.data) == 4
    _lib().kkt_oracle_destroy(h)


@pytest.mark.parametrize("n,m,T", [(4, 1, 16), (8, 2, 16), (12, 4, 10)])
def test_newton_kkt_benchmark_shape(n, m, T):
    dims = rk.newton_kkt_dims(n, m, T)
    model, w, r1, r2, r3, rhs = rk.newton_kkt_problem(dims, seed=n * 100 + m, r2_max=1e2)
    o = KKTOracle(dims)
    assert o.factor(model, w, r1, r2, r3) == 0
    sol = o.solve(model, rhs)
    K = dense_kkt_matrix(dims, model, w, r1, r2, r3)
    dense = np.linalg.solve(K, rhs)
    assert np.linalg.norm(sol - dense) <= 1e-8 * np.linalg.norm(dense)
    res = o.add_Kx_to_y(model, w, r1, r2, r3, sol) - rhs
    assert np.linalg.norm(res) <= 1e-9 * max(1.0, np.linalg.norm(K, np.inf) * np.linalg.norm(sol, np.inf))


def test_batch_entry_matches_single():
    dims = rk.newton_kkt_dims(4, 2, 8)
    model, w, r1, r2, r3, rhs = rk.newton_kkt_problem(dims, seed=5, batch=6, r2_max=1e2)
    r2[3, 0] = -1.0
    o = KKTOracle(dims)
    sol, status = o.batch(model, w, r1, r2, r3, rhs, threads=2)
    assert status.tolist() == [0, 0, 0, 5, 0, 0]
    for p in (0, 5):
        assert o.factor(model[p], w[p], r1[p], r2[p], r3[p]) == 0
        np.testing.assert_array_equal(sol[p], o.solve(model[p], rhs[p]))


def test_reference_schur_variables():
    """CallbackProvider.SolvesBranchedSystemWithSchurVariables (:338-363): theta_dim = 2,
    d2L_dtheta2 = 6 I, K * solution == rhs to 1e-8 through add_Kx_to_y."""
    dims, model, theta_model, (w, r1, r2, r3, rhs) = rk.schur_case()
    o = KKTOracle(dims)
    assert o.factor_theta(model, theta_model, w, r1, r2, r3) == 0
    sol = o.solve_theta(model, theta_model, rhs)
    product = o.add_Kx_to_y_theta(model, theta_model, w, r1, r2, r3, sol)
    assert np.linalg.norm(product - rhs) < 1e-8  # the reference's tolerance, :362
    K = dense_kkt_matrix(dims, model, w, r1, r2, r3, theta_model)
    assert K.shape == (dims.full_dim, dims.full_dim)
    np.testing.assert_allclose(K, K.T, rtol=0, atol=1e-15)
    np.testing.assert_allclose(product, K @ sol, rtol=0, atol=1e-12)
    np.testing.assert_allclose(sol, np.linalg.solve(K, rhs), rtol=1e-9, atol=1e-11)
    # Schur complement not positive definite -> factor() is false (helpers.cpp:404-407)
    bad = rk.initialize_theta_model(dims, -50.0)
    assert o.factor_theta(model, bad, w, r1, r2, r3) == 7


def test_theta_on_a_benchmark_chain():
    """NewtonKKTProblem(n, m, T, p) of newton_kkt_benchmark.cpp:58-262 (theta variant)."""
    base = rk.newton_kkt_dims(6, 2, 10)
    dims = KKTDims(base.parents, base.children, base.sd, base.cd, base.ncd, base.ngd, base.ecd, base.egd,
                   theta_dim=4)
    model, w, r1, r2, r3, rhs, theta_model = rk.newton_kkt_problem(dims, seed=9, r2_max=1e2)
    o = KKTOracle(dims)
    assert o.factor_theta(model, theta_model, w, r1, r2, r3) == 0
    sol = o.solve_theta(model, theta_model, rhs)
    K = dense_kkt_matrix(dims, model, w, r1, r2, r3, theta_model)
    dense = np.linalg.solve(K, rhs)
    assert np.linalg.norm(sol - dense) <= 1e-8 * np.linalg.norm(dense)


def _block_case(name):
    if name == "schur":
        dims, model, theta_model, reg = rk.schur_case()
    elif name == "theta_chain":
        base = rk.newton_kkt_dims(6, 2, 10)
        dims = KKTDims(base.parents, base.children, base.sd, base.cd, base.ncd, base.ngd, base.ecd, base.egd,
                       theta_dim=4)
        model, w, r1, r2, r3, rhs, theta_model = rk.newton_kkt_problem(dims, seed=9, r2_max=1e2)
        reg = (w, r1, r2, r3, rhs)
    else:
        dims, model, reg = rk.reference_case(name)
        theta_model = None
    return dims, model, theta_model, reg


@pytest.mark.parametrize("name", sorted(rk.REFERENCE_CASES) + ["schur", "theta_chain"])
def test_block_operators_match_dense_blocks_and_compose_to_K(name):
    """add_Hx / Cx / CTx / Gx / GTx_to_y (helpers.hpp:20-24, bodies helpers.cpp:978-1368) on the
    reference's four CallbackProvider cases (tests/variable_dimensions_test.cpp:265-363) and a
    benchmark chain with theta: each equals its dense block times x, accumulates (y += ...), and the
    five together with the regularization diagonal are add_Kx_to_y (helpers.cpp:953-976)."""
    from oracle.kkt import dense_kkt_blocks
    dims, model, theta_model, (w, r1, r2, r3, _) = _block_case(name)
    o = KKTOracle(dims)
    H, C, G = dense_kkt_blocks(dims, model, theta_model)
    rng = np.random.default_rng(3)
    xd, yd, zd = dims.x_dim + dims.p, dims.y_dim, dims.z_dim
    vx, vy, vz = rng.standard_normal(xd), rng.standard_normal(yd), rng.standard_normal(zd)
    mats = {"Hx": (H, vx), "Cx": (C, vx), "CTx": (C.T, vy), "Gx": (G, vx), "GTx": (G.T, vz)}
    outs = {}
    for op, (mat, vec) in mats.items():
        y0 = rng.standard_normal(mat.shape[0])
        got = o.add_block_to_y(op, model, vec, y=y0, theta_model=theta_model)
        np.testing.assert_allclose(got, y0 + mat @ vec, rtol=0, atol=1e-12, err_msg=op)
        outs[op] = got - y0
    # composition: K [vx; vy; vz]
    full = np.concatenate([vx, vy, vz])
    want_x = outs["Hx"] + outs["CTx"] + outs["GTx"] + r1 * vx
    want = np.concatenate([want_x, outs["Cx"] - r2 * vy, outs["Gx"] - (w + r3) * vz])
    if dims.p > 0:
        got = o.add_Kx_to_y_theta(model, theta_model, w, r1, r2, r3, full)
    else:
        got = o.add_Kx_to_y(model, w, r1, r2, r3, full)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
