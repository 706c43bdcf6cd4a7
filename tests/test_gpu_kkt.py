"""GPU parity of the batched Newton-KKT step (include/sip_kkt_amd.h) against
the CPU oracle (oracle/kkt_oracle.c) and the reference's own property
(tests/variable_dimensions_test.cpp:135-181: K * solution == rhs to 1e-9).

Tolerance: fp64 on both sides; the GPU accumulates each output element in the
reference's order but with fused multiply-adds, the oracle without
(-ffp-contract=off): agreement is required to 1e-9 relative.
"""
import numpy as np
import pytest
import torch

from oracle.kkt import KKTOracle, dense_kkt_matrix
from tests import reference_kkt_problems as rk

pytestmark = pytest.mark.gpu
REL = 1e-9


def _make(dims, batch):
    from sip_optimal_control_amd import BatchedNewtonKKT
    return BatchedNewtonKKT(dims.parents, dims.children, dims.sd, dims.cd, dims.ncd, dims.ngd, dims.ecd, dims.egd,
                            batch=batch, root=dims.root, theta_dim=dims.p)


def _dev(*arrays):
    return [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda() for a in arrays]


def _batchify(batch, *arrays):
    """batch copies with a problem-dependent scaling so that problems differ."""
    out = []
    for a in arrays:
        out.append(np.stack([a * (1.0 + 0.01 * p) for p in range(batch)]))
    return out


@pytest.mark.parametrize("name", sorted(rk.REFERENCE_CASES))
def test_reference_cases(name):
    dims, model, (w, r1, r2, r3, rhs) = rk.reference_case(name)
    batch = 3
    model_b, w_b, r1_b, r2_b, r3_b, rhs_b = _batchify(batch, model, w, r1, r2, r3, rhs)
    kkt = _make(dims, batch)
    assert (kkt.x_dim, kkt.y_dim, kkt.z_dim, kkt.model_len) == (dims.x_dim, dims.y_dim, dims.z_dim, dims.model_len)
    assert kkt.kernel_name == "tree:general + staged condensation"
    d = _dev(model_b, w_b, r1_b, r2_b, r3_b, rhs_b)
    status = kkt.factor(*d[:5])
    assert status.cpu().tolist() == [0] * batch
    sol = kkt.solve(d[0], d[5])
    prod = kkt.add_Kx_to_y(*d[:5], sol).cpu().numpy()
    sol = sol.cpu().numpy()
    o = KKTOracle(dims)
    for p in range(batch):
        assert o.factor(model_b[p], w_b[p], r1_b[p], r2_b[p], r3_b[p]) == 0
        ref = o.solve(model_b[p], rhs_b[p])
        assert np.abs(sol[p] - ref).max() <= REL * max(1.0, np.abs(ref).max())
        assert np.linalg.norm(prod[p] - rhs_b[p]) < 1e-9  # the reference's own tolerance
        K = dense_kkt_matrix(dims, model_b[p], w_b[p], r1_b[p], r2_b[p], r3_b[p])
        np.testing.assert_allclose(prod[p], K @ sol[p], rtol=0, atol=1e-12)
    # the fused entry point gives the same answer
    sol2, status2 = kkt.factor_solve(*d)
    assert status2.cpu().tolist() == [0] * batch
    np.testing.assert_allclose(sol2.cpu().numpy(), sol, rtol=1e-12, atol=1e-13)
    # second right-hand side on the same factorization (tests/lqr_test.cpp:431-450 style)
    kkt.factor(*d[:5])
    rhs2 = np.cos(np.arange(batch * dims.kkt_dim)).reshape(batch, -1)
    sol3 = kkt.solve(d[0], _dev(rhs2)[0]).cpu().numpy()
    for p in range(batch):
        assert o.factor(model_b[p], w_b[p], r1_b[p], r2_b[p], r3_b[p]) == 0
        ref = o.solve(model_b[p], rhs2[p])
        assert np.abs(sol3[p] - ref).max() <= REL * max(1.0, np.abs(ref).max())


def test_offsets_match_the_reference_ordering():
    dims, _, _ = rk.reference_case("chain_node_edge_constraints")
    kkt = _make(dims, 1)
    from sip_optimal_control_amd.kkt import EDGE_BLOCKS, NODE_BLOCKS, VECTOR_TABLES
    for t, name in enumerate(VECTOR_TABLES):
        count = dims.E if name in ("x_control", "y_edge_c", "z_edge") else dims.N
        assert [kkt.vector_offset(t, i) for i in range(count)] == dims.off[name][:count]
        with pytest.raises(IndexError):
            kkt.vector_offset(t, count)
    for b, name in enumerate(NODE_BLOCKS):
        assert [kkt.model_offset(b, i) for i in range(dims.N)] == dims.node_off[name]
    for b, name in enumerate(EDGE_BLOCKS):
        assert [kkt.model_offset(len(NODE_BLOCKS) + b, e) for e in range(dims.E)] == dims.edge_off[name]


@pytest.mark.parametrize("n,m,T,batch", [(4, 2, 16, 64), (12, 4, 50, 32), (8, 3, 20, 16), (6, 2, 12, 8),
                                         (16, 4, 10, 8), (13, 5, 6, 4), (20, 3, 5, 3)])
def test_newton_kkt_benchmark_shapes(n, m, T, batch):
    """Uniform chains (benchmarks/newton_kkt_benchmark.cpp:58-83): packed chain layout; the Riccati
    part on an exact fused kernel (13,5 among them) or on the general engine (20,3).  n + m > 16
    exercises the multi-tile rank updates of the chain condensation kernel."""
    dims = rk.newton_kkt_dims(n, m, T)
    arrays = rk.newton_kkt_problem(dims, seed=100 * n + m, batch=batch, r2_max=1e2)
    kkt = _make(dims, batch)
    assert kkt.kernel_name.startswith("chain:")
    d = _dev(*arrays)
    sol, status = kkt.factor_solve(*d)
    assert status.cpu().tolist() == [0] * batch
    sol = sol.cpu().numpy()
    o = KKTOracle(dims)
    ref, ref_status = o.batch(*arrays, threads=4)
    assert ref_status.tolist() == [0] * batch
    scale = np.abs(ref).max(axis=1, keepdims=True)
    assert (np.abs(sol - ref) / scale).max() <= REL
    # split entry points agree with the fused one
    kkt.factor(*d[:5])
    sol_split = kkt.solve(d[0], d[5]).cpu().numpy()
    assert (np.abs(sol_split - ref) / scale).max() <= REL
    # K * sol == rhs through the GPU operator
    prod = kkt.add_Kx_to_y(*d[:5], _dev(sol)[0]).cpu().numpy()
    for p in (0, batch - 1):
        K = dense_kkt_matrix(dims, *[a[p] for a in arrays[:5]])
        bound = 1e-9 * np.linalg.norm(K, np.inf) * np.abs(sol[p]).max()
        assert np.abs(prod[p] - arrays[5][p]).max() <= bound
        np.testing.assert_allclose(prod[p], K @ sol[p], rtol=0, atol=bound)


def test_full_benchmark_regularization_range():
    """r2 log-uniform in [1e-3, 1e9] as the benchmark draws it (:249-262): compare with the oracle."""
    dims = rk.newton_kkt_dims(12, 4, 50)
    batch = 16
    arrays = rk.newton_kkt_problem(dims, seed=3, batch=batch)
    kkt = _make(dims, batch)
    sol, status = kkt.factor_solve(*_dev(*arrays))
    ref, ref_status = KKTOracle(dims).batch(*arrays, threads=4)
    assert status.cpu().tolist() == ref_status.tolist() == [0] * batch
    scale = np.abs(ref).max(axis=1, keepdims=True)
    assert (np.abs(sol.cpu().numpy() - ref) / scale).max() <= 1e-8


def test_false_paths_and_status_codes():
    dims = rk.newton_kkt_dims(4, 2, 6)
    batch = 6
    model, w, r1, r2, r3, rhs = rk.newton_kkt_problem(dims, seed=11, batch=batch, r2_max=1e2)
    r2[1, dims.off["y_edge_c"][2]] = 0.0          # helpers.cpp:281-286
    r2[2, dims.off["y_dyn"][3] + 1] = -2.0        # helpers.cpp:256-261
    w[3, dims.off["z_edge"][1]] = -r3[3, dims.off["z_edge"][1]]  # w + r3 == 0, :289-296
    nodes, edges = dims.unpack_model(model[4])
    edges[1]["d2L_du2"] = -5.0 * np.eye(dims.cd[1])  # G not positive definite -> G_FACTORIZATION_FAILURE
    model[4] = dims.pack_model(nodes, edges)
    kkt = _make(dims, batch)
    d = _dev(model, w, r1, r2, r3, rhs)
    sentinel = torch.full((batch, dims.kkt_dim), 7.0, dtype=torch.float64, device="cuda")
    sol, status = kkt.factor_solve(*d, sol=sentinel.clone())
    ref, ref_status = KKTOracle(dims).batch(model, w, r1, r2, r3, rhs)
    assert status.cpu().tolist() == ref_status.tolist() == [0, 5, 5, 5, 3, 0]
    sol = sol.cpu().numpy()
    for p in range(batch):
        if ref_status[p] == 0:
            assert np.abs(sol[p] - ref[p]).max() <= REL * np.abs(ref[p]).max()
        else:
            assert (sol[p] == 7.0).all()  # untouched
    # split path reports the same statuses
    assert kkt.factor(*d[:5]).cpu().tolist() == [0, 5, 5, 5, 3, 0]


def test_invalid_input_is_latched():
    from sip_optimal_control_amd import BatchedNewtonKKT
    kkt = BatchedNewtonKKT([0, 0], [1, 2], [2, 1, 3], [1, 2], edge_c_dims=[-1, 1], batch=2)
    assert kkt.input_status == 6 and kkt.kkt_dim == 0
    assert kkt.factor(None, None, None, None, None).cpu().tolist() == [6, 6]
    dag = BatchedNewtonKKT([0, 1], [2, 2], [2, 1, 3], [1, 2], batch=2)  # variable_dimensions_test.cpp:210-214
    assert dag.input_status == 4
    assert dag.factor(None, None, None, None, None).cpu().tolist() == [4, 4]


def test_null_constraint_dims_and_single_node():
    from sip_optimal_control_amd import BatchedNewtonKKT
    dims = rk.KKTDims([0, 1], [1, 2], [2, 1, 3], [1, 2])
    model = rk.initialize_model(dims)
    w, r1, r2, r3, rhs = rk.regularization(dims)
    kkt = BatchedNewtonKKT(dims.parents, dims.children, dims.sd, dims.cd, batch=1)
    assert kkt.z_dim == 0
    d = _dev(model[None], r1[None], r2[None], rhs[None])
    sol, status = kkt.factor_solve(d[0], None, d[1], d[2], None, d[3])
    o = KKTOracle(dims)
    assert o.factor(model, w, r1, r2, r3) == 0 and status.cpu().tolist() == [0]
    np.testing.assert_allclose(sol.cpu().numpy()[0], o.solve(model, rhs), rtol=1e-10, atol=1e-12)
    # one node, no edges: x = -(Q + r1 + ...)^-1-type solve through the root formulas
    one = rk.KKTDims([], [], [3], [], node_c=[1], node_g=[2])
    model1 = rk.initialize_model(one)
    w1, r11, r21, r31, rhs1 = rk.regularization(one)
    k1 = BatchedNewtonKKT([], [], [3], [], [1], [2], batch=1)
    d1 = _dev(model1[None], w1[None], r11[None], r21[None], r31[None], rhs1[None])
    sol1, st1 = k1.factor_solve(*d1)
    K = dense_kkt_matrix(one, model1, w1, r11, r21, r31)
    assert st1.cpu().tolist() == [0]
    np.testing.assert_allclose(sol1.cpu().numpy()[0], np.linalg.solve(K, rhs1), rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("n,m,T", [(12, 4, 9), (8, 4, 6), (4, 4, 5)])
def test_packed_condensation_gives_the_same_bits(monkeypatch, n, m, T):
    """Where the sweep has the symmetric-packed split kernel, the fused step's condensation writes Q_mod and R_mod as
    packed lower triangles (it computes only the lower tiles anyway; helpers.cpp:155-158, 353-360 mirror them) and
    the sweep reads them with SIP_LQR_LAYOUT_SYMMETRIC: the same arithmetic on the same numbers as the full squares
    (SIP_KKT_SYM=0), so the same bits -- and, as every path, the oracle's solution to 1e-9."""
    dims = rk.newton_kkt_dims(n, m, T)
    batch = 7
    arrays = rk.newton_kkt_problem(dims, seed=33, batch=batch, r2_max=1e2)
    d = _dev(*arrays)
    packed = _make(dims, batch)
    assert "(A|B in place, Q|R packed)" in packed.kernel_name
    sol_p, st_p = packed.factor_solve(*d)
    sol_p = sol_p.clone()
    monkeypatch.setenv("SIP_KKT_SYM", "0")
    full = _make(dims, batch)
    assert "(A|B in place)" in full.kernel_name and "packed" not in full.kernel_name
    sol_f, st_f = full.factor_solve(*d)
    torch.cuda.synchronize()
    assert st_p.cpu().tolist() == [0] * batch == st_f.cpu().tolist()
    assert torch.equal(sol_p, sol_f)
    ref, ref_status = KKTOracle(dims).batch(*arrays, threads=4)
    assert (ref_status == 0).all()
    assert float(np.abs(sol_p.cpu().numpy() - ref).max() / np.abs(ref).max()) <= 1e-9


@pytest.mark.parametrize("n,m", [(10, 5), (14, 2), (7, 7), (15, 1), (2, 8), (12, 3), (8, 1), (4, 3), (6, 1)])
def test_in_place_jacobians_outside_the_benchmark_grid(monkeypatch, n, m):
    """the step reads ddyn_dx | ddyn_du in place for every staged shape whose A | B block is a whole
    number of 16-byte pieces, not only the benchmark grid -- the same bits as the copying step."""
    dims = rk.newton_kkt_dims(n, m, 6)
    batch = 4
    arrays = rk.newton_kkt_problem(dims, seed=3 * n + m, batch=batch, r2_max=1e2)
    d = _dev(*arrays)
    kkt = _make(dims, batch)
    assert "(A|B in place" in kkt.kernel_name
    sol, st = kkt.factor_solve(*d)
    assert st.cpu().tolist() == [0] * batch
    monkeypatch.setenv("SIP_KKT_SPLIT", "0")
    copying = _make(dims, batch)
    assert "(A|B in place" not in copying.kernel_name
    assert torch.equal(copying.factor_solve(*d)[0], sol)
    ref, ref_status = KKTOracle(dims).batch(*arrays)
    assert (ref_status == 0).all()
    assert (np.abs(sol.cpu().numpy() - ref) / np.abs(ref).max(axis=1, keepdims=True)).max() <= REL


def test_condensation_variants_agree_bitwise(monkeypatch):
    """The table-driven LDS-staged kernels and the direct kernels (fallback for items too large for
    LDS) accumulate every element in the same order with the same operations; the uniform-chain
    kernels (matrix-pipe rank updates) agree with them to rounding."""
    dims = rk.newton_kkt_dims(6, 2, 9)
    batch = 5
    arrays = rk.newton_kkt_problem(dims, seed=21, batch=batch, r2_max=1e2)
    d = _dev(*arrays)
    sols = {}
    for variant in ("", "tables", "direct"):
        monkeypatch.setenv("SIP_KKT_VARIANT", variant)
        kkt = _make(dims, batch)
        assert {"": "chain condensation", "tables": "staged condensation",
                "direct": "direct condensation"}[variant] in kkt.kernel_name
        sol, st = kkt.factor_solve(*d)
        assert st.cpu().tolist() == [0] * batch
        kkt.factor(*d[:5])
        sols[variant] = (sol.clone(), kkt.solve(d[0], d[5]).clone())
    assert torch.equal(sols["tables"][0], sols["direct"][0]) and torch.equal(sols["tables"][1], sols["direct"][1])
    # the fused step of a chain plan reads ddyn_dx | ddyn_du in place (sip_lqr_factor_solve_split) instead of
    # copying them into the sweep's inputs (helpers.cpp:365-366): the same arithmetic on the same numbers
    monkeypatch.setenv("SIP_KKT_VARIANT", "")
    assert "(A|B in place)" in _make(dims, batch).kernel_name
    monkeypatch.setenv("SIP_KKT_SPLIT", "0")
    copying = _make(dims, batch)
    assert "(A|B in place)" not in copying.kernel_name
    assert torch.equal(copying.factor_solve(*d)[0], sols[""][0])
    monkeypatch.delenv("SIP_KKT_SPLIT")
    for pair in (sols[""], sols["tables"]):  # fused and split entry points (the split solve is the
        tol = 1e-12 * pair[0].abs().amax(dim=1, keepdim=True)  # vector-only sweep: same to rounding)
        assert bool(((pair[0] - pair[1]).abs() <= tol).all())
    # the chain kernels sum the rank updates on the matrix pipe: same to rounding
    scale = sols["tables"][0].abs().amax(dim=1, keepdim=True)
    assert float(((sols[""][0] - sols["tables"][0]).abs() / scale).max()) <= 1e-12
    # interior-node constraints too, odd dimensions (scalar LDS paths of the chain kernels)
    T = 5
    odd = rk.KKTDims(list(range(T)), list(range(1, T + 1)), [5] * (T + 1), [3] * T, node_c=[1] * T + [2],
                     node_g=[3] * T + [0], edge_c=[3] * T, edge_g=[1] * T)
    arrays = rk.newton_kkt_problem(odd, seed=4, batch=3, r2_max=1e2)
    d = _dev(*arrays)
    ref, ref_status = KKTOracle(odd).batch(*arrays)
    got = {}
    for variant in ("", "direct"):
        monkeypatch.setenv("SIP_KKT_VARIANT", variant)
        kkt = _make(odd, 3)
        got[variant], st = kkt.factor_solve(*d)
        assert st.cpu().tolist() == ref_status.tolist() == [0, 0, 0]
    assert float(((got[""] - got["direct"]).abs() / got["direct"].abs().amax(dim=1, keepdim=True)).max()) <= 1e-12
    assert (np.abs(got[""].cpu().numpy() - ref) / np.abs(ref).max(axis=1, keepdims=True)).max() <= REL
    # a branching tree with variable dimensions through the table-driven variants
    tdims, model, (w, r1, r2, r3, rhs) = rk.reference_case("branch_sibling_edges")
    td = _dev(model[None], w[None], r1[None], r2[None], r3[None], rhs[None])
    sol_d, _ = _make(tdims, 1).factor_solve(*td)
    monkeypatch.delenv("SIP_KKT_VARIANT")
    sol_s, _ = _make(tdims, 1).factor_solve(*td)
    assert torch.equal(sol_d, sol_s)


def test_pipelined_condensation_is_bitwise_the_one_stage_kernel(monkeypatch):
    """SIP_KKT_PIPE = stages per wavefront of the software-pipelined condensation (0 = one stage per
    wavefront): same arithmetic, so the solutions are bit-identical -- also when batch x stages is
    not a multiple of the stages per wavefront, and with a stage image of an odd number of 16-byte
    pieces per lane."""
    for (n, m, T, batch) in ((6, 2, 9, 5), (12, 4, 10, 7), (4, 4, 3, 3)):
        dims = rk.newton_kkt_dims(n, m, T)
        arrays = rk.newton_kkt_problem(dims, seed=5, batch=batch, r2_max=1e2)
        d = _dev(*arrays)
        sols = {}
        for pipe in ("0", "1", "4", "6", "1000"):
            monkeypatch.setenv("SIP_KKT_PIPE", pipe)
            kkt = _make(dims, batch)
            sol, st = kkt.factor_solve(*d)
            assert st.cpu().tolist() == [0] * batch
            kkt.factor(*d[:5])
            sols[pipe] = (sol.clone(), kkt.solve(d[0], d[5]).clone())
        for pipe in ("1", "4", "6", "1000"):
            assert torch.equal(sols[pipe][0], sols["0"][0]) and torch.equal(sols[pipe][1], sols["0"][1]), (n, m, pipe)


def test_theta_schur_reference_case():
    """CallbackProvider.SolvesBranchedSystemWithSchurVariables (variable_dimensions_test.cpp:338-363):
    theta_dim = 2 on the branched tree; K * solution == rhs to 1e-8, and agreement with the oracle."""
    dims, model, theta_model, (w, r1, r2, r3, rhs) = rk.schur_case()
    batch = 3
    mb, tb, wb, r1b, r2b, r3b, rb = _batchify(batch, model, theta_model, w, r1, r2, r3, rhs)
    kkt = _make(dims, batch)
    assert kkt.theta_len == dims.theta_len and kkt.full_dim == dims.full_dim
    from oracle.kkt import THETA_EDGE_BLOCKS, THETA_NODE_BLOCKS
    for b, name in enumerate(THETA_NODE_BLOCKS):
        assert [kkt.theta_offset(b, i) for i in range(dims.N)] == dims.theta_node_off[name]
    for b, name in enumerate(THETA_EDGE_BLOCKS):
        assert [kkt.theta_offset(len(THETA_NODE_BLOCKS) + b, e) for e in range(dims.E)] == dims.theta_edge_off[name]
    d = _dev(mb, tb, wb, r1b, r2b, r3b, rb)
    assert kkt.factor_theta(*d[:6]).cpu().tolist() == [0] * batch
    sol = kkt.solve_theta(d[0], d[1], d[6])
    prod = kkt.add_Kx_to_y_theta(*d[:6], sol).cpu().numpy()
    sol = sol.cpu().numpy()
    o = KKTOracle(dims)
    for p in range(batch):
        assert o.factor_theta(mb[p], tb[p], wb[p], r1b[p], r2b[p], r3b[p]) == 0
        ref = o.solve_theta(mb[p], tb[p], rb[p])
        assert np.abs(sol[p] - ref).max() <= REL * max(1.0, np.abs(ref).max())
        assert np.linalg.norm(prod[p] - rb[p]) < 1e-8  # the reference's tolerance, :362
        K = dense_kkt_matrix(dims, mb[p], wb[p], r1b[p], r2b[p], r3b[p], tb[p])
        np.testing.assert_allclose(prod[p], K @ sol[p], rtol=0, atol=1e-12)
    # Schur complement not positive definite on one problem -> status 7, its sol untouched
    bad = np.stack([tb[0], rk.initialize_theta_model(dims, -50.0), tb[2]])
    db = _dev(bad)[0]
    assert kkt.factor_theta(d[0], db, *d[2:6]).cpu().tolist() == [0, 7, 0]
    sentinel = torch.full((batch, dims.full_dim), 3.0, dtype=torch.float64, device="cuda")
    out = kkt.solve_theta(d[0], db, d[6], sol=sentinel).cpu().numpy()
    assert (out[1] == 3.0).all() and np.abs(out[0] - sol[0]).max() <= 1e-12


@pytest.mark.parametrize("n,m,T,p", [(6, 2, 10, 4), (12, 4, 20, 8), (4, 1, 5, 9)])
def test_theta_on_benchmark_chains(n, m, T, p):
    """NewtonKKTProblem(n, m, T, p) (newton_kkt_benchmark.cpp:58-262, the Theta benchmarks)."""
    base = rk.newton_kkt_dims(n, m, T)
    dims = rk.KKTDims(base.parents, base.children, base.sd, base.cd, base.ncd, base.ngd, base.ecd, base.egd,
                      theta_dim=p)
    batch = 4
    model, w, r1, r2, r3, rhs, theta_model = rk.newton_kkt_problem(dims, seed=7 + n, batch=batch, r2_max=1e2)
    kkt = _make(dims, batch)
    d = _dev(model, theta_model, w, r1, r2, r3, rhs)
    assert kkt.factor_theta(*d[:6]).cpu().tolist() == [0] * batch
    sol = kkt.solve_theta(d[0], d[1], d[6]).cpu().numpy()
    o = KKTOracle(dims)
    for q in range(batch):
        assert o.factor_theta(model[q], theta_model[q], w[q], r1[q], r2[q], r3[q]) == 0
        ref = o.solve_theta(model[q], theta_model[q], rhs[q])
        assert np.abs(sol[q] - ref).max() <= 1e-8 * np.abs(ref).max()


@pytest.mark.parametrize("case", ["family", "odd", "one_edge", "wide_theta", "many_columns"])
def test_fused_theta_passes_equal_the_generic_ones(monkeypatch, case):
    """Uniform chains run the theta Schur complement in fused passes (kkt_theta_chain_kernels.hpp: J_theta is read
    from the theta arena where it is needed and never assembled; the Schur complement is summed from stage partials).
    Same results as the generic passes (J_theta in memory) to rounding and as the oracle to 1e-8; a failed problem
    keeps its status and its solution untouched."""
    if case == "family":
        base = rk.newton_kkt_dims(8, 3, 7)
        kw = dict(node_c=base.ncd, node_g=base.ngd, edge_c=base.ecd, edge_g=base.egd)
        sd, cd, T, p = base.sd, base.cd, 7, 8
    elif case == "odd":  # interior-node constraints, odd dimensions, odd p: the unaligned copy paths
        T, p = 5, 3
        sd, cd = [5] * (T + 1), [3] * T
        kw = dict(node_c=[1] * T + [2], node_g=[3] * T + [0], edge_c=[3] * T, edge_g=[1] * T)
    elif case == "one_edge":
        T, p = 1, 2
        sd, cd = [4] * 2, [2]
        kw = dict(node_c=[2, 1], node_g=[0, 2], edge_c=[1], edge_g=[2])
    elif case == "many_columns":  # more (column, row) pairs than the lanes prefetch (2 x 64): the tail loops
        T, p = 3, 20
        sd, cd = [6] * (T + 1), [2] * T
        kw = dict(node_c=[1] * T + [3], node_g=[2] * T + [4], edge_c=[3] * T, edge_g=[4] * T)
    else:  # more columns than one multi-rhs launch carries, more (a, b) pairs than lanes
        T, p = 4, 11
        sd, cd = [6] * (T + 1), [2] * T
        kw = dict(node_c=[0] * T + [3], node_g=[0] * T + [4], edge_c=[3] * T, edge_g=[4] * T)
    dims = rk.KKTDims(list(range(T)), list(range(1, T + 1)), sd, cd, theta_dim=p, **kw)
    batch = 5
    model, w, r1, r2, r3, rhs, theta_model = rk.newton_kkt_problem(dims, seed=11, batch=batch, r2_max=1e2)
    theta_model[2] = rk.initialize_theta_model(dims, -50.0)  # an indefinite Schur complement: status 7
    d = _dev(model, theta_model, w, r1, r2, r3, rhs)
    got, prods = {}, {}
    for fused in ("1", "0"):
        monkeypatch.setenv("SIP_KKT_THETA_FUSED", fused)
        kkt = _make(dims, batch)
        assert ("fused theta passes" in kkt.kernel_name) == (fused == "1")
        st = kkt.factor_theta(*d[:6]).cpu().tolist()
        assert st == [0, 0, 7, 0, 0]
        sentinel = torch.full((batch, dims.full_dim), 3.0, dtype=torch.float64, device="cuda")
        got[fused] = kkt.solve_theta(d[0], d[1], d[6], sol=sentinel).cpu().numpy()
        assert (got[fused][2] == 3.0).all()
        # y += K x with the theta blocks (stage-parallel for chains, apply_theta_chain_kernel), all blocks and one by one
        xs = _dev(np.random.default_rng(5).standard_normal((batch, dims.full_dim)))[0]
        prods[fused] = [kkt.add_Kx_to_y_theta(*d[:6], xs).cpu().numpy()]
        for op, (src, dst) in kkt.BLOCK_SPACES.items():
            xv = _dev(np.random.default_rng(6).standard_normal((batch, kkt.space_dim(src, True))))[0]
            prods[fused].append(kkt.add_block_to_y(op, d[0], xv, theta_model=d[1]).cpu().numpy())
    scale = np.abs(got["0"]).max(axis=1, keepdims=True)
    assert (np.abs(got["1"] - got["0"]) / scale).max() <= 1e-11
    for a, b in zip(prods["1"], prods["0"]):
        assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max())
    o = KKTOracle(dims)
    for q in (0, 1, 3, 4):
        assert o.factor_theta(model[q], theta_model[q], w[q], r1[q], r2[q], r3[q]) == 0
        ref = o.solve_theta(model[q], theta_model[q], rhs[q])
        assert np.abs(got["1"][q] - ref).max() <= 1e-8 * np.abs(ref).max()


def test_full_size_properties():
    """Batch 4096 at the f1 benchmark shape (n=12, m=4, T=50, c=6, g=8): size-independent
    properties instead of an oracle pass -- K * sol == rhs through the GPU operator, linearity of
    solve in the right-hand side, idempotence of a repeated factor+solve, and a sampled oracle check."""
    from sip_optimal_control_amd import BatchedNewtonKKT, synthetic
    n, m, T, batch = 12, 4, 50, 4096
    c, g = n // 2, 2 * m
    dd = dict(parents=list(range(T)), children=list(range(1, T + 1)), state_dims=[n] * (T + 1),
              control_dims=[m] * T, node_c_dims=[0] * T + [c], node_g_dims=[0] * T + [g],
              edge_c_dims=[c] * T, edge_g_dims=[g] * T)
    kkt = BatchedNewtonKKT(batch=batch, **dd)
    model, w, r1, r2, r3, rhs = synthetic.make_newton_kkt_batch(kkt, seed=5, r2_max=1e2, **dd)
    sol, status = kkt.factor_solve(model, w, r1, r2, r3, rhs)
    assert int((status != 0).sum()) == 0
    sol = sol.clone()
    prod = kkt.add_Kx_to_y(model, w, r1, r2, r3, sol)
    res = (prod - rhs).norm(dim=1) / rhs.norm(dim=1)
    assert float(res.max()) < 1e-9
    # linearity: solve(a b1 + b2) == a solve(b1) + solve(b2) on the same factorization
    kkt.factor(model, w, r1, r2, r3)
    b2 = torch.roll(rhs, 1, dims=0)
    s1 = kkt.solve(model, rhs).clone()
    s2 = kkt.solve(model, b2).clone()
    s12 = kkt.solve(model, (0.5 * rhs + b2).contiguous())
    scale = (0.5 * s1 + s2).abs().amax(dim=1, keepdim=True)
    assert float(((s12 - (0.5 * s1 + s2)).abs() / scale).max()) < 1e-9
    # idempotence of the fused step
    sol_again, _ = kkt.factor_solve(model, w, r1, r2, r3, rhs)
    assert torch.equal(sol_again, sol)
    # sampled oracle check
    od = rk.newton_kkt_dims(n, m, T)
    pick = [0, 1234, batch - 1]
    ref, st = KKTOracle(od).batch(*[a[pick].cpu().numpy() for a in (model, w, r1, r2, r3, rhs)])
    assert st.tolist() == [0, 0, 0]
    got = sol[pick].cpu().numpy()
    assert (np.abs(got - ref) / np.abs(ref).max(axis=1, keepdims=True)).max() <= REL


def test_step_is_hip_graph_capturable():
    """The fused Newton-KKT step and the Riccati sweep only enqueue work on the caller's stream (no
    host synchronisation, no host-side allocation), so a Newton loop can be captured once in a
    hipGraph and replayed: new right-hand sides are picked up from the captured buffers.  Captured on
    a side stream (torch's rule), replayed both there and on the default stream -- the usual
    torch.cuda.graph flow.  (A hipMemsetAsync inside the step used to become a memset node that a
    default-stream replay skipped: csrc/stream_fill.hpp.)"""
    from sip_optimal_control_amd import BatchedChainLQR, ChainShape, synthetic
    dims = rk.newton_kkt_dims(12, 4, 20)
    batch = 16
    arrays = rk.newton_kkt_problem(dims, seed=77, batch=batch, r2_max=1e2)
    kkt = _make(dims, batch)
    d = _dev(*arrays)
    sol = torch.zeros(batch, dims.kkt_dim, dtype=torch.float64, device="cuda")
    shape = ChainShape(12, 4, 20)
    mats, vecs = synthetic.make_chain_batch(shape, batch, seed=5, device="cuda:0")
    lqr = BatchedChainLQR(12, 4, 20, batch)
    lsol, lgains = lqr.empty_sol(), lqr.empty_gains()
    oracle_kkt = KKTOracle(dims)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # eager on the side stream (also loads the code objects before capture)
        kkt.factor_solve(*d, sol=sol)
        lqr.factor_solve(mats, vecs, lsol, lgains)
    side.synchronize()
    eager_lqr = lsol.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        kkt.factor_solve(*d, sol=sol)
        lqr.factor_solve(mats, vecs, lsol, lgains)
    for stream in (side, torch.cuda.current_stream(), side, torch.cuda.current_stream()):
        with torch.cuda.stream(stream):  # a new right-hand side in the captured buffer, then replay
            d[5].copy_(torch.roll(d[5], 1, dims=0))
            lsol.zero_()
            sol.zero_()
            graph.replay()
        torch.cuda.synchronize()
        assert kkt.status.cpu().tolist() == [0] * batch
        ref, st = oracle_kkt.batch(*[a.cpu().numpy() for a in d])
        assert st.tolist() == [0] * batch
        assert (np.abs(sol.cpu().numpy() - ref) / np.abs(ref).max(axis=1, keepdims=True)).max() <= REL
        assert torch.equal(lsol, eager_lqr)


def _block_problem(name, batch):
    """(dims, [model, w, r1, r2, r3, rhs] per-batch arrays, theta_model or None)."""
    if name in rk.REFERENCE_CASES:
        dims, model, reg = rk.reference_case(name)
        return dims, _batchify(batch, model, *reg), None
    if name == "schur":
        dims, model, theta_model, reg = rk.schur_case()
        arrays = _batchify(batch, model, *reg)
        return dims, arrays, _batchify(batch, theta_model)[0]
    n, m, T, p = {"chain": (12, 4, 9, 0), "chain_theta": (6, 2, 10, 4), "chain_wide": (20, 3, 5, 0)}[name]
    dims = rk.newton_kkt_dims(n, m, T)
    if p:
        dims = rk.KKTDims(dims.parents, dims.children, dims.sd, dims.cd, dims.ncd, dims.ngd, dims.ecd, dims.egd,
                          theta_dim=p)
    out = rk.newton_kkt_problem(dims, seed=31 + n, batch=batch, r2_max=1e2)
    return dims, list(out[:6]), (out[6] if p else None)


@pytest.mark.parametrize("name", sorted(rk.REFERENCE_CASES) + ["schur", "chain", "chain_theta", "chain_wide"])
def test_block_operators(name):
    """sip_kkt_add_{Hx,Cx,CTx,Gx,GTx}_to_y (CallbackProvider::add_*x_to_y, helpers.hpp:20-24, bodies
    helpers.cpp:978-1368; the callbacks SIP gets one by one, sip_optimal_control.cpp:147-190) on the
    reference's four CallbackProvider cases (tests/variable_dimensions_test.cpp:265-363: chain,
    sibling edges, zero-dimensional root, Schur variables) and on benchmark chains (LDS-staged chain
    kernel; with theta; n > 16), against the oracle's restatement of each body: they accumulate
    (y += ...), and summed with the regularization diagonal they are add_Kx_to_y (helpers.cpp:953-976)."""
    batch = 5
    dims, arrays, theta_model = _block_problem(name, batch)
    model, w, r1, r2, r3 = arrays[:5]
    kkt = _make(dims, batch)
    if name in ("chain", "chain_theta", "chain_wide"):
        assert "chain condensation" in kkt.kernel_name
    o = KKTOracle(dims)
    rng = np.random.default_rng(17)
    theta = theta_model is not None
    d_model, d_theta = _dev(model)[0], (_dev(theta_model)[0] if theta else None)
    vec = {s: rng.standard_normal((batch, kkt.space_dim(s, theta))) for s in "xyz"}
    parts = {}
    for op, (src, dst) in kkt.BLOCK_SPACES.items():
        y0 = rng.standard_normal((batch, kkt.space_dim(dst, theta)))
        got = kkt.add_block_to_y(op, d_model, _dev(vec[src])[0], y=_dev(y0)[0], theta_model=d_theta).cpu().numpy()
        for p in range(batch):
            ref = o.add_block_to_y(op, model[p], vec[src][p], y=y0[p], theta_model=theta_model[p] if theta else None)
            scale = max(1.0, np.abs(ref).max(initial=0.0))
            assert np.abs(got[p] - ref).max(initial=0.0) <= REL * scale, (op, p)
        parts[op] = got - y0
    # the five blocks and the diagonal add up to the whole operator, as computed by the GPU's add_Kx_to_y
    full = np.concatenate([vec["x"], vec["y"], vec["z"]], axis=1)
    if theta:
        whole = kkt.add_Kx_to_y_theta(d_model, d_theta, *_dev(w, r1, r2, r3, full)).cpu().numpy()
    else:
        whole = kkt.add_Kx_to_y(d_model, *_dev(w, r1, r2, r3, full)).cpu().numpy()
    want = np.concatenate([parts["Hx"] + parts["CTx"] + parts["GTx"] + r1 * vec["x"], parts["Cx"] - r2 * vec["y"],
                           parts["Gx"] - (w + r3) * vec["z"]], axis=1)
    assert np.abs(whole - want).max() <= REL * max(1.0, np.abs(want).max())
