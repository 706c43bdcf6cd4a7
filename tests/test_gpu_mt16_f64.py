"""fp64 on the n = 32 matrix-core kernel (chain_factor_solve_mt16<32, M>/f64, v_mfma_f64_16x16x4_f64): BASELINE's C4
shape in the reference's own precision, and every 16 < n < 32, m <= 8 embedded in it (VERDICT r02 missing #2: these
ran on the general engine at ~2 % of roofline).  Tolerance as everywhere in fp64: 1e-9 max-abs relative to the
oracle's block, statuses exact."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
TOL = 1e-9


def _rel(a, b):
    scale = np.abs(b).max(axis=1, keepdims=True)
    scale[scale == 0] = 1.0
    return float((np.abs(a - b) / scale).max())


def _make(n, m, T, batch, seed):
    from sip_optimal_control_amd import ChainShape, synthetic
    return synthetic.make_chain_batch(ChainShape(n, m, T), batch, seed=seed, device="cuda:0",
                                      dtype=torch.float64, cross_term=0.01)


@pytest.mark.parametrize("n,m,T,batch", [(32, 8, 100, 6), (32, 4, 30, 5), (32, 8, 0, 3), (32, 8, 1, 2),
                                         (17, 1, 12, 4), (20, 3, 25, 7), (24, 8, 16, 3), (31, 5, 9, 5), (32, 7, 11, 3)])
def test_matches_the_oracle(oracle_lib, n, m, T, batch):
    from sip_optimal_control_amd import BatchedChainLQR
    mats, vecs = _make(n, m, T, batch, seed=3000 + 31 * n + m)
    solver = BatchedChainLQR(n, m, T, batch)
    assert "mt16" in solver.kernel_name and solver.kernel_name.count("/f64") == 1
    assert ("embedding" in solver.kernel_name) == ((n, m) not in ((32, 8), (32, 4)))
    sol, gains, status = solver.factor_solve(mats, vecs)
    torch.cuda.synchronize()
    ref_sol, ref_gains, ref_status = oracle_lib.chain_batch(n, m, T, mats.cpu().numpy(), vecs.cpu().numpy())
    np.testing.assert_array_equal(status.cpu().numpy(), ref_status)
    assert (ref_status == 0).all()
    es = _rel(sol.cpu().numpy(), ref_sol)
    eg = _rel(gains.cpu().numpy(), ref_gains) if T > 0 else 0.0
    print(f"mt16/f64 ({n},{m},T={T}): sol {es:.2e}, gains {eg:.2e}")
    assert es <= TOL and eg <= TOL
    # split entry points (they re-run the sweep): same statuses, same solution
    g2, st2 = solver.factor(mats)
    sol2 = solver.solve(mats, vecs, g2)
    sol2 = sol2[0] if isinstance(sol2, tuple) else sol2
    torch.cuda.synchronize()
    np.testing.assert_array_equal(st2.cpu().numpy(), ref_status)
    assert _rel(sol2.cpu().numpy(), ref_sol) <= TOL


def test_injected_failures_report_the_reference_status(oracle_lib):
    """FactorStatus (lqr.hpp:68-74) with injected failures, precedence at a node (G before delta before F:
    lqr.cpp:696-701, 722-727) and "first failing node in postorder", exact against the oracle."""
    from sip_optimal_control_amd import BatchedChainLQR, ChainShape
    n, m, T, batch = 32, 8, 12, 10
    shape = ChainShape(n, m, T)
    mats, vecs = _make(n, m, T, batch, seed=78)
    eye_m = torch.eye(m, dtype=torch.float64, device="cuda:0").reshape(-1)
    eye_n = torch.eye(n, dtype=torch.float64, device="cuda:0").reshape(-1)

    def R(i): o = shape.mats_off(i)["R"]; return slice(o, o + m * m)
    def Q(i): o = shape.mats_off(i)["Q"]; return slice(o, o + n * n)
    def delta(i, j): return shape.mats_off(i)["delta"] + j

    expected = [0] * batch
    mats[1, R(5)] = -1e4 * eye_m;                                    expected[1] = 3
    mats[2, delta(T, 3)] = 0.0;                                      expected[2] = 1
    mats[3, delta(4, 31)] = -1.0;                                    expected[3] = 1
    mats[4, Q(T)] = -1e4 * eye_n;                                    expected[4] = 2
    mats[5, Q(6)] = -1e6 * eye_n;                                    expected[5] = 2
    mats[6, R(5)] = -1e4 * eye_m; mats[6, delta(5, 0)] = 0.0;        expected[6] = 3
    mats[7, delta(5, 0)] = 0.0; mats[7, Q(5)] = -1e6 * eye_n;        expected[7] = 1
    mats[8, delta(9, 2)] = 0.0; mats[8, R(2)] = -1e4 * eye_m;        expected[8] = 1
    mats[9, R(8)] = -1e4 * eye_m; mats[9, delta(3, 1)] = 0.0;        expected[9] = 3
    solver = BatchedChainLQR(n, m, T, batch)
    _, _, status = solver.factor_solve(mats, vecs)
    torch.cuda.synchronize()
    _, _, ref_status = oracle_lib.chain_batch(n, m, T, mats.cpu().numpy(), vecs.cpu().numpy())
    assert list(ref_status) == expected
    np.testing.assert_array_equal(status.cpu().numpy(), ref_status)


def test_c4_shape_in_fp64_beats_the_general_engine_tenfold(oracle_lib, monkeypatch):
    """batch 4096, T = 100, n = 32, m = 8 in fp64: every status SUCCESS, sampled problems
    against the oracle, and the launch at least 10x faster than the general engine on the same data."""
    from sip_optimal_control_amd import BatchedChainLQR
    n, m, T, batch = 32, 8, 100, 4096
    mats, vecs = _make(n, m, T, batch, seed=17)
    solver = BatchedChainLQR(n, m, T, batch)
    sol, gains, status = solver.factor_solve(mats, vecs)
    torch.cuda.synchronize()
    assert bool((status == 0).all())
    pick = [0, 1, 777, 2048, 4095]
    ref_sol, ref_gains, _ = oracle_lib.chain_batch(n, m, T, mats[pick].cpu().numpy(), vecs[pick].cpu().numpy())
    assert _rel(sol[pick].cpu().numpy(), ref_sol) <= TOL and _rel(gains[pick].cpu().numpy(), ref_gains) <= TOL

    def timed(s, reps):
        s.factor_solve(mats, vecs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            s.factor_solve(mats, vecs)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    fast = timed(solver, 3)
    monkeypatch.setenv("SIP_LQR_VARIANT", "general")
    general = BatchedChainLQR(n, m, T, batch)
    assert "tree_generic" in general.kernel_name
    slow = timed(general, 1)
    print(f"C4 shape in fp64: mt16 {fast:.2f} ms, general engine {slow:.2f} ms, x{slow / fast:.1f}")
    assert slow / fast >= 10.0
