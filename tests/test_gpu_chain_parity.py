"""Parity of the HIP batched chain kernel (through the C ABI) with the CPU oracle.

Tolerance (fp64): max |gpu - oracle| / max |oracle| <= 1e-9 on x,u,y (sol) and
K,k (gains), per problem -- SURVEY.md section 8(c); statuses must match exactly.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

TOL = 1e-9


def _run(n, m, T, batch, seed, cross_term=0.01):
    from sip_optimal_control_amd import BatchedChainLQR, ChainShape, synthetic
    shape = ChainShape(n, m, T)
    mats, vecs = synthetic.make_chain_batch(shape, batch, seed=seed, device="cuda:0",
                                            cross_term=cross_term)
    solver = BatchedChainLQR(n, m, T, batch, device="cuda:0")
    sol, gains, status = solver.factor_solve(mats, vecs)
    torch.cuda.synchronize()
    return shape, mats.cpu().numpy(), vecs.cpu().numpy(), sol.cpu().numpy(), \
        gains.cpu().numpy(), status.cpu().numpy()


def _rel(a, b):
    scale = np.abs(b).max(axis=1, keepdims=True)
    scale[scale == 0] = 1.0
    return (np.abs(a - b) / scale).max()


@pytest.mark.parametrize("n,m,T,batch", [
    (12, 4, 50, 64), (12, 4, 50, 61), (4, 2, 20, 1), (4, 2, 20, 7), (1, 1, 1, 5),
    (2, 1, 2, 9), (3, 2, 3, 4), (8, 3, 16, 33), (12, 4, 0, 3), (12, 4, 1, 2),
])
def test_factor_solve_matches_oracle(oracle_lib, n, m, T, batch):
    shape, mats, vecs, sol, gains, status = _run(n, m, T, batch, seed=100 + n + T)
    ref_sol, ref_gains, ref_status = oracle_lib.chain_batch(n, m, T, mats, vecs)
    assert (ref_status == 0).all()
    np.testing.assert_array_equal(status, ref_status)
    assert _rel(sol, ref_sol) <= TOL
    if T > 0:
        assert _rel(gains, ref_gains) <= TOL


@pytest.mark.parametrize("n", [4, 6, 8, 12, 16])
@pytest.mark.parametrize("m", [1, 2, 3, 4])
def test_reference_benchmark_grid_has_fused_kernels(oracle_lib, n, m):
    """The (state_dim, control_dim) grid of the reference's benchmarks (lqr_benchmark.cpp:537-545,
    newton_kkt_benchmark.cpp:264-273) on dedicated kernels; n = 16 in distributed-vector mode."""
    from sip_optimal_control_amd import BatchedChainLQR
    T, batch = 16, 13
    assert "qw16" in BatchedChainLQR(n, m, T, batch, device="cuda:0").kernel_name
    shape, mats, vecs, sol, gains, status = _run(n, m, T, batch, seed=1000 + 10 * n + m)
    ref_sol, ref_gains, ref_status = oracle_lib.chain_batch(n, m, T, mats, vecs)
    np.testing.assert_array_equal(status, ref_status)
    assert (ref_status == 0).all()
    assert _rel(sol, ref_sol) <= TOL and _rel(gains, ref_gains) <= TOL


@pytest.mark.parametrize("n,m", [(8, 8), (12, 8), (14, 4), (14, 8), (15, 4), (15, 8), (16, 8)])
def test_large_host_kernels(oracle_lib, n, m):
    """The larger instantiations (hosts of the embedding of n <= 15, m <= 8 shapes)."""
    from sip_optimal_control_amd import BatchedChainLQR
    T, batch = 12, 9
    name = BatchedChainLQR(n, m, T, batch, device="cuda:0").kernel_name
    assert f"qw16<{n},{m}," in name and "embedding" not in name
    shape, mats, vecs, sol, gains, status = _run(n, m, T, batch, seed=2000 + 10 * n + m)
    ref_sol, ref_gains, ref_status = oracle_lib.chain_batch(n, m, T, mats, vecs)
    np.testing.assert_array_equal(status, ref_status)
    assert (ref_status == 0).all()
    assert _rel(sol, ref_sol) <= TOL and _rel(gains, ref_gains) <= TOL


@pytest.mark.parametrize("lo,hi", [(1e-9, 1e-7), (1e-6, 1e-4), (1e-3, 1e-1), (1e2, 1e6)])
def test_accuracy_tracks_the_reference_algorithm_over_delta(oracle_lib, lo, hi):
    """Dual regularization from SIP's typical r2 (1e-9) to the 1e6 of the Newton-KKT benchmark's draw:
    against an independent dense KKT solve the fused kernel is as accurate as the reference's
    algorithm (the oracle) -- which itself degrades to ~1e-9 for tiny delta (W = D^-1/2 (I - F^-1)
    D^-1/2 cancels there).  Measured on MI355X: 2.9e-9 / 2.3e-9, 2.9e-12 / 2.6e-12, 1.1e-14 / 1.2e-14,
    7.4e-13 / 1.1e-12 (kernel / oracle) for the four ranges."""
    from oracle import dense_kkt
    from sip_optimal_control_amd import BatchedChainLQR, ChainShape, synthetic
    n, m, T, batch = 12, 4, 50, 6
    sh = ChainShape(n, m, T)
    mats, vecs = synthetic.make_chain_batch(sh, batch, seed=3, device="cuda:0", cross_term=0.01)
    gen = torch.Generator(device="cuda:0")
    gen.manual_seed(7)
    for i in range(T + 1):
        off = sh.mats_off(i)["delta"]
        u = torch.rand(batch, n, generator=gen, device="cuda:0", dtype=torch.float64)
        mats[:, off:off + n] = torch.exp(np.log(lo) + (np.log(hi) - np.log(lo)) * u)
    sol, gains, status = BatchedChainLQR(n, m, T, batch).factor_solve(mats, vecs)
    torch.cuda.synchronize()
    hm, hv, got = mats.cpu().numpy(), vecs.cpu().numpy(), sol.cpu().numpy()
    ref_sol, _, ref_status = oracle_lib.chain_batch(n, m, T, hm, hv)
    assert status.cpu().tolist() == ref_status.tolist() == [0] * batch
    par, ch = list(range(T)), list(range(1, T + 1))
    for p in range(2):
        blocks = dense_kkt.chain_blocks_from_packed(n, m, T, hm[p], hv[p])
        x, u_, y = dense_kkt.solve(par, ch, [n] * (T + 1), [m] * T, blocks)
        dense = np.concatenate([np.concatenate([x[i], y[i]] + ([u_[i]] if i < T else [])) for i in range(T + 1)])
        scale = np.abs(dense).max()
        e_gpu, e_ref = np.abs(got[p] - dense).max() / scale, np.abs(ref_sol[p] - dense).max() / scale
        assert e_gpu <= 3.0 * e_ref + 1e-13, (e_gpu, e_ref)


def test_kkt_residual_full_size(oracle_lib):
    """BASELINE C2 shape at full batch: KKT residual of every GPU solution."""
    from oracle import dense_kkt
    n, m, T, batch = 12, 4, 50, 1024
    shape, mats, vecs, sol, gains, status = _run(n, m, T, batch, seed=7)
    assert (status == 0).all()
    par, ch = list(range(T)), list(range(1, T + 1))
    worst = 0.0
    for p in range(0, batch, 37):
        b = dense_kkt.chain_blocks_from_packed(n, m, T, mats[p], vecs[p])
        x, u, y = dense_kkt.chain_sol_from_packed(n, m, T, sol[p])
        worst = max(worst, dense_kkt.residual_norm(par, ch, [n] * (T + 1), [m] * T, b, x, u, y))
    assert worst < 1e-9, worst


def test_full_size_linearity_and_idempotence():
    """Size-independent properties at BASELINE C3 (batch 4096): the solution is
    linear in the right-hand side (q, r, c): sol(a*vecs1 + vecs2) = a*sol(vecs1) + sol(vecs2);
    gains' K part does not depend on it; a second identical launch is bitwise identical."""
    from sip_optimal_control_amd import BatchedChainLQR, ChainShape, synthetic
    n, m, T, batch = 12, 4, 50, 4096
    shape = ChainShape(n, m, T)
    mats, v1 = synthetic.make_chain_batch(shape, batch, seed=31, device="cuda:0")
    _, v2 = synthetic.make_chain_batch(shape, batch, seed=32, device="cuda:0")
    solver = BatchedChainLQR(n, m, T, batch)
    s1, g1, st = solver.factor_solve(mats, v1)
    s1, g1 = s1.clone(), g1.clone()
    assert bool((st == 0).all())
    s1b, g1b, _ = solver.factor_solve(mats, v1)
    assert torch.equal(s1, s1b) and torch.equal(g1, g1b)          # idempotent, deterministic
    s2 = solver.factor_solve(mats, v2)[0].clone()
    g2 = solver.empty_gains()
    s3 = solver.factor_solve(mats, (2.5 * v1 + v2).contiguous(), gains=g2)[0]
    torch.cuda.synchronize()
    scale = s3.abs().amax(dim=1, keepdim=True)
    assert float(((s3 - (2.5 * s1 + s2)).abs() / scale).max()) < 1e-9
    K1 = g1.view(batch, T, shape.gain)[:, :, :m * n]
    K2 = g2.view(batch, T, shape.gain)[:, :, :m * n]
    assert torch.equal(K1, K2)                                      # K depends on mats only


def test_empty_horizon_and_single_problem():
    """T = 0 (root only) and batch = 1 edge cases through the fused kernel."""
    from sip_optimal_control_amd import BatchedChainLQR, ChainShape, synthetic
    from oracle import oracle
    for (n, m, T, batch) in [(12, 4, 0, 1), (4, 2, 0, 5), (12, 4, 1, 1)]:
        shape = ChainShape(n, m, T)
        mats, vecs = synthetic.make_chain_batch(shape, batch, seed=5, device="cuda:0")
        solver = BatchedChainLQR(n, m, T, batch)
        sol, gains, status = solver.factor_solve(mats, vecs)
        torch.cuda.synchronize()
        ref_sol, _, ref_status = oracle.chain_batch(n, m, T, mats.cpu().numpy(), vecs.cpu().numpy())
        np.testing.assert_array_equal(status.cpu().numpy(), ref_status)
        assert _rel(sol.cpu().numpy(), ref_sol) <= TOL


@pytest.mark.parametrize("n,m", [(12, 4), (6, 3), (11, 3), (12, 6), (13, 5), (15, 7)])
def test_odd_and_large_shapes_at_full_batch(oracle_lib, n, m):
    """BASELINE's headline shape (12, 4) -- every one of the 4096 problems against the oracle, not a sample
    (VERDICT r02 weak #1a) -- and the LDS-staged kernels of odd dimensions (16-byte pieces from 8-byte-aligned sources, 4-byte pieces for
    the gains / the terminal delta) and of the shapes that keep two rollout buffers, at BASELINE's batch and
    horizon: every problem against the oracle, and u_i = K_i x_i + k_i (lqr.cpp:856-857)."""
    T, batch = 50, 4096
    shape, mats, vecs, sol, gains, status = _run(n, m, T, batch, seed=77 + 3 * n + m)
    from sip_optimal_control_amd import BatchedChainLQR
    assert "staged" in BatchedChainLQR(n, m, T, batch, device="cuda:0").kernel_name
    ref_sol, ref_gains, ref_status = oracle_lib.chain_batch(n, m, T, mats, vecs)
    np.testing.assert_array_equal(status, ref_status)
    assert (ref_status == 0).all()
    assert _rel(sol, ref_sol) <= TOL and _rel(gains, ref_gains) <= TOL
    vs = 2 * n + m
    body = sol[:, :T * vs].reshape(batch, T, vs)
    x, u = body[:, :, :n], body[:, :, 2 * n:]
    g = gains.reshape(batch, T, m * n + m)
    K, k = g[:, :, :m * n].reshape(batch, T, n, m), g[:, :, m * n:]  # K column-major m x n: (j, c) at c * m + j
    u_ref = np.einsum("btcj,btc->btj", K, x) + k
    assert (np.abs(u - u_ref) / np.maximum(np.abs(u).max(axis=(1, 2), keepdims=True), 1.0)).max() <= 1e-9
