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
