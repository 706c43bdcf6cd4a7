"""sip_lqr_factor_solve_split: the fused chain sweep with the dynamics Jacobians read in place
(helpers.cpp:365-366 copies ddyn_dx / ddyn_du into the LQR inputs; the Newton-KKT step hands them over where
the model callback left them).  Same arithmetic as sip_lqr_factor_solve, so results must be bitwise equal
to it on the same problems, and within 1e-9 of the CPU oracle like every fp64 chain kernel."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

TOL = 1e-9
SPLIT_SHAPES = [(n, m) for n in (4, 6, 8, 12) for m in (1, 2, 3, 4)]


def _rel(a, b):
    scale = np.abs(b).max(axis=1, keepdims=True)
    scale[scale == 0] = 1.0
    return (np.abs(a - b) / scale).max()


def _problem(n, m, T, batch, seed):
    from sip_optimal_control_amd import BatchedChainLQR, ChainShape, synthetic
    shape = ChainShape(n, m, T)
    mats, vecs = synthetic.make_chain_batch(shape, batch, seed=seed, device="cuda:0", cross_term=0.01)
    return BatchedChainLQR(n, m, T, batch, device="cuda:0"), mats, vecs


@pytest.mark.parametrize("n,m", SPLIT_SHAPES)
@pytest.mark.parametrize("T,batch", [(50, 37), (1, 5), (7, 4), (2, 1)])
def test_split_equals_packed_and_oracle(oracle_lib, n, m, T, batch):
    solver, mats, vecs = _problem(n, m, T, batch, seed=7 * n + m + T)
    if not solver.has_split:
        pytest.skip("no split kernel for this build (tools/ab_build.sh carries (12, 4) only)")
    sol, gains, status = (t.clone() for t in solver.factor_solve(mats, vecs))
    qmr, ab = solver.split_inputs(mats)
    sol2, gains2, status2 = solver.factor_solve_split(qmr, ab, vecs)
    torch.cuda.synchronize()
    assert torch.equal(status, status2) and int(status.abs().sum()) == 0
    assert torch.equal(sol, sol2) and torch.equal(gains, gains2)  # the same instructions on the same numbers
    ref_sol, ref_gains, ref_status = oracle_lib.chain_batch(n, m, T, mats.cpu().numpy(), vecs.cpu().numpy())
    assert _rel(sol2.cpu().numpy(), ref_sol) <= TOL and _rel(gains2.cpu().numpy(), ref_gains) <= TOL


@pytest.mark.parametrize("n", list(range(1, 16)))
def test_every_staged_shape_with_whole_pieces_has_a_split_kernel(n):
    """the in-place A | B form for every staged shape n <= 15, m <= 8 whose A | B block is a whole
    number of 16-byte pieces (n (n + m) even; the others would start on odd 8-byte offsets at every other stage and
    keep the copy).  Bitwise equal to the packed sweep."""
    for m in range(1, 9):
        T, batch = 5, 6
        solver, mats, vecs = _problem(n, m, T, batch, seed=100 * n + m)
        if not BatchedChainLQR_full_build(solver):
            pytest.skip("single-kernel diagnostic build")
        assert solver.has_split == (n * (n + m) % 2 == 0), (n, m)
        if not solver.has_split:
            continue
        sol, gains, status = (t.clone() for t in solver.factor_solve(mats, vecs))
        qmr, ab = solver.split_inputs(mats)
        sol2, gains2, status2 = solver.factor_solve_split(qmr, ab, vecs)
        torch.cuda.synchronize()
        assert torch.equal(status, status2) and int(status.abs().sum()) == 0, (n, m)
        assert torch.equal(sol, sol2) and torch.equal(gains, gains2), (n, m)


def BatchedChainLQR_full_build(solver):
    return "qw16" in solver.kernel_name


def test_split_reads_strided_jacobians():
    """A | B inside a larger per-stage record of a larger per-problem arena (what the model arena of the
    Newton-KKT step looks like): strides come from the caller."""
    n, m, T, batch = 12, 4, 50, 64
    solver, mats, vecs = _problem(n, m, T, batch, seed=3)
    if not solver.has_split:
        pytest.skip("no split kernel")
    qmr, ab = solver.split_inputs(mats)
    arena = torch.full((batch, T + 3, 936), float("nan"), dtype=torch.float64, device="cuda:0")
    view = arena[:, 1:T + 1, 520:520 + n * (n + m)]
    view.copy_(ab)
    sol, gains, status = (t.clone() for t in solver.factor_solve(mats, vecs))
    sol2, gains2, status2 = solver.factor_solve_split(qmr, view, vecs)
    torch.cuda.synchronize()
    assert int(status2.abs().sum()) == 0
    assert torch.equal(sol, sol2) and torch.equal(gains, gains2)


@pytest.mark.parametrize("n,m", [(12, 3), (12, 4), (5, 3), (8, 1)])
def test_split_reads_jacobians_at_odd_offsets_and_strides(n, m):
    """A | B at places that are only 8-byte aligned -- an odd offset inside a record of odd length inside a problem
    arena of odd length, as the Newton-KKT model arena has them for odd m: a stage's A | B is a whole number of
    16-byte pieces, and the LDS-DMA copies pieces exactly from 8-byte aligned sources (tools/ubench/lds_dma_align.hip).
    Bitwise the packed sweep."""
    T, batch = 9, 11
    solver, mats, vecs = _problem(n, m, T, batch, seed=5 * n + m)
    if not solver.has_split:
        pytest.skip("no split kernel")
    qmr, ab = solver.split_inputs(mats)
    rec, off = 2 * n * (n + m) + 7, 3  # odd record length, odd offset
    per = (T + 2) * rec + (1 - ((T + 2) * rec) % 2)  # odd problem stride
    flat = torch.full((batch * per + 2,), float("nan"), dtype=torch.float64, device="cuda:0")
    base = 1 if (flat.data_ptr() // 8) % 2 == 0 else 2  # ... from a base whose A | B lands on an odd scalar
    arena = flat[base:base + batch * per].view(batch, per)[:, :(T + 2) * rec].unflatten(1, (T + 2, rec))
    view = arena[:, 1:T + 1, off:off + n * (n + m)]
    assert view.stride(0) % 2 == 1 and view.stride(1) % 2 == 1 and view.data_ptr() % 16 == 8
    view.copy_(ab)
    sol, gains, status = (t.clone() for t in solver.factor_solve(mats, vecs))
    sol2, gains2, status2 = solver.factor_solve_split(qmr, view, vecs)
    torch.cuda.synchronize()
    assert int(status2.abs().sum()) == 0
    assert torch.equal(sol, sol2) and torch.equal(gains, gains2)


def test_split_statuses():
    """Factorization failures are reported as by the packed kernel (lqr.hpp:68-74)."""
    n, m, T, batch = 12, 4, 20, 16
    solver, mats, vecs = _problem(n, m, T, batch, seed=11)
    if not solver.has_split:
        pytest.skip("no split kernel")
    stg = n * n + n + n * n + 2 * n * m + m * m
    mats[3, 5 * stg + n * n + 2] = -1.0        # delta <= 0 at node 5
    mats[7, 9 * stg: 9 * stg + n * n] *= -50.0  # indefinite Q at node 9
    r = 11 * stg + n * n + n + n * n + 2 * n * m
    mats[9, r: r + m * m] *= -80.0              # indefinite R at edge 11
    _, _, status = solver.factor_solve(mats, vecs)
    status = status.clone()
    qmr, ab = solver.split_inputs(mats)
    _, _, status2 = solver.factor_solve_split(qmr, ab, vecs)
    torch.cuda.synchronize()
    assert torch.equal(status, status2)
    assert int(status[3]) == 1 and int(status[7]) != 0 and int(status[9]) != 0


def test_unsupported_plans_say_so():
    from sip_optimal_control_amd import BatchedChainLQR
    assert not BatchedChainLQR(16, 3, 4, 2, device="cuda:0").has_split     # a direct kernel
    assert not BatchedChainLQR(5, 2, 4, 2, device="cuda:0").has_split      # n (n + m) odd: A | B not in 16-byte pieces
    assert not BatchedChainLQR(32, 8, 4, 2, dtype=torch.float32, device="cuda:0").has_split


def test_split_at_the_headline_size():
    """BASELINE.json's C3 (batch 4096, T = 50, n = 12, m = 4): the split sweep against the packed one, bitwise,
    and the property the packed kernel is checked with at this size -- u_i = K_i x_i + k_i (lqr.cpp:856-857)."""
    n, m, T, batch = 12, 4, 50, 4096
    solver, mats, vecs = _problem(n, m, T, batch, seed=5)
    if not solver.has_split:
        pytest.skip("no split kernel")
    sol, gains, status = (t.clone() for t in solver.factor_solve(mats, vecs))
    qmr, ab = solver.split_inputs(mats)
    sol2, gains2, status2 = solver.factor_solve_split(qmr, ab, vecs)
    torch.cuda.synchronize()
    assert int(status2.abs().sum()) == 0
    assert torch.equal(sol, sol2) and torch.equal(gains, gains2)
    vs = 2 * n + m
    body = sol2[:, :T * vs].reshape(batch, T, vs)
    x, u = body[:, :, :n], body[:, :, 2 * n:]
    g = gains2.reshape(batch, T, m * n + m)
    K = g[:, :, :m * n].reshape(batch, T, n, m)  # column-major m x n: element (j, c) at c * m + j
    k = g[:, :, m * n:]
    u_ref = torch.einsum("btcj,btc->btj", K, x) + k
    scale = u.abs().amax(dim=(1, 2), keepdim=True).clamp_min(1.0)
    assert float(((u - u_ref).abs() / scale).max()) <= 1e-9


def test_split_rejects_problem_strides_beyond_32_bit_offsets():
    """The four problems of a wavefront are addressed by 32-bit byte offsets from the first one's A | B
    (include/sip_lqr_amd.h): a larger stride is an error, not a silent wrap."""
    import ctypes
    n, m, T, batch = 12, 4, 3, 4
    solver, mats, vecs = _problem(n, m, T, batch, seed=9)
    if not solver.has_split:
        pytest.skip("no split kernel")
    qmr, ab = solver.split_inputs(mats)
    sol, gains = solver.empty_sol(), solver.empty_gains()
    args = lambda stride: (solver._plan, ctypes.c_void_p(qmr.data_ptr()), ctypes.c_void_p(ab.data_ptr()),
                           ctypes.c_int64(stride), ctypes.c_int64(ab.stride(1)), ctypes.c_void_p(vecs.data_ptr()),
                           ctypes.c_void_p(sol.data_ptr()), ctypes.c_void_p(gains.data_ptr()),
                           ctypes.c_void_p(solver.status.data_ptr()), ctypes.c_void_p(solver.workspace.data_ptr()),
                           ctypes.c_void_p(0))
    limit = ((1 << 32) - 1 - 8 * n * (n + m)) // 24
    assert solver._lib.sip_lqr_factor_solve_split(*args(limit + 2)) == -1   # SIP_LQR_ERR_INVALID_ARGUMENT
    assert solver._lib.sip_lqr_factor_solve_split(*args(1 << 29)) == -1
    assert solver._lib.sip_lqr_factor_solve_split(*args(ab.stride(0))) == 0
    torch.cuda.synchronize()
