"""The general GPU engine behind the chain C ABI: shapes / dtypes without a
dedicated kernel, the split factor()/solve() entry points, fp32 (BASELINE C4).

fp64 tolerance: 1e-9 max-abs relative (as everywhere).  fp32 is judged the way
SURVEY.md 8(c) prescribes: KKT residual of the fp32 solution, evaluated in
fp64, relative to the norm of the KKT right-hand side -- measured ~1e-5 at
C4 (n=32, m=8, T=100, delta >= 1e-3); asserted < 2e-4."""
import os

import numpy as np
import pytest

from oracle import dense_kkt

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _rel(a, b):
    scale = np.abs(b).max(axis=1, keepdims=True)
    scale[scale == 0] = 1.0
    return (np.abs(a - b) / scale).max()


def _make(n, m, T, batch, seed, dtype=None):
    from sip_optimal_control_amd import ChainShape, synthetic
    dtype = dtype or torch.float64
    return synthetic.make_chain_batch(ChainShape(n, m, T), batch, seed=seed, device="cuda:0",
                                      dtype=dtype, cross_term=0.01)


@pytest.mark.parametrize("n,m,T,batch", [(17, 4, 16, 5), (5, 3, 7, 9), (7, 1, 4, 3), (15, 16, 3, 2)])
def test_general_engine_fp64_matches_oracle(oracle_lib, monkeypatch, n, m, T, batch):
    from sip_optimal_control_amd import BatchedChainLQR
    monkeypatch.setenv("SIP_LQR_VARIANT", "general")  # also for the shapes with a fused kernel
    mats, vecs = _make(n, m, T, batch, seed=50 + n)
    solver = BatchedChainLQR(n, m, T, batch)
    assert "tree_generic" in solver.kernel_name
    sol, gains, status = solver.factor_solve(mats, vecs)
    torch.cuda.synchronize()
    ref_sol, ref_gains, ref_status = oracle_lib.chain_batch(n, m, T, mats.cpu().numpy(), vecs.cpu().numpy())
    np.testing.assert_array_equal(status.cpu().numpy(), ref_status)
    assert _rel(sol.cpu().numpy(), ref_sol) <= 1e-9
    assert _rel(gains.cpu().numpy(), ref_gains) <= 1e-9


@pytest.mark.parametrize("n,m,T,batch,host", [
    (5, 3, 7, 9, "<6,3,staged>"), (7, 1, 4, 3, "<8,1,staged>"), (10, 3, 20, 13, "<12,3,staged>"),
    (11, 4, 12, 5, "<12,4,staged>"), (9, 2, 9, 7, "<12,2,staged>"), (3, 1, 6, 4, "<3,2,staged>"),
    (5, 3, 0, 2, "<6,3,staged>"), (13, 4, 10, 6, "<14,4,staged>"), (13, 5, 8, 5, "<14,8,staged>"),
    (10, 6, 8, 5, "<12,8,staged>"), (15, 3, 5, 3, "<15,4,staged>"), (16, 6, 6, 5, "<16,8,direct>")])
def test_embedding_in_the_next_fused_kernel(oracle_lib, monkeypatch, n, m, T, batch, host):
    """Without the exact kernels of qw16_extra.hip (SIP_LQR_EXTRA=0; diagnostic builds leave them
    out) a uniform shape runs on the next larger fused kernel: the extra states and controls
    decouple exactly, so the real components match the oracle as usual.  Fused and split entry
    points, one failing problem."""
    from sip_optimal_control_amd import BatchedChainLQR, ChainShape
    monkeypatch.setenv("SIP_LQR_EXTRA", "0")
    mats, vecs = _make(n, m, T, batch, seed=500 + 10 * n + m)
    if T > 1 and batch > 2:
        off = ChainShape(n, m, T).mats_off(1)["R"]
        mats[2, off:off + m * m] = -torch.eye(m, dtype=torch.float64, device="cuda:0").reshape(-1)  # G failure
    solver = BatchedChainLQR(n, m, T, batch)
    assert host in solver.kernel_name and "embedding" in solver.kernel_name
    sol, gains, status = solver.factor_solve(mats, vecs)
    torch.cuda.synchronize()
    ref_sol, ref_gains, ref_status = oracle_lib.chain_batch(n, m, T, mats.cpu().numpy(), vecs.cpu().numpy())
    np.testing.assert_array_equal(status.cpu().numpy(), ref_status)
    ok = ref_status == 0
    assert _rel(sol.cpu().numpy()[ok], ref_sol[ok]) <= 1e-9
    if T > 0:
        assert _rel(gains.cpu().numpy()[ok], ref_gains[ok]) <= 1e-9
        g2, st2 = solver.factor(mats)
        s2 = solver.solve(mats, vecs, g2)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(st2.cpu().numpy(), ref_status)
        assert _rel(s2.cpu().numpy()[ok], ref_sol[ok]) <= 1e-9
        assert _rel(g2.cpu().numpy()[ok], ref_gains[ok]) <= 1e-9


@pytest.mark.parametrize("n", list(range(1, 17)))
def test_every_shape_up_to_16x8_has_an_exact_kernel(oracle_lib, n):
    """qw16_extra.hip: every fp64 chain shape n <= 16, m <= 8 runs on its own instantiation of the
    fused kernel (staged when n and m are even, n <= 14).  Fused and split entry points against
    the oracle, one problem with an indefinite R (G failure) in the batch."""
    from sip_optimal_control_amd import BatchedChainLQR, ChainShape
    for m in range(1, 9):
        T, batch = 3 + (n + m) % 5, 9
        mats, vecs = _make(n, m, T, batch, seed=7000 + 10 * n + m)
        off = ChainShape(n, m, T).mats_off(1)["R"]
        mats[2, off:off + m * m] = -1e3 * torch.eye(m, dtype=torch.float64, device="cuda:0").reshape(-1)
        solver = BatchedChainLQR(n, m, T, batch)
        assert f"qw16<{n},{m}," in solver.kernel_name and "embedding" not in solver.kernel_name
        sol, gains, status = solver.factor_solve(mats, vecs)
        g2, st2 = solver.factor(mats)
        s2 = solver.solve(mats, vecs, g2)
        torch.cuda.synchronize()
        ref_sol, ref_gains, ref_status = oracle_lib.chain_batch(n, m, T, mats.cpu().numpy(), vecs.cpu().numpy())
        np.testing.assert_array_equal(status.cpu().numpy(), ref_status)
        np.testing.assert_array_equal(st2.cpu().numpy(), ref_status)
        ok = ref_status == 0
        assert ok.sum() == batch - 1
        for got_s, got_g in ((sol, gains), (s2, g2)):
            assert _rel(got_s.cpu().numpy()[ok], ref_sol[ok]) <= 1e-9, (n, m)
            assert _rel(got_g.cpu().numpy()[ok], ref_gains[ok]) <= 1e-9, (n, m)


def test_forced_general_engine_equals_fused_kernel(oracle_lib):
    """Same C3-shaped inputs through the dedicated fused kernel and through the general engine."""
    from sip_optimal_control_amd import BatchedChainLQR
    n, m, T, batch = 12, 4, 50, 17
    mats, vecs = _make(n, m, T, batch, seed=9)
    fused = BatchedChainLQR(n, m, T, batch)
    os.environ["SIP_LQR_VARIANT"] = "general"
    try:
        general = BatchedChainLQR(n, m, T, batch)
    finally:
        del os.environ["SIP_LQR_VARIANT"]
    assert "qw16" in fused.kernel_name and "tree_generic" in general.kernel_name
    s1, g1, _ = fused.factor_solve(mats, vecs)
    s2, g2, _ = general.factor_solve(mats, vecs)
    torch.cuda.synchronize()
    assert _rel(s1.cpu().numpy(), s2.cpu().numpy()) <= 1e-9
    assert _rel(g1.cpu().numpy(), g2.cpu().numpy()) <= 1e-9


@pytest.mark.parametrize("split", ["", "general"])
def test_split_factor_then_repeated_solve(oracle_lib, monkeypatch, split):
    """sip_lqr_factor once, sip_lqr_solve twice with different right-hand sides
    (tests/lqr_test.cpp:431-450; CallbackProvider::factor / ::solve): on the fused kernel
    (default for shapes that have one) and on the general engine (SIP_LQR_SPLIT=general)."""
    from sip_optimal_control_amd import BatchedChainLQR
    monkeypatch.setenv("SIP_LQR_SPLIT", split)
    n, m, T, batch = 12, 4, 20, 6
    mats, vecs = _make(n, m, T, batch, seed=77)
    _, vecs2 = _make(n, m, T, batch, seed=78)
    solver = BatchedChainLQR(n, m, T, batch)
    gains, status = solver.factor(mats)
    sol_a = solver.solve(mats, vecs, gains).clone()
    k_a = gains.clone()
    sol_b = solver.solve(mats, vecs2, gains).clone()
    torch.cuda.synchronize()
    assert (status.cpu().numpy() == 0).all()
    for v, s, g in ((vecs, sol_a, k_a), (vecs2, sol_b, gains)):
        ref_sol, ref_gains, _ = oracle_lib.chain_batch(n, m, T, mats.cpu().numpy(), v.cpu().numpy())
        assert _rel(s.cpu().numpy(), ref_sol) <= 1e-9
        assert _rel(g.cpu().numpy(), ref_gains) <= 1e-9


@pytest.mark.parametrize("split", ["", "general"])
def test_split_solve_skips_failed_problems(oracle_lib, monkeypatch, split):
    from sip_optimal_control_amd import BatchedChainLQR, ChainShape
    monkeypatch.setenv("SIP_LQR_SPLIT", split)
    n, m, T, batch = 4, 2, 5, 4
    shape = ChainShape(n, m, T)
    mats, vecs = _make(n, m, T, batch, seed=3)
    bad = mats.clone()
    off = shape.mats_off(2)["R"]
    bad[1, off:off + m * m] = -torch.eye(m, dtype=torch.float64, device="cuda:0").reshape(-1)   # G failure
    bad[2, shape.mats_off(T)["delta"]] = 0.0                                                     # invalid delta
    solver = BatchedChainLQR(n, m, T, batch)
    gains, status = solver.factor(bad)
    sol = solver.empty_sol().fill_(-7.0)
    solver.solve(bad, vecs, gains, sol)
    torch.cuda.synchronize()
    st = status.cpu().numpy()
    assert list(st) == [0, 3, 1, 0]
    ref_sol, _, ref_status = oracle_lib.chain_batch(n, m, T, bad.cpu().numpy(), vecs.cpu().numpy())
    np.testing.assert_array_equal(st, ref_status)
    out = sol.cpu().numpy()
    if split == "general":  # the general engine leaves failed problems untouched, like the oracle;
        assert (out[1] == -7.0).all() and (out[2] == -7.0).all()   # the fused sweep leaves them unspecified
    assert _rel(out[[0, 3]], ref_sol[[0, 3]]) <= 1e-9
    # the fused launch reports the same statuses
    _, _, st2 = solver.factor_solve(bad, vecs)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(st2.cpu().numpy(), ref_status)


def test_fp32_c4_shape_kkt_residual(oracle_lib):
    """BASELINE config 4 shape (n=32, m=8, T=100, fp32), small batch."""
    from sip_optimal_control_amd import BatchedChainLQR
    n, m, T, batch = 32, 8, 100, 4
    mats64, vecs64 = _make(n, m, T, batch, seed=1000)
    mats, vecs = mats64.float(), vecs64.float()
    solver = BatchedChainLQR(n, m, T, batch, dtype=torch.float32)
    assert solver.kernel_name.endswith("/f32")
    sol, gains, status = solver.factor_solve(mats, vecs)
    torch.cuda.synchronize()
    assert (status.cpu().numpy() == 0).all()
    par, ch = list(range(T)), list(range(1, T + 1))
    worst = 0.0
    for p in range(batch):
        # residual of the fp32 solution against the fp32-rounded problem, in fp64
        blocks = dense_kkt.chain_blocks_from_packed(n, m, T, mats[p].double().cpu().numpy(),
                                                    vecs[p].double().cpu().numpy())
        x, u, y = dense_kkt.chain_sol_from_packed(n, m, T, sol[p].double().cpu().numpy())
        res = dense_kkt.residual_norm(par, ch, [n] * (T + 1), [m] * T, blocks, x, u, y)
        rhs = np.sqrt(sum(float(v @ v) for k in ("q", "r", "c") for v in blocks[k]))
        worst = max(worst, res / rhs)
    print("fp32 C4 relative KKT residual:", worst)
    assert worst < 2e-4, worst
    # and close to the fp64 oracle solution of the same (rounded) problem
    ref_sol, _, _ = oracle_lib.chain_batch(n, m, T, mats.double().cpu().numpy(), vecs.double().cpu().numpy())
    assert _rel(sol.double().cpu().numpy(), ref_sol) < 5e-3


def test_randomized_stress_sample():
    """tests/stress.py (random chain / tree / Newton-KKT shapes against the oracle), a small sample."""
    import subprocess
    import sys
    proc = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                        "tests", "stress.py"),
                           "--seed", "11", "--chains", "40", "--trees", "8", "--kkt", "12"],
                          capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-2000:]
    assert "0 failures" in proc.stdout


def test_first_call_of_a_general_engine_plan_can_be_the_captured_one(oracle_lib):
    """include/sip_lqr_amd.h: the compute entry points only enqueue kernels.  The general engine's
    offset tables are uploaded by sip_lqr_plan_create (not lazily at first use), so the very first
    factor_solve of a plan may run under stream capture; replaying the graph gives the oracle's
    result.  (The code object is loaded beforehand by another plan of the same engine.)"""
    from sip_optimal_control_amd import BatchedChainLQR
    n, m, T, batch = 19, 9, 9, 7   # m > 8: no fused kernel and no embedding
    mats, vecs = _make(n, m, T, batch, seed=91)
    warm = BatchedChainLQR(n, m, T, batch)
    assert "tree_generic" in warm.kernel_name
    warm.factor_solve(mats, vecs)
    torch.cuda.synchronize()
    solver = BatchedChainLQR(n, m, T, batch)          # a fresh plan: nothing of it has run yet
    sol, gains = solver.empty_sol().zero_(), solver.empty_gains().zero_()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        solver.factor_solve(mats, vecs, sol, gains)
    torch.cuda.synchronize()
    assert float(sol.abs().max()) == 0.0               # captured, not executed
    graph.replay()
    torch.cuda.synchronize()
    ref_sol, ref_gains, ref_status = oracle_lib.chain_batch(n, m, T, mats.cpu().numpy(), vecs.cpu().numpy())
    np.testing.assert_array_equal(solver.status.cpu().numpy(), ref_status)
    assert _rel(sol.cpu().numpy(), ref_sol) <= 1e-9 and _rel(gains.cpu().numpy(), ref_gains) <= 1e-9


def test_plan_device_is_explicit_and_left_alone(oracle_lib):
    """A bare "cuda" means torch's current device (not device 0), the plan launches on its own
    ordinal (DeviceGuard in the C ABI) and the caller's current device is what it was."""
    from sip_optimal_control_amd import BatchedChainLQR
    from sip_optimal_control_amd._lib import LQRLibraryError, resolve_device
    cur = torch.cuda.current_device()
    assert resolve_device("cuda") == torch.device("cuda", cur)
    solver = BatchedChainLQR(4, 2, 5, 3, device="cuda")
    assert solver.device.index == cur
    mats, vecs = _make(4, 2, 5, 3, seed=12)
    sol, _, status = solver.factor_solve(mats, vecs)
    torch.cuda.synchronize()
    assert torch.cuda.current_device() == cur
    ref_sol, _, _ = oracle_lib.chain_batch(4, 2, 5, mats.cpu().numpy(), vecs.cpu().numpy())
    assert _rel(sol.cpu().numpy(), ref_sol) <= 1e-9
    with pytest.raises(LQRLibraryError):               # an ordinal the machine does not have
        BatchedChainLQR(4, 2, 5, 3, device=f"cuda:{torch.cuda.device_count()}")


@pytest.mark.parametrize("n,m,T,batch,cols", [(12, 4, 50, 37, 8), (12, 4, 7, 5, 11), (4, 2, 9, 6, 3), (8, 3, 6, 9, 8),
                                             (16, 4, 5, 5, 2), (6, 1, 4, 3, 1), (10, 3, 6, 4, 5), (12, 4, 0, 3, 2),
                                             # round 3: every shape n <= 16, m <= 8 has the multi-rhs kernel
                                             (5, 3, 8, 6, 8), (13, 5, 6, 5, 9), (15, 8, 5, 4, 3), (9, 6, 7, 5, 8),
                                             (16, 8, 4, 5, 4), (1, 1, 6, 4, 2), (7, 8, 5, 3, 8), (11, 2, 9, 5, 16),
                                             # beyond: the n = 32 kernel (and an embedding in it), column by column
                                             (32, 8, 5, 3, 2), (20, 3, 4, 3, 2)])
def test_solve_multi_carries_every_right_hand_side(oracle_lib, n, m, T, batch, cols):
    """sip_lqr_solve_multi (the multi-rhs block of solve_stagewise_kkt_matrix, helpers.cpp:521-665):
    factor once, then `cols` right-hand sides in one sweep per 8 columns (chain_mrhs.hpp; more than 8,
    a staged and a direct factor state, n = 16, odd and large shapes among the cases; n > 16 has no such kernel
    and goes column by column behind the same entry point), each column against the oracle's solve of that
    right-hand side."""
    from sip_optimal_control_amd import BatchedChainLQR, ChainShape
    shape = ChainShape(n, m, T)
    mats, _ = _make(n, m, T, batch, seed=700 + n)
    if T > 1 and batch > 2:  # one failing problem: its columns are left alone, the others are exact
        off = shape.mats_off(1)["R"]
        mats[2, off:off + m * m] = -1e4 * torch.eye(m, dtype=torch.float64, device="cuda:0").reshape(-1)
    gen = torch.Generator(device="cuda:0").manual_seed(5)
    vecs_cols = torch.randn(cols, batch, shape.vecs_len, dtype=torch.float64, device="cuda:0", generator=gen)
    solver = BatchedChainLQR(n, m, T, batch)
    assert (solver.solve_multi_workspace_bytes(cols) > 0) == (n <= 16 and m <= 8)   # one sweep per 8 columns
    gains, status = solver.factor(mats)
    sol_cols = solver.solve_multi(mats, vecs_cols, gains)
    torch.cuda.synchronize()
    st = status.cpu().numpy()
    ok = st == 0
    assert ok.sum() >= batch - 1
    for col in range(cols):
        ref_sol, _, ref_status = oracle_lib.chain_batch(n, m, T, mats.cpu().numpy(), vecs_cols[col].cpu().numpy())
        np.testing.assert_array_equal(st, ref_status)
        assert _rel(sol_cols[col].cpu().numpy()[ok], ref_sol[ok]) <= 1e-9, col
    # the single-rhs entry point still works on the same factorization afterwards
    one = solver.solve(mats, vecs_cols[0].contiguous(), gains)
    torch.cuda.synchronize()
    ref_sol, _, _ = oracle_lib.chain_batch(n, m, T, mats.cpu().numpy(), vecs_cols[0].cpu().numpy())
    assert _rel(one.cpu().numpy()[ok], ref_sol[ok]) <= 1e-9
