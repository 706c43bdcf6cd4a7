"""Parity of the fp32 / n = 32 matrix-core kernels (BASELINE C4) -- chain_factor_solve_mt16 (16 x 16 tiles,
sweep factorisations; the one plans get) and chain_factor_solve_mf32 (32 x 32 tiles, round 1; kept behind
SIP_LQR_VARIANT=mf32) -- beyond the KKT residual of tests/test_gpu_general_chain.py:

* FactorStatus (lqr.hpp:68-74) with injected failures, exact against the oracle on the fp32-rounded
  problem, including the precedence at a node (G before delta before F: lqr.cpp:696-701, 722-727)
  and "first failing node in postorder";
* K, k against the oracle on the rounded problem;
* the committed golden vectors of the C4 shape (tests/golden/chain_c4_n32_m8_T100.npz);
* the full C4 batch (4096) through size-independent properties: linearity of the solve in the
  right-hand side, idempotence (same launch twice), sampled KKT residuals.

Stated fp32 tolerances (measured values are printed): x, u, y and K, k within 1e-4 max-abs relative
to the max-abs of the oracle's block for that problem; KKT residual relative to the right-hand-side
norm < 2e-4 (delta as small as 1e-3 and T = 100)."""
import os

import numpy as np
import pytest

from oracle import dense_kkt

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

N, M, T = 32, 8, 100
TOL = 1e-4
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rel(a, b):
    scale = np.abs(b).max(axis=1, keepdims=True)
    scale[scale == 0] = 1.0
    return float((np.abs(a - b) / scale).max())


def _make(batch, seed, T_=T):
    from sip_optimal_control_amd import ChainShape, synthetic
    return synthetic.make_chain_batch(ChainShape(N, M, T_), batch, seed=seed, device="cuda:0",
                                      dtype=torch.float32, cross_term=0.01)


@pytest.fixture(params=["mt16", "mf32"])
def variant(request, monkeypatch):
    """mt16: what a plan gets by default; mf32: the round-1 kernel, selected by name."""
    if request.param != "mt16":
        monkeypatch.setenv("SIP_LQR_VARIANT", request.param)
    return request.param


def _solver(batch, T_=T, variant="mt16"):
    from sip_optimal_control_amd import BatchedChainLQR
    s = BatchedChainLQR(N, M, T_, batch, dtype=torch.float32)
    assert variant in s.kernel_name, s.kernel_name
    return s


def test_injected_failures_report_the_reference_status(oracle_lib, variant):
    from sip_optimal_control_amd import ChainShape
    T_ = 12
    shape = ChainShape(N, M, T_)
    batch = 10
    mats, vecs = _make(batch, seed=77, T_=T_)
    eye_m = torch.eye(M, dtype=torch.float32, device="cuda:0").reshape(-1)
    eye_n = torch.eye(N, dtype=torch.float32, device="cuda:0").reshape(-1)

    def R(i): o = shape.mats_off(i)["R"]; return slice(o, o + M * M)
    def Q(i): o = shape.mats_off(i)["Q"]; return slice(o, o + N * N)
    def delta(i, j): return shape.mats_off(i)["delta"] + j

    expected = [0] * batch
    mats[1, R(5)] = -1e4 * eye_m;                     expected[1] = 3  # G at edge 5
    mats[2, delta(T_, 3)] = 0.0;                expected[2] = 1  # delta must be strictly positive (lqr.cpp:478)
    mats[3, delta(4, 31)] = -1.0;               expected[3] = 1
    mats[4, Q(T_)] = -1e4 * eye_n;              expected[4] = 2  # I + sqrt(d) V sqrt(d) indefinite at the leaf
    mats[5, Q(6)] = -1e6 * eye_n;               expected[5] = 2  # ... at an interior node
    mats[6, R(5)] = -1e4 * eye_m; mats[6, delta(5, 0)] = 0.0;        expected[6] = 3  # same node: G first
    mats[7, delta(5, 0)] = 0.0; mats[7, Q(5)] = -1e6 * eye_n;  expected[7] = 1  # same node: delta before F
    mats[8, delta(9, 2)] = 0.0; mats[8, R(2)] = -1e4 * eye_m;        expected[8] = 1  # node 9 comes first in postorder
    mats[9, R(8)] = -1e4 * eye_m; mats[9, delta(3, 1)] = 0.0;        expected[9] = 3  # edge 8 comes first
    solver = _solver(batch, T_, variant)
    _, _, status = solver.factor_solve(mats, vecs)
    torch.cuda.synchronize()
    _, _, ref_status = oracle_lib.chain_batch(N, M, T_, mats.double().cpu().numpy(), vecs.double().cpu().numpy())
    assert list(ref_status) == expected          # the oracle agrees with the construction
    np.testing.assert_array_equal(status.cpu().numpy(), ref_status)
    # split entry points report the same
    _, st2 = solver.factor(mats)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(st2.cpu().numpy(), ref_status)


def test_solution_and_gains_match_the_oracle_on_the_rounded_problem(oracle_lib, variant):
    batch = 6
    mats, vecs = _make(batch, seed=4242)
    solver = _solver(batch, variant=variant)
    sol, gains, status = solver.factor_solve(mats, vecs)
    torch.cuda.synchronize()
    ref_sol, ref_gains, ref_status = oracle_lib.chain_batch(N, M, T, mats.double().cpu().numpy(),
                                                            vecs.double().cpu().numpy())
    assert (ref_status == 0).all() and (status.cpu().numpy() == 0).all()
    es, eg = _rel(sol.double().cpu().numpy(), ref_sol), _rel(gains.double().cpu().numpy(), ref_gains)
    print(f"{variant} vs oracle (fp32-rounded problem): sol {es:.2e}, gains {eg:.2e}")
    assert es < TOL and eg < TOL


def test_golden_c4(variant):
    d = np.load(os.path.join(GOLD, "chain_c4_n32_m8_T100.npz"))
    assert (int(d["n"]), int(d["m"]), int(d["T"])) == (N, M, T)
    batch = d["mats"].shape[0]
    solver = _solver(batch, variant=variant)
    sol, gains, status = solver.factor_solve(torch.from_numpy(d["mats"]).float().cuda(),
                                             torch.from_numpy(d["vecs"]).float().cuda())
    torch.cuda.synchronize()
    assert (status.cpu().numpy() == 0).all()
    es, eg = _rel(sol.double().cpu().numpy(), d["sol"]), _rel(gains.double().cpu().numpy(), d["gains"])
    print(f"{variant} vs golden C4 (fp64 dense KKT of the unrounded problem): sol {es:.2e}, gains {eg:.2e}")
    assert es < TOL and eg < TOL


def test_full_c4_batch_properties(variant):
    """batch 4096 (BASELINE C4): every status SUCCESS; the solve is linear in (q, r, c) (sol(a) +
    sol(b) - sol(0) = sol(a + b) up to fp32 rounding); the same launch twice gives the same bits;
    sampled KKT residuals (evaluated in fp64 on the rounded problem) < 2e-4 of the rhs norm."""
    batch = 4096
    mats, va = _make(batch, seed=9)
    _, vb = _make(batch, seed=10)
    solver = _solver(batch, variant=variant)
    sa, ga, st = solver.factor_solve(mats, va)
    sa, ga = sa.clone(), ga.clone()
    assert bool((st == 0).all())
    sa2, ga2, _ = solver.factor_solve(mats, va)
    assert torch.equal(sa, sa2) and torch.equal(ga, ga2)          # idempotent, no state carried over
    sb = solver.factor_solve(mats, vb)[0].clone()
    s0 = solver.factor_solve(mats, torch.zeros_like(va))[0].clone()
    sab = solver.factor_solve(mats, va + vb)[0].clone()
    torch.cuda.synchronize()
    scale = sab.abs().amax(dim=1, keepdim=True)
    lin = float(((sa + sb - s0 - sab).abs() / scale).max())
    print(variant, "full C4: linearity defect", lin)
    assert lin < TOL
    par, ch = list(range(T)), list(range(1, T + 1))
    worst = 0.0
    for p in (0, 1, 777, 2048, 4095):
        blocks = dense_kkt.chain_blocks_from_packed(N, M, T, mats[p].double().cpu().numpy(),
                                                    va[p].double().cpu().numpy())
        x, u, y = dense_kkt.chain_sol_from_packed(N, M, T, sa[p].double().cpu().numpy())
        res = dense_kkt.residual_norm(par, ch, [N] * (T + 1), [M] * T, blocks, x, u, y)
        rhs = np.sqrt(sum(float(v @ v) for k in ("q", "r", "c") for v in blocks[k]))
        worst = max(worst, res / rhs)
    print(variant, "full C4: worst sampled relative KKT residual", worst)
    assert worst < 2e-4
