"""SIP_LQR_LAYOUT_SYMMETRIC (include/sip_lqr_amd.h): Q and R of `mats` as packed lower triangles.  The reference reads
both triangles of a symmetric Q (lqr.cpp:658) and the lower one of R (Eigen::LLT, lqr.cpp:697); with symmetric inputs
the packed kernel must give the bits of the full-layout kernel (the same arithmetic on the same numbers) and, as
everywhere, the oracle's results to 1e-9."""
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
    shape = ChainShape(n, m, T)
    mats, vecs = synthetic.make_chain_batch(shape, batch, seed=seed, device="cuda:0", cross_term=0.01)
    idx = torch.from_numpy(shape.packed().pack_index()).to("cuda:0")
    return shape, mats, mats[:, idx].contiguous(), vecs


@pytest.mark.parametrize("n,m,T,batch", [(12, 4, 50, 37), (12, 4, 1, 3), (12, 4, 0, 2), (8, 4, 9, 6), (4, 4, 7, 5)])
def test_packed_triangles_give_the_full_layout_bits_and_the_oracle(oracle_lib, n, m, T, batch):
    from sip_optimal_control_amd import BatchedChainLQR
    shape, mats, sym, vecs = _make(n, m, T, batch, seed=900 + n + T)
    full = BatchedChainLQR(n, m, T, batch)
    packed = BatchedChainLQR(n, m, T, batch, symmetric=True)
    assert "sym" in packed.kernel_name and "sym" not in full.kernel_name
    assert packed.shape.mats_len == sym.shape[1] < shape.mats_len
    if T > 1 and batch > 4:  # statuses too: one G failure, one invalid delta
        off = shape.mats_off(1)
        mats[2, off["R"]:off["R"] + m * m] = -1e4 * torch.eye(m, dtype=torch.float64, device="cuda:0").reshape(-1)
        mats[4, off["delta"] + 1] = 0.0
        sym = mats[:, torch.from_numpy(shape.packed().pack_index()).to("cuda:0")].contiguous()
    s1, g1, st1 = (t.clone() for t in full.factor_solve(mats, vecs))
    s2, g2, st2 = packed.factor_solve(sym, vecs)
    torch.cuda.synchronize()
    ref_sol, ref_gains, ref_status = oracle_lib.chain_batch(n, m, T, mats.cpu().numpy(), vecs.cpu().numpy())
    np.testing.assert_array_equal(st2.cpu().numpy(), ref_status)
    ok = torch.from_numpy(ref_status == 0).to("cuda:0")
    assert torch.equal(s1[ok], s2[ok]) and torch.equal(g1[ok], g2[ok])          # bitwise
    okh = ref_status == 0
    assert _rel(s2.cpu().numpy()[okh], ref_sol[okh]) <= TOL
    if T > 0:
        assert _rel(g2.cpu().numpy()[okh], ref_gains[okh]) <= TOL
    # split entry points of a packed plan (they re-use the fused kernel's factor / vector-solve modes)
    gains, status = packed.factor(sym)
    sol = packed.solve(sym, vecs, gains)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(status.cpu().numpy(), ref_status)
    assert _rel(sol.cpu().numpy()[okh], ref_sol[okh]) <= TOL


def test_packed_layout_at_the_headline_size(oracle_lib):
    """BASELINE's C3 (batch 4096, T = 50, n = 12, m = 4): every problem against the oracle."""
    from sip_optimal_control_amd import BatchedChainLQR
    n, m, T, batch = 12, 4, 50, 4096
    shape, mats, sym, vecs = _make(n, m, T, batch, seed=31)
    packed = BatchedChainLQR(n, m, T, batch, symmetric=True)
    sol, gains, status = packed.factor_solve(sym, vecs)
    torch.cuda.synchronize()
    ref_sol, ref_gains, ref_status = oracle_lib.chain_batch(n, m, T, mats.cpu().numpy(), vecs.cpu().numpy(), threads=16)
    np.testing.assert_array_equal(status.cpu().numpy(), ref_status)
    assert (ref_status == 0).all()
    assert _rel(sol.cpu().numpy(), ref_sol) <= TOL and _rel(gains.cpu().numpy(), ref_gains) <= TOL


def test_unsupported_shapes_and_host_side_packing(oracle_lib):
    import ctypes
    from sip_optimal_control_amd import BatchedChainLQR
    from sip_optimal_control_amd.chain import LQRLibraryError
    with pytest.raises(LQRLibraryError):
        BatchedChainLQR(5, 3, 4, 2, symmetric=True)            # odd dimensions: no packed kernel
    with pytest.raises(LQRLibraryError):
        BatchedChainLQR(32, 8, 4, 2, dtype=torch.float32, symmetric=True)
    s = BatchedChainLQR(12, 4, 3, 2, symmetric=True)
    assert s._lib.sip_lqr_plan_layout(s._plan) == 1 and s.has_split  # (the Newton-KKT step's sweep: packed [Q | delta | M | R])
    assert s.solve_multi_workspace_bytes(4) == 0                # several right-hand sides: column by column
