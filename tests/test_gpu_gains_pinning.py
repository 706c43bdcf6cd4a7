"""HIP side of tests/test_gains_pinning.py: the gains every kernel family writes are the control law
u = k + K x (lqr.cpp:856-857) of INDEPENDENT dense-KKT solutions of the same problem over families of
offsets c[j] (see that file for why this pins K and k completely), and, at the full BASELINE sizes,
of the kernel's own (KKT-residual-checked) x and u.

fp64: 1e-9 (relative to max |u|).  fp32 (C4 kernel, problem rounded to fp32 first): 1e-4, the stated
fp32 tolerance of tests/test_gpu_mf32_parity.py (measured value printed)."""
import glob
import os

import numpy as np
import pytest

import reference_problems as rp
from oracle import dense_kkt
from test_gains_pinning import ChainFamilies, chain_gains_from_packed, edge_defect, offset_family

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CHAINS = sorted(glob.glob(os.path.join(GOLD, "chain_*.npz")))


@pytest.mark.parametrize("path", CHAINS, ids=[os.path.basename(p) for p in CHAINS])
def test_hip_gains_are_the_control_law_of_the_dense_kkt_solutions(path):
    from sip_optimal_control_amd import BatchedChainLQR
    d = np.load(path)
    n, m, T = int(d["n"]), int(d["m"]), int(d["T"])
    f32 = n == 32
    dtype = torch.float32 if f32 else torch.float64
    mats = torch.from_numpy(d["mats"][:1]).to(dtype).cuda()
    vecs = torch.from_numpy(d["vecs"][:1]).to(dtype).cuda()
    solver = BatchedChainLQR(n, m, T, 1, dtype=dtype)
    assert ("mt16" in solver.kernel_name) if f32 else ("qw16" in solver.kernel_name)
    _, gains, status = solver.factor_solve(mats, vecs)
    torch.cuda.synchronize()
    assert int(status[0]) == 0
    # the families are solved (fp64, dense) on the problem the kernel really saw
    fam = ChainFamilies(n, m, T, mats[0].double().cpu().numpy(), vecs[0].double().cpu().numpy())
    defect, margin = fam.defect(gains[0].double().cpu().numpy())
    print(f"{os.path.basename(path)} {solver.kernel_name}: control-law defect {defect:.2e}, margin {margin:.1e}")
    assert margin > 1e-6
    assert defect <= (1e-4 if f32 else 1e-9), defect
    # split entry point: the gains of sip_lqr_factor alone (K only; k comes with solve)
    g2, st2 = solver.factor(mats)
    solver.solve(mats, vecs, g2)
    torch.cuda.synchronize()
    defect2, _ = fam.defect(g2[0].double().cpu().numpy())
    assert defect2 <= (1e-4 if f32 else 1e-9), defect2


@pytest.mark.parametrize("name", ["nonuniform_diagonal_delta", "branch_tree", "variable_dimension_branch",
                                  "five_node_variable_tree_eigen"])
def test_tree_engine_gains_are_the_control_law_of_the_dense_kkt_solutions(name):
    from sip_optimal_control_amd.tree import BatchedTreeLQR
    prob = getattr(rp, name)()
    par, ch, sd, cd = prob["parents"], prob["children"], prob["state_dims"], prob["control_dims"]
    s = BatchedTreeLQR(par, ch, sd, cd, batch=1)
    s.pack([prob["blocks"]])
    assert int(s.factor()[0]) == 0
    s.solve()
    torch.cuda.synchronize()
    Ks, ks = s.unpack_gains()
    fam = dense_kkt.OffsetFamilies(par, ch, sd, cd, prob["blocks"])
    for e in range(len(cd)):
        sols = fam.solve(par[e], offset_family(np.asarray(prob["blocks"]["c"][par[e]], dtype=float), seed=e))
        dft, umax, margin = edge_defect(sols, par[e], e, Ks[e], ks[e])
        assert margin > 1e-6 and dft <= 1e-9 * max(umax, 1e-300), (e, dft, margin)


def _own_control_law_defect(n, m, T, sol, gains):
    """max |u_i - K_i x_i - k_i| / max |u| over a packed batch (torch, on the device)."""
    B = sol.shape[0]
    stage = sol[:, :T * (2 * n + m)].reshape(B, T, 2 * n + m)
    x, u = stage[:, :, :n], stage[:, :, 2 * n:]
    g = gains.reshape(B, T, m * n + m)
    K = g[:, :, :m * n].reshape(B, T, n, m)  # column-major m x n: [col][row]
    k = g[:, :, m * n:]
    pred = torch.einsum("btcr,btc->btr", K, x) + k
    return float((u - pred).abs().max() / u.abs().max())


@pytest.mark.parametrize("workload", ["c3", "c4"])
def test_full_size_solution_obeys_its_own_gains(workload):
    """BASELINE C3 / C4 at full batch: u_i = K_i x_i + k_i between the kernel's sol and gains."""
    from sip_optimal_control_amd import BatchedChainLQR, ChainShape, synthetic
    batch, T, n, m, dtype = {"c3": (4096, 50, 12, 4, torch.float64), "c4": (4096, 100, 32, 8, torch.float32)}[workload]
    mats, vecs = synthetic.make_chain_batch(ChainShape(n, m, T), batch, seed=321, device="cuda:0", dtype=dtype,
                                            cross_term=0.01)
    solver = BatchedChainLQR(n, m, T, batch, dtype=dtype)
    sol, gains, status = solver.factor_solve(mats, vecs)
    torch.cuda.synchronize()
    assert bool((status == 0).all())
    defect = _own_control_law_defect(n, m, T, sol.double(), gains.double())
    print(f"{workload}: own control-law defect {defect:.2e}")
    assert defect <= (1e-5 if dtype == torch.float32 else 1e-12), defect
