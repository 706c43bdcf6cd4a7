"""TEST INFRASTRUCTURE (uses the CPU oracle): accuracy of the fused kernel and of the oracle against
a dense KKT solve over ranges of the dual regularization delta (run on the GPU box)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sip_optimal_control_amd import BatchedChainLQR, ChainShape, synthetic
from oracle import oracle, dense_kkt
n, m, T, batch = 12, 4, 50, 8
sh = ChainShape(n, m, T)
for lo, hi in ((1e-9, 1e-7), (1e-6, 1e-4), (1e-3, 1e-1), (1e2, 1e6)):
    mats, vecs = synthetic.make_chain_batch(sh, batch, seed=3, device="cuda:0", cross_term=0.01)
    g = torch.Generator(device="cuda:0"); g.manual_seed(7)
    for i in range(T + 1):
        off = sh.mats_off(i)["delta"]
        u = torch.rand(batch, n, generator=g, device="cuda:0", dtype=torch.float64)
        mats[:, off:off + n] = torch.exp(np.log(lo) + (np.log(hi) - np.log(lo)) * u)
    s = BatchedChainLQR(n, m, T, batch)
    sol, gains, st = s.factor_solve(mats, vecs)
    torch.cuda.synchronize()
    hm, hv = mats.cpu().numpy(), vecs.cpu().numpy()
    rs, rg, rst = oracle.chain_batch(n, m, T, hm, hv)
    gs = sol.cpu().numpy()
    par, ch = list(range(T)), list(range(1, T + 1))
    e_gpu = e_or = 0.0
    for p in range(2):
        b = dense_kkt.chain_blocks_from_packed(n, m, T, hm[p], hv[p])
        x, u_, y = dense_kkt.solve(par, ch, [n] * (T + 1), [m] * T, b)
        ref = np.concatenate([np.concatenate([x[i], y[i]] + ([u_[i]] if i < T else [])) for i in range(T + 1)])
        e_gpu = max(e_gpu, np.abs(gs[p] - ref).max() / np.abs(ref).max())
        e_or = max(e_or, np.abs(rs[p] - ref).max() / np.abs(ref).max())
    d = np.abs(gs - rs).max() / np.abs(rs).max()
    print(f"delta in [{lo:g},{hi:g}]: status {st.cpu().tolist()[:2]} gpu-vs-oracle {d:.2e}  gpu-vs-dense {e_gpu:.2e}  oracle-vs-dense {e_or:.2e}")
