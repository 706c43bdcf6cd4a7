"""Development aid: W of the terminal node as the two fp32 n = 32 kernels spill it, against fp64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sip_optimal_control_amd import BatchedChainLQR, ChainShape, synthetic
n, m, T, batch = 32, 8, 1, 8
shape = ChainShape(n, m, T)
mats, vecs = synthetic.make_chain_batch(shape, batch, seed=5, device="cuda:0", dtype=torch.float32, cross_term=0.01)
hm = mats.double().cpu().numpy()
for var in ("", "mf32"):
    os.environ["SIP_LQR_VARIANT"] = var
    s = BatchedChainLQR(n, m, T, batch, dtype=torch.float32)
    sol, gains, status = s.factor_solve(mats, vecs)
    torch.cuda.synchronize()
    ws = s.workspace.view(torch.float32)[: batch * (T + 1) * (n * n + n)].reshape(batch, T + 1, n * n + n).cpu().numpy()
    worst = 0; worst_d = 0; asym = 0
    for p in range(batch):
        o = shape.mats_off(T)
        Q = hm[p, o["Q"]:o["Q"] + n * n].reshape(n, n).T
        d = hm[p, o["delta"]:o["delta"] + n]
        sd = np.sqrt(d)
        F = np.eye(n) + sd[:, None] * Q * sd[None, :]
        Wt = (np.eye(n) - np.linalg.inv(F)) / sd[:, None] / sd[None, :]
        dump = ws[p, T, : n * n].reshape(4, 64, 4)
        W = np.zeros((n, n))
        for t4 in range(4):
            for lane in range(64):
                for e in range(4):
                    if var == "":
                        I, J = t4 >> 1, t4 & 1
                        j, g = lane & 15, lane >> 4
                        W[16 * I + 4 * g + e, 16 * J + j] = dump[t4, lane, e]
                    else:
                        q = 4 * t4 + e
                        j, h = lane & 31, lane >> 5
                        W[(q & 3) + 8 * (q >> 2) + 4 * h, j] = dump[t4, lane, e]
        sc = np.abs(Wt).max()
        worst = max(worst, np.abs(W - Wt).max() / sc)
        worst_d = max(worst_d, np.abs(np.diag(W - Wt)).max() / sc)
        asym = max(asym, np.abs(W - W.T).max() / sc)
    print(s.kernel_name, "W err", worst, "diag err", worst_d, "asym", asym, "scale", sc)
