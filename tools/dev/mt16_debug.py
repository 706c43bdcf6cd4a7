"""Development aid: the mt16 kernels on tiny horizons against the oracle (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import oracle
from sip_optimal_control_amd import BatchedChainLQR, ChainShape, synthetic

def rel(a, b):
    s = np.abs(b).max(axis=1, keepdims=True); s[s == 0] = 1
    return float((np.abs(a - b) / s).max())

for dtype, var in ((torch.float32, ""), (torch.float32, "mf32")):
    os.environ["SIP_LQR_VARIANT"] = var
    for (n, m, T) in [(32, 8, 0), (32, 8, 1), (32, 8, 2), (32, 8, 5), (32, 8, 20), (32, 8, 100)]:
        batch = 16
        mats, vecs = synthetic.make_chain_batch(ChainShape(n, m, T), batch, seed=5, device="cuda:0", dtype=dtype, cross_term=0.01)
        s = BatchedChainLQR(n, m, T, batch, dtype=dtype)
        sol, gains, status = s.factor_solve(mats, vecs)
        torch.cuda.synchronize()
        rs, rg, rst = oracle.chain_batch(n, m, T, mats.double().cpu().numpy(), vecs.double().cpu().numpy())
        print(s.kernel_name, "T", T, "status", status.cpu().numpy(), "ref", rst, "sol err", rel(sol.double().cpu().numpy(), rs),
              "gains err", rel(gains.double().cpu().numpy(), rg) if T else 0.0, flush=True)
        if False:
            vs = 2 * n + m
            got = sol.double().cpu().numpy()[0]
            for i in range(T + 1):
                print("  stage", i, "x err", np.abs(got[i*vs:i*vs+n] - rs[0][i*vs:i*vs+n]).max(), "y err", np.abs(got[i*vs+n:i*vs+2*n] - rs[0][i*vs+n:i*vs+2*n]).max(),
                      "x ref", np.abs(rs[0][i*vs:i*vs+n]).max())
