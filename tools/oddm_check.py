import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from sip_optimal_control_amd import BatchedChainLQR, ChainShape, synthetic
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(10, 3), (6, 3), (12, 3)]
for (n, m) in shapes:
    for T, batch in ((50, 37), (7, 5), (1, 3), (50, 4096)):
        res = {}
        for variant in ("staged", "direct"):
            os.environ["SIP_LQR_VARIANT"] = variant
            sh = ChainShape(n, m, T)
            mats, vecs = synthetic.make_chain_batch(sh, batch, seed=3 + n + T, device="cuda:0", cross_term=0.01)
            s = BatchedChainLQR(n, m, T, batch, device="cuda:0")
            assert variant in s.kernel_name, s.kernel_name
            sol, gains, st = s.factor_solve(mats, vecs)
            torch.cuda.synchronize()
            assert int(st.abs().sum()) == 0
            res[variant] = (sol.clone(), gains.clone())
        ds = float((res["staged"][0] - res["direct"][0]).abs().max() / res["direct"][0].abs().max())
        dg = float((res["staged"][1] - res["direct"][1]).abs().max() / res["direct"][1].abs().max())
        print(n, m, T, batch, "sol rel diff %.2e gains rel diff %.2e" % (ds, dg))
        assert ds < 1e-11 and dg < 1e-11
print("ok")
