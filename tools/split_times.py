#!/usr/bin/env python3
"""Split entry points at a BASELINE shape (the reference's BM_LQRFactor / BM_LQRSolve,
benchmarks/lqr_benchmark.cpp:590-651): sip_lqr_factor, sip_lqr_solve, fused factor_solve."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sip_optimal_control_amd import BatchedChainLQR, ChainShape, synthetic

n, m, T, batch = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (12, 4, 50, 4096)))
mats, vecs = synthetic.make_chain_batch(ChainShape(n, m, T), batch, seed=1, device="cuda:0")
s = BatchedChainLQR(n, m, T, batch)


def timed(fn, steps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


gains, status = s.factor(mats)
sol = s.empty_sol()
out = {"shape": [n, m, T, batch], "kernel": s.kernel_name,
       "ms_factor": timed(lambda: s.factor(mats)),
       "ms_solve": timed(lambda: s.solve(mats, vecs, gains, sol)),
       "ms_factor_solve": timed(lambda: s.factor_solve(mats, vecs))}
print(json.dumps(out))
