#!/usr/bin/env python3
"""DIAGNOSTIC: sip_kkt_factor / sip_kkt_solve (the split pair) at the f1 benchmark shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sip_optimal_control_amd import BatchedNewtonKKT, synthetic
n, m, T, batch = 12, 4, 50, 4096
c, g = n // 2, 2 * m
dims = dict(parents=list(range(T)), children=list(range(1, T + 1)), state_dims=[n] * (T + 1), control_dims=[m] * T,
            node_c_dims=[0] * T + [c], node_g_dims=[0] * T + [g], edge_c_dims=[c] * T, edge_g_dims=[g] * T)
kkt = BatchedNewtonKKT(batch=batch, **dims)
data = synthetic.make_newton_kkt_batch(kkt, seed=0, r2_max=1e2, **dims)
def timed(fn, steps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps
kkt.factor(*data[:5])
print({"factor_ms": round(timed(lambda: kkt.factor(*data[:5])), 4), "solve_ms": round(timed(lambda: kkt.solve(data[0], data[5])), 4),
       "factor_solve_ms": round(timed(lambda: kkt.factor_solve(*data)), 4)})
