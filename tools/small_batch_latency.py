#!/usr/bin/env python3
"""Latency of one Newton-KKT step (sip_kkt_factor_solve) and one bare Riccati sweep at small batch
sizes, eager launches against a captured hipGraph -- run on the GPU box.  Shape: NewtonKKTProblem
(n = 12, m = 4, T = 50), c = 6, g = 8."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import reference_kkt_problems as rk
from sip_optimal_control_amd import BatchedChainLQR, BatchedNewtonKKT, ChainShape, synthetic

n, m, T = 12, 4, 50
dims = rk.newton_kkt_dims(n, m, T)


def timed(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


print("batch | KKT step eager us | graph us | Riccati sweep eager us | graph us")
for batch in (1, 16, 256, 4096):
    arrays = rk.newton_kkt_problem(dims, seed=1, batch=batch, r2_max=1e2)
    d = [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda() for a in arrays]
    kkt = BatchedNewtonKKT(dims.parents, dims.children, dims.sd, dims.cd, dims.ncd, dims.ngd, dims.ecd, dims.egd,
                           batch=batch, root=dims.root)
    sol = torch.zeros(batch, dims.kkt_dim, dtype=torch.float64, device="cuda")
    mats, vecs = synthetic.make_chain_batch(ChainShape(n, m, T), batch, seed=2, device="cuda:0")
    lqr = BatchedChainLQR(n, m, T, batch)
    lsol, lgains = lqr.empty_sol(), lqr.empty_gains()
    row = []
    for work in (lambda: kkt.factor_solve(*d, sol=sol), lambda: lqr.factor_solve(mats, vecs, lsol, lgains)):
        eager = timed(work)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            work()
        row += [eager, timed(graph.replay)]
    print(f"{batch:5d} | {row[0]:8.1f} | {row[1]:8.1f} | {row[2]:8.1f} | {row[3]:8.1f}", flush=True)
