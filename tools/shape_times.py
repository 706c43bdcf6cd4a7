#!/usr/bin/env python3
"""Fused sweep time of uniform chain shapes (batch 4096, T = 50, fp64) -- run on the GPU box:
python tools/shape_times.py 10,3 13,5 9,2 ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sip_optimal_control_amd import BatchedChainLQR, ChainShape, synthetic

shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(10, 3), (13, 5), (9, 2), (12, 4), (16, 4)]
batch, T = 4096, 50
for n, m in shapes:
    sh = ChainShape(n, m, T)
    mats, vecs = synthetic.make_chain_batch(sh, batch, seed=1, device="cuda:0")
    s = BatchedChainLQR(n, m, T, batch)
    sol, gains = s.empty_sol(), s.empty_gains()
    for _ in range(3):
        s.factor_solve(mats, vecs, sol, gains)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        s.factor_solve(mats, vecs, sol, gains)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    gbs = sh.algorithmic_bytes(8) * batch / (ms * 1e-3) / 1e9
    print(f"({n:2d},{m}) {s.kernel_name:48s} {ms:.3f} ms  {batch / ms / 1e3:.2f} M sweeps/s  {gbs:.0f} GB/s algorithmic ({gbs / 8000:.2f} of HBM peak)")
