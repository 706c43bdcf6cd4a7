#!/usr/bin/env python3
"""PCIe-inclusive rate of one C3 sweep: pinned host buffers -> device, fused
kernel, results -> pinned host.  Never the bench `value`; quoted in DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sip_optimal_control_amd import BatchedChainLQR, ChainShape, synthetic
n, m, T, batch = 12, 4, 50, 4096
shape = ChainShape(n, m, T)
mats, vecs = synthetic.make_chain_batch(shape, batch, seed=1, device="cuda:0")
hm, hv = mats.cpu().pin_memory(), vecs.cpu().pin_memory()
solver = BatchedChainLQR(n, m, T, batch)
sol, gains = solver.empty_sol(), solver.empty_gains()
hs, hg = torch.empty_like(sol, device="cpu").pin_memory(), torch.empty_like(gains, device="cpu").pin_memory()
hst = torch.empty(batch, dtype=torch.int32).pin_memory()
def once():
    mats.copy_(hm, non_blocking=True); vecs.copy_(hv, non_blocking=True)
    solver.factor_solve(mats, vecs, sol, gains)
    hs.copy_(sol, non_blocking=True); hg.copy_(gains, non_blocking=True); hst.copy_(solver.status, non_blocking=True)
    torch.cuda.synchronize()
for _ in range(2): once()
t0 = time.perf_counter(); reps = 5
for _ in range(reps): once()
dt = (time.perf_counter() - t0) / reps
gb = (hm.numel() + hv.numel() + hs.numel() + hg.numel()) * 8 / 1e9
print(f"PCIe-inclusive: {dt*1e3:.1f} ms per sweep of {batch} problems = {batch/dt/1e6:.3f} M sweeps/s ({gb/dt:.1f} GB/s over the host link)")
