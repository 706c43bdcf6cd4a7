#!/usr/bin/env python3
"""Newton-KKT factor + solve over the reference's benchmark grid (newton_kkt_benchmark.cpp:264-273: n in {4, 6, 8},
m in {1, 2, 3, 4}; plus n = 12), T = 50, batch 4096, fp64, one MI355X: time per step, solves/s, fraction of the HBM
roofline on the step's algorithmic bytes, and the residual through the GPU operator.  Markdown table on stdout.

    python tools/kkt_grid_times.py > gpurun_out/kkt_grid.md        (on the GPU box)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sip_optimal_control_amd import BatchedNewtonKKT, synthetic

T, batch = 50, 4096
print("| (n, m) | kernels | step | solves/s | fraction of 8 TB/s (algorithmic bytes) | `solve` alone | `y += K x` | max residual |")
print("|---|---|---|---|---|---|---|---|")
for n in (4, 6, 8, 12):
    for m in (1, 2, 3, 4):
        c, g = max(1, n // 2), max(1, 2 * m)
        dims = dict(parents=list(range(T)), children=list(range(1, T + 1)), state_dims=[n] * (T + 1),
                    control_dims=[m] * T, node_c_dims=[0] * T + [c], node_g_dims=[0] * T + [g],
                    edge_c_dims=[c] * T, edge_g_dims=[g] * T)
        kkt = BatchedNewtonKKT(batch=batch, **dims)
        data = synthetic.make_newton_kkt_batch(kkt, seed=0, r2_max=1e2, **dims)
        sol = torch.zeros(batch, kkt.kkt_dim, dtype=torch.float64, device=kkt.device)

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

        ms = timed(lambda: kkt.factor_solve(*data, sol=sol))
        assert int((kkt.status != 0).sum()) == 0
        prod = kkt.add_Kx_to_y(*data[:5], sol)
        res = float((prod - data[5]).norm(dim=1).max())
        ms_kx = timed(lambda: kkt.add_Kx_to_y(*data[:5], sol, y=prod))
        kkt.factor(*data[:5])
        ms_solve = timed(lambda: kkt.solve(data[0], data[5], sol=sol))
        alg = 8 * (kkt.model_len + 2 * kkt.z_dim + kkt.x_dim + kkt.y_dim + 2 * kkt.kkt_dim)
        name = kkt.kernel_name.replace("chain:chain_factor_solve_", "").replace("/f64 + chain condensation", "")
        print(f"| ({n}, {m}) | `{name}` | {ms:.3f} ms | {batch / ms / 1e3:.2f} M | {batch * alg / (ms * 1e-3) / 8e12:.3f} | "
              f"{ms_solve:.3f} ms | {ms_kx:.3f} ms | {res:.1e} |")
        del kkt, data, sol, prod
        torch.cuda.empty_cache()
