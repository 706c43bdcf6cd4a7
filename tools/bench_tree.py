#!/usr/bin/env python3
"""Throughput of the general tree engine on the reference's variable-shape benchmark family
(BM_LQRVariableFactorSolve, benchmarks/lqr_benchmark.cpp:209-310, 670-744): factor + solve of
`batch` instances of one heterogeneous chain / shallow wide tree / binary tree.

    python tools/bench_tree.py [--batch 4096] [--T 63] [--n 8]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--T", type=int, default=63)
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import numpy as np
    import torch
    import reference_problems as rp
    from sip_optimal_control_amd.tree import BatchedTreeLQR
    out = []
    for shape, name in enumerate(("heterogeneous_chain", "shallow_wide_tree", "binary_tree")):
        rng = np.random.default_rng(17 + 31 * shape)
        prob = rp.variable_benchmark_problem(shape, args.T, args.n, 2, rng)
        s = BatchedTreeLQR(prob["parents"], prob["children"], prob["state_dims"], prob["control_dims"],
                           batch=args.batch)
        s.pack([prob["blocks"]])                       # one instance ...
        s.input[1:] = s.input[0:1]                     # ... replicated over the batch
        def timed(fn):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            assert int((s.status != 0).sum()) == 0
            return e0.elapsed_time(e1) / args.steps

        ms = timed(s.factor_solve)                          # fused size-class kernel (tree_qw16.hpp)
        ms_general = timed(lambda: (s.factor(), s.solve()))  # general engine, factor then solve
        bytes_per = 8 * (s.in_len + s.out_len)
        out.append({"shape": name, "T": args.T, "base_n": args.n, "batch": args.batch, "kernel": s.kernel_name,
                    "ms": ms, "sweeps_per_s": args.batch / (ms * 1e-3),
                    "hbm_frac_algorithmic": args.batch * bytes_per / (ms * 1e-3) / 8e12,
                    "general_engine_ms": ms_general, "general_engine_sweeps_per_s": args.batch / (ms_general * 1e-3)})
    print(json.dumps({"metric": "tree factor+solve sweeps/s (sip_lqr_tree_factor_solve)", "results": out}))


if __name__ == "__main__":
    main()
