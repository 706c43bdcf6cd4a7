#!/usr/bin/env python3
"""DIAGNOSTIC: where the fused tree kernel's time goes, from a -DSIP_TREE_STAMPS build
(tools/tree_ab_build.sh stamps -DSIP_TREE_STAMPS; run with SIP_LQR_LIB=.../libtree_stamps.so on the GPU box).
Prints, per shape of the reference's variable-shape family, the share of wavefront cycles per segment."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
NAMES = ["record + fetch issue", "wait for fetched blocks", "prefetch issue", "edge step (live child)",
         "edge step (child from spill)", "node step", "rollout: loads", "rollout: arithmetic"]


def main():
    import numpy as np
    import torch
    import reference_problems as rp
    from sip_optimal_control_amd.tree import BatchedTreeLQR
    from sip_optimal_control_amd import _lib
    lib = _lib.load_library()
    fn = lib.sip_lqr_tree_debug_segments
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    fn.restype = None
    buf = (ctypes.c_ulonglong * 8)()
    for shape, name in enumerate(("heterogeneous_chain", "shallow_wide_tree", "binary_tree")):
        rng = np.random.default_rng(17 + 31 * shape)
        prob = rp.variable_benchmark_problem(shape, 63, 8, 2, rng)
        s = BatchedTreeLQR(prob["parents"], prob["children"], prob["state_dims"], prob["control_dims"], batch=4096)
        s.pack([prob["blocks"]])
        s.input[1:] = s.input[0:1]
        s.factor_solve()
        torch.cuda.synchronize()
        fn(buf)  # clear
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        s.factor_solve()
        e1.record()
        torch.cuda.synchronize()
        fn(buf)
        tot = float(sum(buf))
        print(f"{name}: {e0.elapsed_time(e1):.3f} ms (stamped build), 100 MHz ticks per wavefront {tot / 1024:.0f}")
        for k in range(8):
            print(f"   {NAMES[k]:32s} {100.0 * buf[k] / tot:5.1f} %   {buf[k] / 1024 / 100.0:8.1f} us per wavefront")


if __name__ == "__main__":
    main()
