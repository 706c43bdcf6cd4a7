#!/usr/bin/env python3
"""Segment breakdown of condense_chain_pipe_kernel from a diagnostic build:
    tools/kkt_ab_build.sh stamps -DSIP_KKT_STAMPS
    SIP_LQR_LIB=sip_optimal_control_amd/lib/diag/libkkt_stamps.so python tools/kkt_stamps.py   (on the GPU box)
Cycles (s_memtime) per segment, summed over the wavefronts of all launches of one tests/bench_kkt.py run."""
import ctypes
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench_kkt.py", "--steps", "10", "--cpu-seconds", "0"]
runpy.run_path(os.path.join(ROOT, "tests", "bench_kkt.py"), run_name="__main__")
lib = ctypes.CDLL(os.environ["SIP_LQR_LIB"])  # the copy the run used (same path: same handle)
buf = (ctypes.c_ulonglong * 16)()
lib.sip_kkt_debug_segments(buf)
names = ["prologue (first item's loads issued)", "wait for the image, image -> LDS", "commit (weights, rhs rows) + barrier",
         "next item: records, small reads, image loads issued", "compute (tiles, epilogue, copy-out, rhs)", "closing barrier"]
tot = sum(buf[k] for k in range(6))
for k in range(6):
    print("%-55s %14d  %5.1f %%" % (names[k], buf[k], 100.0 * buf[k] / max(tot, 1)))
