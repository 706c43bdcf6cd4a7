#!/usr/bin/env python3
"""Phase breakdown of the fused kernel from the stamped diagnostic build.
Run on the GPU box:  SIP_LQR_LIB=sip_optimal_control_amd/lib/diag/libsip_lqr_amd.so python tools/stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sip_optimal_control_amd import BatchedChainLQR, ChainShape, synthetic
from sip_optimal_control_amd._lib import load_library

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
c4 = len(sys.argv) > 2 and sys.argv[2] == "c4"
n, m, T = (32, 8, 100) if c4 else (12, 4, 50)
if len(sys.argv) > 3:
    T = int(sys.argv[3])  # horizon override: python tools/stamps.py 4096 c3 20
dtype = torch.float32 if c4 else torch.float64
shape = ChainShape(n, m, T)
mats, vecs = synthetic.make_chain_batch(shape, batch, seed=1, device="cuda:0", dtype=dtype)
solver = BatchedChainLQR(n, m, T, batch, dtype=dtype, device="cuda:0")
lib = load_library()
waves = batch if c4 else (batch + 3) // 4
st = torch.zeros(waves * (24 + 128), dtype=torch.int64, device="cuda:0")
fn = ctypes.CDLL(os.environ["SIP_LQR_LIB"]).sip_lqr_debug_set_stamps
fn.argtypes = [ctypes.c_void_p]
sol = solver.empty_sol(); gains = solver.empty_gains()
for it in range(5):
    fn(ctypes.c_void_p(st.data_ptr()))
    solver.factor_solve(mats, vecs, sol, gains)
    torch.cuda.synchronize()
raw = st.cpu().numpy()
s = raw[:waves * 24].reshape(waves, 24).astype(np.float64)
tot = s[:, 4] - s[:, 0]
print("kernel:", solver.kernel_name)
print("waves", waves, "cycles per wave: total median %.0f  min %.0f max %.0f" % (np.median(tot), tot.min(), tot.max()))
print("  terminal node      %.0f" % np.median(s[:, 1] - s[:, 0]))
print("  backward loop      %.0f  (per stage %.0f), of which waiting for the stage DMA %.0f (per stage %.0f)" % (
    np.median(s[:, 2] - s[:, 1]), np.median(s[:, 2] - s[:, 1]) / T, np.median(s[:, 5]), np.median(s[:, 5]) / T))
print("  root + sync        %.0f" % np.median(s[:, 3] - s[:, 2]))
print("  forward loop       %.0f  (per stage %.0f), of which waiting for the stage DMA %.0f (per stage %.0f)" % (
    np.median(s[:, 4] - s[:, 3]), np.median(s[:, 4] - s[:, 3]) / T, np.median(s[:, 6]), np.median(s[:, 6]) / T))
span = (s[:, 4].max() - s[:, 0].min())
print("  first start -> last end (shader clocks, per-XCD counters may differ): %.0f" % span)
print("  start skew: %.0f" % (s[:, 0].max() - s[:, 0].min()))
names = ["wait DMA", "LDS reads + DMA issue", "F = W [A|t] (+g store)", "Hc, G products", "LDL(G)",
         "H, K solve, gains store", "V += A^T F + K^T H", "status/t/vch", "node_factor (S, LDL F, F^-1, W)",
         "W store + loop tail", "-", "-"]
print("  backward segments, cycles per stage (median over waves):")
for k in range(10):
    print("    %-36s %7.0f" % (names[k], np.median(s[:, 8 + k]) / T))

if not c4:
    tr = raw[waves * 24:].reshape(waves, 128).astype(np.float64)
    for name, lo, end in (("backward", 0, s[:, 2]), ("forward", 64, s[:, 4])):
        cols = tr[:, lo:lo + T]
        if lo == 0:
            cols = cols[:, ::-1]  # the backward loop runs i = T-1 .. 0
        t = np.concatenate([cols, end[:, None]], axis=1)
        d = np.median(np.diff(t, axis=1), axis=0)
        print("  %s: cycles of each stage in loop order (median over waves):" % name)
        for a in range(0, T, 10):
            print("    " + " ".join("%6.0f" % v for v in d[a:a + 10]))
