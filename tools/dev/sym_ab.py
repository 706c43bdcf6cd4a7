"""Development aid: full against symmetric-packed layout of the C3 kernel, interleaved launches on one box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sip_optimal_control_amd import BatchedChainLQR, ChainShape, synthetic
n, m, T, batch = 12, 4, 50, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
shape = ChainShape(n, m, T)
mats, vecs = synthetic.make_chain_batch(shape, batch, seed=1, device="cuda:0")
sym = mats[:, torch.from_numpy(shape.packed().pack_index()).to("cuda:0")].contiguous()
full, packed = BatchedChainLQR(n, m, T, batch), BatchedChainLQR(n, m, T, batch, symmetric=True)
sol, gains = full.empty_sol(), full.empty_gains()
def timed(s, a, reps=30):
    for _ in range(3):
        s.factor_solve(a, vecs, sol, gains)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        s.factor_solve(a, vecs, sol, gains)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for r in range(4):
    print("round", r, "full %.4f ms   symmetric-packed %.4f ms" % (timed(full, mats), timed(packed, sym)), flush=True)
