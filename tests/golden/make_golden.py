#!/usr/bin/env python3
"""Generates the committed golden fixtures (run in the build container):

    python tests/golden/make_golden.py

The reference ships no golden vectors; these come from two
independent numpy derivations -- a dense KKT solve (oracle/dense_kkt.py, the
construction of the reference's own dense check, tests/lqr_test.cpp:859-929)
for x, u, y and a solve()-based Riccati (oracle/numpy_riccati.py) for K, k --
NOT from the C oracle and NOT from the HIP kernels they are used to check.

  chain_<name>.npz : packed inputs (mats, vecs) + sol (x|y|u packed) + gains
  tree_<name>.npz  : flattened blocks + x, u, y of the reference's tree fixtures
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import dense_kkt, numpy_riccati  # noqa: E402
from sip_optimal_control_amd import ChainShape, synthetic  # noqa: E402
import reference_problems as rp  # noqa: E402

CHAINS = {  # name -> (n, m, T, seeds, cross_term)
    "c1_n4_m2_T20": (4, 2, 20, [0, 1, 2], 0.01),
    "c3_n12_m4_T50": (12, 4, 50, [0, 1], 0.01),
    "c3_n12_m4_T50_M0": (12, 4, 50, [2], 0.0),     # the benchmark's M = 0
    "c4_n32_m8_T100": (32, 8, 100, [0], 0.01),
}


def pack_sol(x, u, y, n, m, T):
    out = []
    for i in range(T + 1):
        out += [x[i], y[i]]
        if i < T:
            out.append(u[i])
    return np.concatenate(out)


def main():
    for name, (n, m, T, seeds, cross) in CHAINS.items():
        shape = ChainShape(n, m, T)
        mats_all, vecs_all, sol_all, gains_all = [], [], [], []
        for seed in seeds:
            mats, vecs = synthetic.make_chain_batch(shape, 1, seed=1000 + seed, cross_term=cross)
            mats, vecs = mats[0].numpy(), vecs[0].numpy()
            blocks = dense_kkt.chain_blocks_from_packed(n, m, T, mats, vecs)
            par, ch = list(range(T)), list(range(1, T + 1))
            x, u, y = dense_kkt.solve(par, ch, [n] * (T + 1), [m] * T, blocks)
            res = dense_kkt.residual_norm(par, ch, [n] * (T + 1), [m] * T, blocks, x, u, y)
            Ks, ks, _, _ = numpy_riccati.chain_gains(blocks, n, m, T)
            gains = np.concatenate([np.concatenate([K.reshape(-1, order="F"), k]) for K, k in zip(Ks, ks)])
            print(f"{name} seed {seed}: dense KKT residual {res:.2e}")
            mats_all.append(mats), vecs_all.append(vecs)
            sol_all.append(pack_sol(x, u, y, n, m, T)), gains_all.append(gains)
        np.savez_compressed(os.path.join(HERE, f"chain_{name}.npz"), n=n, m=m, T=T,
                            mats=np.stack(mats_all), vecs=np.stack(vecs_all),
                            sol=np.stack(sol_all), gains=np.stack(gains_all))
    trees = {"nonuniform_diagonal_delta": rp.nonuniform_diagonal_delta(),
             "branch_tree": rp.branch_tree(),
             "variable_dimension_branch": rp.variable_dimension_branch(),
             "five_node_variable_tree": rp.five_node_variable_tree_eigen()}
    for name, prob in trees.items():
        x, u, y = dense_kkt.solve(prob["parents"], prob["children"], prob["state_dims"],
                                  prob["control_dims"], prob["blocks"])
        np.savez_compressed(os.path.join(HERE, f"tree_{name}.npz"),
                            parents=prob["parents"], children=prob["children"],
                            state_dims=prob["state_dims"], control_dims=prob["control_dims"],
                            x=np.concatenate(x), u=np.concatenate(u), y=np.concatenate(y),
                            **{f"blk_{k}": np.concatenate([np.asarray(b).reshape(-1, order="F") for b in v])
                               for k, v in prob["blocks"].items()})
        print(f"tree {name}: stored")


if __name__ == "__main__":
    main()
