#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (uses the CPU oracle).  Randomized parity stress on the GPU box: many (n, m, T, batch) chain shapes through the fused /
embedded / general paths (fused and split entry points, injected failures), random trees with
per-node dimensions, and Newton-KKT problems with random constraint dimensions -- each against the
CPU oracle.  Prints one line per failure and a summary; exit code 1 on any failure.

    python tests/stress.py [--seed 0] [--chains 120] [--trees 25] [--kkt 40]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch


def rel(a, b):
    scale = np.abs(b).max(axis=-1, keepdims=True)
    scale[scale == 0] = 1.0
    return float((np.abs(a - b) / scale).max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--chains", type=int, default=120)
    ap.add_argument("--trees", type=int, default=25)
    ap.add_argument("--kkt", type=int, default=40)
    ap.add_argument("--max-horizon", type=int, default=13, help="chains: T is drawn from [0, max-horizon]")
    ap.add_argument("--max-batch", type=int, default=22, help="chains: batch is drawn from [1, max-batch]")
    args = ap.parse_args()
    from oracle import oracle
    from oracle.kkt import KKTDims, KKTOracle
    from sip_optimal_control_amd import BatchedChainLQR, BatchedNewtonKKT, ChainShape, synthetic
    from sip_optimal_control_amd.tree import BatchedTreeLQR
    import reference_kkt_problems as rk
    rng = np.random.default_rng(args.seed)
    failures, kernels = [], {}

    for it in range(args.chains):
        n, m = int(rng.integers(1, 19)), int(rng.integers(1, 10))
        T, batch = int(rng.integers(0, args.max_horizon + 1)), int(rng.integers(1, args.max_batch + 1))
        sh = ChainShape(n, m, T)
        mats, vecs = synthetic.make_chain_batch(sh, batch, seed=1000 + it, device="cuda:0", cross_term=0.02)
        bad = None
        if T > 0 and batch > 1 and rng.random() < 0.4:  # one failing problem
            bad = int(rng.integers(0, batch))
            kind = int(rng.integers(0, 3))
            stage = int(rng.integers(0, T))
            if kind == 0:
                off = sh.mats_off(stage)["R"]
                mats[bad, off:off + m * m] = -torch.eye(m, dtype=torch.float64, device="cuda:0").reshape(-1)
            elif kind == 1:
                mats[bad, sh.mats_off(stage)["delta"]] = -1.0
            else:
                off = sh.mats_off(stage)["Q"]
                mats[bad, off:off + n * n] = -1e3 * torch.eye(n, dtype=torch.float64, device="cuda:0").reshape(-1)
        os.environ["SIP_LQR_EXTRA"] = "0" if rng.random() < 0.25 else "1"  # a quarter through the embedding
        s = BatchedChainLQR(n, m, T, batch)
        os.environ.pop("SIP_LQR_EXTRA")
        kernels[s.kernel_name.split("/")[0].split(" embedding")[0] + (" (embedded)" if "embedding" in s.kernel_name else "")] = 1
        sol, gains, st = s.factor_solve(mats, vecs)
        torch.cuda.synchronize()
        ref_sol, ref_gains, ref_st = oracle.chain_batch(n, m, T, mats.cpu().numpy(), vecs.cpu().numpy())
        tag = f"chain n={n} m={m} T={T} batch={batch} bad={bad} [{s.kernel_name}]"
        if st.cpu().numpy().tolist() != ref_st.tolist():
            failures.append(tag + f": status {st.cpu().numpy().tolist()} vs {ref_st.tolist()}")
            continue
        ok = ref_st == 0
        if ok.any():
            e = rel(sol.cpu().numpy()[ok], ref_sol[ok])
            eg = rel(gains.cpu().numpy()[ok], ref_gains[ok]) if T > 0 else 0.0
            if not (e <= 1e-9 and eg <= 1e-9):
                failures.append(tag + f": fused rel err sol {e:.2e} gains {eg:.2e}")
            if T > 0:
                g2, st2 = s.factor(mats)
                s2 = s.solve(mats, vecs, g2)
                torch.cuda.synchronize()
                e2 = rel(s2.cpu().numpy()[ok], ref_sol[ok])
                if st2.cpu().numpy().tolist() != ref_st.tolist() or not e2 <= 1e-9:
                    failures.append(tag + f": split rel err {e2:.2e} status {st2.cpu().numpy().tolist()}")

    for it in range(args.trees):
        N = int(rng.integers(2, 14))
        parents = [int(rng.integers(0, e + 1)) for e in range(N - 1)]
        children = list(range(1, N))
        sd = [int(rng.integers(0 if i else 1, 8)) for i in range(N)]
        cd = [int(rng.integers(1, 5)) for _ in range(N - 1)]
        batch = int(rng.integers(1, 6))
        probs = []
        for b in range(batch):
            blocks = {k: [] for k in ("Q", "M", "R", "q", "r", "A", "B", "c", "delta")}
            for nn in sd:
                S = rng.normal(size=(nn, nn))
                blocks["Q"].append(S.T @ S + 1e-3 * np.eye(nn)); blocks["q"].append(rng.normal(size=nn))
                blocks["c"].append(rng.normal(size=nn)); blocks["delta"].append(1e-3 + 0.1 * rng.random(nn))
            for e, mm in enumerate(cd):
                np_, nc = sd[parents[e]], sd[children[e]]
                G = rng.normal(size=(mm, mm))
                blocks["A"].append(0.3 * rng.normal(size=(nc, np_))); blocks["B"].append(0.3 * rng.normal(size=(nc, mm)))
                blocks["M"].append(0.05 * rng.normal(size=(np_, mm))); blocks["R"].append(G.T @ G + 1.01 * np.eye(mm))
                blocks["r"].append(rng.normal(size=mm))
            probs.append(blocks)
        s = BatchedTreeLQR(parents, children, sd, cd, batch=batch)
        s.pack(probs)
        st = s.factor()
        s.solve()
        torch.cuda.synchronize()
        st = st.cpu().numpy()
        for b in range(batch):
            lqr = oracle.TreeLQR(parents, children, sd, cd, probs[b])
            if lqr.factor() != st[b]:
                failures.append(f"tree {it} parents={parents} sd={sd} cd={cd}: status {st[b]}")
                continue
            if st[b] != 0:
                continue
            xo, uo, yo = lqr.solve()
            x, u, y = s.unpack_solution(b)
            for a, bb in list(zip(x, xo)) + list(zip(u, uo)) + list(zip(y, yo)):
                if bb.size and np.abs(a - bb).max() > 1e-10 * max(1.0, np.abs(bb).max()):
                    failures.append(f"tree {it} parents={parents} sd={sd} cd={cd}: err {np.abs(a - bb).max():.2e}")
                    break

    for it in range(args.kkt):
        chain = rng.random() < 0.6
        if chain:
            n, m, T = int(rng.integers(1, 18)), int(rng.integers(1, 7)), int(rng.integers(1, 9))
            cn, gn = (int(rng.integers(0, 3)), int(rng.integers(0, 3))) if rng.random() < 0.5 else (0, 0)
            dims = KKTDims(list(range(T)), list(range(1, T + 1)), [n] * (T + 1), [m] * T,
                           node_c=[cn] * T + [int(rng.integers(0, 4))], node_g=[gn] * T + [int(rng.integers(0, 4))],
                           edge_c=[int(rng.integers(0, 5))] * T, edge_g=[int(rng.integers(0, 6))] * T)
        else:
            N = int(rng.integers(2, 9))
            parents = [int(rng.integers(0, e + 1)) for e in range(N - 1)]
            dims = KKTDims(parents, list(range(1, N)), [int(rng.integers(1, 6)) for _ in range(N)],
                           [int(rng.integers(1, 4)) for _ in range(N - 1)],
                           node_c=[int(rng.integers(0, 3)) for _ in range(N)], node_g=[int(rng.integers(0, 3)) for _ in range(N)],
                           edge_c=[int(rng.integers(0, 3)) for _ in range(N - 1)],
                           edge_g=[int(rng.integers(0, 3)) for _ in range(N - 1)])
        batch = int(rng.integers(1, 7))
        arrays = rk.newton_kkt_problem(dims, seed=3000 + it, batch=batch, r2_max=1e2)
        kkt = BatchedNewtonKKT(dims.parents, dims.children, dims.sd, dims.cd, dims.ncd, dims.ngd, dims.ecd, dims.egd,
                               batch=batch)
        d = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in arrays]
        sol, st = kkt.factor_solve(*d)
        torch.cuda.synchronize()
        ref, ref_st = KKTOracle(dims).batch(*arrays)
        tag = f"kkt {'chain' if chain else 'tree'} sd={dims.sd} cd={dims.cd} nc={dims.ncd} ng={dims.ngd} ec={dims.ecd} eg={dims.egd} [{kkt.kernel_name}]"
        kernels["kkt:" + kkt.kernel_name.split("+")[-1].strip()] = 1
        if st.cpu().numpy().tolist() != ref_st.tolist():
            failures.append(tag + f": status {st.cpu().numpy().tolist()} vs {ref_st.tolist()}")
            continue
        e = rel(sol.cpu().numpy(), ref)
        if not e <= 1e-8:
            failures.append(tag + f": rel err {e:.2e}")
        kkt.factor(*d[:5])
        e2 = rel(kkt.solve(d[0], d[5]).cpu().numpy(), ref)
        if not e2 <= 1e-8:
            failures.append(tag + f": split rel err {e2:.2e}")

    for f in failures:
        print("FAIL", f)
    print(f"stress: {args.chains} chains, {args.trees} trees, {args.kkt} kkt problems; {len(failures)} failures; "
          f"paths exercised: {sorted(kernels)}")
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
