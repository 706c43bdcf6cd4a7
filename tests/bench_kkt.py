#!/usr/bin/env python3
"""TEST / MEASUREMENT INFRASTRUCTURE (times the CPU oracle beside the GPU path and checks against it).
Newton-KKT factor+solve throughput on one MI355X: the
loop body of the reference's BM_NewtonKKTFactorSolve
(benchmarks/newton_kkt_benchmark.cpp:316-324) over a batch of NewtonKKTProblem
(n, m, T) instances (c = n/2 equality and g = 2m inequality rows per edge and
on the last node), device-resident.

    python tests/bench_kkt.py [--n 12 --m 4 --T 50 --batch 4096] [--steps 30]

Prints one JSON line: solves/s, the per-launch split (condense+rhs | Riccati |
recover), the HBM roofline of the whole step (algorithmic bytes = model +
w, r1, r2, r3, b read once, sol written once) and the oracle on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _time(fn, steps, warmup=2):
    import torch
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def theta_main(args):
    import numpy as np
    import torch
    from sip_optimal_control_amd import BatchedNewtonKKT
    from oracle.kkt import KKTDims, KKTOracle
    from tests import reference_kkt_problems as rk
    n, m, T, batch, p = args.n, args.m, args.T, args.batch, args.theta
    base = rk.newton_kkt_dims(n, m, T)
    dims = KKTDims(base.parents, base.children, base.sd, base.cd, base.ncd, base.ngd, base.ecd, base.egd, theta_dim=p)
    few = 8  # distinct problems (host generator), tiled over the batch
    model, w, r1, r2, r3, rhs, theta_model = rk.newton_kkt_problem(dims, seed=0, batch=few, r2_max=1e2)
    tile = lambda a: torch.from_numpy(np.ascontiguousarray(np.tile(a, (batch // few, 1)))).cuda()
    d = [tile(a) for a in (model, theta_model, w, r1, r2, r3, rhs)]
    kkt = BatchedNewtonKKT(dims.parents, dims.children, dims.sd, dims.cd, dims.ncd, dims.ngd, dims.ecd, dims.egd,
                           batch=batch, theta_dim=p)
    sol = torch.zeros(batch, dims.full_dim, dtype=torch.float64, device="cuda")
    ms_factor = _time(lambda: kkt.factor_theta(*d[:6]), args.steps)
    assert int((kkt.status != 0).sum()) == 0
    ms_solve = _time(lambda: kkt.solve_theta(d[0], d[1], d[6], sol=sol), args.steps)
    ms_split0 = _time(lambda: kkt.factor(d[0], d[2], d[3][:, :dims.x_dim].contiguous(), d[4], d[5]), args.steps)
    yk = torch.zeros(batch, dims.full_dim, dtype=torch.float64, device="cuda")
    # y += K x with the theta blocks: the loop body of BM_NewtonKKTThetaResidual (newton_kkt_benchmark.cpp:417-441)
    ms_apply = _time(lambda: kkt.add_Kx_to_y_theta(*d[:6], sol, y=yk), args.steps)
    o = KKTOracle(dims)
    assert o.factor_theta(model[0], theta_model[0], w[0], r1[0], r2[0], r3[0]) == 0
    ref = o.solve_theta(model[0], theta_model[0], rhs[0])
    err = float(np.abs(sol[0].cpu().numpy() - ref).max() / np.abs(ref).max())
    print(json.dumps({"metric": "newton_kkt_theta_factor_and_solve_ms", "theta_dim": p, "batch": batch,
                      "config": f"NewtonKKTProblem(n={n}, m={m}, T={T}, p={p})", "riccati": kkt.kernel_name,
                      "ms_factor_theta": ms_factor, "ms_solve_theta": ms_solve, "ms_factor_stagewise": ms_split0, "ms_add_Kx_to_y_theta": ms_apply,
                      "factor_theta_per_sec": batch / (ms_factor * 1e-3), "solve_theta_per_sec": batch / (ms_solve * 1e-3),
                      "max_rel_err_vs_oracle": err}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=12)
    ap.add_argument("--m", type=int, default=4)
    ap.add_argument("--T", type=int, default=50)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--theta", type=int, default=0,
                    help="theta_dim p > 0: time the split factor_theta / solve_theta path instead "
                         "(NewtonKKTProblem(n, m, T, p), newton_kkt_benchmark.cpp:388-449)")
    args = ap.parse_args()
    if args.theta > 0:
        return theta_main(args)
    import numpy as np
    import torch
    from sip_optimal_control_amd import BatchedNewtonKKT, synthetic
    n, m, T, batch = args.n, args.m, args.T, args.batch
    c, g = max(1, n // 2), max(1, 2 * m)
    dims = dict(parents=list(range(T)), children=list(range(1, T + 1)), state_dims=[n] * (T + 1),
                control_dims=[m] * T, node_c_dims=[0] * T + [c], node_g_dims=[0] * T + [g],
                edge_c_dims=[c] * T, edge_g_dims=[g] * T)
    kkt = BatchedNewtonKKT(batch=batch, **dims)
    data = synthetic.make_newton_kkt_batch(kkt, seed=0, r2_max=1e2, **dims)
    sol = torch.zeros(batch, kkt.kkt_dim, dtype=torch.float64, device=kkt.device)
    for _ in range(args.warmup):
        kkt.factor_solve(*data, sol=sol)
    torch.cuda.synchronize()
    assert int((kkt.status != 0).sum()) == 0, "factorization failed on the synthetic batch"
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(args.steps):
        kkt.factor_solve(*data, sol=sol)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / args.steps
    # residual through the GPU operator (the benchmark's residual_norm counter, :110-124)
    prod = kkt.add_Kx_to_y(*data[:5], sol)
    res = float((prod - data[5]).norm(dim=1).max())
    # ... and its own rate: y += K x, the loop body of BM_NewtonKKTResidual (newton_kkt_benchmark.cpp:417-441)
    ms_apply = _time(lambda: kkt.add_Kx_to_y(*data[:5], sol, y=prod), max(5, args.steps // 2))
    alg_bytes = 8 * (kkt.model_len + 2 * kkt.z_dim + kkt.x_dim + kkt.y_dim + 2 * kkt.kkt_dim)
    out = {
        "metric": "newton_kkt_factor_solves_per_sec", "value": batch / (ms * 1e-3), "unit": "solves/s",
        "ms_per_step": ms, "n_gpus": 1, "steps": args.steps, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"NewtonKKTProblem(n={n}, m={m}, T={T}), c={c}, g={g}, batch {batch}",
                   "riccati": kkt.kernel_name},
        "max_residual_norm": res,
        "ms_add_Kx_to_y": ms_apply, "Kx_per_sec": batch / (ms_apply * 1e-3),
        "roofline": {"bound": "hbm", "achieved": batch * alg_bytes / (ms * 1e-3) / 1e9, "peak": 8000.0,
                     "unit": "GB/s", "frac": batch * alg_bytes / (ms * 1e-3) / 1e9 / 8000.0, "traffic": None},
        "algorithmic_bytes_per_solve": alg_bytes,
    }
    if args.cpu_seconds > 0:
        from oracle.kkt import KKTDims, KKTOracle
        sys.path.insert(0, ROOT)
        from bench import usable_cores
        od = KKTDims(dims["parents"], dims["children"], dims["state_dims"], dims["control_dims"],
                     dims["node_c_dims"], dims["node_g_dims"], dims["edge_c_dims"], dims["edge_g_dims"])
        cores = usable_cores()
        sample = min(batch, max(128, 8 * cores))
        host = [a[:sample].cpu().numpy() for a in data]
        o = KKTOracle(od)
        o.batch(*[a[:4] for a in host], threads=1)
        t0 = time.perf_counter()
        ref, st = o.batch(*host, threads=cores)
        once = time.perf_counter() - t0
        reps = max(1, int(args.cpu_seconds / max(once, 1e-6)))
        t0 = time.perf_counter()
        for _ in range(reps):
            o.batch(*host, threads=cores)
        dt = time.perf_counter() - t0
        err = float(np.abs(sol[:sample].cpu().numpy() - ref).max() / np.abs(ref).max())
        out["cpu_baseline"] = {"value": sample * reps / dt, "unit": "solves/s", "cores": cores, "kind": "port",
                               "sample": f"{sample} problems x {reps} on {cores} threads, oracle/kkt_oracle.c"}
        out["max_rel_err_vs_oracle"] = err
    print(json.dumps(out))


if __name__ == "__main__":
    main()
