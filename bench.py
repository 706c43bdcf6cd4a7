#!/usr/bin/env python3
"""Riccati sweeps/sec of the MI355X batched regularized-LQR path.

One "step" = one fused factor+solve sweep (the loop body of the reference's
BM_LQRFactorSolve, benchmarks/lqr_benchmark.cpp:653-663) over one batch of
synthetic chain problems (recipe of lqr_benchmark.cpp:61-98) already resident
in HBM, per GPU; for N > 1 GPUs the batch is sharded by problem (weak scaling:
every rank owns `batch` problems) and each step's gains (K, k) are all-gathered
over RCCL/xGMI, pipelined against the next step's compute.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3]

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# BASELINE.json configs: name -> (batch, T, n, m, dtype)
WORKLOADS = {
    "c1": (1, 20, 4, 2, "f64"),
    "c2": (1024, 50, 12, 4, "f64"),
    "c3": (4096, 50, 12, 4, "f64"),
    "c4": (4096, 100, 32, 8, "f32"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS) + ["kkt"],
                    help="c1..c4: BASELINE.json configs (Riccati sweeps); kkt: the Newton-KKT step around "
                         "the sweep (SURVEY.md 8 row f1, NewtonKKTProblem(12, 4, 50), batch 4096)")
    ap.add_argument("--batch", type=int, default=0, help="override per-GPU batch")
    ap.add_argument("--no-gather", action="store_true",
                    help="multi-GPU: skip the RCCL all-gather of the gains (the default N > 1 run reports "
                         "both: `value` with the gather, `no_gather` beside it)")
    ap.add_argument("--gather-chunks", type=int, default=8,
                    help="multi-GPU: chunk gathers per sweep (SURVEY.md 8(e): 8 x 512 problems)")
    ap.add_argument("--gather-mode", default="both", choices=["both", "allgather", "mesh"],
                    help="multi-GPU: form of the exchange (sharding.GainsAllGather.mode); both: time each, "
                         "`value` is the better one and the line carries both")
    ap.add_argument("--blocks", type=int, default=5,
                    help="timed blocks of --steps steps each; the line reports the MEDIAN block "
                         "(ms_per_step, value) with min / max beside it")
    ap.add_argument("--layout", default="auto", choices=["auto", "full", "sym"],
                    help="layout of `mats` in HBM (include/sip_lqr_amd.h): full squares as LQR::Input holds them, or "
                         "Q and R as packed lower triangles (SIP_LQR_LAYOUT_SYMMETRIC); auto: sym where the shape has "
                         "the kernel and the launch is bandwidth-bound (c3), and the line then carries the full-layout figure too")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="target wall time of each cpu_baseline leg")
    return ap.parse_args()


def library_version():
    """'sip_lqr_amd <ver> (gfx950) SIPLQRSRC=<sha256 of the sources the library was built from>'."""
    from sip_optimal_control_amd._lib import load_library
    return load_library().sip_lqr_version().decode()


def recorded_traffic(key):
    """HBM bytes per launch from the committed PMC passes (profiles/traffic.json), with where they come
    from: counters are collected in separate rocprofv3 --pmc runs (tools/profile_gpu.sh), never inside a
    bench run, so the line says so instead of passing the number off as measured live."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        rec = json.load(open(path)).get(key)
    except Exception:
        rec = None
    if not rec:
        return None, None
    src = (f"profiles/traffic.json[{key}]: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
           f"({rec.get('profile', 'source file not recorded')}), NOT measured in this run")
    return rec["hbm_bytes_per_launch"], src


def usable_cores():
    """Host cores this process may really use: min(affinity, cgroup CPU quota)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    env = os.environ.get("SIP_LQR_BENCH_CORES")
    if env:
        cores = int(env)
    return max(1, cores)


def cpu_baseline(shape, mats, vecs, seconds):
    """Time the CPU oracle ("port": Eigen-free restatement of lqr.cpp, see
    oracle/lqr_oracle.h) on a bounded sample of the same workload."""
    from oracle import oracle
    import numpy as np
    cores = usable_cores()
    n, m, T = shape.n, shape.m, shape.T
    sample = min(mats.shape[0], max(256, 16 * cores))
    hm = mats[:sample].cpu().numpy().astype(np.float64)
    hv = vecs[:sample].cpu().numpy().astype(np.float64)
    oracle.chain_batch(n, m, T, hm[:8], hv[:8], threads=1, want_gains=True)  # warm

    def timed(threads, budget, native):
        # calibrate, then repeat the sample until ~budget seconds have passed
        t0 = time.perf_counter()
        oracle.chain_batch(n, m, T, hm, hv, threads=threads, native=native)
        once = time.perf_counter() - t0
        reps = max(1, int(budget / max(once, 1e-6)))
        t0 = time.perf_counter()
        for _ in range(reps):
            oracle.chain_batch(n, m, T, hm, hv, threads=threads, native=native)
        dt = time.perf_counter() - t0
        return sample * reps / dt, reps

    # two builds of the same C restatement: the parity oracle (-march=x86-64-v3 -ffp-contract=off) and the
    # flags BASELINE.md section 3 promises for the timed baseline (-O3 -march=native, FMA on); `value` is the
    # faster of the two
    try:
        oracle.chain_batch(n, m, T, hm[:8], hv[:8], threads=1, native=True)
        native_ok = True
    except Exception as exc:  # no compiler on the host: say so, keep the parity build's figure
        native_ok, native_err = False, repr(exc)
    one_p, _ = timed(1, seconds * 0.15, False)
    all_p, _ = timed(cores, seconds * 0.25, False)
    out = {"unit": "sweeps/s", "cores": cores, "kind": "port",
           "parity_build": {"flags": "-O3 -march=x86-64-v3 -ffp-contract=off", "value": all_p, "value_1core": one_p}}
    if native_ok:
        one_n, reps1 = timed(1, seconds * 0.25, True)
        all_n, repsn = timed(cores, seconds * 0.35, True)
        out["native_build"] = {"flags": "-O3 -march=native -ffp-contract=fast", "value": all_n, "value_1core": one_n}
        out["value"], out["value_1core"] = max(all_n, all_p), max(one_n, one_p)
    else:
        out["native_build"] = {"error": native_err}
        out["value"], out["value_1core"] = all_p, one_p
    out["sample"] = (f"{sample} problems of the same synthetic batch, repeated for ~{seconds:.0f} s in all on {cores} "
                     f"threads (OpenMP over problems) and on 1 thread; oracle/lqr_oracle.c (Eigen-free restatement "
                     f"of lqr.cpp; the Eigen reference binary cannot be built here), two compiler-flag builds")
    return out


def kkt_cpu_baseline(dims, data, seconds):
    """The CPU oracle of the Newton-KKT callbacks (oracle/kkt_oracle.c) on a bounded sample."""
    from oracle.kkt import KKTDims, KKTOracle
    cores = usable_cores()
    od = KKTDims(dims["parents"], dims["children"], dims["state_dims"], dims["control_dims"], dims["node_c_dims"],
                 dims["node_g_dims"], dims["edge_c_dims"], dims["edge_g_dims"])
    sample = min(data[0].shape[0], max(128, 8 * cores))
    host = [a[:sample].cpu().numpy() for a in data]
    o = KKTOracle(od)
    o.batch(*[a[:4] for a in host], threads=1)
    t0 = time.perf_counter()
    ref, _ = o.batch(*host, threads=cores)
    once = time.perf_counter() - t0
    reps = max(1, int(seconds / max(once, 1e-6)))
    t0 = time.perf_counter()
    for _ in range(reps):
        o.batch(*host, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": sample * reps / dt, "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": f"{sample} problems x {reps} on {cores} threads, oracle/kkt_oracle.c (Eigen-free "
                      f"restatement of CallbackProvider::factor + solve)"}, ref


def kkt_main(args):
    """One step = CallbackProvider::factor + ::solve (the loop body of BM_NewtonKKTFactorSolve,
    benchmarks/newton_kkt_benchmark.cpp:316-324) of every problem of the batch, device-resident."""
    import numpy as np
    import torch
    from sip_optimal_control_amd import BatchedNewtonKKT, synthetic
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        raise SystemExit("--workload kkt is a one-GPU measurement")
    n, m, T = 12, 4, 50
    batch = args.batch or 4096
    c, g = n // 2, 2 * m
    dims = dict(parents=list(range(T)), children=list(range(1, T + 1)), state_dims=[n] * (T + 1),
                control_dims=[m] * T, node_c_dims=[0] * T + [c], node_g_dims=[0] * T + [g],
                edge_c_dims=[c] * T, edge_g_dims=[g] * T)
    torch.cuda.set_device(0)
    kkt = BatchedNewtonKKT(batch=batch, **dims)
    data = synthetic.make_newton_kkt_batch(kkt, seed=0, r2_max=1e2, **dims)
    sol = torch.zeros(batch, kkt.kkt_dim, dtype=torch.float64, device=kkt.device)
    for _ in range(args.warmup):
        kkt.factor_solve(*data, sol=sol)
    torch.cuda.synchronize()
    block_s = []  # `blocks` timed blocks of exactly `steps` steps; the line reports the median block
    for _ in range(max(1, args.blocks)):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            kkt.factor_solve(*data, sol=sol)
        torch.cuda.synchronize()
        block_s.append(time.perf_counter() - t0)
    elapsed = sorted(block_s)[len(block_s) // 2]
    ms = elapsed / args.steps * 1e3
    alg_bytes = 8 * (kkt.model_len + 2 * kkt.z_dim + kkt.x_dim + kkt.y_dim + 2 * kkt.kkt_dim)
    achieved = batch * alg_bytes / (ms * 1e-3) / 1e9
    kkt_traffic, kkt_traffic_src = recorded_traffic(f"kkt:{kkt.kernel_name}") if batch == 4096 else (None, None)
    out = {
        "metric": "Newton-KKT factor+solves/sec", "value": batch * args.steps / elapsed, "unit": "solves/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "blocks": max(1, args.blocks),
        "ms_per_step_min": min(block_s) / args.steps * 1e3, "ms_per_step_max": max(block_s) / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"kkt: NewtonKKTProblem(n={n}, m={m}, T={T}), c={c}, g={g}, batch {batch}",
                   "kernels": kkt.kernel_name, "all_status_success": bool((kkt.status == 0).all().item())},
        # whole step (3 launches + the Riccati sweep): algorithmic bytes = model + w, r1, r2, r3, b read
        # once + sol written once; PMC traffic of the step in profiles/r01_kkt/traffic_end.md
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": kkt_traffic, "traffic_source": kkt_traffic_src,
                     "algorithmic_bytes_per_launch": alg_bytes * batch},
    }
    out["config"]["library"] = library_version()
    if not args.no_cpu_baseline:
        out["cpu_baseline"], ref = kkt_cpu_baseline(dims, data, args.cpu_seconds)
        got = sol[:ref.shape[0]].cpu().numpy()
        out["max_rel_err_vs_oracle"] = float(np.abs(got - ref).max() / np.abs(ref).max())
    print(json.dumps(out), flush=True)


def free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` from a plain shell (no WORLD_SIZE): start N fresh rank processes, one
    per GPU, and relay rank 0's JSON line.  This parent never imports torch and never touches the GPU
    (a process that has initialised the GPU must not spawn-and-replace, and the children need the
    devices to themselves); it exits non-zero if any rank does."""
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # rank 0 inherits stdout (the one JSON line); the other ranks' stdout goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    deadline = time.time() + float(os.environ.get("SIP_LQR_BENCH_TIMEOUT", "1500"))
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs) or time.time() > deadline:
            # one rank failed (or the run hangs): the others would wait in a collective forever
            time.sleep(10)
            for p in procs:  # exact PIDs of the children started above
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    rcs = [p.wait() for p in procs]
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        raise SystemExit(f"bench.py --gpus {args.gpus}: ranks failed (rank, exit code): {bad}")


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    if args.workload == "kkt":
        return kkt_main(args)
    import torch
    import torch.distributed as dist
    from sip_optimal_control_amd import BatchedChainLQR, ChainShape, synthetic
    from sip_optimal_control_amd.sharding import GainsAllGather

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE=1")
    # one rank per GPU; (modulo only matters for rehearsals with more ranks than GPUs)
    ndev = max(1, torch.cuda.device_count())
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    backend = None
    if world > 1:
        # nccl == RCCL on ROCm.  RCCL refuses two ranks on one device, so a rehearsal with more ranks
        # than GPUs (the one-GPU box) runs the same control flow over gloo (host-staged exchange) and
        # says so in the line.
        backend = os.environ.get("SIP_LQR_BENCH_BACKEND") or ("nccl" if world <= ndev else "gloo")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:  # rehearsal of the N > 1 control flow on a box without N GPUs
            dist.init_process_group(backend)

    batch, T, n, m, dt = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    dtype = torch.float64 if dt == "f64" else torch.float32
    esize = 8 if dt == "f64" else 4
    shape = ChainShape(n, m, T)
    mats, vecs = synthetic.make_chain_batch(shape, batch, seed=1234 + rank, device=device, dtype=dtype)
    full_mats, full_solver = mats, None
    # (auto: where the launch is bandwidth-bound; at batch 1024 -- one wavefront per CU, latency-bound -- the packed
    # triangles' address arithmetic costs 1 % instead of saving 6 %)
    sym = args.layout == "sym" or (args.layout == "auto" and dt == "f64" and (n, m) == (12, 4) and batch >= 2048)
    if sym:
        # the same problems with Q and R as packed lower triangles (the reference takes Q symmetric, lqr.cpp:658,
        # and reads the lower triangle of R only, lqr.cpp:697): 576 B less per problem-stage at (12, 4)
        import numpy as np  # noqa: F401
        idx = torch.from_numpy(shape.packed().pack_index()).to(device)
        mats = full_mats[:, idx].contiguous()
        full_solver = BatchedChainLQR(n, m, T, batch, dtype=dtype, device=device)
    solver = BatchedChainLQR(n, m, T, batch, dtype=dtype, device=device, symmetric=sym)
    sol = solver.empty_sol()
    gather = world > 1 and not args.no_gather
    modes = [] if not gather else (["allgather", "mesh"] if args.gather_mode == "both" else [args.gather_mode])
    gains = [solver.empty_gains(), solver.empty_gains()]

    compute = torch.cuda.current_stream(device)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    def run(ag, solver=solver, mats=mats):
        """W warmup steps, then `blocks` timed blocks of exactly K steps, each between barrier +
        synchronize fences; per block the max-over-ranks seconds.  Returns (list of block seconds,
        mean kernel ms of this rank over all timed steps)."""
        events = []

        def step(i, timed):
            # gains are double-buffered; with the gather on, buffer i%2 is only rewritten once every
            # chunk gather of sweep i-2 has drained it
            out_gains = ag.acquire(i) if ag else gains[i & 1]
            if timed:
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(compute)
            solver.factor_solve(mats, vecs, sol, out_gains)
            if timed:
                e1.record(compute)
                events.append((e0, e1))
            if ag:
                ag.mark_ready(i)  # one launch produces every chunk: all chunks ready behind it
                ag.launch(i)      # chunk exchanges on the side stream, overlapping sweep i+1

        it = 0
        for _ in range(args.warmup):
            step(it, False)
            it += 1
        if ag:
            ag.finish()
        block_s = []
        for _ in range(max(1, args.blocks)):
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step(it, True)
                it += 1
            if ag:
                ag.finish()
            torch.cuda.synchronize(device)
            local = time.perf_counter() - t0
            fence()
            el = torch.tensor([local], dtype=torch.float64, device=device)
            if world > 1:
                dist.all_reduce(el, op=dist.ReduceOp.MAX)
            block_s.append(float(el.item()))
        return block_s, sum(a.elapsed_time(b) for a, b in events) / max(1, len(events))

    def summary(block_s, kernel_ms):
        med = sorted(block_s)[len(block_s) // 2]
        return {"value": world * batch * args.steps / med, "unit": "sweeps/s", "ms_per_step": med / args.steps * 1e3,
                "ms_per_step_min": min(block_s) / args.steps * 1e3, "ms_per_step_max": max(block_s) / args.steps * 1e3,
                "kernel_ms": kernel_ms}

    results = {}
    chunks = min(args.gather_chunks, batch)
    for mode in modes:
        ag = GainsAllGather(batch, shape.gains_len, dtype, device, chunks=chunks, mode=mode)
        results[mode] = summary(*run(ag))
        del ag
    if gather or not results:  # the same K steps without the exchange: independent shards, the compute-only bound
        results["no_gather"] = summary(*run(None))
    full_layout = None
    if sym and world == 1:  # the same sweeps on the full squares, timed the same way, beside the headline
        full_layout = summary(*run(None, full_solver, full_mats))
        full_layout["kernel"] = full_solver.kernel_name
    best = max(modes, key=lambda k: results[k]["value"]) if modes else "no_gather"
    head = results[best]
    elapsed, kernel_ms = head["ms_per_step"] * 1e-3 * args.steps, head["kernel_ms"]

    status_ok = bool((solver.status == 0).all().item())
    if rank == 0:
        alg_bytes = shape.algorithmic_bytes(esize)
        achieved = alg_bytes * batch / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_src = recorded_traffic(f"{args.workload}:{solver.kernel_name}")
        out = {
            "metric": "Riccati sweeps/sec (backward+forward)",
            "value": world * batch * args.steps / elapsed,
            "unit": "sweeps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            # each of `blocks` blocks times exactly `steps` steps between fences; value / ms_per_step are
            # the MEDIAN block's (box-to-box and launch-to-launch noise: VERDICT r02 weak #11)
            "blocks": max(1, args.blocks), "ms_per_step_min": head["ms_per_step_min"],
            "ms_per_step_max": head["ms_per_step_max"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dt,
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: batch={batch}/GPU x horizon={T}, nx={n}, nu={m}, {dt}, "
                            f"chain, fused factor+solve, mats layout: " +
                            ("Q, R as packed lower triangles (SIP_LQR_LAYOUT_SYMMETRIC)" if sym else "full squares") +
                            (f", all-gather of gains ({best} form) over {backend}" if gather else ""),
                "global_batch": world * batch, "horizon": T, "nx": n, "nu": m,
                "parallelism": f"batch-sharded x{world}" +
                               (f"+allgather(K,k) in {chunks} chunks/sweep, {best}" if gather else ""),
                "kernel": solver.kernel_name, "all_status_success": status_ok, "library": library_version(),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": alg_bytes * batch,
                # rate of the bytes the kernel really moves (PMC traffic / live launch time);
                # MI355X_MICROARCH.md: a float4 copy sustains ~6.3 TB/s on this part
                "traffic_gbs": (traffic / (kernel_ms * 1e-3) / 1e9) if traffic else None,
            },
        }
        if dt == "f32":
            # C4 sits above the fp32 ridge -> matrix-core roofline
            # (dense f32 MFMA peak 157.3 TFLOP/s, MI355X_MICROARCH.md); HBM fraction kept alongside
            tflops = shape.algorithmic_flops() * batch / (kernel_ms * 1e-3) / 1e12
            out["roofline"] = {
                "bound": "mfma", "achieved": tflops, "peak": 157.3, "unit": "TFLOP/s",
                "frac": tflops / 157.3, "traffic": traffic, "traffic_source": traffic_src, "kernel_ms": kernel_ms,
                "algorithmic_flops_per_launch": shape.algorithmic_flops() * batch,
                "hbm_achieved_gbs": achieved, "hbm_frac": achieved / HBM_PEAK_GBS,
            }
        if gather:
            # per sweep every rank sends its shard and receives (world - 1) shards of gains over xGMI
            shard = batch * shape.gains_len * esize
            out["gather"] = {"bytes_received_per_rank_per_sweep": (world - 1) * shard, "chunks": chunks,
                             "backend": backend + ("" if backend == "nccl" else " (REHEARSAL: host-staged, not xGMI)"),
                             "form_reported": best, "kernel_ms_beside_the_gather": kernel_ms}
            for mode in modes:
                out["gather"][mode] = results[mode]
            out["no_gather"] = results["no_gather"]
        if full_layout is not None:
            full_layout["roofline_frac"] = alg_bytes * batch / (full_layout["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["full_layout"] = full_layout
            out["roofline"]["bytes_per_launch_in_this_layout"] = batch * esize * (
                shape.packed().mats_len + 2 * shape.vecs_len + shape.gains_len)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(shape, full_mats, vecs, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
