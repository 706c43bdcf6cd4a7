#!/usr/bin/env python3
"""Turns what tools/evidence.sh left under gpurun_out/ into the committed evidence under profiles/:
bench lines, the rocprofv3 summaries of the three chain workloads (tools/summarize_prof.py), the per-kernel tables of
the Newton-KKT step and of the theta path (kernel trace + FETCH_SIZE / WRITE_SIZE passes), and the step's entry in
profiles/traffic.json.

    python tools/publish_profiles.py [tag]        (default r03)
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_line(path):
    lines = [l for l in open(path).read().strip().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def kernel_table(base):
    """[(kernel name, calls, average us, read MB, written MB)], sipamd kernels, by total time."""
    ks = list(csv.DictReader(open(glob.glob(base + "/kt/*/*_kernel_stats.csv")[0])))

    def pmc(kind):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(glob.glob(base + f"/{kind}/*/*_counter_collection.csv")[0])):
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        return {k: sum(v) / len(v) for k, v in agg.items()}

    fetch, write = pmc("fetch"), pmc("write")
    rows = []
    for r in sorted(ks, key=lambda r: -float(r["AverageNs"]) * int(r["Calls"])):
        k = r["Name"]
        if "sipamd" in k:
            rows.append((k, int(r["Calls"]), float(r["AverageNs"]) / 1e3, 2 * fetch.get(k, 0) * 1024 / 1e6,
                         write.get(k, 0) * 1024 / 1e6))
    return rows


def short(name):
    name = name.replace("void ", "").replace("sipamd::", "")
    return name[:name.index("(")] if "(" in name else name


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    out, prof, prof_out = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "gpurun_out", "prof"), os.path.join(ROOT, "profiles")
    for w in ("c2", "c3", "c4", "kkt"):
        shutil.copy(os.path.join(out, f"bench_{w}.json"), os.path.join(prof_out, f"{tag}_bench_{w}.json"))
    shutil.copy(os.path.join(out, "bench_theta8.json"), os.path.join(prof_out, f"{tag}_theta8_bench.json"))
    shutil.copy(os.path.join(out, "bench_tree.json"), os.path.join(prof_out, f"{tag}_tree_bench.json"))
    with open(os.path.join(prof_out, f"{tag}_bench_2rank_gloo_rehearsal.json"), "w") as f:
        f.write(json.dumps(last_line(os.path.join(out, "bench_g2_rehearsal.json"))) + "\n")
    for sub, wl, key in ((f"{tag}_c3", "c3", "true, true, false, true>"), (f"{tag}_c3full", "c3", "true, true, false, false>"),
                         (f"{tag}_c2", "c2", "true, true, false, false>"), (f"{tag}_c4", "c4", "mt16")):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarize_prof.py"), os.path.join(prof, sub), sub, wl, key],
                              stdout=subprocess.DEVNULL)
    # ---- Newton-KKT step ----
    kkt = last_line(os.path.join(out, "bench_kkt.json"))
    rows = kernel_table(os.path.join(prof, f"{tag}_kkt"))
    step = [r for r in rows if r[1] >= 10]  # the launches of the timed steps (not the one-off residual check)
    rd, wr, us = sum(r[3] for r in step), sum(r[4] for r in step), sum(r[2] for r in step)
    alg = kkt["roofline"]["algorithmic_bytes_per_launch"]
    os.makedirs(os.path.join(prof_out, f"{tag}_kkt"), exist_ok=True)
    shutil.copy(glob.glob(os.path.join(prof, f"{tag}_kkt", "kt", "*", "*_kernel_stats.csv"))[0],
                os.path.join(prof_out, f"{tag}_kkt", "step_kernel_stats.csv"))
    with open(os.path.join(prof_out, f"{tag}_kkt", "step.md"), "w") as f:
        f.write(f"# Newton-KKT step (f1) -- `tools/profile_kkt.sh {tag}_kkt`, table by `tools/publish_profiles.py`\n\n")
        f.write(f"Workload: {kkt['config']['workload']}, fp64 (`tests/bench_kkt.py`).  Kernels: {kkt['config']['kernels']}.\n")
        f.write("`rocprofv3 --kernel-trace --stats`, `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate passes (KiB, mean over the\n"
                "launches); read bytes = 2 x FETCH_SIZE (gfx950 correction for 16-B-per-lane streaming reads), written = WRITE_SIZE.\n\n")
        f.write("| kernel | calls | avg duration (us) | read (MB) | written (MB) |\n|---|---|---|---|---|\n")
        for r in rows:
            f.write(f"| `{short(r[0])}` | {r[1]} | {r[2]:.1f} | {r[3]:.0f} | {r[4]:.0f} |\n")
        f.write(f"\nStep total (kernels launched once per step): {rd:.0f} MB read + {wr:.0f} MB written = {rd + wr:.0f} MB, "
                f"{us:.0f} us of kernel time under the profiler.\nAlgorithmic bytes of the step: {alg / 1e6:.0f} MB -> "
                f"**{(rd + wr) * 1e6 / alg:.2f}x** (round 2: 2.73x, round 1: 3.2x).\n")
        f.write(f"`bench.py --workload kkt` on the same box: **{kkt['ms_per_step']:.3f} ms per step, {kkt['value'] / 1e6:.2f} M solves/s, "
                f"frac {kkt['roofline']['frac']:.3f}** (`profiles/{tag}_bench_kkt.json`).\n"
                "History of the three big launches (condensation / sweep / recovery): round 2 474 / 328 / 193 us; round 3 with the\n"
                "generic instantiations 472 / 327 / 195 us; with the compile-time benchmark-family instantiations the figures above.\n")
    grid = os.path.join(out, "kkt_grid.md")
    if os.path.exists(grid):
        with open(os.path.join(prof_out, f"{tag}_kkt", "grid.md"), "w") as f:
            f.write("# Newton-KKT factor + solve over the reference's benchmark grid -- `tools/kkt_grid_times.py`\n\n"
                    "NewtonKKTProblem(n, m, T = 50) with c = n / 2 and g = 2 m rows per edge and on the last node "
                    "(newton_kkt_benchmark.cpp:59-80, 264-273), batch 4096, fp64, HIP events over 20 steps after 3.\n"
                    "Odd m: the stage items and the places of `A | B` in the model arena are odd numbers of scalars -- the pipelined "
                    "condensation and the in-place sweep take them from 8-byte aligned addresses (round 3; before, those shapes "
                    "ran the one-stage generic condensation and copied `A | B`: (12, 3) 1.09 ms, (12, 1) 0.92, (8, 3) 0.78, (8, 1) 0.70, "
                    "(4, 3) 0.63, (4, 1) 0.55 ms).\n\n")
            f.write(open(grid).read())
    traffic_path = os.path.join(prof_out, "traffic.json")
    t = json.load(open(traffic_path))
    t[f"kkt:{kkt['config']['kernels']}"] = {
        "correction": "2*FETCH_SIZE + WRITE_SIZE (KiB), MI355X_MICROARCH.md HBM section; sum over the launches of one step",
        "hbm_bytes_per_launch": (rd + wr) * 1e6, "profile": f"profiles/{tag}_kkt/step.md"}
    json.dump(t, open(traffic_path, "w"), indent=1, sort_keys=True)
    # ---- theta path ----
    th = last_line(os.path.join(out, "bench_theta8.json"))
    rows = kernel_table(os.path.join(prof, f"{tag}_theta8"))
    os.makedirs(os.path.join(prof_out, f"{tag}_theta8"), exist_ok=True)
    shutil.copy(glob.glob(os.path.join(prof, f"{tag}_theta8", "kt", "*", "*_kernel_stats.csv"))[0],
                os.path.join(prof_out, f"{tag}_theta8", "kernel_stats.csv"))
    with open(os.path.join(prof_out, f"{tag}_theta8", "factor_theta.md"), "w") as f:
        f.write(f"# theta Schur complement of a uniform chain, p = 8 (f2) -- `tools/profile_kkt.sh {tag}_theta8 --theta 8`, table by `tools/publish_profiles.py`\n\n")
        f.write(f"Workload: {th['config']}, batch {th['batch']}, fp64 (`tests/bench_kkt.py --theta 8`: `factor_theta`, `solve_theta`, the\n"
                "stagewise `factor` and `add_Kx_to_y_theta` in turn).  Kernel trace + separate FETCH_SIZE / WRITE_SIZE passes\n"
                "(read = 2 x FETCH_SIZE, KiB).  Fused passes of `kkt_theta_chain_kernels.hpp` (`J_theta` never assembled).\n\n")
        f.write("| kernel | calls | avg duration (us) | read (MB) | written (MB) |\n|---|---|---|---|---|\n")
        for r in rows:
            f.write(f"| `{short(r[0])}` | {r[1]} | {r[2]:.1f} | {r[3]:.0f} | {r[4]:.0f} |\n")
        f.write(f"\nBy HIP events in the same script (`profiles/{tag}_theta8_bench.json`): **`factor_theta` {th['ms_factor_theta']:.2f} ms**, "
                f"`solve_theta` {th['ms_solve_theta']:.2f} ms, stagewise `factor` {th['ms_factor_stagewise']:.2f} ms"
                + (f", `add_Kx_to_y_theta` {th['ms_add_Kx_to_y_theta']:.2f} ms" if "ms_add_Kx_to_y_theta" in th else "") + ".\n"
                "`factor_theta` = strip + stagewise factor (condense_chain_pipe_kernel<false>, chain_factor_solve_qw16 in factor mode)\n"
                "+ theta_rhs_chain_kernel + chain_solve_mrhs_qw16 + theta_recover_chain_kernel + theta_schur_reduce_kernel.\n"
                "Round 2 (generic passes): `theta_jacobian` 452 + `condense<rhs>` 414, multi-rhs sweep 687, `recover<cols>` 471 +\n"
                "`theta_schur_small` 359 us; `factor_theta` 3.3-3.4 ms; `add_Kx_to_y_theta` 2.89 ms.\n")
    print("published", tag)


if __name__ == "__main__":
    main()
