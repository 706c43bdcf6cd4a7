#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs written by tools/profile_gpu.sh into the committed
evidence under profiles/:  <tag>_kernel_stats.csv, <tag>_summary.md and an
entry in profiles/traffic.json (read back by bench.py for roofline.traffic).

HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE from
separate --pmc passes, both in KiB; on gfx950 FETCH_SIZE tallies the 128-byte
requests of wide (16 B/lane) coalesced reads at 64 B, so it is doubled; WRITE_SIZE
is exact for 16 B/lane streaming stores.  (The kernel's stores are 8-16 B/lane,
its loads all 16 B/lane LDS-DMA pieces.)

usage: tools/summarize_prof.py gpurun_out/prof/<tag> <tag> <workload> [kernel-substring]
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(path, key):
    agg = collections.defaultdict(list)
    files = glob.glob(os.path.join(path, "*", "*_counter_collection.csv"))
    if not files:
        return {}, None
    name = None
    for r in csv.DictReader(open(files[0])):
        if key in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            name = r["Kernel_Name"]
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}, name


def main():
    src, tag, workload = sys.argv[1], sys.argv[2], sys.argv[3]
    key = sys.argv[4] if len(sys.argv) > 4 else "chain_factor_solve"
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    stats = glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv"))[0]
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        for r in rows[:12]:
            r = dict(r)
            r["Name"] = r["Name"][:160]
            w.writerow(r)
    krow = next(r for r in rows if key in r["Name"])
    fetch, kname = counters(os.path.join(src, "fetch"), key)
    write, _ = counters(os.path.join(src, "write"), key)
    sq, _ = counters(os.path.join(src, "sq"), key)
    mfma, _ = counters(os.path.join(src, "mfma"), key)
    for k2, v2 in mfma.items():
        sq.setdefault(k2 if k2 != "GRBM_GUI_ACTIVE" else "GRBM_GUI_ACTIVE (mfma pass)", v2)
    fetch_kib = fetch.get("FETCH_SIZE", (0, 0))[0]
    write_kib = write.get("WRITE_SIZE", (0, 0))[0]
    hbm = (2.0 * fetch_kib + write_kib) * 1024.0
    bench = {}
    log = os.path.join(src, "kt.log")
    if os.path.exists(log):
        for line in open(log):
            if line.startswith("{") and "roofline" in line:
                bench = json.loads(line)
    kernel = bench.get("config", {}).get("kernel", "")
    alg = bench.get("roofline", {}).get("algorithmic_bytes_per_launch")
    tpath = os.path.join(out_dir, "traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    traffic[f"{workload}:{kernel}"] = {
        "hbm_bytes_per_launch": hbm, "fetch_size_kib_raw": fetch_kib, "write_size_kib": write_kib,
        "correction": "2*FETCH_SIZE + WRITE_SIZE (KiB), MI355X_MICROARCH.md HBM section",
        "profile": f"profiles/{tag}_summary.md"}
    json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
    with open(os.path.join(out_dir, f"{tag}_summary.md"), "w") as f:
        f.write(f"# rocprofv3 summary `{tag}` -- workload {workload}\n\n")
        f.write("Command (on the MI355X box): `tools/profile_gpu.sh %s` = `rocprofv3 --kernel-trace --stats` and "
                "separate `--pmc` passes around `python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline`.\n\n" % tag)
        f.write(f"Kernel: `{kname or krow['Name']}`\n\n")
        f.write("| quantity | value |\n|---|---|\n")
        f.write(f"| calls | {krow['Calls']} |\n| average duration (kernel trace) | {float(krow['AverageNs'])/1e3:.1f} us |\n")
        f.write(f"| min / max duration | {float(krow['MinNs'])/1e3:.1f} / {float(krow['MaxNs'])/1e3:.1f} us |\n")
        if bench:
            f.write(f"| bench.py kernel_ms (HIP events, same run) | {bench['roofline']['kernel_ms']*1e3:.1f} us |\n")
            f.write(f"| bench.py value | {bench['value']:.4g} {bench['unit']} |\n")
            f.write(f"| algorithmic bytes per launch | {alg} |\n")
        f.write(f"| FETCH_SIZE (raw, KiB, mean of {fetch.get('FETCH_SIZE',(0,0))[1]} launches) | {fetch_kib:.0f} |\n")
        f.write(f"| WRITE_SIZE (KiB) | {write_kib:.0f} |\n")
        f.write(f"| HBM traffic per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 | {hbm:.4g} B |\n")
        if alg:
            f.write(f"| traffic / algorithmic | {hbm/alg:.2f} |\n")
        for k in sorted(sq):
            f.write(f"| {k} (mean/launch) | {sq[k][0]:.4g} |\n")
        if sq.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[0] > 0 and "GRBM_GUI_ACTIVE" in sq:
            # matrix-pipe cycles (64 per v_mfma_f32_32x32x2_f32) over all SIMD-cycles of the launch:
            # GRBM_GUI_ACTIVE sums the 8 XCDs, 1024 SIMDs on the chip
            util = sq["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (sq["GRBM_GUI_ACTIVE"][0] / 8.0 * 1024.0)
            f.write(f"| MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs) | {util:.3f} |\n")
            small = "mt16" in (kname or krow["Name"])  # chain_mt16: v_mfma_f32_16x16x4_f32, 2048 flop, 32 cycles each
            per = 16 * 16 * 4 * 2 if small else 32 * 32 * 2 * 2
            flops = sq.get("SQ_INSTS_MFMA", (0, 0))[0] * per
            f.write(f"| MFMA flop/s (f32 {'16x16x4' if small else '32x32x2'}: {per} flop each) | "
                    f"{flops / (float(krow['AverageNs']) * 1e-9) / 1e12:.1f} TFLOP/s of 157.3 peak |\n")
            falg = bench.get("roofline", {}).get("algorithmic_flops_per_launch")
            if falg:
                f.write(f"| matrix-pipe flops per launch / algorithmic flops | {flops / falg:.2f} |\n")
        if "GRBM_GUI_ACTIVE" in sq:
            clk = sq["GRBM_GUI_ACTIVE"][0] / 8.0 / (float(krow["AverageNs"]) * 1e-9) / 1e9
            f.write(f"| effective clock = GRBM_GUI_ACTIVE / 8 / duration | {clk:.2f} GHz |\n")
    print(open(os.path.join(out_dir, f"{tag}_summary.md")).read())


if __name__ == "__main__":
    main()
