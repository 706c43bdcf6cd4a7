#!/usr/bin/env python3
"""Per-kernel means of every counter found in the rocprofv3 --pmc CSVs under a directory."""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per_dispatch = collections.defaultdict(float)
    names = {}
    for row in csv.DictReader(open(path)):
        key = (path, row["Dispatch_Id"], row["Counter_Name"])
        per_dispatch[key] += float(row["Counter_Value"])
        names[(path, row["Dispatch_Id"])] = row["Kernel_Name"]
    for (pth, disp, ctr), val in per_dispatch.items():
        acc[names[(pth, disp)]][ctr].append(val)
want = sys.argv[2:] or ["chain_kernel", "pipe_kernel", "weights_kernel", "qw16", "mf32", "mt16", "staged_kernel", "theta"]
for kern, ctrs in acc.items():
    if not any(w in kern for w in want):
        continue
    print(kern[:100])
    for ctr in sorted(ctrs):
        v = ctrs[ctr]
        print(f"    {ctr:28s} {sum(v) / len(v):16.4g}  (n={len(v)})")
