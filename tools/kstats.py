#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 --kernel-trace --stats output directory."""
import csv
import glob
import sys

for path in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(path)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s}  avg {float(r['AverageNs']) / 1e3:9.1f} us  {r['Percentage']}%")
