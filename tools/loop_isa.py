#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a -save-temps listing.
    python tools/loop_isa.py <listing.s> [kernel-name-regex] [--min N]
Prints, for every backward branch of the kernel, the instruction count between its target and itself and the
commonest mnemonics, then the register / scratch footer of the kernel."""
import collections
import re
import sys


def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else r"chain_factor_solve_qw16ILi12ELi4ELb1ELb1E"
    least = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 150
    txt = open(path).read()
    m = re.search(r"^(\S*%s\S*):" % pat, txt, re.M)
    start = m.end()
    end = txt.index("s_endpgm", start)
    body = txt[start:end].split("\n")
    labels = {}
    for i, l in enumerate(body):
        mm = re.match(r"^(\.LBB\d+_\d+):", l)
        if mm:
            labels[mm.group(1)] = i
    print(m.group(1))
    for i, l in enumerate(body):
        mm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            lo = labels[mm.group(1)]
            ins = [x.split()[0] for x in body[lo:i + 1] if x.startswith("\t") and not x.strip().startswith((".", ";"))]
            if len(ins) < least:
                continue
            print("loop", mm.group(1), "lines", lo, i, "instructions", len(ins))
            print("  ", sorted(collections.Counter(ins).items(), key=lambda t: -t[1])[:36])
    foot = txt[end:end + 4000]
    for key in ("NumVgprs", "NumAgprs", "ScratchSize", "Occupancy", "LDSByteSize"):
        mm = re.search(r"; %s: (\d+)" % key, foot)
        print(key, mm.group(1) if mm else None)


if __name__ == "__main__":
    main()
