#!/usr/bin/env python3
"""Average the counters of a rocprofv3 --pmc pass per kernel:
   python tools/pmc_summary.py <dir with *_counter_collection.csv> [kernel-name substring ...]"""
import collections
import csv
import glob
import os
import sys


def main(root, pats):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            name = r.get("Kernel_Name", "")
            if pats and not any(p in name for p in pats):
                continue
            short = name.split("(")[0].replace("void ", "").replace("sp::", "")[:60]
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        for c in sorted(acc[k]):
            v = acc[k][c]
            print(f"{k:60s} {c:28s} n={len(v):4d} mean={sum(v) / len(v):18.1f} min={min(v):16.1f} max={max(v):16.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
