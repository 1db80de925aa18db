#!/usr/bin/env python3
"""Condense a rocprofv3 `*_kernel_stats.csv` into a short table (kernel names truncated)."""
import csv
import sys


def short(name, n=70):
    for pre in ("void ", "sp::"):
        if name.startswith(pre):
            name = name[len(pre):]
    if name.startswith("at::native::"):
        name = "torch:" + name[12:]
    return name if len(name) <= n else name[: n - 3] + "..."


def main(path, top=25):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"{'kernel':70s} {'calls':>7s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'total_ms':>9s} {'%':>6s}")
    for r in rows[:top]:
        print(f"{short(r['Name']):70s} {int(r['Calls']):7d} {float(r['AverageNs']) / 1e3:10.2f} "
              f"{float(r['MinNs']) / 1e3:9.2f} {float(r['MaxNs']) / 1e3:9.2f} "
              f"{float(r['TotalDurationNs']) / 1e6:9.2f} {100 * float(r['TotalDurationNs']) / total:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25)
