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




def steady(trace_path, patterns, min_us=100.0):
    """Per-kernel stats over the launches of a `*_kernel_trace.csv` whose duration exceeds `min_us`
    (drops graph-capture warm-ups at the padded fill length) for the first pattern; plain stats for
    the rest."""
    import collections
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(trace_path)):
        name = r.get("Kernel_Name") or r.get("Name") or ""
        for p in patterns:
            if p in name:
                dur[p].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    for i, p in enumerate(patterns):
        d = dur[p]
        if i == 0:
            d = [x for x in d if x > min_us]
        if d:
            print(f"#   {p:28s} n={len(d):5d}  avg {sum(d) / len(d):8.1f} us  min {min(d):8.1f}  max {max(d):8.1f}")


if __name__ == "__main__":
    # prof_summary.py <kernel_stats.csv> [top]            -> table
    # prof_summary.py --steady <kernel_trace.csv> <pattern> [pattern ...]
    if sys.argv[1] == "--steady":
        steady(sys.argv[2], sys.argv[3:])
    else:
        main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25)
