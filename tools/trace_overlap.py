#!/usr/bin/env python3
"""Did two kinds of kernels actually run at the same time?  Reads a rocprofv3 --kernel-trace CSV (columns Kernel_Name,
Start_Timestamp, End_Timestamp) and reports, for kernels matching pattern A (default: the decode range kernel) and pattern B
(default: the library's GEMM kernels, Cijk_*), the time each kind was in flight and the time BOTH were.

  python tools/trace_overlap.py <kernel_trace.csv> [--a decode_mfma_range] [--b Cijk_]
Used with tools/probe_overlap.py (round 6): the two-stream graph's branches are only an experiment if they overlap."""
import argparse
import csv


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def total(iv):
    return sum(e - s for s, e in iv)


def intersect(a, b):
    i = j = 0
    out = []
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            out.append([s, e])
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--a", default="decode_mfma_range")
    ap.add_argument("--b", default="Cijk_")
    a = ap.parse_args()
    A, B = [], []
    with open(a.csv) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            s, e = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
            if a.a in name:
                A.append((s, e))
            elif a.b in name:
                B.append((s, e))
    ua, ub = union(A), union(B)
    both = total(intersect(ua, ub))
    print(f"{a.a}: {len(A)} launches, in flight {total(ua) / 1e6:.2f} ms; {a.b}: {len(B)} launches, in flight {total(ub) / 1e6:.2f} ms; "
          f"both in flight {both / 1e6:.2f} ms = {100 * both / max(total(ua), 1):.1f} % of A's time, "
          f"{100 * both / max(total(ub), 1):.1f} % of B's")


if __name__ == "__main__":
    main()
