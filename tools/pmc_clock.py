#!/usr/bin/env python3
"""Effective shader clock of a kernel from a GRBM_GUI_ACTIVE pass (MI355X_MICROARCH.md 'DVFS give-back':
clock ~ GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time) joined with the kernel trace of the same run:
   python tools/pmc_clock.py <dir with *_counter_collection.csv and *_kernel_trace.csv> <kernel-name substring>"""
import csv
import glob
import os
import sys


def main(root, pat):
    dur = {}
    for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if pat in r.get("Kernel_Name", ""):
                dur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
    rows = []
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if pat in r.get("Kernel_Name", "") and r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
                rows.append((float(r["Counter_Value"]), dur[r["Dispatch_Id"]]))
    if not rows:
        print("# no GRBM_GUI_ACTIVE rows joined with the kernel trace")
        return
    clk = [v / 8.0 / t / 1e9 for v, t in rows]
    t = [x[1] * 1e3 for x in rows]
    print(f"# effective shader clock over {len(rows)} dispatches of {pat}: mean {sum(clk) / len(clk):.3f} GHz "
          f"(min {min(clk):.3f}, max {max(clk):.3f}); dispatch time mean {sum(t) / len(t):.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
