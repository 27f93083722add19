#!/usr/bin/env python
"""Kernel timeline of the LAST pass in a rocprofv3 --kernel-trace CSV (start, duration, gap to the previous kernel,
workgroups, name): python tools/tuning/timeline.py trace.csv [marker-kernel-substring]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "planar_to_ndhwc"
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
seq = rows[idx[-1]:] if idx else rows
t0 = int(seq[0]["Start_Timestamp"])
tot, prev = 0, None
for r in seq:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    prev = e
    tot += e - s
    wgs = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} gap {gap:6.1f} wg {wgs:6d} {n}")
print("sum", tot / 1e3, "span", (int(seq[-1]["End_Timestamp"]) - t0) / 1e3)
