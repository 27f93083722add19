#!/usr/bin/env python
"""Print the per-kernel summary of a rocprofv3 --kernel-trace --stats run: summarize.py DIR|kernel_stats.csv [STEPS]."""
import csv
import glob
import sys


def main(d, steps):
    f = d if d.endswith(".csv") else glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# {f}: total kernel time {tot / 1e6:.1f} ms over {steps} steps = {tot / 1e6 / steps:.1f} ms/step")
    print(f"{'kernel':100s} {'calls':>7s} {'ms/step':>9s} {'avg_us':>10s} {'%':>6s}")
    for r in rows[:30]:
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        print(f"{name[:100]:100s} {r['Calls']:>7s} {float(r['TotalDurationNs']) / 1e6 / steps:9.2f} "
              f"{float(r['AverageNs']) / 1e3:10.1f} {float(r['Percentage']):6.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
