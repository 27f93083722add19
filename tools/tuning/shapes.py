"""Fold a rocprofv3 kernel trace by launch shape: python shapes.py kernel_trace.csv [DELIMITER_KERNEL]  (see shapes.sh).
Steps are cut at the launches of a kernel that runs once per step (default: the 5x5x5 filter gradient); the first cut
interval (optimizer state initialisation, allocator warm-up) and everything outside the cuts (side probes) is dropped."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
delim = sys.argv[2] if len(sys.argv) > 2 else "wgrad_tile_kernel<3, 16, 1"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cuts = [int(r["Start_Timestamp"]) for r in rows if delim in r["Kernel_Name"]]
if len(cuts) < 3:
    sys.exit(f"need >= 3 launches of {delim!r} to cut steady-state steps, found {len(cuts)}")
lo, hi, steps = cuts[1], cuts[-1], len(cuts) - 2
acc = collections.defaultdict(list)
n_l = 0
for r in rows:
    t0 = int(r["Start_Timestamp"])
    if not (lo <= t0 < hi):
        continue
    n_l += 1
    n = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n)
    n = n.replace("void at::native::", "at::")[:64]
    grid = tuple(int(r[k]) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z") if k in r) or (int(r.get("Grid_Size", 0)),)
    wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)
    lds = int(r.get("LDS_Block_Size", 0) or 0)
    acc[(n, grid, wg, lds)].append(int(r["End_Timestamp"]) - t0)
tot = sum(sum(v) for v in acc.values())
print(f"# {n_l / steps:.0f} launches and {tot / 1e6 / steps:.2f} ms of kernels per step ({steps} steady steps, {(hi - lo) / 1e6 / steps:.2f} ms "
      f"apart; profiled clocks)")
print(f"{'kernel':64s} {'workgroups':>10s} {'LDS KB':>7s} {'calls/step':>10s} {'avg us':>9s} {'ms/step':>8s}")
for (n, grid, wg, lds), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    g = 1
    for x in grid:
        g *= max(x, 1)
    nwg = g // max(wg, 1) if wg else g
    print(f"{n:64s} {nwg:10d} {lds / 1024:7.1f} {len(v) / steps:10.2f} {sum(v) / len(v) / 1e3:9.1f} {sum(v) / 1e6 / steps:8.3f}")
