#!/bin/bash
# usage: tools/tuning/gaps.sh NAME [bench args] -> gpurun_out/NAME_gaps.txt: idle time between consecutive kernels of the
# LAST step of a rocprofv3 --kernel-trace of bench.py (largest gaps, and their sum against the step's span)
name=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/gp_$name -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/${name}_gaps.log 2>&1 || exit 1
f=$(find /tmp/gp_$name -name '*kernel_trace.csv' | head -1)
python - "$f" > $GRAFT_REPO_ROOT/gpurun_out/${name}_gaps.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps are delimited by the generator's first kernel of an iteration: planar_to_ndhwc of the LR input
idx = [i for i, r in enumerate(rows) if "planar_to_ndhwc" in r["Kernel_Name"]]
# last full G+D pair: find the marker positions; a step has several markers - take the last 1/4 of the trace instead
n = len(rows)
seq = rows[n - n // 4:]
t0, t1 = int(seq[0]["Start_Timestamp"]), int(seq[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seq)
gaps = []
prev = None
for r in seq:
    s = int(r["Start_Timestamp"])
    if prev is not None:
        gaps.append((s - int(prev["End_Timestamp"]), prev["Kernel_Name"], r["Kernel_Name"]))
    prev = r
short = lambda k: k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
print(f"last quarter of the trace: {len(seq)} kernels, span {(t1 - t0) / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms")
pos = [g for g in gaps if g[0] > 0]
print(f"gaps > 0: {len(pos)}, sum {sum(g[0] for g in pos) / 1e6:.2f} ms; > 5 us: {sum(1 for g in pos if g[0] > 5000)} sum {sum(g[0] for g in pos if g[0] > 5000) / 1e6:.2f} ms")
import collections
by = collections.Counter()
for g in pos:
    by[(short(g[1]), short(g[2]))] += g[0]
for (a, b), v in by.most_common(25):
    print(f"{v / 1e3:9.1f} us  {a}  ->  {b}")
PY
cat $GRAFT_REPO_ROOT/gpurun_out/${name}_gaps.txt
