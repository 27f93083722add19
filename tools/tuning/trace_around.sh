#!/bin/bash
# usage: tools/tuning/trace_around.sh NAME PATTERN [bench args] -> gpurun_out/NAME_around.txt: for every dispatch whose kernel
# name contains PATTERN in the LAST step of a rocprofv3 --kernel-trace of bench.py, the two kernels before and after it
name=$1; pat=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/ta_$name -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/${name}_around.log 2>&1 || exit 1
f=$(find /tmp/ta_$name -name '*kernel_trace.csv' | head -1)
python - "$f" "$pat" > $GRAFT_REPO_ROOT/gpurun_out/${name}_around.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda k: k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "")[:90]
hits = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
t0 = int(rows[0]["Start_Timestamp"])
for i in hits:
    print("----")
    for j in range(max(0, i - 2), min(len(rows), i + 3)):
        r = rows[j]
        print(f'{"=>" if j == i else "  "} {(int(r["Start_Timestamp"]) - t0) / 1e6:10.3f} ms {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:10.1f} us  {short(r["Kernel_Name"])}')
PY
cat $GRAFT_REPO_ROOT/gpurun_out/${name}_around.txt
