#!/bin/bash
# usage: tools/tuning/trace_kernel.sh NAME PATTERN [bench args] -> gpurun_out/NAME_trace.txt: every dispatch of the kernels
# whose name contains PATTERN (start offset us, duration us, grid, workgroup), from a rocprofv3 --kernel-trace of bench.py
name=$1; pat=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$name -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/${name}_trace.log 2>&1 || exit 1
f=$(find /tmp/tr_$name -name '*kernel_trace.csv' | head -1)
python - "$f" "$pat" > $GRAFT_REPO_ROOT/gpurun_out/${name}_trace.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]:
        print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:12.1f} us  {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} us  '
              f'grid {r["Grid_Size_X"]}x{r["Grid_Size_Y"]}  wg {r["Workgroup_Size_X"]}  {r["Kernel_Name"][:60]}')
PY
