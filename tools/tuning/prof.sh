#!/bin/bash
# usage: tools/tuning/prof.sh NAME [bench args]  -> gpurun_out/NAME_kernel_stats.csv + NAME.log
name=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-side "$@" > $GRAFT_REPO_ROOT/gpurun_out/$name.log 2>&1
rc=$?
f=$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${name}_kernel_stats.csv
grep '^{' $GRAFT_REPO_ROOT/gpurun_out/$name.log | cut -c1-200
exit $rc
