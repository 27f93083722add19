#!/bin/bash
# usage (GPU box): bash tools/tuning/shapes.sh NAME [bench args] -> gpurun_out/NAME_shapes.txt: the kernel trace of a short
# bench run folded by (kernel, grid, LDS bytes) - one line per launch SHAPE with calls per step, mean duration and ms per
# step.  The kernel-stats table folds every shape of an instantiation into one line; this one shows which layer it is.
name=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/shapes_$name -o run -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fp32-side "$@" > $R/gpurun_out/${name}_shapes.log 2>&1 || exit 1
f=$(find /tmp/shapes_$name -name '*kernel_trace.csv' | head -1)
python $R/tools/tuning/shapes.py "$f" > $R/gpurun_out/${name}_shapes.txt
head -70 $R/gpurun_out/${name}_shapes.txt
