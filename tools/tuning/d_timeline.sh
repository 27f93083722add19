#!/bin/bash
# usage (GPU box): bash tools/tuning/d_timeline.sh NAME -> gpurun_out/NAME_d_timeline.txt: kernel timeline of the LAST pass of
# tools/tuning/time_d.py (discriminator alone at the benchmark's size: forward + parameter gradients, train mode)
name=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/dtl_$name -o run -- python $R/tools/tuning/time_d.py > $R/gpurun_out/${name}_d_timeline.log 2>&1 || exit 1
f=$(find /tmp/dtl_$name -name '*kernel_trace.csv' | head -1)
python $R/tools/tuning/timeline.py "$f" > $R/gpurun_out/${name}_d_timeline.txt
tail -130 $R/gpurun_out/${name}_d_timeline.txt
