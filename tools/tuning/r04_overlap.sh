#!/bin/bash
# kernel trace of the step with the trunk's filter gradients on a second stream -> gpurun_out/TAG_overlap.txt
tag=$1
cd /tmp && export TMPDIR=/tmp
WSR_WGRAD_STREAM=3 rocprofv3 --kernel-trace --output-format csv -d /tmp/ov_$tag -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${tag}_overlap.log 2>&1
f=$(find /tmp/ov_$tag -name '*kernel_trace.csv' | head -1)
python $GRAFT_REPO_ROOT/tools/tuning/overlap.py "$f" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_overlap.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/${tag}_overlap.txt
