#!/bin/bash
# usage: tools/tuning/pmc_step.sh NAME   -> gpurun_out/NAME_{fetch,write,mfma}.csv : three counter passes over bench.py
# (separate --pmc passes: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2 - MI355X_MICROARCH.md, rocprofv3 PMC slots)
name=$1; shift
cd /tmp && export TMPDIR=/tmp
run() {  # $1 tag, rest counters
  tag=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_${name}_$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-side > $GRAFT_REPO_ROOT/gpurun_out/${name}_$tag.log 2>&1 || return 1
  cp $(find /tmp/pmc_${name}_$tag -name '*counter_collection.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/${name}_$tag.csv
}
run fetch FETCH_SIZE && run write WRITE_SIZE && run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
ls -la $GRAFT_REPO_ROOT/gpurun_out/${name}_*.csv
