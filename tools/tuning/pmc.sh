#!/bin/bash
# usage: tools/tuning/pmc.sh NAME "case ..."   three counter passes (SQ waits + LDS, MFMA / instruction mix, GRBM_GUI_ACTIVE for the clock the chip held) over tools/bench_conv.py cases
name=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc1_$name -o p -- python $GRAFT_REPO_ROOT/tools/bench_conv.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/${name}_1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --output-format csv -d /tmp/pmc2_$name -o p -- python $GRAFT_REPO_ROOT/tools/bench_conv.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/${name}_2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc3_$name -o p -- python $GRAFT_REPO_ROOT/tools/bench_conv.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/${name}_3.log 2>&1 || exit 1
cp $(find /tmp/pmc3_$name -name '*counter_collection.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/${name}_pmc3.csv
cp $(find /tmp/pmc1_$name -name '*counter_collection.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/${name}_pmc1.csv
cp $(find /tmp/pmc2_$name -name '*counter_collection.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/${name}_pmc2.csv
ls -la $GRAFT_REPO_ROOT/gpurun_out/${name}_pmc*.csv
