#!/bin/bash
# usage (GPU box): bash tools/tuning/r04_lds_budget.sh TAG  -> gpurun_out/TAG_lds_budget.txt: wave-cycle split, LDS pipe
# activity, MFMA busy cycles and the clock held, for the MFMA-bound kernels of the step
set -e
tag=${1:-r04_l}
R=$GRAFT_REPO_ROOT
bash $R/tools/tuning/pmc.sh $tag hr0 lr hr0w rdbw
cd $R && python tools/tuning/pmc_sum.py $tag > gpurun_out/${tag}_lds_budget.txt
cat gpurun_out/${tag}_lds_budget.txt
