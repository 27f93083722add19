#!/bin/bash
# usage (on the GPU box): bash tools/tuning/final_batch.sh TAG   -> everything profiles/README.md lists for the round's
# final build, under gpurun_out/ (copy into profiles/ afterwards): kernel stats, counter passes, driver-style line, presets
set -e
tag=${1:-r03_h}
R=$GRAFT_REPO_ROOT
bash $R/tools/tuning/prof.sh $tag
bash $R/tools/tuning/pmc_step.sh $tag
cd $R
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_default.json.log 2> gpurun_out/${tag}_bench_default.err
python bench.py --gpus 1 --single-rank-group --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_single_rank_rccl.json.log 2> /dev/null
rm -f gpurun_out/${tag%_h}_presets.jsonl
for c in C1 C1b C2 C3p C4 C5b C5c C5lit; do
  python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline 2> /dev/null | grep '^{' >> gpurun_out/${tag%_h}_presets.jsonl
done
python bench.py --config C3p --dtype fp32 --steps 3 --warmup 1 --no-cpu-baseline 2> /dev/null | grep '^{' >> gpurun_out/${tag%_h}_presets.jsonl
bash $R/tools/tuning/prof.sh ${tag%_h}_c2 --config C2
bash $R/tools/tuning/prof.sh ${tag}_c1b --config C1b
echo final batch done
