#!/bin/bash
# usage (on the GPU box): bash tools/tuning/final_batch.sh TAG [1|2]   -> everything profiles/README.md lists for the
# round's final build, under gpurun_out/ (copy into profiles/ afterwards with collect_profiles.sh).  Part 1: kernel stats,
# counter passes, driver-style line, single-rank RCCL line; part 2: presets + the C2 / C1b kernel tables (two gpurun calls:
# one call is limited to 20 minutes)
set -e
tag=${1:-r05_h}
part=${2:-1}
R=$GRAFT_REPO_ROOT
cd $R
if [ "$part" = "1" ]; then
  bash $R/tools/tuning/prof.sh $tag
  bash $R/tools/tuning/pmc_step.sh $tag
  cd $R
  python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_default.json.log 2> gpurun_out/${tag}_bench_default.err
  python bench.py --gpus 1 --single-rank-group --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-side > gpurun_out/${tag}_bench_single_rank_rccl.json.log 2> /dev/null
else
  rm -f gpurun_out/${tag%_h}_presets.jsonl
  for c in C1 C1b C1c C2 C3p C3lit C4 C5b C5c C5lit C6; do
    echo "[final_batch] preset $c" >&2
    python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-side 2> /dev/null | grep '^{' >> gpurun_out/${tag%_h}_presets.jsonl
  done
  python bench.py --config C3p --dtype fp32 --steps 3 --warmup 1 --no-cpu-baseline 2> /dev/null | grep '^{' >> gpurun_out/${tag%_h}_presets.jsonl
  bash $R/tools/tuning/prof.sh ${tag%_h}_c2 --config C2
  bash $R/tools/tuning/prof.sh ${tag}_c1b --config C1b
  bash $R/tools/tuning/shapes.sh ${tag%_h}_c1c_launch --config C1c > /dev/null
  bash $R/tools/tuning/shapes.sh ${tag%_h}_p_launch > /dev/null
fi
echo final batch part $part done
