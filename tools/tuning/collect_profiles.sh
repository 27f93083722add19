#!/bin/bash
# usage (in the build container, after tools/tuning/final_batch.sh TAG ran on the GPU box): copies the judged artefacts
# from gpurun_out/ into profiles/ and prints the per-kernel tables
set -e
tag=${1:-r05_h}
r=${tag%_h}
cd /root/repo
for n in $tag ${r}_c2 ${tag}_c1b; do
  cp gpurun_out/${n}_kernel_stats.csv profiles/${n}_kernel_stats.csv
  python profiles/summarize.py profiles/${n}_kernel_stats.csv 4 > profiles/${n}_kernel_stats.txt
done
grep '^{' gpurun_out/$tag.log | head -1 > profiles/${tag}_bench.json.log
grep '^{' gpurun_out/${tag}_bench_default.json.log > profiles/${tag}_bench_default.json.log
grep '^{' gpurun_out/${tag}_bench_single_rank_rccl.json.log > profiles/${tag}_bench_single_rank_rccl.json.log
cp gpurun_out/${r}_presets.jsonl profiles/${r}_presets.jsonl
python tools/tuning/pmc_step_sum.py $tag gpurun_out/${tag}_kernel_stats.csv > profiles/${tag}_hbm_kernels.txt
for f in gpurun_out/parity_*.json; do cp $f profiles/${r}_$(basename $f); done
for n in c1c_launch p_launch; do [ -f gpurun_out/${r}_${n}_shapes.txt ] && cp gpurun_out/${r}_${n}_shapes.txt profiles/${r}_${n}_shapes.txt; done
ls -la profiles/${r}_* | wc -l
