#!/bin/bash
# same-device A/B of the trunk's filter gradients on a second stream (WSR_WGRAD_STREAM = ring size, 0 = in line)
# usage (on the GPU box): bash tools/tuning/r04_ab_stream.sh TAG "0 3 2 0 3" [bench args]
tag=$1; variants=$2; shift 2
out=gpurun_out/${tag}_ab_stream.jsonl
: > $out
for v in $variants; do
  echo "[ab] WSR_WGRAD_STREAM=$v" >&2
  WSR_WGRAD_STREAM=$v python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-fp32-side "$@" 2>gpurun_out/${tag}_ab_stream_$v.err | \
    python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'WSR_WGRAD_STREAM': '$v', 'ms_per_step': d['ms_per_step'], 'hr0_ms': d['roofline']['avg_launch_ms'], 'host_issue_ms': d['config']['host_issue_ms_per_step'], 'peak_hbm_gb': d['config']['peak_hbm_gb']}))" | tee -a $out
done
