set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r06_h_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r06_h_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r06_h_gpu_tests.log
for r in 1 0 1 0; do
  echo "== C3p WSR_FUSED_RAGAN=$r"; WSR_FUSED_RAGAN=$r python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-side 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['host_issue_ms_per_step'])"
done
