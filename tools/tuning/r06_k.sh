set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "discriminator or batchnorm or trace or run_train or bn" > gpurun_out/r06_k_tests.log 2>&1 || { tail -40 gpurun_out/r06_k_tests.log; exit 1; }
tail -2 gpurun_out/r06_k_tests.log
for c in C3p C1b; do for r in 1 0 1 0; do
  echo "== $c WSR_FUSED_BN_STATS=$r"; WSR_FUSED_BN_STATS=$r python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-side 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
