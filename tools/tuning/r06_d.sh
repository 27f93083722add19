set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "trace or iteration or guards or labels or noise or dropout or discriminator or run_train or bit or reproduc" > gpurun_out/r06_d_tests.log 2>&1 || { tail -30 gpurun_out/r06_d_tests.log; exit 1; }
tail -3 gpurun_out/r06_d_tests.log
python tools/tuning/torchops.py 2>/dev/null | head -3
for c in C3p C4; do for d in 0 1; do
  echo "== $c WSR_CT_DIET=$d"; WSR_CT_DIET=$d python bench.py --config $c --steps 8 --warmup 3 --no-cpu-baseline --no-fp32-side 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['host_issue_ms_per_step'])"
done; done
HP_NZ=10 python tools/tuning/hostprof2.py > gpurun_out/r06_d_hostprof_c1b.txt 2>&1 || true
head -60 gpurun_out/r06_d_hostprof_c1b.txt
