set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_networks.py -m gpu -x -q -k "grouped_by_source or substacks" 2>&1 | tail -15
for r in 1 0 1 0; do
  echo "== C3p WSR_FWD_REGROUP=$r"; WSR_FWD_REGROUP=$r python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-side 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
