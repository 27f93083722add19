cd $GRAFT_REPO_ROOT
for c in C1b C1; do for r in 1 0 1 0 1 0; do
  echo "== $c WSR_FUSED_BN_STATS=$r"; WSR_FUSED_BN_STATS=$r python bench.py --config $c --steps 60 --warmup 10 --no-cpu-baseline --no-fp32-side 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
