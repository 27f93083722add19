cd $GRAFT_REPO_ROOT
for c in C1b C1 C1c; do for r in a b a b; do
  if [ $r = a ]; then unset WSR_LIB_PATH; else export WSR_LIB_PATH=$GRAFT_REPO_ROOT/build_ab/lib_noepf.so; fi
  echo "== $c lib=$r"; python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-side 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done
