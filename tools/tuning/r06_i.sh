set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "generator or trace or dense or stacked or bit or dgrad or conv_fwd" > gpurun_out/r06_i_tests.log 2>&1 || { tail -30 gpurun_out/r06_i_tests.log; exit 1; }
tail -2 gpurun_out/r06_i_tests.log
for r in a b a b; do
  if [ $r = a ]; then unset WSR_LIB_PATH; else export WSR_LIB_PATH=$GRAFT_REPO_ROOT/build_ab/lib_noepf.so; fi
  echo "== C3p lib=$r (a: early epilogue operands, b: without)"; python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-side 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
unset WSR_LIB_PATH
for r in a b; do
  if [ $r = a ]; then unset WSR_LIB_PATH; else export WSR_LIB_PATH=$GRAFT_REPO_ROOT/build_ab/lib_noepf.so; fi
  echo "== C4 lib=$r"; python bench.py --config C4 --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-side 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  echo "== C1b lib=$r"; python bench.py --config C1b --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-side 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
