cd /tmp && export TMPDIR=/tmp
for n in 2 8; do
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fc_$n -o run -- python $GRAFT_REPO_ROOT/bench.py --steps $n --warmup 1 --no-cpu-baseline --no-fp32-side > /dev/null 2>&1
f=$(find /tmp/fc_$n -name '*kernel_stats.csv' | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/fc_$n.csv
done
