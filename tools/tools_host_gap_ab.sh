# Host work exposed between two forwards (the range guard's read-back wait): A/B of the early read-back, cfg2 (headline) and cfg3,
# and the idle gaps of a kernel trace (tools_trace_gaps.py).
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/hostgap; mkdir -p $OUT
python -m pytest tests/test_gpu_aggregators.py -m gpu -q -x -k "range or guard or nonfinite or graph" 2>&1 | tail -2
for W in cfg2 cfg3; do
B="bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-kernel-timing"
for r in 1 2 3; do
MSNET_EARLY_READBACK=0 python $B | cut -c1-150 | sed "s/^/$W late  /"
python $B | cut -c1-150 | sed "s/^/$W early /"
done
MSNET_RANGE_CHECK=0 python $B | cut -c1-150 | sed "s/^/$W noguard /"
done
cd /tmp && export TMPDIR=/tmp
for W in cfg2 cfg3; do
rocprofv3 --kernel-trace --output-format csv -d $OUT/$W -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-timing > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
echo "== cfg2"; python tools/tools_trace_gaps.py $(find $OUT/cfg2 -name "*kernel_trace.csv" | head -1) vprep_kernel
echo "== cfg3"; python tools/tools_trace_gaps.py $(find $OUT/cfg3 -name "*kernel_trace.csv" | head -1)
find $OUT -name "*.csv" -delete
