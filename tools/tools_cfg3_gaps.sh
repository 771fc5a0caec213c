# Config #3 (PSMNet aggregator): what is not in any kernel.  bench with / without the range-guard read-back, with HIP graphs, and a
# kernel trace of each reduced to idle gaps (tools_trace_gaps.py).
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/cfg3gaps; mkdir -p $OUT
B="bench.py --workload cfg3 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-kernel-timing"
for r in 1 2; do
python $B | cut -c1-140 | sed 's/^/eager           /'
MSNET_RANGE_CHECK=0 python $B | cut -c1-140 | sed 's/^/eager, no guard /'
python $B --graph | cut -c1-140 | sed 's/^/graph           /'
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/eager -- python3 $GRAFT_REPO_ROOT/$B > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/graph -- python3 $GRAFT_REPO_ROOT/$B --graph > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for m in eager graph; do echo "== $m"; python tools/tools_trace_gaps.py $(find $OUT/$m -name "*kernel_trace.csv" | head -1); done
find $OUT -name "*.csv" -size +20M -delete
