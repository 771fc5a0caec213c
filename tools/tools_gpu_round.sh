#!/bin/bash
# One GPU-box round: parity tests, verbose bench, rocprofv3 kernel stats.  Usage: gpurun -- bash tools/tools_gpu_round.sh TAG
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
python -m pytest tests -m gpu -q --timeout 900 -s 2>&1 | grep -E "max\|disp|rel err|full size|passed|failed|FAILED|Error|error" | tail -60 > $OUT/${TAG}_pytest.log
python bench.py --steps 5 --warmup 2 --verbose > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python bench.py --steps 5 --warmup 2 --verbose --workload cfg3 --no-cpu-baseline > $OUT/${TAG}_bench_cfg3.json 2> $OUT/${TAG}_bench_cfg3.err
python bench.py --steps 5 --warmup 2 --precision fp32 --no-cpu-baseline > $OUT/${TAG}_bench_fp32.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err
find $OUT/${TAG}_prof -name "*kernel_stats*" | head -3
# keep only the small summaries
find $OUT/${TAG}_prof -type f ! -name "*stats*" ! -name "*kernel_trace*" -delete
echo round-done
