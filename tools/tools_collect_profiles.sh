#!/bin/bash
# Copies the summaries of one gpurun round (tools_gpu_round.sh TAG) from gpurun_out/ into profiles/.  Usage: tools_collect_profiles.sh TAG
TAG=$1
cd "$(dirname "$0")/.."
cp gpurun_out/${TAG}_bench.json profiles/${TAG}_bench.json
cp gpurun_out/${TAG}_bench_fp32.json profiles/${TAG}_bench_fp32.json
cp gpurun_out/${TAG}_bench_cfg3.json profiles/${TAG}_bench_cfg3.json
grep -E "ms/step|kernels" gpurun_out/${TAG}_bench.err > profiles/${TAG}_bench_breakdown.txt
grep -E "ms/step|kernels" gpurun_out/${TAG}_bench_cfg3.err > profiles/${TAG}_bench_cfg3_breakdown.txt
cp gpurun_out/${TAG}_pytest.log profiles/${TAG}_parity_errors.txt
cp $(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1) profiles/${TAG}_kernel_stats.csv
python tools/tools_trace_summary.py $(find gpurun_out/${TAG}_prof -name "*kernel_trace.csv" | head -1) > profiles/${TAG}_per_launch.txt
ls -la profiles/${TAG}_*
