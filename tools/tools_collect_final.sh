#!/bin/bash
# Copies the summaries of tools_gpu_final.sh TAG from gpurun_out/ into profiles/ and derives the JSON files bench.py reads.
TAG=${1:-r06f}
RND=${TAG:0:3}
cd "$(dirname "$0")/.."
for f in bench bench_cfg3 bench_cfg5 bench_cfg4 bench_selflaunch benchv; do cp gpurun_out/${TAG}_$f.json profiles/${TAG}_$f.json; done
grep -E "ms/step|kernels" gpurun_out/${TAG}_benchv.err > profiles/${TAG}_bench_breakdown.txt
grep -E "ms/step|kernels" gpurun_out/${TAG}_bench_cfg3.err > profiles/${TAG}_bench_cfg3_breakdown.txt
cp gpurun_out/${TAG}_pytest.log profiles/${TAG}_parity_errors.txt
STATS=$(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1)
cp $STATS profiles/${TAG}_kernel_stats.csv
cp gpurun_out/${TAG}_pmc/traffic.json profiles/${RND}_pmc_traffic.json
cp gpurun_out/${TAG}_pmc/traffic.txt profiles/${TAG}_pmc_traffic.txt
python tools/tools_pmc_summary.py gpurun_out/${TAG}_pmc > profiles/${TAG}_pmc_sq_raw.txt 2>&1
python tools/tools_profile_facts.py profiles/${TAG}_kernel_stats.csv gpurun_out/${TAG}_pmc > profiles/${RND}_profile_facts.json
ls -la profiles/${TAG}_* profiles/r03_*
