#!/bin/bash
# Final GPU-box round of a build: rocprofv3 kernel stats + PMC passes first, then parity tests, default + verbose bench, other configs.
# Usage: gpurun -- bash tools/tools_gpu_final.sh TAG      (then tools/tools_collect_final.sh TAG on the build host)
TAG=${1:-r06f}
RND=${TAG:0:3}            # r06f -> r06: the two sha-gated files bench.py reads are named per round
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err
find $OUT/${TAG}_prof -type f ! -name "*kernel_stats*" -delete
cd $GRAFT_REPO_ROOT
bash tools/tools_pmc_traffic.sh ${TAG}_pmc > $OUT/${TAG}_pmc.log 2>&1
# what bench.py quotes from these passes (sha-gated): written into this box's profiles/ so the bench lines below carry them; the
# build host derives the same two files from the merged gpurun_out/ (tools_collect_final.sh) and commits them
cp $OUT/${TAG}_pmc/traffic.json profiles/${RND}_pmc_traffic.json
python tools/tools_profile_facts.py $(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_pmc > profiles/${RND}_profile_facts.json
python -m pytest tests -m gpu -q --timeout 1800 -s 2>&1 | grep -E "max\||rel err|full size|cfg|likelihood|K gate|EXACT|passed|failed|FAILED|Error|error" | tail -400 > $OUT/${TAG}_pytest.log
python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python bench.py --steps 10 --warmup 3 --verbose --no-cpu-baseline --no-extras > $OUT/${TAG}_benchv.json 2> $OUT/${TAG}_benchv.err
python bench.py --steps 10 --warmup 3 --verbose --workload cfg3 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_cfg3v.json 2> $OUT/${TAG}_bench_cfg3.err
python bench.py --steps 20 --warmup 5 --workload cfg3 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_cfg3.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --workload cfg5 --batch-per-gpu 2 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_cfg5.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --batch-per-gpu 4 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_cfg4.json 2>/dev/null
python bench.py --gpus 1 --self-launch --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_selflaunch.json 2>/dev/null
tail -3 $OUT/${TAG}_pytest.log; cut -c1-200 $OUT/${TAG}_bench.json
echo round-done
