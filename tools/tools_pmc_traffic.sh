#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, kernel trace only) of every kernel of the default bench step,
# then profiles-ready JSON for bench.py's roofline.traffic fields.  Usage: gpurun -- bash tools/tools_pmc_traffic.sh TAG
TAG=${1:-pmctraffic}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing > $OUT/sq2.log 2>&1
find $OUT -type f ! -name "*counter_collection*" ! -name "*.log" -delete
python3 $GRAFT_REPO_ROOT/tools/tools_pmc_traffic.py $OUT > $OUT/traffic.json 2> $OUT/traffic.txt
cat $OUT/traffic.txt | head -40
echo pmc-traffic-done
