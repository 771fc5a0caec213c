#!/bin/bash
# SQ counters of every kernel of one bench step (rocprofv3 --pmc, kernel-trace only).  Usage: gpurun -- bash tools/tools_pmc_bench.sh TAG
TAG=${1:-pmcb}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/p1.log 2>&1
find $OUT -type f ! -name "*counter_collection*" ! -name "*.log" -delete
echo pmc-done
