# HBM traffic of the volume-build kernels: separate --pmc passes for FETCH_SIZE and WRITE_SIZE (TCC slots), kernel trace only.
# Usage: gpurun -- bash tools/tools_pmc_volume.sh TAG [cfg2]
TAG=${1:-pmcvol}; CFG=${2:-cfg2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/tools/tools_volume_bench.py $CFG 5 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/tools/tools_volume_bench.py $CFG 5 > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -- python3 $GRAFT_REPO_ROOT/tools/tools_volume_bench.py $CFG 5 > $OUT/sq.log 2>&1
find $OUT -type f ! -name "*counter_collection*" ! -name "*.log" -delete
python3 $GRAFT_REPO_ROOT/tools/tools_pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | grep -v "^$" | head -60
echo pmc-volume-done
