# HBM traffic of the dominant launch (conv3dbn_2 on the split-fp16 kernel): separate PMC passes, kernel-trace only.
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/tools/tools_layer_bench.py s1_32_32 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/tools/tools_layer_bench.py s1_32_32 > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -- python3 $GRAFT_REPO_ROOT/tools/tools_layer_bench.py s1_32_32 > $OUT/sq.log 2>&1
find $OUT -type f ! -name "*counter_collection*" ! -name "*.log" -delete
echo done
