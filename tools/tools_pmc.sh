# PMC passes for one layer (arg1 = layer name of tools_layer_bench.py, arg2 = tag)
L=${1:-s1_32_32}; TAG=${2:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p1 -- python3 $GRAFT_REPO_ROOT/tools/tools_layer_bench.py $L > $OUT/p1.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d $OUT/p2 -- python3 $GRAFT_REPO_ROOT/tools/tools_layer_bench.py $L > $OUT/p2.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/p3 -- python3 $GRAFT_REPO_ROOT/tools/tools_layer_bench.py $L > $OUT/p3.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/p4 -- python3 $GRAFT_REPO_ROOT/tools/tools_layer_bench.py $L > $OUT/p4.log 2>&1
find $OUT -name "*.csv" | head; find $OUT -type f ! -name "*counter_collection*" ! -name "*.log" -delete
echo pmc-done
