cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_aggregators.py -m gpu -q -k winograd 2>&1 | tail -2
for r in 1 2; do
MSNET_LB_WD=0 python tools/tools_layer_bench.py s1_32_32 s1_32_32q 2>&1 | grep " ms " | sed 's/^/direct   /'
python tools/tools_layer_bench.py s1_32_32 s1_32_32q 2>&1 | grep " ms " | sed 's/^/winograd /'
done
python bench.py --steps 10 --warmup 3 --verbose --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "co32|kernels"
python bench.py --workload cfg3 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-150
