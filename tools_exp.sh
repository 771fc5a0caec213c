cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_aggregators.py -m gpu -q -x --timeout 600 -s 2>&1 | grep -E "128|passed|failed|Error" | tail -8
python tools_layer_bench.py s1_128 s1_64_64 d_64_64 2>&1 | grep -v amdgpu.ids
