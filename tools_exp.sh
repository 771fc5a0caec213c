cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_aggregators.py -m gpu -q -x --timeout 600 2>&1 | tail -3
python tools_layer_bench.py c8 s1_32_32 s1_64_64 s1_64_64b 2>&1 | grep -v amdgpu.ids
python tools_stamps.py s1_32_32 2>&1 | grep -v amdgpu.ids
