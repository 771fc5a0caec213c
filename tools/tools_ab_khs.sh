cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_aggregators.py -m gpu -q -k "conv3d or ragged or golden_end_to_end or sliding" 2>&1 | tail -2
bash tools/tools_ab_layers.sh s1_32_32 s1_32_32q 2>&1 | grep -E "round|ms "
bash tools/tools_ab.sh r03n libx_old.so 2>&1 | grep -E "co32|diff|kernels|=="
