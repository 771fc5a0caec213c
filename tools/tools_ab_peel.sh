cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_aggregators.py -m gpu -q -k "conv3d or ragged or golden_end_to_end or sliding" 2>&1 | tail -2
bash tools/tools_ab_layers.sh s2_32_64 s1_64_64 s2_64_64 2>&1 | grep -E "round|ms "
bash tools/tools_ab.sh r03m libx_old.so 2>&1 | grep -E "s2_f16s|co64|diff|kernels|=="
