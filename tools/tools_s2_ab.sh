cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_aggregators.py -m gpu -q -k "stride2 or ragged or small_magn or golden_end_to_end" 2>&1 | tail -3
python tools/tools_layer_bench.py s2_32_64 s2_64_64 s2_32_64s 2>&1 | grep " ms "
MSNET_HIP_LIB=$PWD/ms-nets_amd/libx_s2padded.so python tools/tools_layer_bench.py s2_32_64 s2_64_64 s2_32_64s 2>&1 | grep " ms "
bash tools/tools_ab.sh r03l libx_s2padded.so 2>&1 | grep -E "s2_f16s|diff|kernels|=="
