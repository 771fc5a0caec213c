cd $GRAFT_REPO_ROOT
python tools/tools_layer_bench.py s2_32_64 s2_64_64 2>&1 | grep " ms "
for so in ms-nets_amd/libx_*.so; do echo "== $so"; MSNET_HIP_LIB=$PWD/$so python tools/tools_layer_bench.py s2_32_64 s2_64_64 2>&1 | grep " ms "; done
