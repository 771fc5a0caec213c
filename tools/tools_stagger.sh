cd $GRAFT_REPO_ROOT
for st in 0 2 4 8 16; do echo "stagger $st"; MSNET_EXP_STAGGER=$st MSNET_HIP_LIB=$PWD/ms-nets_amd/libx_stagger.so python tools/tools_layer_bench.py s1_64_64 s2_32_64 s2_64_64 s1_64_64b 2>&1 | grep " ms "; done
