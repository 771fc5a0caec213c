#!/bin/bash
# Times tools_layer_bench layers against every experiment build ms-nets_amd/libx_*.so.  Usage: tools_variants.sh layer...
cd $GRAFT_REPO_ROOT
python tools/tools_layer_bench.py "$@" 2>&1 | grep -E "ms "
for so in ms-nets_amd/libx_*.so; do
  echo "== $so"
  MSNET_HIP_LIB=$PWD/$so python tools/tools_layer_bench.py "$@" 2>&1 | grep -E "ms "
done
