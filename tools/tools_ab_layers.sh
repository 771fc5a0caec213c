#!/bin/bash
# Interleaved A/B of layer timings: shipped library vs each ms-nets_amd/libx_*.so, three rounds.  Usage: tools_ab_layers.sh layer...
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  echo "-- round $round: shipped"; python tools/tools_layer_bench.py "$@" 2>&1 | grep " ms "
  for so in ms-nets_amd/libx_*.so; do echo "-- round $round: $so"; MSNET_HIP_LIB=$PWD/$so python tools/tools_layer_bench.py "$@" 2>&1 | grep " ms "; done
done
