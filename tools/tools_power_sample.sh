#!/bin/bash
# Samples rocm-smi power / clocks while a conv layer loops: is the chip at its power cap under the conv kernels?
# Usage: gpurun -- bash tools/tools_power_sample.sh [layer]
cd $GRAFT_REPO_ROOT
L=${1:-s1_32_32}
rocm-smi --showmaxpower 2>&1 | grep -v "^=\|^$" | head -4
for mode in random relu zeros; do
  python tools/tools_power_loop.py $L $mode 6 &
  PID=$!
  sleep 3.5
  for k in 1 2 3; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' '; echo; sleep 0.5; done
  wait $PID
done
