#!/bin/bash
# A/B of an experiment build against the shipped library on one box: disparity difference + verbose bench of both.
# Usage: gpurun -- bash tools/tools_ab.sh TAG libx_name.so [more libs...]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python tools/tools_ab_disp.py /tmp/base.npy 2>&1 | tail -1
python bench.py --steps 10 --warmup 3 --verbose --no-cpu-baseline --no-extras > $OUT/${TAG}_base.json 2> $OUT/${TAG}_base.err
grep -E "ms/step|kernels" $OUT/${TAG}_base.err
for so in "$@"; do
  echo "== $so"
  MSNET_HIP_LIB=$PWD/ms-nets_amd/$so python tools/tools_ab_disp.py /tmp/x.npy /tmp/base.npy 2>&1 | tail -2
  MSNET_HIP_LIB=$PWD/ms-nets_amd/$so python bench.py --steps 10 --warmup 3 --verbose --no-cpu-baseline --no-extras > $OUT/${TAG}_${so}.json 2> $OUT/${TAG}_${so}.err
  grep -E "ms/step|kernels" $OUT/${TAG}_${so}.err
done
