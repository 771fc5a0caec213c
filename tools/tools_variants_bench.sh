#!/bin/bash
# bench.py --verbose kernel lines matching $1 for the shipped library and every experiment build ms-nets_amd/libx_*.so
cd $GRAFT_REPO_ROOT
PAT=${1:-kernels}
python bench.py --steps 5 --warmup 2 --verbose --no-cpu-baseline 2>&1 | grep -E "$PAT"
for so in ms-nets_amd/libx_*.so; do
  echo "== $so"
  MSNET_HIP_LIB=$PWD/$so python bench.py --steps 5 --warmup 2 --verbose --no-cpu-baseline 2>&1 | grep -E "$PAT"
done
