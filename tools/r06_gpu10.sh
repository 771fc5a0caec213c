#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06k; mkdir -p $O
for i in 1 2; do
  MSNET_HIP_LIB=$PWD/ms-nets_amd/libx_pre_softarg.so timeout 300 python tools/r06_softarg_ab.py >> $O/pre.txt 2>&1
  timeout 300 python tools/r06_softarg_ab.py >> $O/new.txt 2>&1
done
A="--no-cpu-baseline --no-extras --steps 20 --warmup 5"
for i in 1 2 3; do
  MSNET_HIP_LIB=$PWD/ms-nets_amd/libx_pre_softarg.so timeout 300 python bench.py $A > $O/pre_cfg2_$i.json 2>/dev/null
  timeout 300 python bench.py $A > $O/new_cfg2_$i.json 2>/dev/null
  MSNET_HIP_LIB=$PWD/ms-nets_amd/libx_pre_softarg.so timeout 300 python bench.py $A --workload cfg3 > $O/pre_cfg3_$i.json 2>/dev/null
  timeout 300 python bench.py $A --workload cfg3 > $O/new_cfg3_$i.json 2>/dev/null
done
grep -h "digest" $O/pre.txt | head -6; echo; grep -h "digest" $O/new.txt | head -6
