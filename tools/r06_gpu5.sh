#!/bin/bash
# round 6, fifth GPU call: planar-16 pair probe (fixed), bench.py with everything set up in front of the first step
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06e; mkdir -p $O
MSNET_HIP_LIB=$PWD/ms-nets_amd/libx_planar16.so timeout 600 python tools/r06_planar_probe.py > $O/planar_probe.txt 2>&1; echo "planar rc $?" >> $O/rc.txt
A="--no-cpu-baseline --no-extras --steps 20 --warmup 5"
for i in 1 2 3 4; do
  timeout 300 python bench_r05_copy.py $A > $O/old_cfg2_$i.json 2>/dev/null
  timeout 300 python bench.py $A > $O/new_cfg2_$i.json 2>/dev/null
  timeout 300 python bench_r05_copy.py $A --workload cfg3 > $O/old_cfg3_$i.json 2>/dev/null
  timeout 300 python bench.py $A --workload cfg3 > $O/new_cfg3_$i.json 2>/dev/null
done
tail -8 $O/planar_probe.txt
