#!/bin/bash
# round 6, fourth GPU call: same-box A/B of bench.py (round 5's script, with its idle gap in front of the timed region, against this round's),
# and the planar-16 pair probe
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06d; mkdir -p $O
A="--no-cpu-baseline --no-extras --steps 20 --warmup 3"
for i in 1 2 3; do
  timeout 300 python bench_r05_copy.py $A > $O/old_cfg2_$i.json 2>/dev/null
  timeout 300 python bench.py $A > $O/new_cfg2_$i.json 2>/dev/null
  timeout 300 python bench_r05_copy.py $A --workload cfg3 > $O/old_cfg3_$i.json 2>/dev/null
  timeout 300 python bench.py $A --workload cfg3 > $O/new_cfg3_$i.json 2>/dev/null
done
MSNET_HIP_LIB=$PWD/ms-nets_amd/libx_planar16.so timeout 600 python tools/r06_planar_probe.py > $O/planar_probe.txt 2>&1; echo "planar rc $?" >> $O/rc.txt
cat $O/planar_probe.txt | tail -8
