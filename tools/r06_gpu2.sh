#!/bin/bash
# round 6, second GPU call: full-size parity (gate X on every case), slow-step diagnosis (python gc on / off), first-layer 3-WG variant A/B
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fullsize_golden.py -q -s > $O/pytest_fullsize.log 2>&1; echo "pytest_fullsize rc $?" >> $O/rc.txt
B="--no-cpu-baseline --no-extras --steps 30 --warmup 5 --verbose"
for i in 1 2; do
  timeout 300 python bench.py $B --keep-gc > $O/cfg2_gc_$i.json 2>$O/cfg2_gc_$i.err
  timeout 300 python bench.py $B > $O/cfg2_nogc_$i.json 2>$O/cfg2_nogc_$i.err
  timeout 300 python bench.py $B --workload cfg3 --keep-gc > $O/cfg3_gc_$i.json 2>$O/cfg3_gc_$i.err
  timeout 300 python bench.py $B --workload cfg3 > $O/cfg3_nogc_$i.json 2>$O/cfg3_nogc_$i.err
done
for i in 1 2 3; do
  timeout 200 python tools/tools_layer_bench.py c8 >> $O/c8_layer_shipped.txt 2>&1
  MSNET_HIP_LIB=$PWD/ms-nets_amd/libx_c8wg3.so timeout 200 python tools/tools_layer_bench.py c8 >> $O/c8_layer_wg3.txt 2>&1
done
B2="--no-cpu-baseline --no-extras --steps 20 --warmup 5"
for i in 1 2 3; do
  timeout 300 python bench.py $B2 > $O/step_shipped_$i.json 2>/dev/null
  MSNET_HIP_LIB=$PWD/ms-nets_amd/libx_c8wg3.so timeout 300 python bench.py $B2 > $O/step_wg3_$i.json 2>/dev/null
done
cat $O/rc.txt
