#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_s2c8; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_aggregators.py -q -s -k "c8_chunk" 2>&1 | grep -E "rel err|passed|failed|Error|error" | tail -12
for i in 1 2 3; do
  MSNET_S2C8=0 timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ws   %.2f maps/s'%d['value'])"
  MSNET_S2C8=1 timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('s2c8 %.2f maps/s'%d['value'])"
done
MSNET_S2C8=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verbose 2>&1 >/dev/null | grep -E "s2_f16s|kernels"
MSNET_S2C8=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verbose 2>&1 >/dev/null | grep -E "s2_f16s|kernels"
