#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_s2w16; mkdir -p $O
export MSNET_HIP_LIB=$PWD/ms-nets_amd/libx_s2w16.so
MSNET_S2_W16=1 timeout 600 python -m pytest tests/test_gpu_aggregators.py -q -k "stride2 or s2 or fuzz or golden" 2>&1 | tail -4
for i in 1 2 3; do
  timeout 120 python tools/tools_layer_bench.py s2_32_64 s2_64_64 2>&1 | grep split-fp16 >> $O/base.txt
  MSNET_S2_W16=1 timeout 120 python tools/tools_layer_bench.py s2_32_64 s2_64_64 2>&1 | grep split-fp16 >> $O/w16.txt
done
echo base; cat $O/base.txt; echo w16; cat $O/w16.txt
python tools/tools_ab_disp.py /tmp/base.npy 2>&1 | tail -1
MSNET_S2_W16=1 python tools/tools_ab_disp.py /tmp/x.npy /tmp/base.npy 2>&1 | tail -1
for i in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base %.2f maps/s'%d['value'])"
  MSNET_S2_W16=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w16  %.2f maps/s'%d['value'])"
done
