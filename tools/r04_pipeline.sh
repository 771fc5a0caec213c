#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sequential %.2f maps/s %.3f ms %.0f W'%(d['value'], d['ms_per_step'], d['power']['power_w']))"
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --pipeline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipelined  %.2f maps/s %.3f ms %.0f W'%(d['value'], d['ms_per_step'], d['power']['power_w']), d['metric'])"
done
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --pipeline --batch-per-gpu 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipelined B=2 %.2f maps/s'%d['value'])"
