#!/bin/bash
cd $GRAFT_REPO_ROOT
for a in "" "--workload cfg3" "--workload cfg5 --batch-per-gpu 2" "--batch-per-gpu 4 --steps 5"; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $a 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f maps/s'%d['value'], d['power'])"
done
timeout 600 python -m pytest tests/test_gpu_bench_contract.py -q 2>&1 | tail -3
