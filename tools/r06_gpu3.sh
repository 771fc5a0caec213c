#!/bin/bash
# round 6, third GPU call: bench.py without the idle gap in front of its timed region (series in order), then the whole GPU suite
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06c; mkdir -p $O
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 > $O/cfg2_$i.json 2>$O/cfg2_$i.err
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 --workload cfg3 > $O/cfg3_$i.json 2>$O/cfg3_$i.err
done
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 --verbose > $O/cfg2_v.json 2>$O/cfg2_v.err
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 --verbose --workload cfg3 > $O/cfg3_v.json 2>$O/cfg3_v.err
timeout 2400 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc $?" >> $O/rc.txt
tail -3 $O/pytest_gpu.log
cat $O/rc.txt
