#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_final; mkdir -p $O
for i in 1 2; do timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_$i.txt 2>&1; tail -1 $O/pytest_$i.txt; done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print('%.2f maps/s %.3f ms'%(d['value'], d['ms_per_step']), d['roofline']['family'], d['roofline']['frac'], d['roofline'].get('frac_rocprof'), d['power'], d['cpu_baseline']['value'])"
