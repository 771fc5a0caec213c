#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.txt 2>&1
echo "suite rc=$?" >> $O/pytest_gpu.txt
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_gpu.txt | tail -20
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_cl.json 2> $O/bench_cl.err
MSNET_HIP_LIB=$PWD/ms-nets_amd/libx_r03.so timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --volume-layout ncdhw > $O/bench_r03lib.json 2> $O/bench_r03lib.err
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_cl2.json 2> $O/bench_cl2.err
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload cfg3 --verbose > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04c/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "%.2f maps/s %.3f ms" % (d["value"], d["ms_per_step"]), d["roofline"].get("family"), "%.3f" % d["roofline"]["frac"], d.get("power"))
    except Exception as e:
        print(f, "ERR", e)
PY
