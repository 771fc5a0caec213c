#!/bin/bash
# VERDICT r03 #6, measured: the 64->64 stride-1 layers as four Winograd-depth launches each (bench.py --wd64, MSNET_WD64=1) vs the direct kernel.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_wd64; mkdir -p $O
for i in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_direct_$i.json 2>/dev/null
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --wd64 > $O/bench_wd64_$i.json 2>/dev/null
done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --wd64 --verbose > $O/bench_wd64_v.json 2> $O/bench_wd64_v.err
grep -E "ms/step|kernels" $O/bench_wd64_v.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_wd64/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "%.2f maps/s %.3f ms"%(d["value"], d["ms_per_step"]), d["power"]["power_w"], d["power"]["sclk_mhz"])
    except Exception as e: print(f, "ERR", e)
PY
