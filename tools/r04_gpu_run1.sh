#!/bin/bash
# Round 4, first GPU pass: the new tests first, then the whole suite, then bench lines (default = channels-last hand-over,
# A/B against the NCDHW hand-over on the same box), cfg3, batch 2.   gpurun --timeout 1500 -- bash tools/r04_gpu_run1.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
{
  echo "== sysfs power / clock files"
  for c in /sys/class/drm/card*/device; do echo $c; ls $c/hwmon/*/ 2>/dev/null | tr '\n' ' '; echo; cat $c/hwmon/*/power1_average $c/hwmon/*/power1_input $c/hwmon/*/freq1_input 2>&1 | head -5; head -3 $c/pp_dpm_sclk 2>&1; cat $c/numa_node 2>&1; done
  nproc; lscpu | grep -i "numa\|model name" | head
} > $O/sysfs.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_volume.py tests/test_gpu_aggregators.py -x -q -k "channels_last or forward_ndhwc or concurrent or data_parallel" > $O/pytest_new.txt 2>&1
echo "new tests rc=$?" >> $O/pytest_new.txt
timeout 1200 python -m pytest tests -m gpu -q -s -x > $O/pytest_gpu.txt 2>&1
echo "suite rc=$?" >> $O/pytest_gpu.txt
for i in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_cl_$i.json 2> $O/bench_cl_$i.err
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --volume-layout ncdhw > $O/bench_nc_$i.json 2> $O/bench_nc_$i.err
done
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --verbose > $O/bench_verbose.json 2> $O/bench_verbose.err
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --verbose --volume-layout ncdhw > $O/bench_verbose_nc.json 2> $O/bench_verbose_nc.err
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload cfg3 --verbose > $O/bench_cfg3.json 2> $O/bench_cfg3.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --batch-per-gpu 4 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --workload cfg5 --batch-per-gpu 2 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/pytest_new.txt $O/pytest_gpu.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04a/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "%.2f maps/s %.3f ms" % (d["value"], d["ms_per_step"]), d["roofline"]["family"], "%.3f" % d["roofline"]["frac"], d.get("power"), (d.get("roofline_step") or {}).get("mfmas_per_algorithmic_product"))
    except Exception as e:
        print(f, "ERR", e)
PY
