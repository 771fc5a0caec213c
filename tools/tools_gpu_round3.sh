#!/bin/bash
# One GPU-box round (round 3).  Usage: gpurun -- bash tools/tools_gpu_round3.sh TAG [pytest-args...]
TAG=${1:-r03}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
python -m pytest ${@:-tests} -m gpu -q --timeout 1800 -s 2>&1 | grep -E "max\||rel err|full size|cfg|likelihood|tap |K gate|passed|failed|FAILED|Error|error|assert" | tail -150 > $OUT/${TAG}_pytest.log
python bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python bench.py --steps 10 --warmup 3 --verbose --no-cpu-baseline --no-extras > $OUT/${TAG}_benchv.json 2> $OUT/${TAG}_benchv.err
tail -5 $OUT/${TAG}_pytest.log; cut -c1-300 $OUT/${TAG}_bench.json; grep -E "ms/step|kernels" $OUT/${TAG}_benchv.err
echo round-done
