#!/bin/bash
# pytest -m gpu on the box, terse log.  Usage: gpurun -- bash tools/tools_gpu_pytest.sh TAG [pytest-args...]
TAG=${1:-t}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
python -m pytest ${@:-tests} -m gpu -q --timeout 1800 -s 2>&1 | grep -E "max\||rel err|full size|cfg|likelihood|K gate|passed|failed|FAILED|Error|error|assert|^E " | tail -200 > $OUT/${TAG}_pytest.log
tail -40 $OUT/${TAG}_pytest.log
