#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
timeout 300 python tools/r04_power_probe.py > $O/power_probe.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.txt 2>&1
echo "suite rc=$?" >> $O/pytest_gpu.txt
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_gpu.txt | tail -20
