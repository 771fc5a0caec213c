#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06g; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fullsize_golden.py -q -s > $O/pytest_fullsize.log 2>&1; echo "rc $?" > $O/rc.txt
grep -E "EXACT|passed|failed" $O/pytest_fullsize.log | cut -c1-250
