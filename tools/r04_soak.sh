#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_soak; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
timeout 300 python -m pytest tests/test_gpu_aggregators.py -q -k "psmnet_forward_ndhwc or concurrent" 2>&1 | tail -2
timeout 1500 python tools/tools_soak.py 2000 > $O/soak.txt 2>&1; echo "soak rc=$?" >> $O/soak.txt; tail -5 $O/soak.txt
