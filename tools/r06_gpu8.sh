#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06i; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/build_smoke.log 2>&1; echo "build+smoke rc $?" > $O/rc.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/rc.txt
timeout 600 python __graft_entry__.py --smoke > $O/main_smoke.log 2>&1; echo "main --smoke rc $?" >> $O/rc.txt
timeout 600 python -m pytest tests/test_abi.py tests/test_gpu_bench_contract.py -q -m "gpu or not gpu" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/rc.txt
grep -h "smoke\]" $O/*.log; tail -2 $O/pytest.log; cat $O/rc.txt
