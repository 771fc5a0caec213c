#!/bin/bash
# round 6, last GPU call: smoke() and the whole GPU suite on the final tree
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06h; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" > $O/rc.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/rc.txt
tail -3 $O/smoke.log; tail -3 $O/pytest_gpu.log; cat $O/rc.txt
