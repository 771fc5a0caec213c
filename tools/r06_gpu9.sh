#!/bin/bash
# rocprofv3 kernel stats of config #3 on the final tree, then the whole GPU suite once more (the tree's last state)
cd "$GRAFT_REPO_ROOT"
O=$GRAFT_REPO_ROOT/gpurun_out/r06j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/prof_cfg3_bench.json 2> $O/prof_cfg3.err
find $O/prof_cfg3 -type f ! -name "*kernel_stats*" -delete
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" > $O/rc.txt
tail -2 $O/pytest_gpu.log; cat $O/rc.txt
