#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06l; mkdir -p $O
for i in 1 2; do
  MSNET_HIP_LIB=$PWD/ms-nets_amd/libx_pre_softarg.so timeout 300 python tools/r06_softarg_ab.py >> $O/pre.txt 2>&1
  timeout 300 python tools/r06_softarg_ab.py >> $O/new.txt 2>&1
done
timeout 900 python -m pytest tests/test_gpu_aggregators.py tests/test_gpu_determinism.py -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" > $O/rc.txt
grep -h "digest" $O/pre.txt | tail -6; echo; grep -h "digest" $O/new.txt | tail -6; tail -2 $O/pytest.log; cat $O/rc.txt
