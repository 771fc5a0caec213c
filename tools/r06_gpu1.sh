#!/bin/bash
# round 6, first GPU call: new parity tests, config #3 A/B (range check folded / graph), CU-mask premise check
mkdir -p gpurun_out/r06a
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06a
timeout 900 python -m pytest tests/test_gpu_golden_volume.py "tests/test_gpu_aggregators.py::test_check_input_range_any_length_and_alignment" "tests/test_gpu_aggregators.py::test_conv_on_module_input_carries_the_range_check" "tests/test_gpu_aggregators.py::test_psmnet_forward_ndhwc_equals_forward" -x -q -s > $O/pytest_new.log 2>&1; echo "pytest_new rc $?" >> $O/rc.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize_golden.py -x -q -s > $O/pytest_fullsize.log 2>&1; echo "pytest_fullsize rc $?" >> $O/rc.txt
for i in 1 2 3; do
  MSNET_IN_FUSED=0 timeout 300 python bench.py --workload cfg3 --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/cfg3_unfused_$i.json 2>$O/cfg3_unfused_$i.err
  timeout 300 python bench.py --workload cfg3 --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/cfg3_fused_$i.json 2>$O/cfg3_fused_$i.err
  timeout 300 python bench.py --workload cfg3 --no-cpu-baseline --no-extras --steps 20 --warmup 5 --graph > $O/cfg3_graph_$i.json 2>$O/cfg3_graph_$i.err
done
timeout 300 python bench.py --workload cfg3 --no-cpu-baseline --no-extras --steps 10 --warmup 3 --verbose > $O/cfg3_verbose.json 2>$O/cfg3_verbose.err
timeout 900 python tools/r06_cumask_ab.py --configs "single:4;two:128/rest:2;two:all/all:2;single:2;two:128/rest:1" --steps 6 --repeats 3 > $O/cumask_ab.txt 2>$O/cumask_ab.err; echo "cumask rc $?" >> $O/rc.txt
cat $O/rc.txt
