#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_variant_libs.py tests/test_gpu_bench_contract.py -q -s > $O/pytest_sel.txt 2>&1
echo "rc=$?" >> $O/pytest_sel.txt
grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_sel.txt | tail
for cfg in 0 1 2 3 4 5 6; do
  echo "== MSNET_BAND_CFG=$cfg" >> $O/band_cfg.txt
  MSNET_HIP_LIB=$PWD/ms-nets_amd/libx_volknobs.so MSNET_BAND_CFG=$cfg timeout 120 python tools/tools_volume_bench.py cfg2 30 ndhwc >> $O/band_cfg.txt 2>&1
done
echo "== cfg5 shapes" >> $O/band_cfg.txt
for cfg in 0 4 5; do
  echo "== cfg5 MSNET_BAND_CFG=$cfg" >> $O/band_cfg.txt
  MSNET_HIP_LIB=$PWD/ms-nets_amd/libx_volknobs.so MSNET_BAND_CFG=$cfg timeout 120 python tools/tools_volume_bench.py cfg5 30 ndhwc >> $O/band_cfg.txt 2>&1
done
grep -v amdgpu.ids $O/band_cfg.txt
