# Sobel-SAD band kernel: which phase costs what (timing-only: skipped phases give wrong results).  Needs libx_volknobs.so (-DEXP_VOLUME_KNOBS).
# Usage: gpurun -- bash tools/r05_band_phases.sh TAG
TAG=${1:-r05band}; OUT=gpurun_out/$TAG.txt
export MSNET_HIP_LIB=$PWD/ms-nets_amd/libx_volknobs.so
for cfgb in 0 4 5; do for skip in 0 1 2 4 3 5 6 7; do
  echo "== MSNET_BAND_CFG=$cfgb MSNET_BAND_SKIP=$skip (1: no vertical pass, 2: no horizontal chain, 4: no boxes)" >> $OUT
  MSNET_BAND_CFG=$cfgb MSNET_BAND_SKIP=$skip python tools/tools_volume_bench.py cfg2 20 ndhwc 2>&1 | grep -v amdgpu.ids | head -1 >> $OUT
done; done
cat $OUT
