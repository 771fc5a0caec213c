# Depth segments per tile column of the Winograd-depth kernel at PSMNet's 48x136x240 (cfg#3: six launches per step), forced through
# MSNET_FORCE_WD_SEG against the launch's own cost model (seg = 12 there).  Usage: gpurun -- bash tools/r05_wd_seg_sweep.sh TAG
TAG=${1:-r05_wd_seg}; OUT=gpurun_out/$TAG.txt
for rep in 1 2; do
for seg in model 1 2 3 4 6 8 12 24; do
  if [ $seg = model ]; then unset MSNET_FORCE_WD_SEG; else export MSNET_FORCE_WD_SEG=$seg; fi
  python bench.py --workload cfg3 --steps 10 --warmup 3 --verbose --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "conv3d_s1_wd_f16s|kernels " | tr '\n' ' ' | sed "s/^/seg=$seg /" >> $OUT
  echo >> $OUT
done; done
cat $OUT
