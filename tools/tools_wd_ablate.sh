# Winograd-depth kernel, timing-only ablations (wrong results) in the hot loop on random and on all-zero data (= full clock, no power limit):
# what the tile requests / the stores cost in CYCLES.
cd $GRAFT_REPO_ROOT
for so in "" $(ls ms-nets_amd/libx_*.so); do
  for mode in random zeros; do
    MSNET_HIP_LIB=${so:+$PWD/$so} python tools/tools_power_loop.py s1_32_32 $mode 3 2>&1 | grep "ms per" | sed "s|^|${so:-shipped} |"
  done
done
