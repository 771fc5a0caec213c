# Which kernels are cycle-bound and which power-bound: hot loop of one layer on random and on all-zero data, time + sclk + power.
cd $GRAFT_REPO_ROOT
for L in c8 s2_32_64 s1_64_64; do
  for mode in random zeros; do
    python tools/tools_power_loop.py $L $mode 5 > /tmp/pl.txt 2>&1 &
    PID=$!
    sleep 3.2
    S=$(rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" | sed 's/.*(\([0-9]*\)Mhz).*/sclk \1 MHz/; s/.*Power (W): /power W /' | tr '\n' ' ')
    wait $PID
    echo "$(grep 'ms per' /tmp/pl.txt)   $S"
  done
done
