# What per-launch timing costs the headline: recorded event pairs (libx_evrec.so = the library one commit earlier) vs events attached to
# the kernels' own dispatch packets (shipped), each with and without timing, interleaved.  Usage: gpurun -- bash tools/r05_event_ab.sh TAG
TAG=${1:-r05_event_ab}; OUT=gpurun_out/$TAG.txt
for i in 1 2 3; do
  for v in "attached:::" "attached_notiming:::--no-kernel-timing" "recorded:$PWD/ms-nets_amd/libx_evrec.so::" "recorded_notiming:$PWD/ms-nets_amd/libx_evrec.so::--no-kernel-timing"; do
    name=${v%%:*}; rest=${v#*:}; lib=${rest%%::*}; flags=${rest#*::}
    MSNET_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-extras --steps 30 $flags 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$name#$i', round(d['value'],2), round(d['ms_per_step'],3), 'dominant family ms/step', round(r['kernels'][0]['ms_per_step'],3) if r['kernels'] else None)" >> $OUT
  done
done
cat $OUT
