# Same-box A/Bs of round 5.  Usage: gpurun -- bash tools/r05_ab.sh TAG
TAG=${1:-r05ab}; OUT=gpurun_out/$TAG; mkdir -p $OUT
OLD=$PWD/ms-nets_amd/libx_r05a.so
for i in 1 2 3; do
  python tools/tools_volume_bench.py cfg2 30 ndhwc >> $OUT/volume.txt 2>&1
  [ -f $OLD ] && MSNET_HIP_LIB=$OLD python tools/tools_volume_bench.py cfg2 30 ndhwc >> $OUT/volume.txt 2>&1
done
python tools/tools_volume_bench.py cfg2 30 ncdhw >> $OUT/volume.txt 2>&1
[ -f $OLD ] && MSNET_HIP_LIB=$OLD python tools/tools_volume_bench.py cfg2 30 ncdhw >> $OUT/volume.txt 2>&1
cat $OUT/volume.txt
val() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[2], round(d['value'],2), round(d['ms_per_step'],3), (d.get('power') or {}).get('power_w'), (d.get('power') or {}).get('sclk_mhz'))" $1 "$2"; }
for i in 1 2; do
  for v in "default:" "notiming:--no-kernel-timing" "ncdhw:--volume-layout ncdhw" "ncdhw_notiming:--volume-layout ncdhw --no-kernel-timing" "idbn:--identity-bn" "steps40:--steps 40"; do
    name=${v%%:*}; flags=${v#*:}
    python bench.py --no-cpu-baseline --no-extras --steps 20 $flags > $OUT/b_${name}_$i.json 2> $OUT/b_${name}_$i.err
    val $OUT/b_${name}_$i.json "$name#$i" >> $OUT/bench.txt
  done
done
cat $OUT/bench.txt
