#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_s2nt; mkdir -p $O
for i in 1 2 3; do
  python tools/tools_layer_bench.py s2_32_64 s2_64_64 2>&1 | grep -v amdgpu >> $O/layers_base.txt
  MSNET_HIP_LIB=$PWD/ms-nets_amd/libx_s2nt.so python tools/tools_layer_bench.py s2_32_64 s2_64_64 2>&1 | grep -v amdgpu >> $O/layers_nt.txt
done
echo base; grep split-fp16 $O/layers_base.txt; echo nt; grep split-fp16 $O/layers_nt.txt
bash tools/tools_ab.sh r04_s2nt/ab libx_s2nt.so 2>&1 | grep -E "diff|s2_f16s|kernels|==" 
cd /tmp && export TMPDIR=/tmp
for lib in base nt; do
  if [ $lib = nt ]; then export MSNET_HIP_LIB=$GRAFT_REPO_ROOT/ms-nets_amd/libx_s2nt.so; fi
  rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_$lib -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing > $O/pmc_$lib.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r04_s2nt"
for lib in ("base","nt"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(O+"/pmc_%s/**/*counter_collection.csv"%lib, recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,c in agg.items():
        if "2, 2, 32, 32, 1, 2" in k:
            print(lib, {n: "%.4g"%(sum(v)/len(v)) for n,v in c.items()}, "fetch x2 MB per launch %.1f"%(2048*sum(c["FETCH_SIZE"])/len(c["FETCH_SIZE"])/1e6))
PY
