#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_tcc2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $C | tr ' ' '+')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing > $O/$tag.log 2>&1
  tail -2 $O/$tag.log
done
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/r04_tcc2"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(O+"/summary.txt","w") as out:
    for k,c in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
        if "msnet" not in k: continue
        out.write(k.replace("msnet::","")[:90].ljust(92)+" calls %d "%len(next(iter(c.values())))+" ".join("%s=%.4g"%(n.replace("_sum",""), sum(v)/len(v)) for n,v in sorted(c.items()))+"\n")
print(open(O+"/summary.txt").read()[:4000])
PY
