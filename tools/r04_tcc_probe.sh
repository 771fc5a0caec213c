#!/bin/bash
# Which request sizes do the conv kernels' L2 misses use?  (Is FETCH_SIZE x 2 right for the stride-2 kernel's 64-byte-per-line loads?)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_tcc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA[0-9A-Z_]*RDREQ[A-Za-z0-9_]*\|TCC_HIT[a-z_]*\|TCC_MISS[a-z_]*\|TCC_REQ[a-z_]*\|TCP_TCC_READ_REQ[a-z_]*\|TCC_BUBBLE[a-z_]*\|TCC_EA0_RD_UNCACHED_32B[a-z_]*" | sort -u > $O/avail.txt
for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $C | tr ' ' '+')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing > $O/$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/r04_tcc"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(O+"/summary.txt","w") as out:
    for k,c in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
        if "msnet" not in k: continue
        out.write(k.replace("msnet::","")[:90].ljust(92)+" ".join("%s=%.4g"%(n.replace("_sum",""), sum(v)/len(v)) for n,v in sorted(c.items()))+"\n")
print(open(O+"/summary.txt").read()[:6000])
PY
cat $O/avail.txt | tr '\n' ' '
