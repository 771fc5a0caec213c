#!/bin/bash
# Round 6: L2 / fabric request counters of every kernel of the default bench step on the final tree -- fabric read
# requests by size, L2 hits / misses / requests, TCP->TCC read requests -- separate --pmc passes, kernel trace only.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_tcc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $C | tr ' ' '+')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing > $O/$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/r06_tcc"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(O+"/summary.txt","w") as out:
    for k,c in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
        if "msnet" not in k: continue
        m={n.replace("_sum",""): sum(v)/len(v) for n,v in c.items()}
        hit = m.get("TCC_HIT",0)/max(1.0,m.get("TCC_HIT",0)+m.get("TCC_MISS",0))
        out.write(k.replace("msnet::","")[:86].ljust(88)+" calls %3d "%len(next(iter(c.values())))+" ".join("%s=%.4g"%(n, v) for n,v in sorted(m.items()))+"  L2_hit=%.2f  fabric_read_MB=%.0f\n"%(hit, (m.get("TCC_EA0_RDREQ_128B",0)*128+m.get("TCC_EA0_RDREQ_64B",0)*64+m.get("TCC_EA0_RDREQ_32B",0)*32)/1e6))
print(open(O+"/summary.txt").read()[:5000])
PY
find $O -type f ! -name "summary.txt" ! -name "*.log" -delete
