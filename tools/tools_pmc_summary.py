import csv, sys, glob, collections
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = r["Kernel_Name"].replace("msnet::", "")[:60]
        if "at::native" in k or "rocclr" in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", f.split("/")[-3])
    for k, cs in agg.items():
        print("  ", k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
