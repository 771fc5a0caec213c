"""Per-launch summary of a rocprofv3 kernel_trace.csv: one line per distinct (kernel, grid) with mean duration."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].replace("msnet::", "").replace("(msnet::ConvArgs)", "")
    if "at::native" in name or "rocclr" in name: continue
    key = (name[:70], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault(key, []).append(d)
tot = 0
for k, v in agg.items():
    print("%-72s grid %9s wg %4s  n=%3d  mean %9.1f us" % (k[0], k[1], k[2], len(v), sum(v) / len(v)))
