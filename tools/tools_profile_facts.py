"""profiles/r03_profile_facts.json (stdout) from one round's rocprofv3 outputs, so bench.py can quote the TRACKED profile next
to its live HIP-event numbers (roofline.frac_rocprof, roofline.sustained_clock_ghz) -- gated on the sha256 of the kernel sources.

    python tools/tools_profile_facts.py <kernel_stats.csv of `rocprofv3 --kernel-trace --stats -- python3 bench.py ...`> <pmc dir with sq/>

kernel_avg_ns: average duration per kernel family (bench.py's family names); the dominant family is conv3dbn_2's launch,
once per map: conv3d_s1_wd_f16s (conv3d_wd_f16s_kernel), or conv3d_s1_f16s_co32 (the slowest conv3d_k3s1_f16s_ws) without it.
sustained_clock_ghz: GRBM_GUI_ACTIVE (summed over the 8 XCDs by rocprofv3) / 8 / the launch's duration, averaged over launches.
"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stats_csv, pmc_dir = sys.argv[1], sys.argv[2]
SOURCES = ["conv3d_f16s.hip", "conv_common.h", "volume_fused.hip", "volume.hip", "tail.hip", "pack.hip"]


def sha(names):
    h = hashlib.sha256()
    for n in names:
        h.update(open(os.path.join(repo, "ms-nets_amd", "csrc", n), "rb").read())
    return h.hexdigest()[:16]


rows = [r for r in csv.DictReader(open(stats_csv))]
# conv3dbn_2's launch: the Winograd-depth kernel where it is taken, else the slowest stride-1 instantiation of the direct kernel
wd = [r for r in rows if "conv3d_wd_f16s_kernel" in r["Name"]]
s1 = [r for r in rows if "conv3d_k3s1_f16s_ws" in r["Name"]]
dom = max(wd, key=lambda r: float(r["AverageNs"])) if wd else (max(s1, key=lambda r: float(r["AverageNs"])) if s1 else None)
DOM_KEY = "conv3d_s1_wd_f16s" if wd else "conv3d_s1_f16s_co32"          # bench.py's launch-family name of that kernel
out = {"workload": "cfg2", "batch_per_gpu": 1, "sources": SOURCES, "source_sha16": sha(SOURCES),
       "kernel_stats_csv": os.path.basename(stats_csv), "kernel_avg_ns": {}, "kernel_names": {}, "sustained_clock_ghz": {}}
if dom:
    out["kernel_avg_ns"][DOM_KEY] = float(dom["AverageNs"])
    out["kernel_names"][DOM_KEY] = dom["Name"]
for r in rows:
    if any(t in r["Name"] for t in ("msnet::", "msnet_")) and r is not dom:
        out["kernel_avg_ns"][r["Name"].replace("msnet::", "")[:100]] = float(r["AverageNs"])

clk = collections.defaultdict(list)
for f in glob.glob(os.path.join(pmc_dir, "sq", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            if dur > 0:
                clk[r["Kernel_Name"]].append(float(r["Counter_Value"]) / 8.0 / dur)
if dom:
    for k, v in clk.items():
        if k == dom["Name"]:
            out["sustained_clock_ghz"][DOM_KEY] = sum(v) / len(v)
avg_ns = {r["Name"]: float(r["AverageNs"]) for r in rows}
for k, v in clk.items():
    # (GRBM_GUI_ACTIVE / 8 / duration is only meaningful for launches long enough to keep all eight XCDs busy throughout)
    if "msnet" in k and avg_ns.get(k, 0.0) >= 2.0e5:
        out["sustained_clock_ghz"].setdefault(k.replace("msnet::", "")[:100], sum(v) / len(v))
print(json.dumps(out, indent=1))
