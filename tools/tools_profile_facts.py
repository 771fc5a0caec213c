"""profiles/r04_profile_facts.json (stdout) from one round's rocprofv3 outputs, so bench.py can quote the TRACKED profile next
to its live HIP-event numbers (roofline.frac_rocprof, roofline.sustained_clock_ghz, roofline_volume.limiter) -- gated on the
sha256 of the kernel sources.

    python tools/tools_profile_facts.py <kernel_stats.csv of `rocprofv3 --kernel-trace --stats -- python3 bench.py ...`> <pmc dir with sq/ [sq2/]> [steps]

family_ns_per_step: nanoseconds per step (= per map at batch 1) of each of bench.py's launch families, summed over the kernels
  that make it up (FAMILIES below; kernel_stats.csv counts the untimed setup / warm-up / post-pass steps too, so totals are
  divided by the number of steps the kernel ran: calls of the once-per-step Winograd launch).
sustained_clock_ghz: GRBM_GUI_ACTIVE (summed over the 8 XCDs by rocprofv3) / 8 / the launch's duration, averaged over launches.
volume_limiter: what the counters say limits the volume build's feature kernel (vector-instruction issue).
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
SOURCES = sorted(f for f in os.listdir(os.path.join(repo, "ms-nets_amd", "csrc")) if f.endswith((".hip", ".h", ".cpp")))
# bench.py family -> predicate on the rocprofv3 kernel name (cfg#2, batch 1: which kernels a family's launches run)
FAMILIES = {
    "conv3d_s1_wd_f16s": lambda n: "conv3d_wd_f16s_kernel" in n,
    "conv3d_s1_f16s_co64": lambda n: ("conv3d_k3s1_f16s_ws<2, 8, 16, 16, 2, 2" in n or "conv3d_k3s1_f16s_ws<2, 4, 32, 32, 2, 2, true" in n
                                      or "conv3d_direct_f16s_kernel<false, 8>" in n),
    "conv3d_s2_f16s": lambda n: ("conv3d_k3s1_f16s_ws<2, 2, 32, 32, 1, 2" in n or "conv3d_k3s1_f16s_ws<2, 4, 16, 16, 1, 2" in n
                                 or "conv3d_direct_f16s_kernel<false, 4>" in n),
    "deconv3d_f16s": lambda n: "deconv3d_k3s2_f16s_ws" in n or "conv3d_direct_f16s_kernel<true" in n,
    "conv3d_s1_c8_f16s": lambda n: "conv3d_c8_f16s_kernel" in n,
    "deconv5_softargmin": lambda n: "deconv5_tail_mfma_kernel" in n or "softargmin_merge_kernel" in n,
}


def sha(names):
    h = hashlib.sha256()
    for n in names:
        h.update(open(os.path.join(repo, "ms-nets_amd", "csrc", n), "rb").read())
    return h.hexdigest()[:16]


rows = [r for r in csv.DictReader(open(stats_csv))]
wd = [r for r in rows if FAMILIES["conv3d_s1_wd_f16s"](r["Name"])]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else (int(wd[0]["Calls"]) if wd else 1)
out = {"workload": "cfg2", "batch_per_gpu": 1, "sources": SOURCES, "source_sha16": sha(SOURCES), "steps_in_trace": steps,
       "kernel_stats_csv": os.path.basename(stats_csv), "family_ns_per_step": {}, "family_launches_per_step": {},
       "kernel_avg_ns": {}, "sustained_clock_ghz": {}}
for fam, pred in FAMILIES.items():
    rs = [r for r in rows if pred(r["Name"])]
    if rs:
        out["family_ns_per_step"][fam] = sum(float(r["TotalDurationNs"]) for r in rs) / steps
        out["family_launches_per_step"][fam] = sum(int(r["Calls"]) for r in rs) / steps
for r in rows:
    if any(t in r["Name"] for t in ("msnet::", "msnet_")):
        out["kernel_avg_ns"][r["Name"].replace("msnet::", "")[:100]] = float(r["AverageNs"])

cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("sq", "sq2"):
    for f in glob.glob(os.path.join(pmc_dir, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            cnt[r["Kernel_Name"]][r["Counter_Name"]].append((float(r["Counter_Value"]), dur))
avg_ns = {r["Name"]: float(r["AverageNs"]) for r in rows}
fam_clk = collections.defaultdict(list)
for k, c in cnt.items():
    g = c.get("GRBM_GUI_ACTIVE")
    # (GRBM_GUI_ACTIVE / 8 / duration is only meaningful for launches long enough to keep all eight XCDs busy throughout)
    if g and "msnet" in k and avg_ns.get(k, 0.0) >= 2.0e5:
        ghz = [v / 8.0 / d for v, d in g if d > 0]
        out["sustained_clock_ghz"][k.replace("msnet::", "")[:100]] = sum(ghz) / len(ghz)
        for fam, pred in FAMILIES.items():
            if pred(k):
                fam_clk[fam] += [(v / 8.0 / d, d) for v, d in g if d > 0]
for fam, v in fam_clk.items():          # duration-weighted over the family's long launches
    out["sustained_clock_ghz"][fam] = sum(c * d for c, d in v) / sum(d for _, d in v)
# the volume build's feature kernel: VALU wave-instructions x 4 cycles / (1024 SIMDs x duration x clock)
feat = [k for k in cnt if "features_cl_kernel" in k or "features4_kernel" in k]
for k in feat:
    c = cnt[k]
    if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        insts = sum(v for v, _ in c["SQ_INSTS_VALU"]) / len(c["SQ_INSTS_VALU"])
        gui = sum(v for v, _ in c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"]) / 8.0          # cycles of the launch
        util = insts * 4.0 / (1024.0 * gui) if gui > 0 else None
        wait = None
        if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
            wait = (sum(v for v, _ in c["SQ_WAIT_ANY"]) / len(c["SQ_WAIT_ANY"])) / (sum(v for v, _ in c["SQ_WAVE_CYCLES"]) / len(c["SQ_WAVE_CYCLES"]))
        out["volume_limiter"] = {"bound": "valu-issue", "kernel": k.replace("msnet::", "")[:60], "valu_wave_insts_per_launch": insts,
                                 "valu_issue_utilisation": util, "wave_cycles_waiting_frac": wait,
                                 "note": "vector-instruction issue, not HBM: VALU wave-instructions x 4 cycles / (1024 SIMDs x the launch's "
                                         "cycles); the HBM roofline (8 TB/s) is what SURVEY 8(d) prices the build against"}
print(json.dumps(out, indent=1))
