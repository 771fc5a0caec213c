"""Aggregates the FETCH_SIZE / WRITE_SIZE passes of tools_pmc_traffic.sh into profiles/r03_pmc_traffic.json (stdout) and a
per-kernel table (stderr).  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at half
their bytes (MI355X_MICROARCH.md, HBM) and is doubled here, WRITE_SIZE is exact for 16-byte-per-lane stores."""
import collections, csv, glob, hashlib, json, os, sys

root = sys.argv[1]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(sub, counter):
    per = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return per


def sha(names):
    h = hashlib.sha256()
    for n in names:
        h.update(open(os.path.join(repo, "ms-nets_amd", "csrc", n), "rb").read())
    return h.hexdigest()[:16]


fetch, write = load("fetch", "FETCH_SIZE"), load("write", "WRITE_SIZE")
sq = {c: load("sq", c) for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")}
rows = []
for k in sorted(set(fetch) | set(write)):
    if "at::native" in k or "rocclr" in k or "peak_" in k:
        continue
    f = fetch.get(k, [0.0]); w = write.get(k, [0.0])
    rows.append((k, len(f), 2048.0 * sum(f) / len(f), 1024.0 * sum(w) / len(w)))
for k, n, fb, wb in sorted(rows, key=lambda r: -(r[2] + r[3])):
    print("%-110s calls %4d  fetch(x2) %9.1f MB  write %9.1f MB" % (k.replace("msnet::", "")[:110], n, fb / 1e6, wb / 1e6), file=sys.stderr)


def pick(pred):
    return [r for r in rows if pred(r[0])]


out = {}
dom = pick(lambda k: "conv3d_wd_f16s_kernel" in k)          # conv3dbn_2: the Winograd-depth kernel ...
if not dom:
    dom = pick(lambda k: "conv3d_k3s1_f16s_ws" in k and "true, 4>" in k.replace("(bool)1", "true"))      # ... or the sliding-window direct kernel
if dom:
    k, n, fb, wb = dom[0]
    mf = sq["SQ_VALU_MFMA_BUSY_CYCLES"].get(k); bz = sq["SQ_BUSY_CYCLES"].get(k); gr = sq["GRBM_GUI_ACTIVE"].get(k)
    out["conv3d_s1_wd_f16s" if "conv3d_wd_f16s_kernel" in k else "conv3d_s1_f16s_co32"] = {
        "workload": "cfg2", "batch_per_gpu": 1, "sources": ["conv3d_f16s.hip", "conv_common.h"], "source_sha16": sha(["conv3d_f16s.hip", "conv_common.h"]),
        "kernel": k, "hbm_bytes": fb + wb, "fetch_bytes_x2": fb, "write_bytes": wb,
        "algorithmic_bytes": 2.0 * 96 * 272 * 480 * 32 * 4,
        "note": "conv3dbn_2 per launch; FETCH_SIZE doubled (gfx950 wide-read correction), WRITE_SIZE exact",
    }
vol = pick(lambda k: any(s in k for s in ("vprep_kernel", "features4_kernel", "sadsob_bandsum_kernel", "sadsob_band_kernel")))
if vol:
    # per MAP: a kernel may be launched more than once per build (features4_kernel: matchers 0-2, then Sobel-SAD), so every
    # kernel's bytes are summed over all its calls and divided by the number of builds (= calls of vprep_kernel)
    maps = max(r[1] for r in vol if "vprep_kernel" in r[0])
    per_map = {r[0]: (r[2] * r[1] / maps, r[3] * r[1] / maps, r[1] / maps) for r in vol}
    fb = sum(v[0] for v in per_map.values()); wb = sum(v[1] for v in per_map.values())
    src = ["volume_fused.hip", "volume.hip"]
    out["volume_build"] = {
        "workload": "cfg2", "batch_per_gpu": 1, "sources": src, "source_sha16": sha(src),
        "kernels": {k: {"launches_per_map": v[2], "fetch_bytes_x2": v[0], "write_bytes": v[1]} for k, v in per_map.items()},
        "hbm_bytes": fb + wb, "fetch_bytes_x2": fb, "write_bytes": wb, "algorithmic_bytes": 4.0 * 8 * 96 * 272 * 480 + 2.0 * 292 * 500,
        "note": "sum over the volume-build kernels per map; their reads are 4-byte-per-lane loads, for which the FETCH_SIZE x2 "
                "correction is uncalibrated (MI355X_MICROARCH.md): fetch_bytes_x2 is an upper bound, write_bytes is exact",
    }
print(json.dumps(out, indent=1))
