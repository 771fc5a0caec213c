"""Aggregates the FETCH_SIZE / WRITE_SIZE passes of tools_pmc_traffic.sh into profiles/r04_pmc_traffic.json (stdout) and a
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
CONV_SRC = sorted(f for f in os.listdir(os.path.join(repo, "ms-nets_amd", "csrc")) if f.startswith(("conv3d_f16s", "conv_f16s", "conv_common")))
dom = pick(lambda k: "conv3d_wd_f16s_kernel" in k)          # conv3dbn_2: the Winograd-depth kernel
if dom:
    k, n, fb, wb = dom[0]
    out["conv3d_s1_wd_f16s"] = {
        "workload": "cfg2", "batch_per_gpu": 1, "sources": CONV_SRC, "source_sha16": sha(CONV_SRC),
        "kernel": k, "hbm_bytes": fb + wb, "fetch_bytes_x2": fb, "write_bytes": wb,
        "algorithmic_bytes": 2.0 * 96 * 272 * 480 * 32 * 4,
        "note": "conv3dbn_2 per launch; FETCH_SIZE doubled (gfx950 wide-read correction), WRITE_SIZE exact",
    }
# the Co = 64 stride-1 family (bench.py's dominant family by total time): bytes per MAP summed over its eight launches
co64 = pick(lambda k: "conv3d_k3s1_f16s_ws<2, 8, 16, 16, 2, 2" in k or "conv3d_k3s1_f16s_ws<2, 4, 32, 32, 2, 2, true" in k or
            "conv3d_direct_f16s_kernel<false, 8>" in k)
if co64 and dom:
    maps = dom[0][1]
    fb = sum(r[2] * r[1] for r in co64) / maps; wb = sum(r[3] * r[1] for r in co64) / maps
    v = [48 * 136 * 240] * 2 + [24 * 68 * 120] * 2 + [12 * 34 * 60] * 2           # 64 -> 64 outputs (+ 2 x 128 -> 128 at 6 x 17 x 30)
    out["conv3d_s1_f16s_co64"] = {
        "workload": "cfg2", "batch_per_gpu": 1, "sources": CONV_SRC, "source_sha16": sha(CONV_SRC),
        "kernels": [r[0] for r in co64], "launches_per_map": sum(r[1] for r in co64) / maps,
        "hbm_bytes": fb + wb, "fetch_bytes_x2": fb, "write_bytes": wb,
        "algorithmic_bytes": sum(2.0 * x * 64 * 4 for x in v) + 2 * 2.0 * 6 * 17 * 30 * 128 * 4,
        "note": "per map, summed over the family's launches; FETCH_SIZE doubled (gfx950 wide-read correction), WRITE_SIZE exact",
    }
vol = pick(lambda k: any(s in k for s in ("vprep_kernel", "features4_kernel", "features_cl_kernel", "sadsob_bandsum_kernel", "sadsob_band_kernel")))
if vol:
    # per MAP: a kernel may be launched more than once per build (features4_kernel: matchers 0-2, then Sobel-SAD), so every
    # kernel's bytes are summed over all its calls and divided by the number of builds (= calls of vprep_kernel)
    maps = max(r[1] for r in vol if "vprep_kernel" in r[0])
    per_map = {r[0]: (r[2] * r[1] / maps, r[3] * r[1] / maps, r[1] / maps) for r in vol}
    fb = sum(v[0] for v in per_map.values()); wb = sum(v[1] for v in per_map.values())
    src = ["volume_fused.hip", "volume.hip"]          # (+ common.h: unchanged across rounds)
    out["volume_build"] = {
        "workload": "cfg2", "batch_per_gpu": 1, "sources": src, "source_sha16": sha(src),
        "kernels": {k: {"launches_per_map": v[2], "fetch_bytes_x2": v[0], "write_bytes": v[1]} for k, v in per_map.items()},
        "hbm_bytes": fb + wb, "fetch_bytes_x2": fb, "write_bytes": wb, "algorithmic_bytes": 4.0 * 8 * 96 * 272 * 480 + 2.0 * 292 * 500,
        "note": "sum over the volume-build kernels per map; their reads are 4-byte-per-lane loads, for which the FETCH_SIZE x2 "
                "correction is uncalibrated (MI355X_MICROARCH.md): fetch_bytes_x2 is an upper bound, write_bytes is exact",
    }
print(json.dumps(out, indent=1))
