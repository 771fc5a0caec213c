"""Times the fused matching-space volume build at a benchmark shape (HIP events via the library's own per-launch
profiler) and prints the per-kernel split.  Also the workload of the volume PMC passes (tools_pmc_volume.sh).
   python tools_volume_bench.py [cfg2|cfg5|cfg1] [reps] [ndhwc]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msnets_amd
from msnets_amd import _lib, cbmv_generator, synthetic

SHAPES = {"cfg2": (272, 480, 96), "cfg5": (192, 624, 96), "cfg1": (128, 256, 32)}

if __name__ == "__main__":
    name = next((a for a in sys.argv[1:] if a in SHAPES), "cfg2")
    reps = next((int(a) for a in sys.argv[1:] if a.isdigit()), 20)
    hh, wh, nd = SHAPES[name]
    dev = torch.device("cuda")
    l, r, _ = synthetic.stereo_pair(hh, wh, nd, seed=0)
    l, r = torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)
    layout = "ndhwc" if "ndhwc" in sys.argv[1:] else "ncdhw"
    vb = cbmv_generator.VolumeBuilder(hh + 20, wh + 20, nd, dev, layout=layout)
    out = torch.empty(vb.out_shape, device=dev)
    for _ in range(3):
        vb(l, r, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        vb(l, r, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    alg = 4.0 * 8 * nd * hh * wh + 2.0 * (hh + 20) * (wh + 20)
    print("%s [%s] volume build: %.3f ms per map back to back = %.0f GB/s algorithmic (%.1f MB)" % (name, layout, ms, alg / ms / 1e6, alg / 1e6))
    _lib.prof_enable(True)
    for _ in range(reps):
        vb(l, r, out=out)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    tot = 0.0
    for k, v in sorted(_lib.prof_collect().items(), key=lambda kv: -kv[1]["ms"]):
        print("  %-22s %8.1f us per map" % (k, 1e3 * v["ms"] / reps))
        tot += v["ms"] / reps
    print("  sum of kernels %.1f us = %.0f GB/s algorithmic" % (1e3 * tot, alg / tot / 1e6))
