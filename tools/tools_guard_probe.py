import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
import msnets_amd
from msnets_amd import _lib, cbmv_generator, hipops, synthetic
from msnets_amd.gcnet_3dcnn import GCNet_CostVolumeAggre
dev = torch.device("cuda")
torch.manual_seed(0)
model = GCNet_CostVolumeAggre(192).eval().to(dev)
vol = synthetic.random_volume((1, 8, 96, 272, 480), seed=0).to(dev)
for mode in ("guard", "noguard", "guard"):
    model.range_check = mode == "guard"
    for _ in range(3):
        model(vol)
    torch.cuda.synchronize()
    ts = []
    for k in range(12):
        t0 = time.perf_counter()
        out = model(vol)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    print(mode, " ".join("%.2f" % t for t in ts), flush=True)
