"""Does the 256 MiB Infinity Cache make a just-written activation slab cheaper to read than HBM?  Times the fused GCNet tail
(deconv5 + soft-argmin, HBM-read-bound) on a D'-slab of the deconvbn4 output (a) right after the slab was written by a copy
kernel (resident in the memory-side cache if it keeps written lines), (b) after 2 GB of unrelated traffic flushed it.
    python tools/tools_mall_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msnets_amd
from msnets_amd import hipops

dev = torch.device("cuda")
H, W, C = 272, 480, 32
wt = (torch.randn((32, 1, 3, 3, 3)) * 0.1).to(dev)
big = torch.empty(512 * 1024 * 1024, device=dev, dtype=torch.float32)      # 2 GB flush buffer
ev = lambda: torch.cuda.Event(enable_timing=True)
for dslab in (4, 8, 12, 16, 24, 96):
    src = torch.rand((1, dslab, H, W, C), device=dev)
    x = torch.empty_like(src)
    mb = x.numel() * 4 / 1e6
    res = {}
    for mode in ("warm", "cold"):
        ts = []
        for _ in range(5):
            x.copy_(src)                       # the producer: writes the slab
            if mode == "cold":
                big.fill_(1.0)                 # 2 GB of other traffic: evicts the slab from the Infinity Cache
            a, b = ev(), ev()
            a.record()
            hipops.deconv5_softargmin(x, wt, 0.1)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        res[mode] = sorted(ts)[len(ts) // 2]
    print("slab D'=%3d (%7.1f MB): tail right after the write %.3f ms (%.2f TB/s), after a 2 GB flush %.3f ms (%.2f TB/s)"
          % (dslab, mb, res["warm"], mb / res["warm"] / 1e3, res["cold"], mb / res["cold"] / 1e3), flush=True)
