"""Would cutting the tail's 96-slice chains into depth segments pay?  The same 1.6 GB of deconvbn4 output as N x D slices:
(1, 96), (2, 48), (3, 32), (4, 24), (6, 16) -- every shape is 624 tiles x N workgroups walking D slices each (a segment boundary's
extra pre-step and the merge pass are NOT in these numbers).  HIP events around 20 launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import msnets_amd  # noqa: E402,F401
from msnets_amd import hipops  # noqa: E402

dev = torch.device("cuda:0")
w = (torch.randn(32, 1, 3, 3, 3) * 0.05).to(dev)
for n, d in ((1, 96), (2, 48), (3, 32), (4, 24), (6, 16), (1, 96)):
    x = torch.rand((n, d, 272, 480, 32), device=dev)
    for _ in range(3):
        hipops.deconv5_softargmin(x, w, 0.1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        hipops.deconv5_softargmin(x, w, 0.1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("N=%d x D'=%2d: %.3f ms = %.2f TB/s" % (n, d, ms, x.numel() * 4 / ms / 1e9), flush=True)
    del x
