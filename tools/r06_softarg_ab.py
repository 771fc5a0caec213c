"""Round 6: the one-exponential online-softmax push against the two-exponential form it replaces -- bit-identity and timing.
MSNET_HIP_LIB selects the library; run once per library, the script prints sha256 digests of the outputs and timings:
  fused GCNet tail (deconv5 + soft-argmin) at 48x136x240x32 -> 544x960, broad and peaky logits
  trilinear soft-argmin at 48x136x240 -> 192x544x960
  PSMNet / GCNet small forwards"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import msnets_amd  # noqa: F401
from msnets_amd import _lib, hipops

print("lib:", _lib.LIB_PATH)
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)


def digest(t):
    return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:16]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


x = (torch.rand((1, 96, 272, 480, 32), generator=g) * 2 - 0.5).to(dev)
for gain in (0.05, 1.0, 8.0):
    w = (torch.randn((32, 1, 3, 3, 3), generator=g) * gain).to(dev)
    wps, wsc = hipops.pow2_prescale(w)
    fn = lambda: hipops.deconv5_softargmin(x, wps, 0.1, wsc)     # noqa: E731
    out = fn()
    print("fused tail gain %-5g  digest %s  range %.2f..%.2f  %.3f ms" % (gain, digest(out), float(out.min()), float(out.max()), timed(fn)))
del x
for scale in (1.0, 30.0):
    c = (torch.randn((1, 48, 136, 240), generator=g) * scale).to(dev)
    fn = lambda: hipops.trilinear_softargmin(c, (192, 544, 960))  # noqa: E731
    out = fn()
    print("trilinear scale %-5g  digest %s  %.3f ms" % (scale, digest(out), timed(fn)))
lg = (torch.randn((2, 64, 40, 56), generator=g) * 5).to(dev)
print("softargmin            digest %s" % digest(hipops.softargmin(lg)))
