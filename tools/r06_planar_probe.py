"""Round 6, VERDICT r05 #4: the (conv3dbn_2 -> block_3d_1.convbn_3d_1) pair with the stride-2 consumer reading 16-channel planes.

Variant library only (-DEXP_WD_PLANAR16: the kernel-side code lives in commit 9b783ab and was removed from the tree afterwards; MSNET_HIP_LIB=ms-nets_amd/libx_planar16.so): the Winograd-depth kernel writes its output a
second time as [2][D][H][W][16] planes (one more drained store per element) and the stride-2 kernel's loaders read such a tensor
(64-byte records: every 16-channel chunk pass is a dense read).  Timed on one box, interleaved, at 96x272x480:
    A  Winograd-depth 32->32 (NDHWC output only)  +  stride-2 32->64 reading NDHWC           (what ships)
    B  Winograd-depth 32->32 writing both copies  +  stride-2 32->64 reading the planes
Checks: the planar copy equals the NDHWC output re-laid-out, the stride-2 output is bit-identical on both routes."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import msnets_amd  # noqa: F401
from msnets_amd import _lib, hipops

lib = _lib.load()
assert hasattr(lib, "msnet_exp_set_planar") or True
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.msnet_exp_set_planar.argtypes = [ctypes.c_void_p, ctypes.c_int]
D, H, W = 96, 272, 480
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
x = torch.rand((1, D, H, W, 32), generator=g).to(dev)
w1 = (torch.randn((32, 32, 3, 3, 3), generator=g) * 0.05).to(dev)
w2 = (torch.randn((64, 32, 3, 3, 3), generator=g) * 0.05).to(dev)
wd = hipops.winograd_depth_weights(w1)
wpk1 = hipops.pack_conv_weight(w1, f16s=True, stride=1)
wpk2 = hipops.pack_conv_weight(w2, f16s=True, stride=2)
y2 = torch.zeros((2, D, H, W, 16), device=dev)


def wdconv():
    return hipops.conv3d_k3(x, wpk1, None, None, 32, stride=1, relu=True, f16s=True, wpk_wd=wd)


def s2conv(inp):
    return hipops.conv3d_k3(inp, wpk2, None, None, 64, stride=2, relu=True, f16s=True)


def timed(fn, reps=10):
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


# correctness first
raw.msnet_exp_set_planar(None, 0)
y = wdconv().clone()
z = s2conv(y).clone()
raw.msnet_exp_set_planar(ctypes.c_void_p(y2.data_ptr()), 0)
yb = wdconv().clone()
torch.cuda.synchronize()
assert torch.equal(yb, y), "NDHWC output changed"
planes = y[0].view(D, H, W, 2, 16).permute(3, 0, 1, 2, 4).contiguous()
assert torch.equal(planes, y2), "planar copy differs: %g" % float((planes - y2).abs().max())
raw.msnet_exp_set_planar(ctypes.c_void_p(y2.data_ptr()), 1)
zb = s2conv(y2.view(1, D, H, W, 32)).clone()
torch.cuda.synchronize()
assert torch.equal(zb, z), "stride-2 output differs on the planar route: %g" % float((zb - z).abs().max())
print("checks passed: planar copy == NDHWC re-laid-out, stride-2 outputs bit-identical")

rows = []
for rep in range(4):
    raw.msnet_exp_set_planar(None, 0)
    a_wd, a_s2 = timed(wdconv), timed(lambda: s2conv(y))
    a_pair = timed(lambda: s2conv(wdconv()))
    raw.msnet_exp_set_planar(ctypes.c_void_p(y2.data_ptr()), 1)
    b_wd, b_s2 = timed(wdconv), timed(lambda: s2conv(y2.view(1, D, H, W, 32)))
    b_pair = timed(lambda: (wdconv(), s2conv(y2.view(1, D, H, W, 32)))[1])
    rows.append((a_wd, a_s2, a_pair, b_wd, b_s2, b_pair))
    print("rep %d  A: wd %.3f  s2 %.3f  pair %.3f ms   |   B (planes): wd %.3f  s2 %.3f  pair %.3f ms" % ((rep,) + rows[-1]), flush=True)
raw.msnet_exp_set_planar(None, 0)
m = [sum(r[i] for r in rows) / len(rows) for i in range(6)]
print("mean   A: wd %.3f  s2 %.3f  pair %.3f ms   |   B (planes): wd %.3f  s2 %.3f  pair %.3f ms   pair B - A = %+.3f ms" % (*m, m[5] - m[2]))
