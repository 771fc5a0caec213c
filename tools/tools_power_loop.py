"""Loops one conv layer for a few seconds (argv: layer, data mode random|relu|zeros, seconds) -- to sample rocm-smi against."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import msnets_amd
from msnets_amd import hipops
import tools_layer_bench as T
name, mode, secs = sys.argv[1], sys.argv[2], float(sys.argv[3])
kind, ci, co, stride, (d, h, w), use_res = T.LAYERS[name]
g = torch.Generator().manual_seed(0)
x = torch.rand((1, d, h, w, ci), generator=g); wt = torch.randn((co, ci, 3, 3, 3), generator=g) * 0.05
if mode == "relu": x = torch.relu(x - 0.5) * 2
if mode == "zeros": x.zero_(); wt.zero_()
x, wt = x.cuda(), wt.cuda()
wpk = hipops.pack_conv_weight(wt, f16s=True, stride=stride)
# 32 -> 32 stride 1: the Winograd-depth kernel, as in the modules (MSNET_LB_WD=0: the direct kernel)
wd = hipops.winograd_depth_weights(wt) if (ci, co, stride) == (32, 32, 1) and os.environ.get("MSNET_LB_WD", "1") != "0" else None
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(50): hipops.conv3d_k3(x, wpk, None, None, co, stride=stride, relu=True, f16s=True, wpk_wd=wd)
    torch.cuda.synchronize(); n += 50
print("%s %s%s: %.3f ms per launch over %.1f s" % (name, mode, " (winograd-depth)" if wd is not None else "", 1e3 * (time.time() - t0) / n, time.time() - t0))
