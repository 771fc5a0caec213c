"""Is a conv kernel power-limited?  Times the same launch on (a) uniform random data, (b) ReLU-like data (half zeros), (c) all
zeros (operands and weights): the instruction stream is identical, only the toggling in the MFMA / LDS / register data paths
differs.  A large gap between (a) and (c) means the chip's power management, not the kernel's stalls, sets the rate.
    python tools/tools_power_probe.py [layer ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import msnets_amd
from msnets_amd import hipops
import tools_layer_bench as T

def run(name, mode, reps=10):
    kind, ci, co, stride, (d, h, w), use_res = T.LAYERS[name]
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    x = torch.rand((1, d, h, w, ci), generator=g)
    wt = torch.randn((co, ci, 3, 3, 3), generator=g) * 0.05
    if mode == "relu":
        x = torch.relu(x - 0.5) * 2
    elif mode == "zeros":
        x.zero_(); wt.zero_()
    elif mode == "zero_x":
        x.zero_()
    x, wt = x.to(dev), wt.to(dev)
    wpk = hipops.pack_conv_weight(wt, f16s=True, stride=stride)
    fn = lambda: hipops.conv3d_k3(x, wpk, None, None, co, stride=stride, relu=True, f16s=True)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    od = [(v - 1) // stride + 1 for v in (d, h, w)]
    fl = 2.0 * 27 * ci * co * od[0] * od[1] * od[2]
    print("%-10s %-7s %8.3f ms  %7.1f TFLOP/s" % (name, mode, ms, fl / ms / 1e9), flush=True)

for name in [a for a in sys.argv[1:] if a in T.LAYERS] or ["s1_32_32", "s1_64_64", "s2_32_64", "c8"]:
    for mode in ("random", "relu", "zero_x", "zeros"):
        run(name, mode)
