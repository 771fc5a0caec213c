"""Times single conv layers at benchmark shapes (HIP events), for kernel A/B work.
   python tools_layer_bench.py [name ...]      MSNET_HIP_LIB=<variant.so> selects another build."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import msnets_amd
from msnets_amd import hipops, _lib

LAYERS = {
    # name: (kind, Ci, Co, stride, (D,H,W) input, residual)
    "c8":        ("conv", 8, 32, 1, (96, 272, 480), False),
    "s1_32_32":  ("conv", 32, 32, 1, (96, 272, 480), False),
    "s1_32_32q": ("conv", 32, 32, 1, (48, 136, 240), False),      # PSMNet's 32->32 layers (quarter resolution)
    "s2_32_64":  ("conv", 32, 64, 2, (96, 272, 480), False),
    "s2_32_64s": ("conv", 32, 64, 2, (8, 272, 480), False),       # same layer on a 134 MB input (fits the 256 MB Infinity Cache)
    "s2_16_64":  ("conv", 16, 64, 2, (96, 272, 480), False),     # (not a network layer: stride-2 staging with full-line requests)
    "s1_64_64":  ("conv", 64, 64, 1, (48, 136, 240), False),
    "s2_64_64":  ("conv", 64, 64, 2, (48, 136, 240), False),
    "s1_64_64b": ("conv", 64, 64, 1, (24, 68, 120), False),
    "s1_128":    ("conv", 128, 128, 1, (6, 17, 30), False),
    "d_64_32":   ("deconv", 64, 32, 2, (48, 136, 240), True),
    "d_64_32n":  ("deconv", 64, 32, 2, (48, 136, 240), False),    # deconvbn4's shape WITHOUT its residual (1.6 GB less read): what the residual path costs
    "d_64_64":   ("deconv", 64, 64, 2, (24, 68, 120), True),
    "d_128_64":  ("deconv", 128, 64, 2, (6, 17, 30), True),
}

def run(name, prec, reps=5):
    kind, ci, co, stride, (d, h, w), use_res = LAYERS[name]
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.rand((1, d, h, w, ci), generator=g).to(dev)
    if kind == "conv":
        wt = (torch.randn((co, ci, 3, 3, 3), generator=g) * 0.05).to(dev)
        f16s = prec == "split-fp16" and bool(_lib.load().msnet_conv3d_k3_f16s_supported(ci, co, stride))
        wpk = hipops.pack_conv_weight(wt, f16s=f16s, stride=stride)
        od = [(v - 1) // stride + 1 for v in (d, h, w)]
        res = torch.rand((1, *od, co), device=dev) if use_res else None
        wd = (hipops.winograd_depth_weights(wt) if f16s and ci == 32 and co == 32 and stride == 1 and
              os.environ.get("MSNET_LB_WD", "1") != "0" else None)         # MSNET_LB_WD=0: the direct split-fp16 kernel
        wd4 = None
        if hipops.USE_WD64 and f16s and ci == 64 and co == 64 and stride == 1:          # MSNET_WD64=1: four Winograd-depth launches
            wd4 = [[hipops.winograd_depth_weights(wt[32 * a:32 * a + 32, 32 * b:32 * b + 32].contiguous()) for b in range(2)] for a in range(2)]
        fn = lambda: hipops.conv3d_k3(x, wpk, None, None, co, stride=stride, relu=True, residual=res, f16s=f16s, wpk_wd=wd, wpk_wd4=wd4)
        vox = od[0] * od[1] * od[2]
    else:
        wt = (torch.randn((ci, co, 3, 3, 3), generator=g) * 0.05).to(dev)
        f16s = prec == "split-fp16" and bool(_lib.load().msnet_deconv3d_k3s2_f16s_supported(ci, co))
        wpk = hipops.pack_conv_weight(wt, transposed=True, f16s=f16s)
        res = torch.rand((1, 2 * d, 2 * h, 2 * w, co), device=dev) if use_res else None
        fn = lambda: hipops.deconv3d_k3s2(x, wpk, None, None, co, relu=True, residual=res, f16s=f16s)
        vox = d * h * w
    for _ in range(2 if reps < 20 else 10): y = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): y = fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * 27 * ci * co * vox
    print("%-10s %-10s f16s=%d  %8.3f ms  %7.1f TFLOP/s" % (name, prec, f16s, ms, fl / ms / 1e9), flush=True)

if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if a in LAYERS] or list(LAYERS)
    precs = [a for a in sys.argv[1:] if a in ("fp32", "split-fp16")] or ["split-fp16"]
    print("lib:", _lib.LIB_PATH)
    for p in precs:
        for n in names:
            run(n, p)
