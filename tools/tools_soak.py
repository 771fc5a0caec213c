"""Soak: the same forward many times on the same input must return the same bits (a hand-off race between loader and MFMA waves of
the persistent kernels would show as a run-to-run difference long before it shows in a tolerance test).
    python tools/tools_soak.py [repeats]   -- config #2 (GCNet, 960x544x192, volume build included) and config #3 (PSMNet aggregator)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msnets_amd import cbmv_generator, synthetic                     # noqa: E402
from msnets_amd.gcnet_3dcnn import GCNet_CostVolumeAggre             # noqa: E402
from msnets_amd.psmnet_3dcnn import PSMNet_CostVolumeAggre           # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
torch.manual_seed(0)
H, W, D = 544, 960, 192
hh, wh, nd = H // 2, W // 2, D // 2
left, right, _ = synthetic.stereo_pair(hh, wh, nd, seed=3)
l, r = torch.from_numpy(left).to(dev), torch.from_numpy(right).to(dev)
builder = cbmv_generator.VolumeBuilder(hh + 20, wh + 20, nd, dev)
g = GCNet_CostVolumeAggre(D).eval().to(dev)
ref = None
bad = 0
for k in range(reps):
    vol = builder(l, r)
    out = g(vol.unsqueeze(0))
    if ref is None:
        ref_vol, ref = vol.clone(), out.clone()
    else:
        if not torch.equal(vol, ref_vol) or not torch.equal(out, ref):
            bad += 1
            print("cfg2 repeat %d differs: volume %s, disparity max diff %.3e" % (k, torch.equal(vol, ref_vol), float((out - ref).abs().max())))
print("cfg2: %d repeats, %d differ" % (reps, bad))
# the channels-last route (round 4): build [D',H',W',8] + forward_ndhwc must return the NCDHW route's bits, every time
builder_cl = cbmv_generator.VolumeBuilder(hh + 20, wh + 20, nd, dev, layout="ndhwc")
ref_cl = ref_vol.permute(1, 2, 3, 0).contiguous()
bad_cl = 0
for k in range(reps):
    vcl = builder_cl(l, r)
    out = g.forward_ndhwc(vcl.unsqueeze(0))
    if not torch.equal(vcl, ref_cl) or not torch.equal(out, ref):
        bad_cl += 1
        print("cfg2 channels-last repeat %d differs: volume %s, disparity max diff %.3e" % (k, torch.equal(vcl, ref_cl), float((out - ref).abs().max())))
print("cfg2 channels-last: %d repeats, %d differ" % (reps, bad_cl))
bad += bad_cl
del g, vol, out, ref, ref_vol, builder, builder_cl, vcl, ref_cl
torch.cuda.empty_cache()
p = PSMNet_CostVolumeAggre(D).eval().to(dev)
x = torch.rand((1, 64, D // 4, H // 4, W // 4), device=dev)
ref = None
bad3 = 0
for k in range(reps):
    out = p(x)
    if ref is None:
        ref = out.clone()
    elif not torch.equal(out, ref):
        bad3 += 1
        print("cfg3 repeat %d differs: max diff %.3e" % (k, float((out - ref).abs().max())))
print("cfg3: %d repeats, %d differ" % (reps, bad3))
sys.exit(1 if bad or bad3 else 0)
