"""Full-size PSMNet / GCNet: HIP (both precisions) vs the fp32 oracle and an fp64 evaluation of the same network."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT_); sys.path.insert(0, os.path.join(ROOT_, "tests", "golden"))
import torch, recipes
import msnets_amd
from msnets_amd import hipops
from msnets_amd.gcnet_3dcnn import GCNet_CostVolumeAggre as G
from msnets_amd.psmnet_3dcnn import PSMNet_CostVolumeAggre as P
from oracle import aggregators as O
torch.set_num_threads(min(64, os.cpu_count()))
for name, shape in (("psmnet", (1, 64, 48, 136, 240)), ("gcnet", (1, 8, 96, 272, 480))):
    case = dict(model=name, seed=21, maxdisp=192, in_shape=shape)
    m = recipes.build_case(case, G, P); sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = recipes.make_input(shape, 21)
    with torch.no_grad():
        f = (lambda s_, x_: O.psmnet_forward(s_, x_, 192, (544, 960))) if name == "psmnet" else (lambda s_, x_: O.gcnet_forward(s_, x_, 192))
        ref32 = f(sd, x)
        ref64 = f({k: v.double() for k, v in sd.items()}, x.double()).float()
    print(name, "oracle fp32 vs fp64: %.3e" % (ref32 - ref64).abs().max().item(), flush=True)
    m = m.cuda()
    for prec in ("fp32", "split-fp16"):
        hipops.set_default_precision(prec)
        got = m(x.cuda()).cpu()
        print(name, prec, "HIP vs oracle fp32: %.3e | HIP vs fp64: %.3e" % ((got - ref32).abs().max().item(), (got - ref64).abs().max().item()), flush=True)
