"""A/B helper for experiment builds: runs the cfg#2 MS-GCNet forward (seeded weights, seeded random volume) on the library
named by MSNET_HIP_LIB (default: the shipped one), saves the disparity map and, if a second path is given, prints the max
difference to it.     python tools/tools_ab_disp.py out.npy [other.npy]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np
import torch

import msnets_amd  # noqa: F401
import recipes
from msnets_amd.gcnet_3dcnn import GCNet_CostVolumeAggre
from msnets_amd.psmnet_3dcnn import PSMNet_CostVolumeAggre

case = dict(model="gcnet", seed=21, maxdisp=192, in_shape=(1, 8, 96, 272, 480))
m = recipes.build_case(case, GCNet_CostVolumeAggre, PSMNet_CostVolumeAggre).cuda()
x = recipes.make_input(case["in_shape"], 21).cuda()
d = m(x).cpu().numpy()
np.save(sys.argv[1], d)
print("lib %s: disparity range %.3f..%.3f" % (os.environ.get("MSNET_HIP_LIB", "default"), d.min(), d.max()))
if len(sys.argv) > 2:
    o = np.load(sys.argv[2])
    print("max |diff| vs %s = %.3e" % (sys.argv[2], np.abs(d - o).max()))
