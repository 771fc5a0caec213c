import numpy as np, torch, sys
sys.path.insert(0, '.')
from oracle import ms_volume as O
from msnets_amd import cbmv_generator as cg
rng = np.random.default_rng(1)
H, W, nd = 48, 80, 16
l = rng.integers(0, 256, (H, W), dtype=np.uint8); r = rng.integers(0, 256, (H, W), dtype=np.uint8)
ref = O.build_ms_volume(l, r, nd); got = cg.build_ms_volume(l, r, nd)
names = ["census", "ncc", "sobel", "sad", "aml_census", "aml_ncc", "aml_sobel", "aml_sad"]
for ch in range(8):
    a, b = got[ch], ref[ch]
    bad = a.view(np.uint32) != b.view(np.uint32)
    print(names[ch], "differ %d/%d" % (bad.sum(), bad.size), "max|diff| %.3e" % np.abs(a.astype(np.float64) - b).max())
    if bad.any() and ch < 4:
        idx = np.argwhere(bad)[:6]
        for d, y, x in idx:
            print("   d=%d y=%d x=%d got=%r ref=%r" % (d, y, x, a[d, y, x], b[d, y, x]))
        print("   bad per d:", bad.reshape(nd, -1).sum(1).tolist())
