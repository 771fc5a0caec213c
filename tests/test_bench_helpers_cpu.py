"""bench.py's algorithmic FLOP counts against the figures SURVEY.md section 8(a) quotes for the reference modules
(a13: MS-GCNet 1065 GMAC = 2.13 TFLOP at config #2, 89 GMAC at config #1, 978 GMAC at config #5; a16: PSMNet aggregator
505 GMAC at config #3) -- the numerators of every MFMA roofline fraction in the JSON line."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def test_flop_counts_match_the_survey():
    import bench
    assert abs(bench.gcnet_flops(544, 960, 192) / 2 - 1065e9) < 1e9
    assert abs(bench.gcnet_flops(256, 512, 64) / 2 - 89e9) < 0.5e9
    assert abs(bench.gcnet_flops(384, 1248, 192) / 2 - 978e9) < 1e9
    assert abs(bench.psmnet_flops(544, 960, 192) / 2 - 505e9) < 0.5e9


def test_flop_count_equals_the_modules_own_layers():
    """The same number from the module definitions: 2 * 27 * Ci * Co * output voxels (input voxels for transposed convs)."""
    import torch
    import bench
    import msnets_amd  # noqa: F401
    from msnets_amd.gcnet_3dcnn import GCNet_CostVolumeAggre
    from msnets_amd.psmnet_3dcnn import PSMNet_CostVolumeAggre

    def count(model, vox0):
        total, vox = 0.0, vox0
        for m in model.modules():            # registration order == execution order for these two modules' conv layers
            if isinstance(m, torch.nn.Conv3d):
                if m.stride[0] == 2:
                    vox //= 8
                total += 2.0 * 27 * m.in_channels * m.out_channels * vox
            elif isinstance(m, torch.nn.ConvTranspose3d):
                total += 2.0 * 27 * m.in_channels * m.out_channels * vox
                vox *= m.stride[0] ** 3
        return total
    g = GCNet_CostVolumeAggre(192)
    assert abs(count(g, 96 * 272 * 480) - bench.gcnet_flops(544, 960, 192)) < 1e6
    # PSMNet: three hourglasses return to the input resolution between the registered layers; classification heads at full size
    p = PSMNet_CostVolumeAggre(192)
    v0 = 48 * 136 * 240
    total = 0.0
    for name, seq in (("dres0", p.dres0), ("dres1", p.dres1)):
        total += count(seq, v0)
    for h in (p.dres2, p.dres3, p.dres4):
        total += count(h, v0)
    for c in (p.classif1, p.classif2, p.classif3):
        total += count(c, v0)
    assert abs(total - bench.psmnet_flops(544, 960, 192)) < 1e6
