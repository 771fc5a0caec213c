"""Synthetic stereo pairs and volumes of the benchmark shapes; no datasets exist
offline.  Pure NumPy/torch-CPU generators -- no arithmetic of the hot path happens here."""
import numpy as np
import torch

BORDER = 10

# name -> (full-res H, W after pad-to-32, maxdisp)
CONFIGS = {
    "cfg1_256x512_d64": (256, 512, 64),
    "cfg2_sceneflow_960x540_d192": (544, 960, 192),
    "cfg5_kitti_1242x375_d192": (384, 1248, 192),
}


def stereo_pair(Hh, Wh, ndisp, seed=0, border=BORDER, bands=8):
    """Half-resolution bordered pair for the matchers: uint8 [Hh+2b, Wh+2b] x2 and the planted per-row-band
    disparity [Hh].  Texture: uniform noise smoothed by a 3x3 box (non-degenerate NCC/ZSAD); the right
    image is the left one shifted by a piecewise-constant disparity in [0, ndisp-1]."""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(Hh + 2, Wh + 2 * ndisp + 2)).astype(np.float32)
    sm = sum(base[dy:dy + Hh, dx:dx + Wh + 2 * ndisp] for dy in range(3) for dx in range(3)) / 9.0
    base = np.clip(np.rint(sm), 0, 255).astype(np.uint8)
    disp_rows = np.zeros(Hh, np.int64)
    edges = np.linspace(0, Hh, bands + 1).astype(int)
    band_d = rng.integers(0, ndisp, size=bands)
    for b in range(bands):
        disp_rows[edges[b]:edges[b + 1]] = band_d[b]
    left = base[:, ndisp:ndisp + Wh]
    right = np.empty_like(left)
    for y in range(Hh):
        right[y] = base[y, ndisp + disp_rows[y]: ndisp + disp_rows[y] + Wh]   # right[x] = left[x + d]
    pad = lambda a: np.pad(a, ((border, border), (border, border)), "constant").astype(np.uint8).copy(order="C")  # noqa: E731
    return pad(left), pad(right), disp_rows


def random_volume(shape, seed=0):
    """torch.rand volume in [0,1] like real MS features (aggregator-only runs)."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g)


def randomize_bn(model, seed=0):
    """Non-identity BatchNorm for seeded random-init weights (at its defaults BN is an identity and a BN-folding
    defect cannot show): running_mean ~ 0.1 N(0,1), running_var and gamma ~ U(0.75, 1.25), beta ~ 0.1 N(0,1), from one seeded
    generator in module order -- the recipe the golden fixtures were generated with (tests/golden/recipes.py)."""
    g = torch.Generator().manual_seed(10_000 + seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm3d):
                c = m.num_features
                m.running_mean.copy_(torch.randn(c, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(c, generator=g) * 0.5 + 0.75)
                m.weight.copy_(torch.rand(c, generator=g) * 0.5 + 0.75)
                m.bias.copy_(torch.randn(c, generator=g) * 0.1)
    return model
