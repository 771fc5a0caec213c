"""Seeded recipes shared by the golden-vector generator (tests/golden/make_aggregator_golden.py, which
imports the REFERENCE and therefore only runs in the build container) and by the tests (which never
touch /root/reference).  Nothing in here reads the reference.

A case is fully determined by (seed, shapes): weights come from torch.manual_seed(seed) + the model
constructor (net_init), BatchNorm statistics / affine from `randomize_bn` (default BN would be an
identity and hide BN bugs), the input from `make_input`.
"""
import hashlib

import numpy as np
import torch

AGG_CASES = {
    # name: dict(model, seed, maxdisp, in_shape [N,C,D',H',W'], plus options)
    "gcnet_small": dict(model="gcnet", seed=1, maxdisp=32, in_shape=(1, 8, 16, 16, 32)),
    "gcnet_batch2": dict(model="gcnet", seed=2, maxdisp=64, in_shape=(2, 8, 32, 32, 64)),
    "gcnet_peaky": dict(model="gcnet", seed=3, maxdisp=32, in_shape=(1, 8, 16, 16, 32), peaky=25.0),
    "gcnet_ragged": dict(model="gcnet", seed=4, maxdisp=32, in_shape=(1, 8, 16, 48, 80)),
    "gcnet_quarter": dict(model="gcnet", seed=5, maxdisp=64, in_shape=(1, 8, 16, 16, 32), quarter=True),
    "psmnet_small": dict(model="psmnet", seed=6, maxdisp=32, in_shape=(1, 64, 8, 16, 16)),
    "psmnet_batch2": dict(model="psmnet", seed=7, maxdisp=32, in_shape=(2, 64, 8, 8, 24)),
}
MAX_SAMPLES = 4096

# Full-size cases (BASELINE.json configs #2, #5, #3), produced by tests/golden/make_fullsize_golden.py from the REFERENCE
# itself: random-init and `peaky` (trained-network-like softmax) weights; `ms_volume`: the input is the MS volume of the
# seeded synthetic stereo pair (oracle/ms_volume.py -- the real feature statistics) instead of torch.rand.
FULL_CASES = {
    "gcnet_cfg2": dict(model="gcnet", seed=31, maxdisp=192, in_shape=(1, 8, 96, 272, 480)),
    "gcnet_cfg2_peaky": dict(model="gcnet", seed=32, maxdisp=192, in_shape=(1, 8, 96, 272, 480), peaky=25.0),
    "gcnet_cfg2_ms_peaky": dict(model="gcnet", seed=33, maxdisp=192, in_shape=(1, 8, 96, 272, 480), peaky=25.0,
                                ms_volume=True),
    "gcnet_cfg5": dict(model="gcnet", seed=34, maxdisp=192, in_shape=(1, 8, 96, 192, 624)),
    "gcnet_cfg5_peaky": dict(model="gcnet", seed=35, maxdisp=192, in_shape=(1, 8, 96, 192, 624), peaky=25.0),
    "psmnet_cfg3": dict(model="psmnet", seed=36, maxdisp=192, in_shape=(1, 64, 48, 136, 240)),
    "psmnet_cfg3_peaky": dict(model="psmnet", seed=37, maxdisp=192, in_shape=(1, 64, 48, 136, 240), peaky=2.0),
    # round 4: a softmax that is UNIMODAL AT THE RIGHT PLACE -- the MS volume of the synthetic pair goes in, every layer keeps
    # its seeded random weights, and `plant_unimodal` routes the census likelihood channel (which peaks at the planted disparity)
    # through trunk channel 0 into deconv5 with gain `unimodal`
    "gcnet_cfg2_ms_unimodal": dict(model="gcnet", seed=46, maxdisp=192, in_shape=(1, 8, 96, 272, 480), ms_volume=True,
                                   unimodal=6.0),
    # round 5: BASELINE.json configs[0] -- one synthetic 256x512 pair, D = 64 -- end to end: the MS volume of the pair into the
    # reference aggregator (random-init, randomised BN), and on the GPU from the two images through the HIP volume build
    "gcnet_cfg1_ms": dict(model="gcnet", seed=51, maxdisp=64, in_shape=(1, 8, 32, 128, 256), ms_volume=True),
}
# Cases whose flat-1e-3 gate is exceeded on part of the map also carry the reference's OWN fp32 noise floor:
# tests/golden/fullsize_<case>_alt.npz = the unmodified reference forward, same weights, same input, under other CPU thread
# counts (another summation order inside oneDNN / ATen's reductions) -- make_fullsize_golden.py --alt
ALT_CASES = ("psmnet_cfg3", "psmnet_cfg3_peaky", "gcnet_cfg2_peaky", "gcnet_cfg5_peaky", "gcnet_cfg2_ms_peaky",
             "gcnet_cfg2_ms_unimodal")
# (the committed fullsize_<case>.npz were generated with 8 threads and oneDNN).  "t1": torch.set_num_threads(1) -- at these
# shapes 1 and 3 threads give the same bits, which differ from the 8-thread run; "nomkldnn":
# torch.backends.mkldnn.flags(enabled=False), i.e. ATen's vol2col + GEMM convolutions instead of oneDNN's direct ones (needs
# 43 GB for the column buffer of MS-GCNet's conv3dbn_2: run it alone in the build container).
ALT_VARIANTS = ("t1", "nomkldnn")
# Round 6: the same cases once more with the unmodified reference moved to FLOAT64 (make_fullsize_golden.py --f64):
# fullsize_<case>_f64.npz holds the mathematically exact disparity map, so the tests can say how far the reference's own float32
# forward and the HIP forward each sit from the truth.
F64_CASES = tuple(FULL_CASES)          # every full-size case (the six ALT_CASES first had them; the random-init ones followed)
FULL_MAX_SAMPLES = 65536


def randomize_bn(model, seed):
    g = torch.Generator().manual_seed(10_000 + seed)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            c = m.num_features
            m.running_mean.copy_(torch.randn(c, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(c, generator=g) * 0.5 + 0.75)
            m.weight.data.copy_(torch.rand(c, generator=g) * 0.5 + 0.75)
            m.bias.data.copy_(torch.randn(c, generator=g) * 0.1)


def make_input(shape, seed):
    g = torch.Generator().manual_seed(20_000 + seed)
    return torch.rand(shape, generator=g)


def plant_unimodal(model, gain, src_channels=(4, 5, 6, 7)):
    """MS-GCNet only.  Trunk channel 0 becomes a pass-through of the SUM of the input channels `src_channels` (4..7 = the four
    likelihood channels of the MS volume, each ~1 at the matching disparity and ~0 elsewhere, cbmv_generator.py:301-304):
    conv3dbn_1 adds them with centre taps, conv3dbn_2 copies the result with a centre tap, both with an
    identity BN, deconvbn4 contributes nothing to channel 0 so that the additive skip (gcnet_3dcnn.py:123) hands it to deconv5,
    and deconv5 reads channel 0 through a tent filter (0.5, 1, 0.5 per axis = linear x2 up-sampling of the half-resolution
    likelihood) times `gain`.  Every other weight keeps its seeded random value and all 31 other trunk channels still feed
    deconv5, so all 19 convs run on full-mantissa data and add O(5) of structured logit noise under the planted peak."""
    with torch.no_grad():
        for seq, srcs in ((model.conv3dbn_1, src_channels), (model.conv3dbn_2, (0,))):
            conv, bn = seq[0], seq[1]
            conv.weight.data[0].zero_()
            for src in srcs:
                conv.weight.data[0, src, 1, 1, 1] = 1.0
            bn.running_mean[0], bn.running_var[0] = 0.0, 1.0
            bn.weight.data[0], bn.bias.data[0] = 1.0, 0.0
        conv, bn = model.deconvbn4[0], model.deconvbn4[1]
        conv.weight.data[:, 0].zero_()                       # ConvTranspose3d weight is [Ci, Co, 3, 3, 3]
        bn.running_mean[0], bn.running_var[0] = 0.0, 1.0
        bn.weight.data[0], bn.bias.data[0] = 1.0, 0.0
        t = torch.tensor([0.5, 1.0, 0.5])
        model.deconv5.weight.data[0, 0] = float(gain) * t.view(3, 1, 1) * t.view(1, 3, 1) * t.view(1, 1, 3)


def apply_options(model, case):
    """Post-construction tweaks of a case; `peaky` scales deconv5 so the softmax is sharply peaked
    (trained-network-like), the regime where reduced-precision convs fail by pixels."""
    if case.get("unimodal"):
        plant_unimodal(model, case["unimodal"])
    if case.get("peaky") and case["model"] == "gcnet":
        model.deconv5.weight.data.mul_(case["peaky"])
    elif case.get("peaky"):                 # PSMNet: the three classification heads' last conv (psmnet_3dcnn.py:110-122)
        for head in (model.classif1, model.classif2, model.classif3):
            head[2].weight.data.mul_(case["peaky"])


def state_sha256(sd):
    h = hashlib.sha256()
    for k in sorted(sd.keys()):
        v = sd[k]
        h.update(k.encode())
        h.update(np.ascontiguousarray(v.detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def sample(t, max_samples=MAX_SAMPLES):
    """Deterministic strided sample of a tensor (<= max_samples values) + its flat stride."""
    flat = t.detach().reshape(-1)
    stride = max(1, flat.numel() // max_samples)
    return flat[::stride].cpu().numpy().astype(np.float32), stride


def build_case(case, gcnet_cls, psmnet_cls):
    """Construct the seeded model of a case with the given classes (reference's or ours)."""
    torch.manual_seed(case["seed"])
    if case["model"] == "gcnet":
        m = gcnet_cls(case["maxdisp"], is_quarter_input_size=bool(case.get("quarter", False)))
    else:
        m = psmnet_cls(case["maxdisp"])
    randomize_bn(m, case["seed"])
    apply_options(m, case)
    return m.eval()


def out_hw(case):
    n, c, d, h, w = case["in_shape"]
    if case["model"] == "psmnet":
        return 4 * h, 4 * w
    s = 4 if case.get("quarter") else 2
    return s * h, s * w


def full_input(case):
    """Input volume of a FULL_CASES entry: torch.rand, or (ms_volume) the oracle's MS volume of the seeded synthetic pair --
    the caller passes it in because this file must not import oracle/ (the tests do)."""
    return make_input(case["in_shape"], case["seed"])


# ---------------------------------------------------------------------------------------------------------------------
# Round 6: the matching-space volume cases of tests/golden/make_volume_golden.py (the reference's own Python glue,
# cbmv_generator.py:27-79,84-254,258-308, around the oracle's natives).  Three tiny bordered pairs: uniform noise, a shifted
# texture (census / ZSAD minima at a planted disparity), and large constant regions (NCC's non-finite branch, all-equal
# census windows, all-sentinel rows in the left D' columns).
VOLUME_CASES = {
    "random": dict(kind="random", seed=61, hw=(36, 56), ndisp=12, board=10),
    "shifted": dict(kind="shifted", seed=62, hw=(40, 64), ndisp=16, board=10),
    "flat": dict(kind="flat", seed=63, hw=(33, 47), ndisp=8, board=10),
}


def volume_pair(case):
    """-> bordered uint8 images [h + 2b, w + 2b] x 2 of a VOLUME_CASES entry (the fixtures store them too)."""
    h, w = case["hw"]
    nd, b = case["ndisp"], case["board"]
    rng = np.random.default_rng(case["seed"])
    if case["kind"] == "random":
        l = rng.integers(0, 256, (h, w), dtype=np.uint8)
        r = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif case["kind"] == "shifted":
        base = rng.integers(0, 256, (h + 2, w + nd + 2)).astype(np.float32)
        sm = sum(base[dy:dy + h, dx:dx + w + nd] for dy in range(3) for dx in range(3)) / 9.0
        base = np.clip(np.rint(sm), 0, 255).astype(np.uint8)
        shift = np.where(np.arange(h) < h // 2, 5, nd - 3)            # two row bands, two disparities
        l = base[:, :w]
        r = np.stack([base[y, shift[y]: shift[y] + w] for y in range(h)])   # right[x] = left[x + d]
    elif case["kind"] == "flat":
        l = np.full((h, w), 90, np.uint8)
        r = np.full((h, w), 90, np.uint8)
        l[: h // 2, : w // 2] = rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8)
        r[h // 3:, w // 3:] = rng.integers(0, 256, (h - h // 3, w - w // 3), dtype=np.uint8)
        l[-6:, -9:] = 255                                              # a saturated corner next to the zero border
    else:
        raise ValueError(case["kind"])
    pad = lambda a: np.pad(a, ((b, b), (b, b)), "constant").astype(np.uint8).copy(order="C")  # noqa: E731
    return pad(l), pad(r)


def arrays_sha256(arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.shape).encode() + str(a.dtype).encode())
        h.update(a.tobytes())
    return h.digest()
