"""BASELINE.json configs #2, #4 (per-GPU share) and #5 end to end on the GPU at their full sizes, through the C ABI:

  images (bordered uint8 half-res pair) -> msnet_build_volume -> MS-GCNet (HIP) -> disparity

against the oracle's end-to-end result (oracle volume build -> oracle GCNet forward, CPU fp32) on the same seeded
weights (randomised BN statistics, tests/golden/recipes.py) and the same synthetic pairs.

  cfg#2  960x540 -> 960x544, D=192, batch 1               (cbmv_generator.py:780-788, gcnet_3dcnn.py:97-141)
  cfg#4  the same shape at batch 4 per GPU (32 over 8 GPUs): N=4 equals four N=1 forwards bit for bit and
         sample 0 matches the oracle
  cfg#5  KITTI 1242x375 -> 1248x384: bordered 212x644 images, D'=96, volume [8,96,192,624]

Gates: volume cost channels bit-exact, likelihood channels 2e-6 (GPU expf vs glibc expf); disparity 1e-3 abs
(BASELINE.json north_star).  The 8-GPU legs of #4/#5 need an 8-GPU node; what one GPU does in them is what runs here.
"""
import os

import numpy as np
import pytest
import torch

import recipes
from oracle import aggregators as oracle
from oracle import ms_volume as O

pytestmark = pytest.mark.gpu
DISP_TOL = 1e-3
AML_TOL = 2e-6
WEIGHT_SEED = 21


def _classes():
    from msnets_amd.gcnet_3dcnn import GCNet_CostVolumeAggre
    from msnets_amd.psmnet_3dcnn import PSMNet_CostVolumeAggre
    return GCNet_CostVolumeAggre, PSMNet_CostVolumeAggre


def _model(maxdisp=192):
    case = dict(model="gcnet", seed=WEIGHT_SEED, maxdisp=maxdisp, in_shape=None)
    m = recipes.build_case(case, *_classes())
    return m, {k: v.clone() for k, v in m.state_dict().items()}


def _bitexact(a, b, what):
    bad = a.view(np.uint32) != b.view(np.uint32)
    assert not bad.any(), "%s: %d / %d values differ" % (what, int(bad.sum()), bad.size)


def _check_volume(got, ref, tag):
    assert got.shape == ref.shape
    for ch, nm in enumerate(["census", "ncc", "sobel", "sad"]):
        _bitexact(got[ch], ref[ch], "%s cost channel %s" % (tag, nm))
    err = float(np.abs(got[4:] - ref[4:]).max())
    print("%s: likelihood channels max|err| = %.2e" % (tag, err))
    assert err <= AML_TOL


def _oracle_e2e(Hh, Wh, nd, seed, sd):
    """-> (left, right, oracle volume, oracle disparity) for one synthetic pair."""
    from msnets_amd import synthetic
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    left, right, _ = synthetic.stereo_pair(Hh, Wh, nd, seed=seed)
    vol = O.build_ms_volume(left, right, nd)
    with torch.no_grad():
        disp = oracle.gcnet_forward(sd, torch.from_numpy(vol).unsqueeze(0), 2 * nd)
    return left, right, vol, disp


@pytest.fixture(scope="module")
def cfg2_oracle():
    _, sd = _model()
    return _oracle_e2e(272, 480, 96, 0, sd)


def test_cfg2_end_to_end_images_to_disparity(gpu, cfg2_oracle):
    """Config #2: two 292x500 bordered images -> VolumeBuilder -> GCNet_CostVolumeAggre, all on the device, vs the
    oracle's end-to-end disparity."""
    from msnets_amd import cbmv_generator as cg
    left, right, vol_ref, disp_ref = cfg2_oracle
    model, _ = _model()
    model = model.cuda()
    builder = cg.VolumeBuilder(292, 500, 96, "cuda")
    vol = builder(torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda())
    _check_volume(vol.cpu().numpy(), vol_ref, "cfg2")
    disp = model(vol.unsqueeze(0)).cpu()
    assert disp.shape == disp_ref.shape == (1, 544, 960)
    err = float((disp - disp_ref).abs().max())
    print("cfg2 end to end: max|disp - oracle| = %.3e, range %.2f..%.2f" % (err, float(disp_ref.min()), float(disp_ref.max())))
    assert err <= DISP_TOL
    # the driver's crop (main_msnet.py:585-589) gives the 540-row map
    from msnets_amd import driver_utils
    assert driver_utils.crop_disparity(disp.numpy(), 544, 960, 540, 960).shape == (540, 960)


def test_cfg4_batch4_per_gpu(gpu, cfg2_oracle):
    """Config #4's per-GPU share: batch 4 at [4,8,96,272,480].  The batched forward must equal four single forwards bit
    for bit (samples are independent: eval-mode BN, main_msnet.py:534) and sample 0 must match the oracle."""
    from msnets_amd import cbmv_generator as cg, synthetic
    left0, right0, _, disp_ref = cfg2_oracle
    model, _ = _model()
    model = model.cuda()
    builder = cg.VolumeBuilder(292, 500, 96, "cuda")
    vol = torch.empty((4, 8, 96, 272, 480), device="cuda", dtype=torch.float32)
    for b in range(4):
        l, r = (left0, right0) if b == 0 else synthetic.stereo_pair(272, 480, 96, seed=b)[:2]
        builder(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda(), out=vol[b])
    batched = model(vol)
    assert batched.shape == (4, 544, 960)
    for b in range(4):
        single = model(vol[b:b + 1])
        assert torch.equal(single[0], batched[b]), "sample %d: batched forward differs from the single forward" % b
    assert float((batched[1] - batched[0]).abs().max()) > 1.0          # the samples really are different pairs
    err = float((batched[0].cpu() - disp_ref[0]).abs().max())
    print("cfg4 batch 4: sample 0 max|disp - oracle| = %.3e" % err)
    assert err <= DISP_TOL


def test_cfg5_kitti_shape(gpu):
    """Config #5: 1242x375 padded to 1248x384 (cbmv_generator.py:780-788) -> bordered 212x644 images, D'=96 ->
    volume [8,96,192,624] -> disparity [384,1248]; plus the driver's crop back to 375x1242."""
    from msnets_amd import cbmv_generator as cg, driver_utils
    model, sd = _model()
    left, right, vol_ref, disp_ref = _oracle_e2e(192, 624, 96, 5, sd)
    assert left.shape == (212, 644)
    vol = cg.build_ms_volume(torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda(), 96)
    assert tuple(vol.shape) == (8, 96, 192, 624)
    _check_volume(vol.cpu().numpy(), vol_ref, "cfg5")
    model = model.cuda()
    disp = model(vol.unsqueeze(0)).cpu()
    assert disp.shape == disp_ref.shape == (1, 384, 1248)
    err = float((disp - disp_ref).abs().max())
    print("cfg5 end to end: max|disp - oracle| = %.3e" % err)
    assert err <= DISP_TOL
    # batch 2 per GPU (16 over 8 GPUs) == two single forwards
    two = model(torch.stack([vol, vol.flip(-1).contiguous()]))
    assert torch.equal(two[0], disp[0].cuda())
    assert torch.equal(two[1], model(vol.flip(-1).contiguous().unsqueeze(0))[0])
    assert driver_utils.crop_disparity(disp.numpy(), 384, 1248, 375, 1242).shape == (375, 1242)


def test_cfg1_plumbing_shape(gpu):
    """Config #1 (BASELINE.json configs[0], the reference's own CPU-runnable case): 256x512, D=64 -> bordered 148x276 images,
    volume [8,32,128,256], disparity [256,512]; end to end vs the oracle."""
    from msnets_amd import cbmv_generator as cg
    model, sd = _model(64)
    left, right, vol_ref, disp_ref = _oracle_e2e(128, 256, 32, 7, sd)
    assert left.shape == (148, 276)
    vol = cg.build_ms_volume(torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda(), 32)
    _check_volume(vol.cpu().numpy(), vol_ref, "cfg1")
    disp = model.cuda()(vol.unsqueeze(0)).cpu()
    assert disp.shape == disp_ref.shape == (1, 256, 512)
    err = float((disp - disp_ref).abs().max())
    print("cfg1 end to end: max|disp - oracle| = %.3e" % err)
    assert err <= DISP_TOL


def test_power_of_two_scaling_is_exact_at_full_size(gpu):
    """A size-independent property of the split-fp16 conv at the full benchmark shape (no oracle needed): with zero shift,
    scaling the input by a power of two scales the output by exactly that power of two -- hi = fp16(x) and
    lo = fp16((x - hi) * 2^11) both scale exactly while nothing leaves the fp16 normal range, and fp32 accumulation of exactly
    scaled terms is exactly scaled.  Checked bit for bit on conv3dbn_2's shape (32->32, 96x272x480, sliding-window kernel), on
    the stride-2 32->64 layer and on the 64->32 transposed layer."""
    from msnets_amd import hipops
    g = torch.Generator().manual_seed(5)
    for kind, ci, co, stride, dims in (("conv", 32, 32, 1, (96, 272, 480)), ("conv", 32, 64, 2, (96, 272, 480)),
                                       ("deconv", 64, 32, 2, (48, 136, 240))):
        d, h, w = dims
        x = torch.rand((1, d, h, w, ci), generator=g).cuda()
        if kind == "conv":
            wt = (torch.randn((co, ci, 3, 3, 3), generator=g) * 0.05).cuda()
            wpk = hipops.pack_conv_weight(wt, f16s=True, stride=stride)
            f = lambda t: hipops.conv3d_k3(t, wpk, None, None, co, stride=stride, relu=True, f16s=True)     # noqa: E731
        else:
            wt = (torch.randn((ci, co, 3, 3, 3), generator=g) * 0.05).cuda()
            wpk = hipops.pack_conv_weight(wt, transposed=True, f16s=True)
            f = lambda t: hipops.deconv3d_k3s2(t, wpk, None, None, co, relu=True, f16s=True)               # noqa: E731
        y1 = f(x).clone()
        for k in (2.0, 8.0, 64.0):        # (downwards the property ends where hi = fp16(x) turns subnormal: torch.rand has such values)
            assert torch.equal(f(x * k), y1 * k), (kind, ci, co, stride, k)
        assert float(y1.abs().max()) > 0
