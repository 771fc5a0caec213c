"""GPU parity of the aggregator kernels, called through the C ABI (ctypes), against
 (a) plain torch fp32 CPU ops for single layers,
 (b) the golden vectors the reference produced (tests/golden/aggregators_*.npz),
 (c) the oracle restatement on every intermediate activation.
Tolerance on disparity: 1e-3 abs (BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import recipes
from oracle import aggregators as oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DISP_TOL = 1e-3


def _cl(x):      # NCDHW cpu -> NDHWC gpu
    return x.permute(0, 2, 3, 4, 1).contiguous().cuda()


def _nc(y):      # NDHWC gpu -> NCDHW cpu
    return y.cpu().permute(0, 4, 1, 2, 3).contiguous()


def _rel(a, b):
    return float((a - b).abs().max() / max(1e-6, float(b.abs().max())))


def test_layout_roundtrip(gpu):
    from msnets_amd import hipops
    for shape in [(1, 8, 5, 7, 33), (2, 32, 3, 4, 70), (1, 64, 2, 9, 31), (1, 1, 4, 4, 4)]:
        x = torch.rand(shape)
        y = hipops.ncdhw_to_ndhwc(x.cuda())
        assert torch.equal(y.cpu(), x.permute(0, 2, 3, 4, 1).contiguous()), shape
        assert torch.equal(hipops.ndhwc_to_ncdhw(y).cpu(), x), shape


CONV_CASES = [
    # Ci, Co, stride, (N,D,H,W), relu, residual
    (8, 32, 1, (1, 6, 10, 37), True, False),
    (8, 32, 1, (2, 4, 8, 32), False, True),
    (32, 32, 1, (1, 4, 9, 40), True, False),
    (32, 32, 1, (1, 3, 8, 32), True, True),
    (32, 64, 1, (1, 5, 6, 19), True, False),
    (64, 64, 1, (1, 4, 10, 24), True, True),
    (64, 32, 1, (1, 4, 8, 48), True, False),
    (128, 128, 1, (1, 3, 5, 9), True, False),
    (16, 32, 1, (1, 3, 5, 20), False, False),
    (32, 64, 2, (1, 8, 12, 34), True, False),
    (64, 64, 2, (1, 6, 9, 33), True, False),
    (64, 128, 2, (1, 4, 6, 10), True, False),
    (32, 32, 2, (2, 5, 7, 21), False, False),
]


F16S_CASES = [
    (8, 32, (1, 6, 10, 37), True, False),
    (8, 32, (2, 4, 8, 32), False, True),
    (8, 64, (1, 3, 5, 33), True, False),
    (16, 32, (1, 5, 9, 37), True, False),
    (16, 32, (2, 4, 8, 32), False, True),
    (16, 64, (1, 3, 6, 33), True, True),
    (32, 32, (1, 4, 9, 40), True, False),
    (32, 32, (2, 3, 8, 16), True, True),
    (32, 64, (1, 5, 6, 19), True, False),
    (64, 64, (1, 4, 10, 24), True, True),
    (64, 32, (1, 4, 8, 48), False, False),
    (128, 64, (1, 3, 5, 9), True, False),
    (128, 128, (1, 3, 5, 9), True, False),
    (64, 128, (1, 4, 6, 33), True, True),
]


# (the last two: output widths 16 mod 32 with rows in fours -> the 2x4x16 tile of round 4; 8 x 16 x 96 also has depth / row edge tiles)
F16S_S2_CASES = [(32, 64, (1, 8, 12, 34), True, False), (64, 64, (1, 6, 9, 33), True, True), (64, 128, (1, 4, 6, 10), True, False),
                 (32, 64, (1, 7, 16, 96), True, False), (64, 64, (2, 4, 8, 32), True, True),
                 (32, 64, (2, 5, 7, 70), False, False)]


@pytest.mark.parametrize("ci,co,dims,relu,use_res", F16S_S2_CASES)
@pytest.mark.parametrize("direct", ["0", "1"])
def test_conv3d_stride2_split_fp16(gpu, hiplib, ci, co, dims, relu, use_res, direct, monkeypatch):
    from msnets_amd import hipops
    assert hiplib.msnet_conv3d_k3_f16s_supported(ci, co, 2) == 1
    monkeypatch.setenv("MSNET_DIRECT", direct)      # "0": tiled persistent kernel, "1": direct small-layer kernel
    g = torch.Generator().manual_seed(ci * 7 + co)
    n, d, h, w = dims
    x = torch.randn((n, ci, d, h, w), generator=g) * 3
    wt = torch.randn((co, ci, 3, 3, 3), generator=g) * (2.0 / (27 * ci)) ** 0.5
    scale = torch.rand(co, generator=g) + 0.5
    shift = torch.randn(co, generator=g) * 0.1
    ref = F.conv3d(x.double(), wt.double(), None, stride=2, padding=1) * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if use_res:
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    wpk = hipops.pack_conv_weight(wt.cuda(), f16s=True, stride=2)
    y = hipops.conv3d_k3(_cl(x), wpk, scale.cuda(), shift.cuda(), co, stride=2, relu=relu,
                         residual=_cl(res) if use_res else None, f16s=True)
    assert tuple(_nc(y).shape) == tuple(ref.shape)
    err = _rel(_nc(y).double(), ref)
    print("split-fp16 s2 %d->%d rel err %.2e" % (ci, co, err))
    assert err < 5e-6


@pytest.mark.parametrize("ci,co,dims,relu,use_res", F16S_CASES)
@pytest.mark.parametrize("direct", ["0", "1"])
def test_conv3d_layer_split_fp16(gpu, hiplib, ci, co, dims, relu, use_res, direct, monkeypatch):
    """Split-fp16 MFMA path: operands carry 22 bits, so a single layer agrees with the fp64 conv to ~1e-6
    relative (plain fp16 operands would be ~5e-4)."""
    from msnets_amd import hipops
    assert hiplib.msnet_conv3d_k3_f16s_supported(ci, co, 1) == 1
    monkeypatch.setenv("MSNET_DIRECT", direct)      # "0": tiled persistent kernel, "1": direct small-layer kernel
    g = torch.Generator().manual_seed(ci * 31 + co)
    n, d, h, w = dims
    x = torch.randn((n, ci, d, h, w), generator=g) * 3
    wt = torch.randn((co, ci, 3, 3, 3), generator=g) * (2.0 / (27 * ci)) ** 0.5
    scale = torch.rand(co, generator=g) + 0.5
    shift = torch.randn(co, generator=g) * 0.1
    ref = F.conv3d(x.double(), wt.double(), None, padding=1) * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if use_res:
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    wpk = hipops.pack_conv_weight(wt.cuda(), f16s=True)
    y = hipops.conv3d_k3(_cl(x), wpk, scale.cuda(), shift.cuda(), co, relu=relu, residual=_cl(res) if use_res else None,
                         f16s=True)
    err = _rel(_nc(y).double(), ref)
    print("split-fp16 %d->%d rel err %.2e" % (ci, co, err))
    assert err < 5e-6


@pytest.mark.parametrize("co,dims", [(32, (1, 6, 10, 37)), (32, (2, 4, 8, 32)), (64, (1, 3, 5, 33)), (32, (1, 5, 13, 70))])
def test_first_layer_reads_ncdhw(gpu, hiplib, co, dims, monkeypatch):
    """msnet_conv3d_k3_c8_ncdhw_f16s: the first layer straight from the NCDHW volume (no layout-conversion pass).  Same tiles,
    same MFMA order as the NDHWC first-layer kernel => bit-identical to layout conversion + msnet_conv3d_k3_f16s; and within
    5e-6 of the fp64 conv.  An out-of-range input voxel raises the INPUT bit of the overflow word."""
    from msnets_amd import hipops
    monkeypatch.setenv("MSNET_DIRECT", "0")
    g = torch.Generator().manual_seed(co + dims[3])
    n, d, h, w = dims
    x = torch.randn((n, 8, d, h, w), generator=g) * 3
    wt = torch.randn((co, 8, 3, 3, 3), generator=g) * (2.0 / (27 * 8)) ** 0.5
    scale = torch.rand(co, generator=g) + 0.5
    shift = torch.randn(co, generator=g) * 0.1
    ref = F.relu(F.conv3d(x.double(), wt.double(), None, padding=1) * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1))
    wpk = hipops.pack_conv_weight(wt.cuda(), f16s=True)
    y = hipops.conv3d_c8_ncdhw(x.cuda(), wpk, scale.cuda(), shift.cuda(), co, relu=True)
    y2 = hipops.conv3d_k3(hipops.ncdhw_to_ndhwc(x.cuda()), wpk, scale.cuda(), shift.cuda(), co, relu=True, f16s=True)
    assert torch.equal(y, y2)
    assert _rel(_nc(y).double(), ref) < 5e-6
    guard = hipops.RangeGuard(torch.device("cuda"))
    with guard:
        hipops.conv3d_c8_ncdhw(x.cuda(), wpk, scale.cuda(), shift.cuda(), co, relu=True)
    assert guard.word() == 0
    xb = x.clone()
    xb[n - 1, 5, d - 1, h - 1, w - 1] = 7.0e4
    with guard:
        hipops.conv3d_c8_ncdhw(xb.cuda(), wpk, scale.cuda(), shift.cuda(), co, relu=True)
    assert guard.word() & hipops.RangeGuard.INPUT


WD_CASES = [((1, 9, 18, 70), True, False), ((2, 8, 16, 64), True, True), ((1, 3, 5, 33), False, False), ((1, 12, 7, 37), True, True),
            ((1, 2, 4, 32), True, False), ((1, 17, 21, 96), True, False)]


@pytest.mark.parametrize("dims,relu,use_res", WD_CASES)
def test_conv3d_layer_winograd_depth(gpu, hiplib, dims, relu, use_res, monkeypatch):
    """msnet_conv3d_k3_wd_f16s (Winograd F(2,3) along depth, 32 -> 32 stride 1): against the fp64 conv at the split-fp16 gate
    (5e-6 of the layer's magnitude) and against the direct split-fp16 kernel (not bit-identical: different summation; both are
    within the gate of fp64).  Ragged shapes: odd depths, heights that are not multiples of 4, widths that are not multiples of 32."""
    from msnets_amd import hipops
    monkeypatch.setenv("MSNET_DIRECT", "0")
    g = torch.Generator().manual_seed(sum(dims))
    n, d, h, w = dims
    x = torch.relu(torch.randn((n, 32, d, h, w), generator=g)) * 3
    wt = torch.randn((32, 32, 3, 3, 3), generator=g) * (2.0 / (27 * 32)) ** 0.5
    scale = torch.rand(32, generator=g) + 0.5
    shift = torch.randn(32, generator=g) * 0.1
    ref = F.conv3d(x.double(), wt.double(), None, padding=1) * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if use_res:
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    assert hiplib.msnet_conv3d_k3_wd_f16s_supported(d, h, w, 32, 32, 1) == 1
    assert hiplib.msnet_conv3d_k3_wd_f16s_supported(d, h, w, 32, 64, 1) == 0 and hiplib.msnet_conv3d_k3_wd_f16s_supported(d, h, w, 32, 32, 2) == 0
    wpk_wd = hipops.winograd_depth_weights(wt.cuda())
    y = hipops.conv3d_k3_winograd_depth(_cl(x), wpk_wd, scale.cuda(), shift.cuda(), relu=relu, residual=_cl(res) if use_res else None)
    err = _rel(_nc(y).double(), ref)
    wpk = hipops.pack_conv_weight(wt.cuda(), f16s=True)
    yd = hipops.conv3d_k3(_cl(x), wpk, scale.cuda(), shift.cuda(), 32, relu=relu, residual=_cl(res) if use_res else None, f16s=True)
    errd = _rel(_nc(yd).double(), ref)
    print("winograd-depth 32->32 %s: rel err %.2e (direct split-fp16 kernel %.2e)" % (dims, err, errd))
    assert err < 5e-6
    # and through conv3d_k3's dispatch (the path the modules take)
    y2 = hipops.conv3d_k3(_cl(x), wpk, scale.cuda(), shift.cuda(), 32, relu=relu, residual=_cl(res) if use_res else None, f16s=True,
                          wpk_wd=wpk_wd)
    assert torch.equal(y2, y)


def test_winograd_depth_full_size_is_deterministic(gpu, hiplib):
    """conv3dbn_2's shape (96 x 272 x 480, 32 -> 32): the persistent kernel's loader / MFMA wave hand-offs (q planes rewritten under
    the running tile, weight groups double-buffered, drained stores of the previous tile) leave no run-to-run freedom -- three
    launches on the same input are bit-identical -- and the result agrees with the direct split-fp16 kernel to the layer gate."""
    from msnets_amd import hipops
    g = torch.Generator().manual_seed(77)
    d, h, w = 96, 272, 480
    x = torch.relu(torch.randn((1, d, h, w, 32), generator=g)).cuda()             # NDHWC as the kernels take it
    wt = torch.randn((32, 32, 3, 3, 3), generator=g) * (2.0 / (27 * 32)) ** 0.5
    scale, shift = (torch.rand(32, generator=g) + 0.5).cuda(), (torch.randn(32, generator=g) * 0.1).cuda()
    assert hiplib.msnet_conv3d_k3_wd_f16s_supported(d, h, w, 32, 32, 1) == 1
    wpk_wd = hipops.winograd_depth_weights(wt.cuda())
    y0 = hipops.conv3d_k3_winograd_depth(x, wpk_wd, scale, shift, relu=True)
    for _ in range(2):
        y = hipops.conv3d_k3_winograd_depth(x, wpk_wd, scale, shift, relu=True)
        assert torch.equal(y, y0)
        del y
    wpk = hipops.pack_conv_weight(wt.cuda(), f16s=True)
    yd = hipops.conv3d_k3(x, wpk, scale, shift, 32, relu=True, f16s=True)
    err = float((yd - y0).abs().max() / yd.abs().max())
    print("winograd-depth vs direct split-fp16 at 96x272x480: max rel diff %.2e" % err)
    assert err < 5e-6


@pytest.mark.parametrize("seed", range(8))
def test_conv_kernels_random_ragged_shapes(gpu, hiplib, monkeypatch, seed):
    """Random small, ragged volumes (odd depths, widths that are not multiples of 16, single rows) through every tiled
    split-fp16 kernel family and the direct kernel: stride 1 and 2, transposed, with and without residual."""
    from msnets_amd import hipops
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(1, 3)); d = int(rng.integers(1, 8)); h = int(rng.integers(1, 14)); w = int(rng.integers(1, 71))
    ci, co = [(32, 32), (32, 64), (64, 64), (64, 32), (64, 128), (128, 128), (8, 32), (32, 32)][seed]
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((n, ci, d, h, w), generator=g) * 2
    for direct in ("0", "1"):
        monkeypatch.setenv("MSNET_DIRECT", direct)
        for stride in (1, 2):
            if not hiplib.msnet_conv3d_k3_f16s_supported(ci, co, stride):
                continue
            wt = torch.randn((co, ci, 3, 3, 3), generator=g) * (2.0 / (27 * ci)) ** 0.5
            shift = torch.randn(co, generator=g) * 0.1
            ref = F.conv3d(x.double(), wt.double(), None, stride=stride, padding=1) + shift.double().view(1, -1, 1, 1, 1)
            res = torch.randn(ref.shape, generator=g)
            ref = F.relu(ref + res.double())
            wpk = hipops.pack_conv_weight(wt.cuda(), f16s=True, stride=stride)
            y = hipops.conv3d_k3(_cl(x), wpk, None, shift.cuda(), co, stride=stride, relu=True, residual=_cl(res), f16s=True)
            assert _rel(_nc(y).double(), ref) < 5e-6, (ci, co, stride, direct, (n, d, h, w))
        if hiplib.msnet_deconv3d_k3s2_f16s_supported(ci, co):
            wt = torch.randn((ci, co, 3, 3, 3), generator=g) * (2.0 / (27 * ci)) ** 0.5
            ref = F.conv_transpose3d(x.double(), wt.double(), None, stride=2, padding=1, output_padding=1)
            wpk = hipops.pack_conv_weight(wt.cuda(), transposed=True, f16s=True)
            y = hipops.deconv3d_k3s2(_cl(x), wpk, None, None, co, relu=False, residual=None, f16s=True)
            assert _rel(_nc(y).double(), ref) < 5e-6, (ci, co, "deconv", direct, (n, d, h, w))


@pytest.mark.parametrize("dims,seg", [((1, 12, 9, 70), 1), ((1, 12, 9, 70), 3), ((2, 8, 16, 33), 2), ((1, 16, 5, 32), 4),
                                      ((1, 6, 4, 32), 1)])
@pytest.mark.parametrize("use_res", [False, True])
def test_conv3d_sliding_window_kernel(gpu, hiplib, monkeypatch, dims, seg, use_res):
    """The sliding-window variant of the 32->32 kernel (plane slots rotate along d) is normally chosen only for large
    volumes; MSNET_FORCE_SLIDE_SEG forces it with a given number of segments per tile column.  Same bar as the plain kernel,
    and bit-identical to it (same MFMA sequence per output)."""
    from msnets_amd import hipops
    g = torch.Generator().manual_seed(sum(dims) + seg)
    n, d, h, w = dims
    x = torch.randn((n, 32, d, h, w), generator=g) * 3
    wt = torch.randn((32, 32, 3, 3, 3), generator=g) * (2.0 / (27 * 32)) ** 0.5
    shift = torch.randn(32, generator=g) * 0.1
    ref = F.conv3d(x.double(), wt.double(), None, padding=1) + shift.double().view(1, -1, 1, 1, 1)
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if use_res:
        ref = ref + res.double()
    ref = F.relu(ref)
    wpk = hipops.pack_conv_weight(wt.cuda(), f16s=True)
    run = lambda: hipops.conv3d_k3(_cl(x), wpk, None, shift.cuda(), 32, relu=True, residual=_cl(res) if use_res else None, f16s=True)
    monkeypatch.setenv("MSNET_FORCE_SLIDE_SEG", "0")
    plain = run()
    monkeypatch.setenv("MSNET_FORCE_SLIDE_SEG", str(seg))
    slid = run()
    torch.cuda.synchronize()
    assert _rel(_nc(slid).double(), ref) < 5e-6
    assert torch.equal(plain, slid)


@pytest.mark.parametrize("ci,co,stride,dims,relu,use_res", CONV_CASES)
def test_conv3d_layer(gpu, ci, co, stride, dims, relu, use_res):
    from msnets_amd import hipops
    g = torch.Generator().manual_seed(ci * 1000 + co + stride)
    n, d, h, w = dims
    x = torch.randn((n, ci, d, h, w), generator=g)
    wt = torch.randn((co, ci, 3, 3, 3), generator=g) * (2.0 / (27 * ci)) ** 0.5
    scale = torch.rand(co, generator=g) + 0.5
    shift = torch.randn(co, generator=g) * 0.1
    ref = F.conv3d(x, wt, None, stride=stride, padding=1) * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1)
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if use_res:
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    wpk = hipops.pack_conv_weight(wt.cuda())
    y = hipops.conv3d_k3(_cl(x), wpk, scale.cuda(), shift.cuda(), co, stride=stride, relu=relu,
                         residual=_cl(res) if use_res else None)
    assert tuple(_nc(y).shape) == tuple(ref.shape)
    assert _rel(_nc(y), ref) < 2e-5


DECONV_CASES = [
    (128, 64, (1, 3, 5, 9), True, True),
    (64, 64, (1, 4, 6, 20), True, True),
    (64, 64, (1, 3, 5, 33), True, False),
    (64, 32, (1, 4, 9, 17), True, True),
    (64, 32, (2, 3, 4, 32), False, True),
    (32, 32, (1, 3, 6, 18), False, False),
    (32, 64, (1, 2, 5, 35), True, False),
]


@pytest.mark.parametrize("ci,co,dims,relu,use_res", DECONV_CASES)
def test_deconv3d_layer(gpu, ci, co, dims, relu, use_res):
    from msnets_amd import hipops
    g = torch.Generator().manual_seed(ci * 77 + co)
    n, d, h, w = dims
    x = torch.randn((n, ci, d, h, w), generator=g)
    wt = torch.randn((ci, co, 3, 3, 3), generator=g) * (2.0 / (27 * ci)) ** 0.5
    scale = torch.rand(co, generator=g) + 0.5
    shift = torch.randn(co, generator=g) * 0.1
    ref = F.conv_transpose3d(x, wt, None, stride=2, padding=1, output_padding=1)
    ref = ref * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1)
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if use_res:
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    wpk = hipops.pack_conv_weight(wt.cuda(), transposed=True)
    y = hipops.deconv3d_k3s2(_cl(x), wpk, scale.cuda(), shift.cuda(), co, relu=relu, residual=_cl(res) if use_res else None)
    assert tuple(_nc(y).shape) == tuple(ref.shape)
    assert _rel(_nc(y), ref) < 2e-5


@pytest.mark.parametrize("ci,co,dims,relu,use_res", [(64, 32, (1, 4, 9, 17), True, True), (64, 32, (2, 3, 4, 32), False, True),
                                                   (64, 64, (1, 4, 6, 20), True, True), (64, 64, (1, 3, 5, 33), True, False),
                                                   (128, 64, (1, 3, 5, 14), True, True), (32, 32, (2, 2, 3, 9), False, False),
                                                   (128, 96, (1, 2, 4, 7), True, False)])
@pytest.mark.parametrize("direct", ["0", "1"])
def test_deconv3d_layer_split_fp16(gpu, hiplib, ci, co, dims, relu, use_res, direct, monkeypatch):
    from msnets_amd import hipops
    assert hiplib.msnet_deconv3d_k3s2_f16s_supported(ci, co) == 1
    monkeypatch.setenv("MSNET_DIRECT", direct)      # "0": tiled persistent kernel, "1": direct small-layer kernel
    g = torch.Generator().manual_seed(ci * 13 + co)
    n, d, h, w = dims
    x = torch.randn((n, ci, d, h, w), generator=g) * 3
    wt = torch.randn((ci, co, 3, 3, 3), generator=g) * (2.0 / (27 * ci)) ** 0.5
    scale = torch.rand(co, generator=g) + 0.5
    shift = torch.randn(co, generator=g) * 0.1
    ref = F.conv_transpose3d(x.double(), wt.double(), None, stride=2, padding=1, output_padding=1)
    ref = ref * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if use_res:
        ref = ref + res.double()
    if relu:
        ref = F.relu(ref)
    wpk = hipops.pack_conv_weight(wt.cuda(), transposed=True, f16s=True)
    y = hipops.deconv3d_k3s2(_cl(x), wpk, scale.cuda(), shift.cuda(), co, relu=relu, residual=_cl(res) if use_res else None,
                             f16s=True)
    err = _rel(_nc(y).double(), ref)
    print("split-fp16 deconv %d->%d rel err %.2e" % (ci, co, err))
    assert err < 5e-6


# (round 4: the kernel walks depth SEGMENTS -- 13, 20 and 48 slices give uneven last segments and 2 .. 8 segments per tile)
@pytest.mark.parametrize("dims,use_add", [((1, 5, 9, 33), False), ((2, 4, 8, 40), True), ((1, 1, 3, 5), False),
                                          ((1, 13, 7, 31), True), ((2, 20, 6, 30), False), ((1, 48, 12, 61), True)])
def test_conv3d_cout1_head(gpu, dims, use_add):
    from msnets_amd import hipops
    g = torch.Generator().manual_seed(5)
    n, d, h, w = dims
    x = torch.randn((n, 32, d, h, w), generator=g)
    wt = torch.randn((1, 32, 3, 3, 3), generator=g) * 0.05
    ref = F.conv3d(x, wt, None, padding=1).squeeze(1)
    add = torch.randn(ref.shape, generator=g) if use_add else None
    if use_add:
        ref = ref + add
    y = hipops.conv3d_k3_cout1(_cl(x), wt.cuda(), add.cuda() if use_add else None)
    assert _rel(y.cpu(), ref) < 2e-5


@pytest.mark.parametrize("dims,stride", [((1, 4, 9, 35), 2), ((2, 3, 8, 32), 2), ((1, 3, 4, 6), 4)])
def test_deconv5_logits_and_fused_tail(gpu, dims, stride):
    from msnets_amd import hipops
    g = torch.Generator().manual_seed(9)
    n, d, h, w = dims
    x = torch.randn((n, 32, d, h, w), generator=g)
    wt = torch.randn((32, 1, 3, 3, 3), generator=g) * 0.2
    bias = 0.37
    ref = F.conv_transpose3d(x, wt, torch.tensor([bias]), stride=stride, padding=1, output_padding=stride - 1).squeeze(1)
    logits = hipops.deconv3d_cout1(_cl(x), wt.cuda(), bias, stride=stride)
    assert tuple(logits.shape) == tuple(ref.shape)
    assert _rel(logits.cpu(), ref) < 2e-5
    ref_disp = oracle.soft_argmin(ref)
    assert float((hipops.softargmin(logits).cpu() - ref_disp).abs().max()) < 1e-4
    if stride == 2:
        fused = hipops.deconv5_softargmin(_cl(x), wt.cuda(), bias)
        assert float((fused.cpu() - ref_disp).abs().max()) < 1e-4


@pytest.mark.parametrize("dims,gain", [((1, 20, 9, 40), 0.2), ((2, 32, 16, 33), 0.2), ((1, 48, 7, 31), 3.0), ((1, 5, 8, 8), 0.2)])
def test_fused_tail_depth_segments(gpu, monkeypatch, dims, gain):
    """msnet_deconv5_softargmin_ws cuts a tile's D slices into depth segments whose online-softmax states meet in a merge pass.
    Every segment count -- 1 (the single chain), ragged (20 slices in 3 runs of 7, 7, 6), one slice per run -- must give the
    oracle's disparity at the tolerance of the single chain, broad and peaky softmax alike; the segment boundary (the carry of
    slice p0 - 1 into logit 2 p0 - 1) is where an off-by-one would show.  And the count depends on the sample's shape only: a
    batch returns the bits of its samples' single forwards."""
    from msnets_amd import hipops
    g = torch.Generator().manual_seed(19)
    n, d, h, w = dims
    x = torch.randn((n, 32, d, h, w), generator=g)
    wt = torch.randn((32, 1, 3, 3, 3), generator=g) * gain
    bias = -0.21
    ref = oracle.soft_argmin(F.conv_transpose3d(x, wt, torch.tensor([bias]), stride=2, padding=1, output_padding=1).squeeze(1))
    xcl, wg = _cl(x), wt.cuda()
    outs = {}
    for segs in (1, 2, 3, 4, d):
        monkeypatch.setenv("MSNET_TAIL_SEGS", str(segs))
        outs[segs] = hipops.deconv5_softargmin(xcl, wg, bias).cpu()
        err = float((outs[segs] - ref).abs().max())
        print("tail segments %d (D'=%d): max |disp - oracle| %.2e" % (segs, d, err))
        assert err < (1e-4 if gain < 1 else 5e-4), (segs, err)
    assert float((outs[3] - outs[1]).abs().max()) < (2e-5 if gain < 1 else 2e-4)
    monkeypatch.setenv("MSNET_TAIL_SEGS", "3")
    if n > 1:
        for i in range(n):
            assert torch.equal(hipops.deconv5_softargmin(xcl[i:i + 1].contiguous(), wg, bias).cpu()[0], outs[3][i])
    monkeypatch.delenv("MSNET_TAIL_SEGS")
    auto = hipops.deconv5_softargmin(xcl, wg, bias).cpu()          # the library's own choice for this shape
    assert float((auto - ref).abs().max()) < (1e-4 if gain < 1 else 5e-4)


def test_softargmin_peaky_and_flat(gpu):
    from msnets_amd import hipops
    g = torch.Generator().manual_seed(3)
    logits = torch.randn((2, 192, 5, 7), generator=g) * 30.0     # peaky
    assert float((hipops.softargmin(logits.cuda()).cpu() - oracle.soft_argmin(logits)).abs().max()) < 1e-4
    flat = torch.zeros((1, 64, 3, 3))
    assert float((hipops.softargmin(flat.cuda()).cpu() - 31.5).abs().max()) < 1e-4


@pytest.mark.parametrize("in_dhw,out_dhw", [((8, 16, 16), (32, 64, 64)), ((8, 8, 24), (32, 32, 96)), ((5, 7, 9), (17, 30, 33)),
                                            ((48, 17, 30), (192, 68, 120))])
def test_trilinear_softargmin(gpu, in_dhw, out_dhw):
    from msnets_amd import hipops
    g = torch.Generator().manual_seed(11)
    cost = torch.randn((2, 1) + in_dhw, generator=g) * 4
    ref = oracle.soft_argmin(F.interpolate(cost, list(out_dhw), mode="trilinear", align_corners=True).squeeze(1))
    got = hipops.trilinear_softargmin(cost.squeeze(1).cuda(), out_dhw)
    assert float((got.cpu() - ref).abs().max()) < 5e-4   # fp32 noise on disparities ~100 at D=192; gate is 1e-3


def _our_classes():
    from msnets_amd.gcnet_3dcnn import GCNet_CostVolumeAggre
    from msnets_amd.psmnet_3dcnn import PSMNet_CostVolumeAggre
    return GCNet_CostVolumeAggre, PSMNet_CostVolumeAggre


@pytest.fixture(params=["split-fp16", "fp32"])
def precision(request):
    from msnets_amd import hipops
    old = hipops.get_default_precision()
    hipops.set_default_precision(request.param)
    yield request.param
    hipops.set_default_precision(old)


@pytest.mark.parametrize("name", sorted(recipes.AGG_CASES))
def test_golden_end_to_end(gpu, name, precision):
    """HIP module vs the reference's own output on the same seeded weights and input, for both conv precisions."""
    case = recipes.AGG_CASES[name]
    gold = np.load(os.path.join(GOLD, "aggregators_%s.npz" % name))
    model = recipes.build_case(case, *_our_classes())
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    assert recipes.state_sha256(sd) == str(gold["state_sha256"])
    x = recipes.make_input(case["in_shape"], case["seed"])
    model = model.cuda()
    disp = model(x.cuda()).cpu()            # fused-tail path
    err = float(np.abs(disp.numpy() - gold["disp"]).max())
    print("%s [%s]: max|disp - reference| = %.3e" % (name, precision, err))
    assert err <= DISP_TOL
    # every intermediate activation against the oracle (un-fused tail path, exercises msnet_softargmin too)
    taps_hip, taps_or = {}, {}
    disp2 = model(x.cuda(), taps=taps_hip).cpu()
    assert float(np.abs(disp2.numpy() - gold["disp"]).max()) <= DISP_TOL
    with torch.no_grad():
        if case["model"] == "gcnet":
            oracle.gcnet_forward(sd, x, case["maxdisp"], bool(case.get("quarter")), taps=taps_or)
        else:
            oracle.psmnet_forward(sd, x, case["maxdisp"], recipes.out_hw(case), taps=taps_or)
    assert taps_hip, "no taps captured"
    for t, v in taps_hip.items():
        assert _rel(v.cpu(), taps_or[t]) < 1e-4, t
    for key in gold.files:                  # and against the reference's own sampled activations
        if key.startswith("tap_") and key[4:] in taps_hip:
            s, _ = recipes.sample(taps_hip[key[4:]].cpu())
            assert np.abs(s - gold[key]).max() <= 1e-4 * max(1.0, float(np.abs(gold[key]).max())), key


def test_full_benchmark_shape_matches_oracle(gpu):
    """BASELINE.json config #2's aggregator at full size (960x544, D=192): HIP (default split-fp16 precision) vs the CPU
    oracle on the same seeded weights and a random volume; also a size-independent property (disparity inside [0, D-1]).
    (tests/test_gpu_configs.py runs the same shape end to end from the images.)"""
    case = dict(model="gcnet", seed=21, maxdisp=192, in_shape=(1, 8, 96, 272, 480))
    model = recipes.build_case(case, *_our_classes())
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x = recipes.make_input(case["in_shape"], 21)
    got = model.cuda()(x.cuda()).cpu()
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    with torch.no_grad():
        ref = oracle.gcnet_forward(sd, x, 192)
    err = float((got - ref).abs().max())
    print("gcnet full size: max|disp - oracle| = %.3e, range %.2f..%.2f" % (err, float(ref.min()), float(ref.max())))
    assert got.shape == ref.shape == (1, 544, 960)
    assert float(got.min()) >= 0 and float(got.max()) <= 191
    assert err <= DISP_TOL


def _tail_stats(cost3, maxdisp, out_hw, dtype):
    """The reference's PSMNet tail (psmnet_3dcnn.py:167-174) on given logits in `dtype`: -> (disparity, kappa) with
    kappa_i = sum_d |d - disp_i| p_d, the first-order sensitivity of pixel i's disparity to a perturbation of its logits:
    |delta disp_i| <= kappa_i * max_d |delta logit_d| (softmax-weighted mean; trilinear weights are a convex combination)."""
    c = F.interpolate(cost3.to(dtype), [maxdisp, out_hw[0], out_hw[1]], mode="trilinear", align_corners=True).squeeze(1)
    p = F.softmax(c, 1)
    d = torch.arange(maxdisp, dtype=dtype).view(1, -1, 1, 1)
    disp = torch.sum(p * d, 1)
    kappa = torch.sum(p * (d - disp.unsqueeze(1)).abs(), 1)
    return disp, kappa


def test_cfg3_psmnet_full_size(gpu):
    """BASELINE.json config #3 at full size ([1,64,48,136,240] -> 960x544, D=192), three gates that can each fail:

      G1  the 28-conv trunk: logits cost3 (pre-trilinear, psmnet_3dcnn.py:147) vs the fp32 oracle, relative error <= 1e-5
          (the golden taps use 1e-4; fp32-vs-fp64 of the oracle itself is 9e-7 here);
      G2  the tail alone: HIP trilinear+softmax+regression vs torch's fp32 tail ON THE SAME (HIP) LOGITS;
      G3  end to end vs the fp32 oracle, per pixel: |disp - oracle| <= 1e-3 + kappa_i * (1e-5 * max|logit| + eps_tail) -- the
          G1 BOUND, not the measured logit error -- and >= 95 % of the map within a flat 1e-3.
          (tests/test_gpu_fullsize_golden.py repeats this against the REFERENCE's own map with a 5e-6 bound.)

    Why not a flat 1e-3 end to end: with these random-init weights the softmax over D=192 is broad (kappa up to ~80
    disparities per unit logit), so the oracle's own fp32 tail differs from an fp64 tail ON IDENTICAL LOGITS by 2e-2, and
    only 43 % of the pixels have |oracle fp32 - oracle fp64| <= 1e-4 (measured in the build container; printed below from
    the fp64 tail).  kappa is what tells a conditioning effect from a bug: a well-conditioned pixel gets 1e-3."""
    case = dict(model="psmnet", seed=21, maxdisp=192, in_shape=(1, 64, 48, 136, 240))
    maxdisp, hw = 192, (544, 960)
    model = recipes.build_case(case, *_our_classes())
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x = recipes.make_input(case["in_shape"], 21)
    taps = {}
    got = model.cuda()(x.cuda(), taps=taps).cpu()
    c3 = taps["cost3"].cpu()
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    with torch.no_grad():
        taps_or = {}
        ref = oracle.psmnet_forward(sd, x, maxdisp, hw, taps=taps_or)
        c3_ref = taps_or["cost3"]
        # G1
        dl = float((c3 - c3_ref).abs().max())
        rel = dl / float(c3_ref.abs().max())
        print("psmnet full size: logits cost3 rel err %.3e (abs %.3e, max|logit| %.1f)" % (rel, dl, float(c3_ref.abs().max())))
        assert c3.shape == c3_ref.shape == (1, 1, 48, 136, 240)
        assert rel <= 1e-5
        # G2: same logits through torch's fp32 tail and an fp64 tail
        t32, _ = _tail_stats(c3, maxdisp, hw, torch.float32)
        t64, kappa = _tail_stats(c3, maxdisp, hw, torch.float64)
    kappa = kappa.float()
    eps_tail = 8 * 2.0 ** -24 * float(c3.abs().max())         # fp32 rounding of the 7 lerp operations on a logit of this size
    e_tail = (got - t32).abs()
    noise32 = (t32.double() - t64).abs().float()
    print("psmnet full size: tail alone, HIP vs torch-fp32 on the same logits: max %.3e; torch-fp32 vs fp64 tail: max %.3e; "
          "kappa max %.1f median %.1f" % (float(e_tail.max()), float(noise32.max()), float(kappa.max()), float(kappa.median())))
    assert float((e_tail - (DISP_TOL + kappa * eps_tail)).max()) <= 0
    assert float((got.double() - t64).abs().max()) <= float(noise32.max()) + DISP_TOL   # no further from exact than torch's fp32 tail
    # G3
    err = (got - ref).abs()
    G1_BOUND = 1e-5                                              # the G1 gate itself: the tolerance must not follow the measured error
    dl_bound = G1_BOUND * float(c3_ref.abs().max())
    tol = DISP_TOL + kappa * (dl_bound + eps_tail)
    tight = float((tol <= 2 * DISP_TOL).float().mean())
    print("psmnet full size: max|disp - oracle| = %.3e; per-pixel tolerance 1e-3 + kappa*(%.2e) [from the G1 bound, measured dl %.2e]: "
          "%.1f%% of pixels gated at <= 2e-3, worst err/tol %.3f; pixels with |torch-fp32 - fp64 tail| <= 1e-4: %.1f%%"
          % (float(err.max()), dl_bound + eps_tail, dl, 100 * tight, float((err / tol).max()), 100 * float((noise32 <= 1e-4).float().mean())))
    assert got.shape == ref.shape == (1, 544, 960)
    assert float(got.min()) >= 0 and float(got.max()) <= maxdisp - 1
    assert float((err - tol).max()) <= 0
    assert tight >= 0.10                                         # the gate is not vacuous (bound-based tolerance: fewer pixels sit at <= 2e-3)
    assert float((err <= DISP_TOL).float().mean()) >= 0.95        # and the map as a whole is within the flat gate


def test_gcnet_16_plane_volume(gpu):
    """cbmv_in_planes=16 (the left+right volume of SURVEY 8(f).2): the first layer then has 16 input channels -- one
    16-channel K-step per tap on the split-fp16 kernel (conv3d_s1_c16_f16s).  Checked against the oracle."""
    from msnets_amd.gcnet_3dcnn import GCNet_CostVolumeAggre
    torch.manual_seed(3)
    model = GCNet_CostVolumeAggre(maxdisp=32, cbmv_in_planes=16).eval()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x = torch.rand((1, 16, 16, 16, 48), generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        ref = oracle.gcnet_forward(sd, x, 32)
    model = model.cuda()
    got = model(x.cuda()).cpu()
    assert model._plans("split-fp16")["conv3dbn_1"].f16s        # the 16-channel layer is on the split-fp16 path
    assert tuple(got.shape) == tuple(ref.shape) == (1, 32, 96)
    assert float((got - ref).abs().max()) <= DISP_TOL


def test_psmnet_takes_the_ms_volume(gpu):
    """PSMNet-style aggregator with an 8-channel first layer (the MS volume); checked against the oracle,
    whose dres0 follows the state_dict's shapes."""
    from msnets_amd.psmnet_3dcnn import PSMNet_CostVolumeAggre
    torch.manual_seed(5)
    model = PSMNet_CostVolumeAggre(64, in_planes=8).eval()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x = torch.rand((1, 8, 16, 12, 40), generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        ref = oracle.psmnet_forward(sd, x, 64, (48, 160))
    got = model.cuda()(x.cuda()).cpu()
    assert tuple(got.shape) == tuple(ref.shape) == (1, 48, 160)
    assert float((got - ref).abs().max()) <= DISP_TOL
    with pytest.raises(ValueError):
        model(torch.rand((1, 64, 16, 12, 40)).cuda())


def test_psmnet_all_heads(gpu):
    case = recipes.AGG_CASES["psmnet_small"]
    model = recipes.build_case(case, *_our_classes())
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x = recipes.make_input(case["in_shape"], case["seed"])
    got = model.cuda().forward_all_heads(x.cuda())
    with torch.no_grad():
        ref = oracle.psmnet_forward(sd, x, case["maxdisp"], recipes.out_hw(case), training=True)
    for a, b in zip(got, ref):
        assert float((a.cpu() - b).abs().max()) <= DISP_TOL


def test_reference_checkpoint_keys_load(gpu):
    """A DataParallel-style checkpoint ('module.' prefix, main_msnet.py:199-203) loads after prefix strip and
    the plan cache notices the new weights."""
    G, _ = _our_classes()
    torch.manual_seed(0)
    a = G(32).eval().cuda()
    torch.manual_seed(1)
    b = G(32).eval()
    recipes.randomize_bn(b, 1)
    x = torch.rand(1, 8, 16, 16, 32).cuda()
    d0 = a(x).clone()
    ckpt = {"module." + k: v for k, v in b.state_dict().items()}
    a.load_state_dict({k[len("module."):]: v for k, v in ckpt.items()}, strict=True)
    d1 = a(x)
    with torch.no_grad():
        ref = oracle.gcnet_forward(b.state_dict(), x.cpu(), 32)
    assert float((d1.cpu() - ref).abs().max()) <= DISP_TOL
    assert float((d1 - d0).abs().max()) > 1e-3


def test_data_edits_need_invalidate_plans(gpu):
    """Edits through `.data` do not bump a tensor's version counter, so the plan cache cannot see them (hipops.state_key);
    `invalidate_plans()` is the documented way to pick them up.  copy_ / load_state_dict are seen without it."""
    G, _ = _our_classes()
    torch.manual_seed(0)
    m = G(32).eval().cuda()
    x = torch.rand(1, 8, 16, 16, 32).cuda()
    d0 = m(x).clone()
    m.deconv5.weight.data.mul_(3.0)
    m.invalidate_plans()
    d1 = m(x).clone()
    assert float((d1 - d0).abs().max()) > 1e-3
    with torch.no_grad():
        ref = oracle.gcnet_forward({k: v.cpu() for k, v in m.state_dict().items()}, x.cpu(), 32)
    assert float((d1.cpu() - ref).abs().max()) <= DISP_TOL
    with torch.no_grad():
        m.conv3dbn_2[0].weight.mul_(0.5)                  # a tracked in-place op: seen without invalidate_plans()
    d2 = m(x)
    assert float((d2 - d1).abs().max()) > 1e-4


def _big_activation_model(scale):
    """A GCNet whose conv3dbn_2 output (= res_l20) is `scale` times larger than usual while everything downstream of it is
    scaled back (its two consumers' weights / scale), so the network stays well conditioned and only the activation range
    changes."""
    G, _ = _our_classes()
    torch.manual_seed(11)
    m = G(32).eval()
    recipes.randomize_bn(m, 11)
    with torch.no_grad():
        m.conv3dbn_2[1].weight.mul_(scale)
        m.conv3dbn_2[1].bias.mul_(scale)
        m.block_3d_1.convbn_3d_1[0].weight.div_(scale)
        m.deconvbn4[1].weight.mul_(scale)          # keep deconvbn4's own term visible next to the large skip connection
        m.deconvbn4[1].bias.mul_(scale)
        m.deconv5.weight.div_(scale)
    return m


def test_fp16_range_guard_reruns_on_fp32(gpu):
    """Activations of ~1e6 cannot be split into fp16 hi + lo (hi = inf).  The epilogues raise the library's overflow flag, the
    module warns, repeats the forward on the exact fp32 kernels and stays on them; the result matches the oracle.  With the
    guard switched off the same forward is garbage -- which is what the guard is for."""
    from msnets_amd import hipops
    assert hipops.get_default_precision() == "split-fp16"
    m = _big_activation_model(3.0e5)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.rand((1, 8, 16, 16, 32), generator=torch.Generator().manual_seed(12))
    with torch.no_grad():
        ref = oracle.gcnet_forward(sd, x, 32)
    m = m.cuda()
    with pytest.warns(RuntimeWarning, match="fp16 range"):
        got = m(x.cuda()).cpu()
    assert m._forced_precision == "fp32"
    assert float((got - ref).abs().max()) <= DISP_TOL
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")                  # second call: already on fp32, no new warning, no re-run
        again = m(x.cuda()).cpu()
    assert torch.equal(again, got)
    m.invalidate_plans()                                # back to split-fp16 ...
    m.range_check = False                               # ... without the guard
    bad = m(x.cuda()).cpu()
    assert not bool(torch.isfinite(bad).all()) or float((bad - ref).abs().max()) > 1.0
    # in-range activations never trip it
    G, _ = _our_classes()
    torch.manual_seed(1)
    ok = G(32).eval().cuda()
    import warnings as w2
    with w2.catch_warnings():
        w2.simplefilter("error")
        ok(x.cuda())
    assert ok._forced_precision is None


def test_winograd_input_range_is_guarded(gpu, monkeypatch):
    """The Winograd-depth kernel splits SUMS of two activations (q1 = p1 + p2): inputs between 32752 and 65504 -- finite as fp16
    themselves -- can overflow its `hi` halves.  The guard's limit is therefore 32752 in every epilogue: a conv3dbn_1 output
    of ~48 000 (compensated in conv3dbn_2's weights) trips it, the forward is repeated on fp32 and matches the oracle."""
    monkeypatch.setenv("MSNET_DIRECT", "0")             # tiled kernels at this size: conv3dbn_2 runs the Winograd-depth kernel
    G, _ = _our_classes()
    torch.manual_seed(21)
    m = G(32).eval()
    recipes.randomize_bn(m, 21)
    x = torch.rand((1, 8, 16, 16, 32), generator=torch.Generator().manual_seed(22))
    with torch.no_grad():
        a1 = float(F.relu(m.conv3dbn_1(x)).max())
        scale = 48000.0 / a1
        m.conv3dbn_1[1].weight.mul_(scale)
        m.conv3dbn_1[1].bias.mul_(scale)
        m.conv3dbn_2[0].weight.div_(scale)
        a1 = float(F.relu(m.conv3dbn_1(x)).max())
    assert 33000.0 < a1 < 60000.0                       # inside the fp16 range, outside the split-fp16 kernels' limit
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ref = oracle.gcnet_forward(sd, x, 32)
    m = m.cuda()
    with pytest.warns(RuntimeWarning, match="an activation"):
        got = m(x.cuda()).cpu()
    assert m._forced_precision == "fp32"
    assert float((got - ref).abs().max()) <= DISP_TOL


def test_range_flag_of_the_last_conv_is_seen(gpu):
    """The guard word is read back early -- behind the last launch that has a range check (deconvbn4), under the tail kernel
    (RangeGuard.checkpoint).  A flag raised by that very last launch, and by no other, must still be seen: deconvbn4's output
    scaled to ~1e5 (deconv5 compensates; the folded weights stay inside the fp16 range, so the layer stays on split-fp16), everything upstream in range."""
    G, _ = _our_classes()
    torch.manual_seed(41)
    m = G(32).eval()
    recipes.randomize_bn(m, 41)
    with torch.no_grad():
        m.deconvbn4[1].weight.mul_(1.0e5)
        m.deconvbn4[1].bias.mul_(1.0e5)
        m.deconv5.weight.div_(1.0e5)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.rand((1, 8, 16, 16, 32), generator=torch.Generator().manual_seed(42))
    with torch.no_grad():
        taps_or = {}
        ref = oracle.gcnet_forward(sd, x, 32, taps=taps_or)
    assert float(taps_or["deconvbn3"].abs().max()) < 3.0e4 < float(taps_or["deconvbn4"].abs().max())
    m = m.cuda()
    import warnings
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        got = m(x.cuda()).cpu()
    msgs = [str(w.message) for w in rec]
    assert any("an activation" in t for t in msgs), msgs
    assert not any("folded conv weight" in t for t in msgs), msgs      # deconvbn4 itself ran on the split-fp16 kernel
    assert m._forced_precision == "fp32"
    assert float((got - ref).abs().max()) <= DISP_TOL


def test_nonfinite_module_input_is_reported(gpu):
    """A NaN or inf voxel in the module input raises the INPUT bit of the range guard (the checks compare magnitude bits: v_max
    alone would drop the NaN): a warning names the module input, the call is repeated on fp32, the module stays on split-fp16."""
    G, _ = _our_classes()
    torch.manual_seed(23)
    m = G(32).eval().cuda()
    x = torch.rand((1, 8, 16, 16, 32), generator=torch.Generator().manual_seed(24)).cuda()
    good = m(x).clone()
    for bad in (float("nan"), float("inf"), -4.0e4):
        xb = x.clone()
        xb[0, 3, 7, 5, 11] = bad
        with pytest.warns(RuntimeWarning, match="module input"):
            m(xb)
        assert m._forced_precision is None
    assert torch.equal(m(x), good)


@pytest.mark.parametrize("ci,co,stride,transposed", [(32, 32, 1, False), (64, 64, 1, False), (8, 32, 1, False), (32, 64, 2, False),
                                                      (64, 32, 1, True), (128, 128, 1, False)])
@pytest.mark.parametrize("xs,ws", [(1.0e-5, 1.0), (1.0, 1.0e-5), (1.0e-5, 1.0e-5), (3.0e-4, 1.0e-3)])
def test_split_fp16_small_magnitudes(gpu, ci, co, stride, transposed, xs, ws):
    """Underflow twin of the range guard, per layer: activations and / or BN-folded weights of magnitude ~1e-5, i.e. below
    2^-14 where `hi = fp16(x)` is an fp16 SUBNORMAL.  ConvBNPlan's per-channel power-of-two pre-scale brings every folded
    weight back into the normal range (exact); small activations rely on the fp16 MFMA keeping subnormal inputs (it does on
    gfx950 -- this test is what establishes it) and carry an absolute error floor of 2^-35 per operand (lo's last subnormal
    bit), i.e. ~3e-6 relative at 1e-5.  Gate: 5e-6 of the layer's output magnitude, as for O(1) data."""
    from msnets_amd import hipops
    g = torch.Generator().manual_seed(ci * 13 + co + stride)
    dims = (1, 4, 9, 40) if ci < 128 else (1, 3, 5, 9)
    n, d, h, w = dims
    x = torch.randn((n, ci, d, h, w), generator=g) * 3 * xs
    if transposed:
        conv = torch.nn.ConvTranspose3d(ci, co, 3, stride=2, padding=1, output_padding=1, bias=False)
    else:
        conv = torch.nn.Conv3d(ci, co, 3, stride=stride, padding=1, bias=False)
    bn = torch.nn.BatchNorm3d(co)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (27 * ci)) ** 0.5)
        bn.weight.copy_((torch.rand(co, generator=g) + 0.5) * ws)
        bn.bias.copy_(torch.randn(co, generator=g) * 0.1 * xs * ws)
        bn.running_mean.copy_(torch.randn(co, generator=g) * 0.1 * xs)
        bn.running_var.copy_(torch.rand(co, generator=g) * 0.5 + 0.75)
    conv, bn = conv.eval(), bn.eval()
    with torch.no_grad():
        ref = F.relu(bn.double()(conv.double()(x.double())))
    conv, bn = conv.float().cuda(), bn.float().cuda()
    plan = hipops.ConvBNPlan(conv, bn, transposed=transposed, precision="split-fp16")
    assert plan.f16s
    if transposed:
        y = hipops.deconv3d_k3s2(_cl(x), plan.wpk, plan.scale, plan.shift, co, relu=True, f16s=True)
    else:
        y = hipops.conv3d_k3(_cl(x), plan.wpk, plan.scale, plan.shift, co, stride=stride, relu=True, f16s=True)
    err = float((_nc(y).double() - ref).abs().max() / ref.abs().max())        # (relative to the true magnitude, however small)
    print("split-fp16 small magnitudes %d->%d s%d%s x*%.0e w*%.0e: rel err %.2e (max|ref| %.2e)"
          % (ci, co, stride, " T" if transposed else "", xs, ws, err, float(ref.abs().max())))
    assert float(ref.abs().max()) > 0
    assert err < 5e-6


@pytest.mark.parametrize("direct", [None, "0"])
def test_small_activations_end_to_end(gpu, direct, monkeypatch):
    """Underflow twin of test_fp16_range_guard_reruns_on_fp32 at the module level: conv3dbn_2's output (= the res_l20 skip
    connection) and its folded weights are scaled to ~1e-5, compensated downstream.  The split-fp16 forward must still match
    the oracle to 1e-3 -- with no fallback (no warning, no fp32 re-run)."""
    import warnings
    if direct is not None:
        monkeypatch.setenv("MSNET_DIRECT", direct)      # "0": tiled kernels at this size -- conv3dbn_2 on the Winograd-depth kernel
    m = _big_activation_model(1.0e-5)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.rand((1, 8, 16, 16, 32), generator=torch.Generator().manual_seed(12))
    with torch.no_grad():
        taps_or = {}
        ref = oracle.gcnet_forward(sd, x, 32, taps=taps_or)
    assert float(taps_or["conv3dbn_2"].abs().max()) < 2e-4          # the activations really are in the subnormal-hi range
    m = m.cuda()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        got = m(x.cuda()).cpu()
    assert m._forced_precision is None
    err = float((got - ref).abs().max())
    print("small activations end to end: max|disp - oracle| = %.3e" % err)
    assert err <= DISP_TOL


def test_small_activations_into_winograd_layer(gpu, monkeypatch):
    """The other side of the Winograd-depth layer: its INPUT (conv3dbn_1's output) scaled to ~1e-5, its weights up by the same
    factor.  q = p1 + p2 etc. are formed in fp32 and split like any activation; no fallback, result within 1e-3 of the oracle."""
    import warnings
    monkeypatch.setenv("MSNET_DIRECT", "0")
    G, _ = _our_classes()
    torch.manual_seed(31)
    m = G(32).eval()
    recipes.randomize_bn(m, 31)
    with torch.no_grad():
        m.conv3dbn_1[1].weight.mul_(1.0e-5)
        m.conv3dbn_1[1].bias.mul_(1.0e-5)
        m.conv3dbn_2[0].weight.mul_(1.0e5)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.rand((1, 8, 16, 16, 32), generator=torch.Generator().manual_seed(32))
    with torch.no_grad():
        taps_or = {}
        ref = oracle.gcnet_forward(sd, x, 32, taps=taps_or)
    assert float(taps_or["conv3dbn_1"].abs().max()) < 2e-4
    m = m.cuda()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        got = m(x.cuda()).cpu()
    assert m._forced_precision is None
    err = float((got - ref).abs().max())
    print("small activations into the Winograd-depth layer: max|disp - oracle| = %.3e" % err)
    assert err <= DISP_TOL


def test_fp16_range_of_folded_weights(gpu):
    """A BN scale that pushes a folded weight beyond 65504 (tiny running_var / huge gamma): that layer is planned on the fp32
    MFMA kernel (with a warning) instead of producing an infinite `hi` half."""
    from msnets_amd import hipops
    G, _ = _our_classes()
    torch.manual_seed(13)
    m = G(32).eval()
    recipes.randomize_bn(m, 13)
    with torch.no_grad():
        m.block_3d_2.convbn_3d_2[1].weight.mul_(1.0e7)
        m.block_3d_2.convbn_3d_2[1].bias.mul_(1.0e7)
        m.block_3d_2.convbn_3d_3[0].weight.div_(1.0e7)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.rand((1, 8, 16, 16, 32), generator=torch.Generator().manual_seed(14))
    with torch.no_grad():
        ref = oracle.gcnet_forward(sd, x, 32)
    m = m.cuda()
    with pytest.warns(RuntimeWarning):
        got = m(x.cuda()).cpu()
    assert bool(torch.isfinite(got).all())
    assert float((got - ref).abs().max()) <= DISP_TOL


def test_activation_arena_reuse(gpu):
    """Steady-state forwards allocate no activation memory (hipops.Arena): same buffers, same result; what the module returns
    is a fresh tensor every time; a different input shape re-sizes the arena; taps bypass it."""
    G, _ = _our_classes()
    torch.manual_seed(2)
    m = G(32).eval().cuda()
    x = torch.rand(2, 8, 16, 16, 32).cuda()
    d0 = m(x)
    ptrs = [t.data_ptr() for t in m._arena.bufs]
    assert len(ptrs) >= 18 and m._arena.nbytes() > 0
    d1 = m(x)
    assert [t.data_ptr() for t in m._arena.bufs] == ptrs
    assert d1.data_ptr() != d0.data_ptr() and torch.equal(d0, d1)
    assert d0.data_ptr() not in ptrs
    taps = {}
    d2 = m(x, taps=taps)
    assert float((d2 - d0).abs().max()) < 1e-4              # the tapped forward uses the un-fused tail: rounding-level difference
    assert all(v.data_ptr() not in ptrs for v in taps.values())
    y = torch.rand(1, 8, 16, 32, 32).cuda()
    ref = m(y).clone()
    assert torch.equal(m(x), d0)                              # back to the first shape
    assert torch.equal(m(y), ref)


def test_errors(gpu):
    G, P = _our_classes()
    m = G(32).cuda()
    with pytest.raises(RuntimeError, match="inference only"):
        m(torch.rand(1, 8, 16, 16, 32).cuda())
    m.eval()
    with pytest.raises(AssertionError):
        m(torch.rand(1, 8, 32, 16, 32).cuda())          # 2*D' != maxdisp  (gcnet_3dcnn.py:135)
    with pytest.raises(ValueError):
        P(32).eval().cuda()(torch.rand(1, 8, 8, 16, 16).cuda())


def test_graphed_forward_equals_eager(gpu):
    """module.use_graph: the forward replayed from a captured HIP graph returns the bits of the eager forward, follows the
    CONTENT of the (reused) input buffer, and a new input buffer gets its own capture."""
    from msnets_amd import hipops
    from msnets_amd.gcnet_3dcnn import GCNet_CostVolumeAggre
    from msnets_amd.psmnet_3dcnn import PSMNet_CostVolumeAggre
    torch.manual_seed(3)
    for model, shape in ((GCNet_CostVolumeAggre(32).eval().cuda(), (1, 8, 16, 32, 48)),
                         (PSMNet_CostVolumeAggre(32).eval().cuda(), (1, 64, 8, 12, 20))):
        x = torch.rand(shape, device="cuda")
        ref1 = model(x).clone()
        x2 = torch.rand(shape, device="cuda")
        ref2 = model(x2).clone()
        model.use_graph = True
        buf = x.clone()
        outs = [model(buf) for _ in range(3)]                      # eager, capture + replay, replay
        assert all(torch.equal(o, ref1) for o in outs)
        assert len(model._graphs) == 1 and next(iter(model._graphs.values()))["graph"] is not None
        buf.copy_(x2)
        assert torch.equal(model(buf), ref2)                       # same buffer, new content
        other = x.clone()
        assert torch.equal(model(other), ref1) and torch.equal(model(other), ref1) and len(model._graphs) == 2
        kept = model(buf)
        model(other)
        assert torch.equal(kept, ref2)                             # returned tensors are copies, not the graph's buffer


def test_graphed_forward_range_guard(gpu):
    """The fp16-range guard stays armed under graph replay: an out-of-range INPUT trips it after the replay and that forward is
    repeated on the fp32 kernels -- for this call only (the graphs and the split-fp16 path stay); an out-of-range ACTIVATION
    drops the graphs and keeps the module on fp32."""
    import warnings
    from msnets_amd.gcnet_3dcnn import GCNet_CostVolumeAggre
    torch.manual_seed(4)
    model = GCNet_CostVolumeAggre(32).eval().cuda()
    x = torch.rand((1, 8, 16, 32, 48), device="cuda")
    model.use_graph = True
    a = model(x); b = model(x); c = model(x)
    assert torch.equal(a, b) and torch.equal(b, c)
    x[0, 0, 0, 0, 0] = 1e5
    with pytest.warns(RuntimeWarning, match="module input"):
        y = model(x)
    assert bool(torch.isfinite(y).all()) and model._forced_precision is None and len(model._graphs) == 1
    model.use_graph = False
    with pytest.warns(RuntimeWarning, match="module input"):
        assert torch.equal(y, model(x))                            # eager: the same per-call fallback, the same bits
    model.use_graph = True
    x[0, 0, 0, 0, 0] = 0.5
    with warnings.catch_warnings():
        warnings.simplefilter("error")                             # the sample after the bad one: split-fp16 graph replay again
        z = model(x)
    model.use_graph = False
    assert torch.equal(z, model(x))
    # activation trip under replay: the same buffer now holds values just inside the kernels' range (32752), so the INPUT passes but the
    # first conv's outputs do not: sticky, graphs dropped
    model.use_graph = True
    x.mul_(3.0e4)
    with pytest.warns(RuntimeWarning, match="an activation"):
        w = model(x)
    assert bool(torch.isfinite(w).all()) and model._forced_precision == "fp32" and not model._graphs
    w1 = model(x); w2 = model(x); w3 = model(x)                    # fp32 key: eager, capture, replay
    assert torch.equal(w1, w) and torch.equal(w2, w) and torch.equal(w3, w)


def test_graph_survives_arena_resize(gpu):
    """A captured graph bakes in the addresses of its activation buffers.  Shape A (captured), then shape B (re-sizes the
    module's arena and allocates where A's buffers might have been freed), then shape A again: the replay must still return
    shape A's result -- every graph owns its arena (replay into freed memory)."""
    from msnets_amd.gcnet_3dcnn import GCNet_CostVolumeAggre
    torch.manual_seed(6)
    model = GCNet_CostVolumeAggre(32).eval().cuda()
    xa = torch.rand((1, 8, 16, 32, 48), device="cuda")
    xb = torch.rand((1, 8, 16, 48, 80), device="cuda")
    ref_a, ref_b = model(xa).clone(), model(xb).clone()
    model.use_graph = True
    for _ in range(3):
        assert torch.equal(model(xa), ref_a)                       # eager, capture, replay
    for _ in range(3):
        assert torch.equal(model(xb), ref_b)                       # shape B: module arena re-sized, own graph
    junk = [torch.full((1, 16, 32, 48, 32), float(k), device="cuda") for k in range(8)]   # take whatever memory was released
    assert torch.equal(model(xa), ref_a)                           # replay of graph A
    assert torch.equal(model(xb), ref_b)
    model.use_graph = False
    assert torch.equal(model(xa), ref_a) and torch.equal(model(xb), ref_b)   # eager passes in between do not disturb the graphs
    model.use_graph = True
    assert torch.equal(model(xa), ref_a) and torch.equal(model(xb), ref_b)
    del junk


def test_range_fallback_scope(gpu):
    """the fp32 fallback is sticky only when an ACTIVATION raised the flag, and even then only for the parameter
    state it was raised for.  One bad input sample costs one repeated forward, not the module's speed."""
    import warnings
    G, _ = _our_classes()
    torch.manual_seed(7)
    m = G(32).eval().cuda()
    x = torch.rand((1, 8, 16, 16, 32), device="cuda")
    good = m(x).clone()
    bad = x.clone()
    bad[0, 3, 2, 5, 7] = float("inf")
    with pytest.warns(RuntimeWarning, match="module input"):
        m(bad)
    assert m._forced_precision is None
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert torch.equal(m(x), good)                             # the next sample is back on split-fp16, no warning
    big = _big_activation_model(3.0e5).cuda()
    with pytest.warns(RuntimeWarning, match="an activation"):
        big(x)
    assert big._forced_precision == "fp32"
    big.load_state_dict(m.state_dict())                            # new parameters: the fallback is lifted
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert torch.equal(big(x), good)
    assert big._forced_precision is None


def test_concurrent_forwards_from_two_threads(gpu):
    """The reference wraps the model in nn.DataParallel (main_msnet.py:174): one Python THREAD per replica.  Two module
    instances running forwards concurrently on two threads (here: on the one GPU, each on its own stream) must not hand each
    other's activation buffers out -- results equal the serial ones bit for bit; two threads calling the SAME instance are
    serialised by the module's per-device lock and also get the serial bits."""
    import threading
    G, P = _our_classes()
    torch.manual_seed(11)
    ma, mb = G(32).eval().cuda(), G(64).eval().cuda()
    mp = P(32).eval().cuda()
    xa, xb = torch.rand(2, 8, 16, 32, 48).cuda(), torch.rand(1, 8, 32, 16, 32).cuda()
    xp = torch.rand(1, 64, 8, 12, 20).cuda()
    ref = {"a": ma(xa).clone(), "b": mb(xb).clone(), "p": mp(xp).clone()}
    torch.cuda.synchronize()
    errs = []

    def worker(name, m, x, reps):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(reps):
                    out = m(x)
                    s.synchronize()
                    if not torch.equal(out, ref[name]):
                        errs.append("%s: result differs from the serial forward" % name)
                        return
        except Exception as e:      # noqa: BLE001
            errs.append("%s: %r" % (name, e))
    ts = [threading.Thread(target=worker, args=("a", ma, xa, 20)), threading.Thread(target=worker, args=("b", mb, xb, 20)),
          threading.Thread(target=worker, args=("p", mp, xp, 20)), threading.Thread(target=worker, args=("a", ma, xa, 20))]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not any(t.is_alive() for t in ts)
    assert not errs, errs


def test_data_parallel_wrapper_single_device_works_and_replicas_refuse(gpu):
    """nn.DataParallel over ONE device calls the module itself (no replicas): the reference's wrapper line then just works.
    A replica (what it creates for several devices) raises with the pointer to ms-nets_amd.dist."""
    G, _ = _our_classes()
    torch.manual_seed(12)
    m = G(32).eval().cuda()
    x = torch.rand(1, 8, 16, 16, 32).cuda()
    ref = m(x).clone()
    dp = torch.nn.DataParallel(m, device_ids=[0])
    assert torch.equal(dp(x), ref)
    rep = torch.nn.parallel.replicate(m, [0])[0]
    with pytest.raises(RuntimeError, match="one process per GPU"):
        rep(x)


def test_forward_ndhwc_equals_forward(gpu):
    """GCNet_CostVolumeAggre.forward_ndhwc on the channels-last volume == forward on the NCDHW volume, bit for bit (both
    precisions; 8 and 16 input planes), and an out-of-range / NaN voxel of the channels-last input is caught by the first
    layer's own check (RangeGuard.INPUT: that call falls back to fp32, the module stays on split-fp16)."""
    from msnets_amd import hipops
    G, _ = _our_classes()
    for planes, shape in ((8, (2, 8, 16, 32, 48)), (16, (1, 16, 16, 16, 32))):
        torch.manual_seed(21)
        m = G(32, cbmv_in_planes=planes).eval().cuda()
        x = torch.rand(shape).cuda()
        xcl = x.permute(0, 2, 3, 4, 1).contiguous()
        for prec in ("split-fp16", "fp32"):
            hipops.set_default_precision(prec)
            try:
                assert torch.equal(m.forward_ndhwc(xcl), m(x)), (planes, prec)
            finally:
                hipops.set_default_precision("split-fp16")
    torch.manual_seed(22)
    m = G(32).eval().cuda()
    x = torch.rand(1, 8, 16, 16, 32)
    good = m(x.cuda()).clone()
    bad = x.clone()
    bad[0, 3, 5, 7, 9] = float("nan")
    bcl = bad.permute(0, 2, 3, 4, 1).contiguous().cuda()
    with pytest.warns(RuntimeWarning, match="module input"):
        out = m.forward_ndhwc(bcl)
    assert m._forced_precision is None                      # one bad sample does not move the module to fp32
    assert torch.equal(m.forward_ndhwc(x.permute(0, 2, 3, 4, 1).contiguous().cuda()), good)
    big = x.clone()
    big[0, 1, 2, 3, 4] = 5e4
    with pytest.warns(RuntimeWarning, match="module input"):
        m.forward_ndhwc(big.permute(0, 2, 3, 4, 1).contiguous().cuda())
    with pytest.raises(ValueError):
        m.forward_ndhwc(x.cuda())                           # an NCDHW tensor is not a channels-last volume of 8 planes


def test_winograd_depth_on_channel_slices(gpu):
    """msnet_conv3d_k3_wd_f16s_strided (experiment entry): a 64 -> 64 layer as four Winograd-depth launches on
    32-channel slices of the 64-channel records, partial sums handed over through the residual -- same 5e-6 bound vs the fp64
    convolution as every other split-fp16 layer; odd depth, edge tiles, batch 2."""
    from msnets_amd import hipops
    g = torch.Generator().manual_seed(17)
    for (n, d, h, w), relu in (((1, 5, 9, 40), True), ((2, 4, 8, 33), False)):
        x = torch.randn((n, 64, d, h, w), generator=g)
        wt = torch.randn((64, 64, 3, 3, 3), generator=g) * 0.05
        scale = torch.rand(64, generator=g) + 0.5
        shift = torch.randn(64, generator=g) * 0.1
        ref = F.conv3d(x.double(), wt.double(), None, padding=1) * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)
        if relu:
            ref = F.relu(ref)
        wg = wt.cuda()
        wd4 = [[hipops.winograd_depth_weights(wg[32 * a:32 * a + 32, 32 * b:32 * b + 32].contiguous()) for b in range(2)] for a in range(2)]
        y = hipops.conv3d_k3_wd64(_cl(x), wd4, scale.cuda(), shift.cuda(), relu=relu)
        err = _rel(_nc(y).double(), ref)
        print("WD64 (four strided Winograd-depth launches) rel err %.2e" % err)
        assert err < 5e-6


def test_psmnet_forward_ndhwc_equals_forward(gpu):
    """PSMNet_CostVolumeAggre(in_planes=8).forward_ndhwc on the channels-last MS volume == forward on the NCDHW volume, bit for bit
    (end to end from two images through VolumeBuilder(layout="ndhwc") at quarter resolution); the 64-plane module (the
    reference's own width) reads its channels-last input in place behind one read-only range pass and agrees as well."""
    from msnets_amd import cbmv_generator as cg, synthetic
    _, P = _our_classes()
    left, right, _ = synthetic.stereo_pair(32, 48, 16, seed=9)
    l, r = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    vol = cg.build_ms_volume(l, r, 16)                                    # [8, 16, 32, 48]
    vcl = cg.build_ms_volume(l, r, 16, layout="ndhwc")                    # [16, 32, 48, 8]
    torch.manual_seed(31)
    m = P(64, in_planes=8).eval().cuda()
    assert torch.equal(m.forward_ndhwc(vcl.unsqueeze(0)), m(vol.unsqueeze(0)))
    torch.manual_seed(32)
    m64 = P(32).eval().cuda()
    x = torch.rand(1, 64, 8, 16, 24).cuda()
    good = m64(x).clone()
    assert torch.equal(m64.forward_ndhwc(x.permute(0, 2, 3, 4, 1).contiguous()), good)
    # the 64-plane channels-last input has no conversion pass to carry the fp16-range check: msnet_check_input_range reads it once
    for bad_value in (float("nan"), float("inf"), 4e4):
        bad = x.clone()
        bad[0, 17, 3, 5, 7] = bad_value
        with pytest.warns(RuntimeWarning, match="module input"):
            m64.forward_ndhwc(bad.permute(0, 2, 3, 4, 1).contiguous())
        assert m64._forced_precision is None                              # a bad sample does not move the module to fp32
    assert torch.equal(m64.forward_ndhwc(x.permute(0, 2, 3, 4, 1).contiguous()), good)
    with pytest.raises(ValueError):
        m.forward_ndhwc(vol.unsqueeze(0))



def test_check_input_range_any_length_and_alignment(gpu):
    """msnet_check_input_range takes any count and any float alignment (scalar edges around whole float4 reads); a bad
    value is found wherever it sits -- in the unaligned head, in the body, in the tail."""
    from msnets_amd import hipops
    base = torch.rand(4099, device="cuda")
    for off in (0, 1, 2, 3):
        for n in (1, 2, 3, 4, 5, 7, 8, 1021, 4090):
            for pos in (None, 0, n - 1, n // 2):
                x = base[off:off + n].clone() if off == 0 else base[off:off + n]      # a view: data_ptr() is 4 * off past alignment
                keep = None
                if pos is not None:
                    keep = float(base[off + pos])
                    base[off + pos] = float("inf") if (pos + n) % 2 else 7e4
                    x = base[off:off + n]
                with hipops.RangeGuard(x.device) as g:
                    hipops.check_input_range(x)
                    torch.cuda.synchronize()
                    word = g.word()
                if keep is not None:
                    base[off + pos] = keep
                assert word == (0 if pos is None else hipops.RangeGuard.INPUT), (off, n, pos, word)


def test_conv_on_module_input_carries_the_range_check(gpu):
    """Round 6: msnet_conv3d_k3_in_f16s == msnet_conv3d_k3_f16s bit for bit, and raises RangeGuard.INPUT for a bad module input
    wherever it sits -- on the tiled Co = 32 kernel (64 -> 32 at 8x32x64: the check rides in the loaders, no pass over the
    volume) and on a shape that kernel does not take (64 -> 32 at 4x8x24: the direct kernel behind msnet_check_input_range)."""
    from msnets_amd import hipops
    torch.manual_seed(5)
    for (d, h, w) in ((8, 32, 64), (6, 21, 70), (4, 8, 24)):
        conv = torch.nn.Conv3d(64, 32, 3, padding=1, bias=False)
        bn = torch.nn.BatchNorm3d(32).eval()
        p = hipops.ConvBNPlan(conv.cuda(), bn.cuda(), precision="split-fp16")
        x = torch.rand(2, d, h, w, 64, device="cuda") * 4 - 2
        with hipops.RangeGuard(x.device) as g:
            y_in = hipops.conv3d_k3_in(x, p.wpk, p.scale, p.shift, p.co, relu=True)
            torch.cuda.synchronize()
            assert g.word() == 0
        y = hipops.conv3d_k3(x, p.wpk, p.scale, p.shift, p.co, relu=True, f16s=True)
        assert torch.equal(y_in, y), (d, h, w)
        for pos, val in (((0, 0, 0, 0, 0), float("nan")), ((1, d - 1, h - 1, w - 1, 63), float("inf")), ((0, d // 2, h // 2, w // 2, 37), -5e4),
                         ((1, 0, h - 1, 0, 31), 32752.0)):
            bad = x.clone()
            bad[pos] = val
            with hipops.RangeGuard(x.device) as g:
                hipops.conv3d_k3_in(bad, p.wpk, p.scale, p.shift, p.co, relu=True)
                torch.cuda.synchronize()
                assert g.word() & hipops.RangeGuard.INPUT, (d, h, w, pos, val)
        ok = x.clone()
        ok[0, 1, 2, 3, 4] = 32000.0                      # inside the range: no input bit (the activation bit may rise downstream)
        with hipops.RangeGuard(x.device) as g:
            hipops.conv3d_k3_in(ok, p.wpk, p.scale, p.shift, p.co, relu=True)
            torch.cuda.synchronize()
            assert not (g.word() & hipops.RangeGuard.INPUT)
