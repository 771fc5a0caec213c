"""GPU parity of the matching-space volume kernels against the CPU oracle, through the libmatchers /
libfeatextract / cbmv_generator mirrors (which call the C ABI).  Integer and order-pinned float32 paths must be
bit-exact; the AML channels go through expf (GPU libm vs glibc) and get a 2e-6 abs tolerance on values in [0,1]."""
import numpy as np
import pytest
import torch

from oracle import ms_volume as O

pytestmark = pytest.mark.gpu
SENT = np.float32(2147483648.0)


def _pair(H, W, nd, seed, kind="texture"):
    from msnets_amd import synthetic
    if kind == "texture":
        l, r, _ = synthetic.stereo_pair(H - 20, W - 20, nd, seed=seed)
        return l, r
    rng = np.random.default_rng(seed)
    if kind == "random":
        return rng.integers(0, 256, (H, W), dtype=np.uint8), rng.integers(0, 256, (H, W), dtype=np.uint8)
    if kind == "flat":      # large constant regions: NCC's non-finite branch, all-equal census
        l = np.full((H, W), 90, np.uint8); r = np.full((H, W), 90, np.uint8)
        l[: H // 2, : W // 2] = rng.integers(0, 256, (H // 2, W // 2), dtype=np.uint8)
        r[H // 3:, W // 3:] = rng.integers(0, 256, (H - H // 3, W - W // 3), dtype=np.uint8)
        return l, r
    if kind == "extreme":   # 0/255 checkerboards: largest Sobel responses, float32 integral beyond 2^24
        yy, xx = np.mgrid[0:H, 0:W]
        l = (((yy + xx) & 1) * 255).astype(np.uint8)
        r = (((yy + 2 * xx) & 1) * 255).astype(np.uint8)
        return l, r
    raise ValueError(kind)


CASES = [(68, 100, 16, 0, "texture"), (48, 80, 16, 1, "random"), (57, 93, 12, 2, "flat"), (40, 150, 33, 3, "extreme"),
         (30, 41, 8, 4, "random"), (12, 13, 4, 5, "random"), (292, 500, 96, 6, "texture")]


def _bitexact(a, b, what):
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, what
    bad = a.view(np.uint32) != b.view(np.uint32)
    assert not bad.any(), "%s: %d / %d values differ, max|diff| %.3e" % (
        what, int(bad.sum()), bad.size, float(np.abs(a[bad].astype(np.float64) - b[bad]).max()))


@pytest.mark.parametrize("H,W,nd,seed,kind", CASES)
def test_matchers_bit_exact(gpu, H, W, nd, seed, kind):
    from msnets_amd import libmatchers as mtc, libfeatextract as fte
    l, r = _pair(H, W, nd, seed, kind)
    _bitexact(mtc.census(l, r, nd, 11), O.census(l, r, nd, 11), "census")
    _bitexact(mtc.nccNister(l, r, nd, 3), O.nccNister(l, r, nd, 3), "ncc")
    _bitexact(mtc.zsad(l, r, nd, 5), O.zsad(l, r, nd, 5), "zsad")
    sl, sr = mtc.sobel(l), mtc.sobel(r)
    _bitexact(sl, O.sobel(l), "sobel")
    _bitexact(mtc.sadsob(sl, sr, nd, 5), O.sadsob(O.sobel(l), O.sobel(r), nd, 5), "sadsob")
    z = O.zsad(l, r, nd, 5)
    _bitexact(fte.swap_axes(z), O.swap_axes(z), "swap_axes")


def test_other_window_sizes(gpu):
    from msnets_amd import libmatchers as mtc
    l, r = _pair(50, 70, 10, 7, "random")
    _bitexact(mtc.census(l, r, 10, 5), O.census(l, r, 10, 5), "census w5")
    _bitexact(mtc.census(l, r, 10, 9), O.census(l, r, 10, 9), "census w9")
    _bitexact(mtc.nccNister(l, r, 10, 5), O.nccNister(l, r, 10, 5), "ncc w5")
    _bitexact(mtc.zsad(l, r, 10, 3), O.zsad(l, r, 10, 3), "zsad w3")
    s = O.sobel(l), O.sobel(r)
    _bitexact(mtc.sadsob(s[0], s[1], 10, 7), O.sadsob(s[0], s[1], 10, 7), "sadsob w7")


@pytest.mark.parametrize("sigma", [128.0, 0.02, 20000.0])
def test_extract_likelihood(gpu, sigma):
    from msnets_amd import libfeatextract as fte
    rng = np.random.default_rng(0)
    vol = (rng.random((1000, 96), dtype=np.float32) * {128.0: 120, 0.02: 2, 20000.0: 8192}[sigma]).astype(np.float32)
    vol[rng.random(vol.shape) < 0.2] = SENT          # unreached entries
    vol[5] = SENT                                      # an all-sentinel row -> zeros
    got, ref = fte.extract_likelihood(vol, sigma), O.extract_likelihood(vol, sigma)
    assert np.abs(got - ref).max() <= 2e-6
    assert not got[5].any()
    assert np.all(got[vol == SENT] == 0)
    live = np.ones(len(vol), bool); live[5] = False
    assert np.abs(got[live].sum(1) - 1).max() < 1e-4


@pytest.mark.parametrize("H,W,nd,seed,kind", CASES[:5] + CASES[6:])
def test_fused_volume_build(gpu, H, W, nd, seed, kind):
    """msnet_build_volume == get_costs + extract_features_left of the oracle; cost channels bit-exact."""
    from msnets_amd import cbmv_generator as cg
    l, r = _pair(H, W, nd, seed, kind)
    ref = O.build_ms_volume(l, r, nd)
    got = cg.build_ms_volume(l, r, nd)
    assert got.shape == ref.shape == (8, nd, H - 20, W - 20)
    for ch, nm in enumerate(["census", "ncc", "sobel", "sad"]):
        _bitexact(got[ch], ref[ch], "cost channel " + nm)
    assert np.abs(got[4:] - ref[4:]).max() <= 2e-6
    assert got.min() >= 0 and got.max() <= 1


@pytest.mark.parametrize("H,W,nd,seed,kind", [(68, 100, 16, 0, "texture"), (40, 150, 32, 3, "extreme"), (57, 93, 8, 2, "flat"),
                                              (84, 212, 48, 8, "random"), (292, 500, 96, 6, "texture"),
                                              (45, 1300, 24, 9, "random")])     # wider than a 32-row LDS band: 16-row bands
def test_fast_and_generic_volume_paths_agree(gpu, monkeypatch, H, W, nd, seed, kind):
    """msnet_build_volume has a register-resident fast path for the reference's own windows (volume_fused.hip) and the
    run-time-window kernels (volume.hip, forced here with MSNET_VOLUME_GENERIC=1).  Cost channels must be bit-identical
    between the two and to the oracle; the likelihood channels of the two paths use the same expf and must agree bit for bit
    wherever the value is a normal float."""
    from msnets_amd import cbmv_generator as cg
    l, r = _pair(H, W, nd, seed, kind)
    monkeypatch.setenv("MSNET_VOLUME_GENERIC", "0")
    fast = cg.build_ms_volume(l, r, nd)
    monkeypatch.setenv("MSNET_VOLUME_GENERIC", "1")
    slow = cg.build_ms_volume(l, r, nd)
    ref = O.build_ms_volume(l, r, nd)
    for ch, nm in enumerate(["census", "ncc", "sobel", "sad"]):
        _bitexact(fast[ch], ref[ch], "fast path, cost channel " + nm)
        _bitexact(slow[ch], ref[ch], "generic path, cost channel " + nm)
    # likelihoods: same expf, IEEE division (generic) vs the three-operation division (fast): identical bits except where
    # the quotient is subnormal (< 1.2e-38), where the fma residual is no longer exact
    assert np.abs(fast[4:].astype(np.float64) - slow[4:]).max() < 1.2e-38
    assert np.abs(fast[4:] - ref[4:]).max() <= 2e-6


def test_fused_volume_build_other_windows(gpu):
    """Non-default window sizes take the run-time-window kernels."""
    from msnets_amd import cbmv_generator as cg
    l, r = _pair(64, 90, 12, 11, "random")
    c, n, so, sa = O.get_costs(l, r, 12, 9, 5, 3, 7, 10, 10, 10)
    ref = O.extract_features_left(c, n, so, sa, 128.0, 0.02, 20000.0, 20000.0)
    got = cg.build_ms_volume(l, r, 12, params=dict(censw=9, nccw=5, sadw=3, sobelw=7))
    for ch, nm in enumerate(["census", "ncc", "sobel", "sad"]):
        _bitexact(got[ch], ref[ch], "cost channel " + nm)
    assert np.abs(got[4:] - ref[4:]).max() <= 2e-6


def test_unfused_pipeline_matches_fused(gpu):
    """The reference's own call sequence (get_costs -> extract_features_left) through the drop-in modules."""
    from msnets_amd import cbmv_generator as cg
    l, r = _pair(68, 100, 16, 0, "texture")
    costs = cg.get_costs(l, r, 16, 11, 3, 5, 5, 10, 10, 10)
    ref_costs = O.get_costs(l, r, 16, 11, 3, 5, 5, 10, 10, 10)
    for a, b, nm in zip(costs, ref_costs, ["census", "ncc", "sobel", "sad"]):
        _bitexact(a, b, nm)
    feats = cg.extract_features_left(*costs)
    fused = cg.build_ms_volume(l, r, 16)
    _bitexact(feats[:4], fused[:4], "cost channels fused vs unfused")
    assert np.abs(feats[4:] - fused[4:]).max() <= 2e-6


def test_left_right_features(gpu):
    """get_right_cost + extract_features_lr (the 16-channel volume of is_left_only=False)."""
    from msnets_amd import cbmv_generator as cg, libfeatextract as fte
    l, r = _pair(60, 90, 12, 9, "texture")
    costs = O.get_costs(l, r, 12, 11, 3, 5, 5, 10, 10, 10)
    for c, nm in zip(costs, ["census", "ncc", "sobel", "sad"]):
        _bitexact(fte.get_right_cost(c), O.get_right_cost(c), "get_right_cost " + nm)
    ref = O.extract_features_lr(*costs)
    got = cg.extract_features_lr(*costs)
    assert got.shape == ref.shape == (16, 12, 40, 70)
    for ch in (0, 1, 2, 3, 8, 9, 10, 11):
        _bitexact(got[ch], ref[ch], "cost channel %d" % ch)
    assert np.abs(got - ref).max() <= 2e-6
    with pytest.raises(ValueError):
        fte.get_right_cost(costs[0][0])


def test_planted_disparity_recovered(gpu):
    """Size-independent property at a benchmark shape: census argmin recovers the planted shift."""
    from msnets_amd import cbmv_generator as cg, synthetic
    l, r, drows = synthetic.stereo_pair(128, 256, 32, seed=3)
    vol = cg.build_ms_volume(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda(), 32)
    am = vol[0].argmin(0).cpu().numpy()                  # [H', W']
    ok = am[:, 40:] == drows[:, None]
    assert ok.mean() > 0.9
    aml = vol[4].cpu().numpy()
    assert np.abs(aml.sum(0) - 1).max() < 1e-4           # likelihoods sum to 1 over d wherever a cost exists


def test_volume_errors(gpu):
    from msnets_amd import libmatchers as mtc
    l, r = _pair(30, 40, 8, 0, "random")
    with pytest.raises(TypeError):
        mtc.census(l.astype(np.float32), r.astype(np.float32), 8, 11)
    with pytest.raises(RuntimeError, match="wsize"):
        mtc.census(l, r, 8, 4)
    with pytest.raises(ValueError):
        mtc.zsad(l, r[:, :-1].copy(), 8, 5)


# ---- round 4: the channels-last build (msnet_build_volume_ndhwc) ------------------------------------------------------------
@pytest.mark.parametrize("Hh,Wh,nd,seed", [(32, 64, 16, 0), (20, 70, 8, 1), (37, 129, 32, 2), (16, 200, 96, 3), (5, 33, 24, 4),
                                           (64, 63, 40, 5)])
def test_channels_last_volume_is_the_ncdhw_volume_transposed(gpu, Hh, Wh, nd, seed):
    """VolumeBuilder(layout='ndhwc') [D',H',W',8] == the NCDHW build [8,D',H',W'] permuted, bit for bit (ragged widths, row
    ends inside a 64-pixel segment, D' = 8 .. 96, both register-array instantiations)."""
    from msnets_amd import cbmv_generator as cg, synthetic
    left, right, _ = synthetic.stereo_pair(Hh, Wh, nd, seed=seed)
    l, r = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    ref = cg.build_ms_volume(l, r, nd)
    vb = cg.VolumeBuilder(l.shape[0], l.shape[1], nd, l.device, layout="ndhwc")
    assert vb.native_cl
    got = vb(l, r)
    assert got.shape == (nd, Hh, Wh, 8)
    assert torch.equal(got, ref.permute(1, 2, 3, 0).contiguous())
    out = torch.full_like(got, float("nan"))                 # every element of a caller's buffer is written
    vb(l, r, out=out)
    assert torch.equal(out, got)


def test_channels_last_volume_full_size(gpu):
    """config #2's shape, and the extreme images (flat / checkerboard: sentinel-only rows, NCC's non-finite branch)."""
    from msnets_amd import cbmv_generator as cg, synthetic
    left, right, _ = synthetic.stereo_pair(272, 480, 96, seed=7)
    rng = np.random.default_rng(0)
    flat = np.zeros_like(left)
    flat[10:-10, 10:-10] = 77
    chk = np.zeros_like(left)
    chk[10:-10, 10:-10] = ((np.add.outer(np.arange(272), np.arange(480)) & 1) * 255).astype(np.uint8)
    noise = np.zeros_like(left)
    noise[10:-10, 10:-10] = rng.integers(0, 256, size=(272, 480))
    for a, b in ((left, right), (flat, flat), (chk, np.roll(chk, 1, axis=1)), (noise, flat)):
        l, r = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        ref = cg.build_ms_volume(l, r, 96)
        got = cg.build_ms_volume(l, r, 96, layout="ndhwc")
        assert torch.equal(got, ref.permute(1, 2, 3, 0).contiguous())


def test_channels_last_layout_falls_back_for_other_windows(gpu):
    """Parameters the channels-last kernel does not take (here a 9x9 census window) are built NCDHW and converted: same API."""
    from msnets_amd import cbmv_generator as cg, synthetic
    left, right, _ = synthetic.stereo_pair(24, 48, 16, seed=3)
    l, r = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    vb = cg.VolumeBuilder(l.shape[0], l.shape[1], 16, l.device, params=dict(censw=9), layout="ndhwc")
    assert not vb.native_cl
    ref = cg.build_ms_volume(l, r, 16, params=dict(censw=9))
    assert torch.equal(vb(l, r), ref.permute(1, 2, 3, 0).contiguous())
