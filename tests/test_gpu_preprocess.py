"""GPU parity of the on-device test-time pre-processing (pad, anti-aliased rescale, border) against the oracle:
uint8 results must be identical."""
import numpy as np
import pytest
import torch

from oracle import ms_volume as O

pytestmark = pytest.mark.gpu


def _img(shape, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "random":
        return rng.integers(0, 256, shape, dtype=np.uint8)
    if kind == "white":
        return np.full(shape, 255, np.uint8)
    if kind == "dark":
        return rng.integers(0, 7, shape, dtype=np.uint8)
    if kind == "texture":
        from msnets_amd import synthetic
        l, _, _ = synthetic.stereo_pair(shape[0], shape[1], 16, seed=seed)
        return np.ascontiguousarray(l[10:-10, 10:-10][: shape[0], : shape[1]])
    raise ValueError(kind)


@pytest.mark.parametrize("shape,enc,ds", [((540, 960), 32, 2), ((70, 100), 32, 2), ((64, 96), 32, 4), ((37, 53), 16, 1),
                                          ((33, 65), 64, 2), ((96, 96), 32, 8)])
@pytest.mark.parametrize("kind", ["random", "white", "dark"])
def test_prepare_test_image_bit_exact(gpu, shape, enc, ds, kind):
    from msnets_amd import cbmv_generator as cg
    img = _img(shape, 3, kind)
    ref = O.prepare_test_image(img, enc, ds, 10)
    got = cg.prepare_test_image(img, enc, ds, 10).cpu().numpy()
    assert got.shape == ref.shape and got.dtype == np.uint8
    bad = got != ref
    assert not bad.any(), "%d / %d pixels differ (max |diff| %d)" % (
        int(bad.sum()), bad.size, int(np.abs(got.astype(int) - ref.astype(int)).max()))


def test_down_sampling_input_mirror(gpu):
    from msnets_amd import cbmv_generator as cg
    l, r = _img((128, 192), 1, "random"), _img((128, 192), 2, "random")
    gl, gr = cg.down_sampling_input(0.5, l, r)
    assert isinstance(gl, np.ndarray) and gl.shape == (64, 96)
    assert np.array_equal(gl, O.rescale_explicit(l, 2)) and np.array_equal(gr, O.rescale_explicit(r, 2))
    tl, _ = cg.down_sampling_input(0.25, torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda())
    assert tl.is_cuda and np.array_equal(tl.cpu().numpy(), O.rescale_explicit(l, 4))
    with pytest.raises(ValueError):
        cg.down_sampling_input(0.5, l[:-1], r[:-1])
    with pytest.raises(ValueError):
        cg.down_sampling_input(0.3, l, r)
    with pytest.raises(RuntimeError, match="no CPU"):
        cg.prepare_test_image(torch.from_numpy(l))


def test_images_to_volume(gpu):
    """generate_test_cbmv's whole device part: two grayscale images -> 8-channel volume, vs the oracle's two stages."""
    from msnets_amd import cbmv_generator as cg, synthetic
    l, r, _ = synthetic.stereo_pair(120, 200, 32, seed=5)          # bordered by 10: use the interior as "camera" images
    l, r = np.ascontiguousarray(l[10:-10, 10:-10]), np.ascontiguousarray(r[10:-10, 10:-10])
    vol, (pad_h, pad_w) = cg.generate_test_cbmv_from_images(l, r, encoder_ds=32, maxdisp=32)
    lb, rb = O.prepare_test_image(l, 32, 2, 10), O.prepare_test_image(r, 32, 2, 10)
    ref = O.build_ms_volume(lb, rb, 16)
    got = vol.cpu().numpy()
    assert (pad_h, pad_w) == ((32 - 120 % 32) % 32, (32 - 200 % 32) % 32)
    assert got.shape == ref.shape == (8, 16, (120 + pad_h) // 2, (200 + pad_w) // 2)
    assert np.array_equal(got[:4].view(np.uint32), ref[:4].view(np.uint32))
    assert np.abs(got[4:] - ref[4:]).max() <= 2e-6


def test_epe_badx_on_device(gpu):
    """get_epe_rate (main_msnet.py:708-713) on the device == the host formula on the same maps, including the
    validity mask (0.001 <= gt <= max_disp) and the driver's crop of the padded rows / columns (main_msnet.py:585-589)."""
    import torch
    from msnets_amd import driver_utils as du
    rng = np.random.default_rng(5)
    gt = (rng.random((540, 960), dtype=np.float32) * 230).astype(np.float32)       # some beyond max_disp
    gt[rng.random(gt.shape) < 0.1] = 0.0                                            # invalid pixels
    pred_pad = (rng.random((1, 544, 960), dtype=np.float32) * 192).astype(np.float32)
    pred = du.crop_disparity(pred_pad, 544, 960, 540, 960)
    ref_epe, ref_rate = du.get_epe_rate(gt, pred, 192, 3.0)
    pg = torch.from_numpy(pred_pad).cuda()[0, 4:, :960].contiguous()               # the same crop, on the device
    epe, rate = du.get_epe_rate(torch.from_numpy(gt).cuda(), pg, 192, 3.0)
    assert abs(epe - float(ref_epe)) <= 1e-5 * float(ref_epe) and abs(rate - float(ref_rate)) < 1e-12
    e0, r0 = du.get_epe_rate(torch.zeros(8, 8).cuda(), torch.ones(8, 8).cuda())
    assert np.isnan(e0) and np.isnan(r0)
