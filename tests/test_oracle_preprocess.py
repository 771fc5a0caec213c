"""Test-time pre-processing oracle.  scikit-image is absent, so parity with skimage.transform.rescale
itself is UNPINNED; what is pinned here is that the explicit restatement (the arithmetic the HIP kernel implements)
equals, bit for bit, the scipy.ndimage calls skimage.transform.resize is built from."""
import numpy as np
import pytest

from oracle import ms_volume as O


def _images(shape, seed):
    rng = np.random.default_rng(seed)
    yield "random", rng.integers(0, 256, shape, dtype=np.uint8)
    yield "white", np.full(shape, 255, np.uint8)
    stripes = np.zeros(shape, np.uint8); stripes[::2] = 255
    yield "stripes", stripes
    ramp = np.tile(np.linspace(0, 255, shape[1]).astype(np.uint8), (shape[0], 1))
    yield "ramp", ramp
    dark = rng.integers(0, 7, shape, dtype=np.uint8)       # max far below 255: the clip range matters
    yield "dark", dark


@pytest.mark.parametrize("s", [2, 4, 3])
@pytest.mark.parametrize("shape", [(96, 144), (48, 60), (192, 96)])
def test_explicit_restatement_equals_scipy_calls(s, shape):
    if shape[0] % s or shape[1] % s:
        pytest.skip("size not a multiple of the factor")
    for name, img in _images(shape, s):
        a, b = O.rescale_explicit(img, s), O.rescale_scipy(img, s)
        assert a.shape == (shape[0] // s, shape[1] // s) and a.dtype == np.uint8
        assert np.array_equal(a, b), "%s s=%d: %d pixels differ" % (name, s, int((a != b).sum()))


def test_known_answers():
    # far from the border a constant image stays constant up to the float32 rounding of the taps' sum
    img = np.full((64, 64), 200, np.uint8)
    out = O.rescale_explicit(img, 2)
    assert set(np.unique(out[4:-4, 4:-4])) <= {199, 200}
    assert out[0, 0] < out[8, 8]                           # zero padding (mode='constant') darkens the rim
    assert np.array_equal(O.rescale_explicit(img, 1), img)
    w = O.gaussian_weights(2)
    assert len(w) == 5 and abs(w.sum() - 1) < 1e-15 and np.allclose(w, w[::-1])
    assert len(O.gaussian_weights(4)) == 13


def test_prepare_test_image_geometry():
    rng = np.random.default_rng(1)
    img = rng.integers(1, 256, (70, 100), dtype=np.uint8)
    out = O.prepare_test_image(img, encoder_ds=32, ds=2, board=10)
    assert out.shape == (96 // 2 + 20, 128 // 2 + 20) and out.dtype == np.uint8
    assert not out[:10].any() and not out[-10:].any() and not out[:, :10].any() and not out[:, -10:].any()
    # padding is on TOP (26 rows -> 13 rescaled rows, minus the blur reaching one row in) and RIGHT (28 -> 14 columns)
    assert not out[10:10 + 12, 10:-10].any()
    assert not out[10:-10, -10 - 13:-10].any()
    assert out[10 + 14:-10, 10:10 + 49].all()
    same = O.prepare_test_image(img, encoder_ds=2, ds=1, board=0)
    assert np.array_equal(same, img)
