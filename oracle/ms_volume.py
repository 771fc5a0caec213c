"""ORACLE (test infrastructure, not product code): NumPy/ctypes front-end of oracle/matchers_oracle.c with
the reference's Python glue restated.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this.  PARITY UNPINNED for the natives (census ... extract_likelihood: see matchers_oracle.c header).
The Python glue below (get_costs, extract_features_left, extract_features_lr, build_ms_volume) IS pinned, bit for bit, to
the reference's own cbmv_generator.py run around these natives: tests/golden/volume_*.npz, made by
tests/golden/make_volume_golden.py, checked by tests/test_golden_volume.py.

Restated reference functions (paths under /root/reference/src):
  libmatchers.census / nccNister / zsad / sobel / sadsob  cpp/matchers/matchers.cpp:565-580
  libfeatextract.swap_axes / extract_likelihood           cpp/featextract/featextract.cpp:529-553
  get_costs                                               dataloader/cbmv_generator.py:27-79
  extract_features_left                                   dataloader/cbmv_generator.py:258-308
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_matchers.so")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_libs = {}
_variant = [""]


class variant:
    """`with variant("refflags"):` routes every call below through liboracle_matchers_refflags.so -- the same source compiled
    with the reference's own optimisation flags (Makefile) -- so a test can hold the two builds bit-identical."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        _variant.append(self.name)

    def __exit__(self, *exc):
        _variant.pop()


def lib():
    v = _variant[-1]
    if v not in _libs:
        so = _SO if not v else _SO.replace(".so", "_%s.so" % v)
        if not os.path.exists(so):
            subprocess.check_call(["make", "-s", "-C", _HERE, os.path.basename(so)])     # a variant is built on demand
        L = ctypes.CDLL(so)
        i = ctypes.c_int
        L.oracle_census.argtypes = [_u8p, _u8p, _f32p, i, i, i, i]
        L.oracle_ncc.argtypes = [_u8p, _u8p, _f32p, i, i, i, i]
        L.oracle_zsad.argtypes = [_u8p, _u8p, _f32p, i, i, i, i]
        L.oracle_sobel.argtypes = [_u8p, _f32p, i, i]
        L.oracle_sadsob.argtypes = [_f32p, _f32p, _f32p, i, i, i, i]
        L.oracle_swap_axes.argtypes = [_f32p, _f32p, i, i, i]
        L.oracle_extract_likelihood.argtypes = [_f32p, _f32p, ctypes.c_long, i, ctypes.c_float]
        for f in ("oracle_census", "oracle_ncc", "oracle_zsad", "oracle_sobel", "oracle_sadsob", "oracle_swap_axes",
                  "oracle_extract_likelihood"):
            getattr(L, f).restype = None
        _libs[v] = L
    return _libs[v]


def _img(a):
    a = np.ascontiguousarray(a)
    assert a.dtype == np.uint8 and a.ndim == 2
    return a


def census(left, right, ndisp, wsize):
    left, right = _img(left), _img(right)
    H, W = left.shape
    out = np.empty((H, W, ndisp), np.float32)
    lib().oracle_census(left, right, out, H, W, ndisp, wsize)
    return out


def nccNister(left, right, ndisp, wsize):
    left, right = _img(left), _img(right)
    H, W = left.shape
    out = np.empty((ndisp, H, W), np.float32)
    lib().oracle_ncc(left, right, out, H, W, ndisp, wsize)
    return out


def zsad(left, right, ndisp, wsize):
    left, right = _img(left), _img(right)
    H, W = left.shape
    out = np.empty((ndisp, H, W), np.float32)
    lib().oracle_zsad(left, right, out, H, W, ndisp, wsize)
    return out


def sobel(img):
    img = _img(img)
    out = np.empty(img.shape, np.float32)
    lib().oracle_sobel(img, out, img.shape[0], img.shape[1])
    return out


def sadsob(sobl, sobr, ndisp, wsize):
    sobl = np.ascontiguousarray(sobl, np.float32)
    sobr = np.ascontiguousarray(sobr, np.float32)
    H, W = sobl.shape
    out = np.empty((ndisp, H, W), np.float32)
    lib().oracle_sadsob(sobl, sobr, out, H, W, ndisp, wsize)
    return out


def swap_axes(cost):
    cost = np.ascontiguousarray(cost, np.float32)
    D, H, W = cost.shape
    out = np.empty((H, W, D), np.float32)
    lib().oracle_swap_axes(cost, out, D, H, W)
    return out


def extract_likelihood(vol, sigma):
    vol = np.ascontiguousarray(vol, np.float32)
    P, D = vol.shape
    out = np.empty((P, D), np.float32)
    lib().oracle_extract_likelihood(vol, out, P, D, float(sigma))
    return out


def get_costs(iml, imr, maxdisp=192, censw=11, nccw=3, sadw=5, sobelw=5, board_h=10, board_w_left=10, board_w_right=0):
    """cbmv_generator.py:27-79: four raw costs on the bordered grid, cropped, each [H', W', ndisp] float32.
    Return order is the reference's: census, ncc, sobel-SAD, zsad."""
    costcensus = census(iml, imr, maxdisp, censw).astype(np.float32)
    costncc = swap_axes(nccNister(iml, imr, maxdisp, nccw).astype(np.float32))
    costsad = swap_axes(zsad(iml, imr, maxdisp, sadw).astype(np.float32))
    costsob = swap_axes(sadsob(sobel(iml), sobel(imr), maxdisp, sobelw).astype(np.float32))
    w_end = -board_w_right if board_w_right > 0 else None
    h_end = -board_h if board_h > 0 else None
    crop = lambda c: c[board_h:h_end, board_w_left:w_end, :].copy(order="C")   # noqa: E731
    return crop(costcensus), crop(costncc), crop(costsob), crop(costsad)


def extract_features_left(census_c, ncc_c, sobel_c, sad_c, cens_sigma=128.0, ncc_sigma=0.02, sad_sigma=20000.0,
                          sobel_sigma=20000.0):
    """cbmv_generator.py:258-308: 4 clipped/normalised costs + 4 AML channels -> [8, ndisp, H', W'] float32.
    float32 arithmetic stored through a float64 scratch (:281) and cast back (:308) -- value-preserving.
    sobel_sigma is accepted and ignored, as in the reference (:298,303 use sad_sigma for the Sobel channel)."""
    h, w, nd = census_c.shape
    flat = lambda a: np.reshape(a, [h * w, nd])   # noqa: E731
    census_c, ncc_c, sobel_c, sad_c = flat(census_c), flat(ncc_c), flat(sobel_c), flat(sad_c)
    feats = np.empty((8, h, w, nd), order="C")
    feats[0] = np.reshape(np.clip(census_c, 0., 120.) / 120., [h, w, nd])
    feats[1] = np.reshape((1 + np.clip(ncc_c, -1., 1.)) / 2, [h, w, nd])
    feats[2] = np.reshape(np.clip(sobel_c, 0., 2 ** 13) / float(2 ** 13), [h, w, nd])
    feats[3] = np.reshape(np.clip(sad_c, 0., 2 ** 13) / float(2 ** 13), [h, w, nd])
    feats[4] = np.reshape(extract_likelihood(census_c, cens_sigma), [h, w, nd])
    feats[5] = np.reshape(extract_likelihood(ncc_c, ncc_sigma), [h, w, nd])
    feats[6] = np.reshape(extract_likelihood(sobel_c, sad_sigma), [h, w, nd])
    feats[7] = np.reshape(extract_likelihood(sad_c, sad_sigma), [h, w, nd])
    return feats.transpose((0, 3, 1, 2)).astype(np.float32)


def get_right_cost(cost):
    """featextract.cpp:136-172: res[i,j,d] = cost[i,j+d,d] for j < W-d, everything else = cost[0,0,0]."""
    cost = np.ascontiguousarray(cost, np.float32)
    h, w, nd = cost.shape
    res = np.full_like(cost, cost[0, 0, 0])
    for d in range(nd):
        res[:, : w - d, d] = cost[:, d:, d]
    return res


def extract_features_lr(census_c, ncc_c, sobel_c, sad_c, cens_sigma=128.0, ncc_sigma=0.02, sad_sigma=20000.0,
                        sobel_sigma=20000.0):
    """cbmv_generator.py:84-254: channels 0-7 = the left features, 8-15 = the same features of get_right_cost(cost)."""
    left = extract_features_left(census_c, ncc_c, sobel_c, sad_c, cens_sigma, ncc_sigma, sad_sigma, sobel_sigma)
    right = extract_features_left(get_right_cost(census_c), get_right_cost(ncc_c), get_right_cost(sobel_c),
                                  get_right_cost(sad_c), cens_sigma, ncc_sigma, sad_sigma, sobel_sigma)
    return np.concatenate([left, right], axis=0)


def build_ms_volume(imgl_board, imgr_board, ndisp, board=10):
    """The test-time call sequence of generate_test_cbmv (cbmv_generator.py:826-839) on already
    down-sampled, bordered uint8 images: get_costs(..., 11,3,5,5, 10,10,10) -> extract_features_left."""
    c, n, so, sa = get_costs(imgl_board, imgr_board, ndisp, 11, 3, 5, 5, board, board, board)
    return extract_features_left(c, n, so, sa, 128.0, 0.02, 20000.0, 20000.0)


# ---------------------------------------------------------------------------------------------------------------
# Test-time pre-processing: cbmv_generator.py:780-788 (pad to a multiple of encoder_ds on the
# TOP and RIGHT), :465-482 (down_sampling_input = skimage.transform.rescale(img/255, 1/s, anti_aliasing=True,
# mode='constant', preserve_range=True) * 255 -> uint8) and :819-823 (10-px zero border).
#
# PARITY UNPINNED for the rescale: scikit-image is not installed here and the reference holds no vectors for it.  The
# restatement below follows skimage.transform.resize as published (0.16-0.19): anti-aliasing = scipy.ndimage
# gaussian_filter with sigma = (s-1)/2, mode='constant', cval=0 (truncate 4.0), then an order-1 resample at the
# pixel-centre-aligned coordinates (x + 0.5)*s - 0.5 (for integer s: the mean of the two central source pixels per
# axis, or the centre pixel for odd s), then clipping to the input's value range (cval included).  Two forms:
#   rescale_scipy  : literally those scipy.ndimage calls (gaussian_filter + zoom(order=1, grid_mode=True))
#   rescale_explicit: the same arithmetic spelled out (double accumulation in scipy's correlate1d order, float32
#                     between passes) -- what the HIP kernel implements; tests require the two to agree bit for bit.
# ---------------------------------------------------------------------------------------------------------------
def gaussian_weights(s):
    """scipy.ndimage._gaussian_kernel1d(sigma=(s-1)/2, order 0, radius=int(4*sigma+0.5)) in double."""
    sigma = (s - 1) / 2.0
    radius = int(4.0 * sigma + 0.5)
    if sigma <= 0:
        return np.ones(1, np.float64)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum()


def _correlate_sym(a, w, axis):
    """scipy ni_filters.c NI_Correlate1D, symmetric branch, mode='constant' cval 0, float32 output:
    tmp = x[0]*w[c]; for j = -r..-1: tmp += (x[j] + x[-j]) * w[c+j]   (all in double)."""
    r = len(w) // 2
    a64 = np.moveaxis(a.astype(np.float64), axis, -1)
    pad = np.zeros(a64.shape[:-1] + (r,), np.float64)
    p = np.concatenate([pad, a64, pad], axis=-1)
    n = a64.shape[-1]
    tmp = p[..., r:r + n] * w[r]
    for j in range(-r, 0):
        tmp = tmp + (p[..., r + j:r + j + n] + p[..., r - j:r - j + n]) * w[r + j]
    return np.moveaxis(tmp.astype(np.float32), -1, axis)


def rescale_explicit(img_u8, s):
    """uint8 [H, W] (H, W multiples of s) -> uint8 [H/s, W/s], see the block comment above."""
    s = int(s)
    img_u8 = np.ascontiguousarray(img_u8, dtype=np.uint8)
    if s == 1:
        return img_u8.copy()
    H, W = img_u8.shape
    assert H % s == 0 and W % s == 0
    x = img_u8.astype(np.float32) / np.float32(255.0)
    w = gaussian_weights(s)
    f = _correlate_sym(_correlate_sym(x, w, 0), w, 1)           # gaussian_filter: axis 0 then axis 1
    lo = (s - 1) // 2                                           # floor((x+0.5)*s - 0.5) - x*s
    if s % 2 == 0:                                              # half-integer coordinate: mean of two pixels per axis
        f64 = f.astype(np.float64)
        rows = [f64[lo::s], f64[lo + 1::s]]
        acc = np.zeros((H // s, W // s), np.float64)
        for rr in rows:
            acc += 0.25 * rr[:, lo::s] + 0.25 * rr[:, lo + 1::s]
        out = acc.astype(np.float32)
    else:
        out = f[lo::s, lo::s].copy()
    vmax = np.float32(x.max())
    out = np.minimum(np.maximum(out, np.float32(0.0)), vmax)    # _clip_warp_output (cval = 0 joins the range)
    return (out * np.float32(255.0)).astype(np.uint8)


def rescale_scipy(img_u8, s):
    from scipy import ndimage as ndi
    s = int(s)
    if s == 1:
        return np.ascontiguousarray(img_u8, dtype=np.uint8).copy()
    x = np.ascontiguousarray(img_u8, dtype=np.uint8).astype(np.float32) / np.float32(255.0)
    sigma = (s - 1) / 2.0
    f = ndi.gaussian_filter(x, (sigma, sigma), cval=0, mode="constant")
    out = ndi.zoom(f, (1.0 / s, 1.0 / s), order=1, mode="grid-constant", cval=0, grid_mode=True)
    out = np.clip(out, min(float(x.min()), 0.0), float(x.max())).astype(np.float32)
    return (out * np.float32(255.0)).astype(np.uint8)


def prepare_test_image(img_u8, encoder_ds=32, ds=2, board=10):
    """cbmv_generator.py:780-788 + :811-812 + :819-823 for one grayscale image: pad (top, right) to a multiple of
    encoder_ds, rescale by 1/ds, add a `board`-pixel zero border."""
    img_u8 = np.ascontiguousarray(img_u8, dtype=np.uint8)
    h, w = img_u8.shape
    pad_w = (encoder_ds - w % encoder_ds) % encoder_ds
    pad_h = (encoder_ds - h % encoder_ds) % encoder_ds
    p = np.pad(img_u8, ((pad_h, 0), (0, pad_w)), "constant")
    p = rescale_explicit(p, ds)
    return np.pad(p, ((board, board), (board, board)), "constant").astype(np.uint8).copy(order="C")
