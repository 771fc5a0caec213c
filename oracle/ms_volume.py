"""ORACLE (test infrastructure, not product code): NumPy/ctypes front-end of oracle/matchers_oracle.c with
the reference's Python glue restated.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this.  PARITY UNPINNED for the matcher half (see matchers_oracle.c header).

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
_lib = None
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
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
        _lib = L
    return _lib


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


def build_ms_volume(imgl_board, imgr_board, ndisp, board=10):
    """The test-time call sequence of generate_test_cbmv (cbmv_generator.py:826-839) on already
    down-sampled, bordered uint8 images: get_costs(..., 11,3,5,5, 10,10,10) -> extract_features_left."""
    c, n, so, sa = get_costs(imgl_board, imgr_board, ndisp, 11, 3, 5, 5, board, board, board)
    return extract_features_left(c, n, so, sa, 128.0, 0.02, 20000.0, 20000.0)
