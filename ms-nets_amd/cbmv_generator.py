"""Cost-volume assembly on the GPU, mirroring the two functions of the reference's
src/dataloader/cbmv_generator.py that sit on the hot path:

  get_costs(iml, imr, maxdisp, censw, nccw, sadw, sobelw, board_h, board_w_left, board_w_right)   :27-79
  extract_features_left(census, ncc, sobel, sad, cens_sigma, ncc_sigma, sad_sigma, sobel_sigma)  :258-308

plus the fused entry point the reference does not have: build_ms_volume(imgl_board, imgr_board, ndisp),
which goes from the two bordered uint8 images to the [8, D', H', W'] float32 volume in one C-ABI call
without leaving the device.
"""
import ctypes

import numpy as np
import torch

from . import _lib, libfeatextract as fte, libmatchers as mtc
from ._lib import check, ptr, stream_ptr


def get_default_args_dict():
    """The hot-path constants of cbmv_generator.py:434-462."""
    return dict(censw=11, nccw=3, sadw=5, sobelw=5, cens_sigma=128.0, ncc_sigma=0.02, sad_sigma=20000.0,
                sobel_sigma=20000.0, cbmv_F=8, ds_scale=2, board_h=10)


def _crop(c, board_h, board_w_left, board_w_right):
    w_end = -board_w_right if board_w_right > 0 else None
    h_end = -board_h if board_h > 0 else None
    c = c[board_h:h_end, board_w_left:w_end, :]
    return c.copy(order="C") if isinstance(c, np.ndarray) else c.contiguous()


def get_costs(iml, imr, maxdisp=192, censw=11, nccw=3, sadw=5, sobelw=5, board_h=10, board_w_left=10, board_w_right=0):
    """Four raw matching costs, each [H', W', ndisp] float32, returned as (census, ncc, sobel, sad)."""
    costcensus = mtc.census(iml, imr, maxdisp, censw)
    costncc = fte.swap_axes(mtc.nccNister(iml, imr, maxdisp, nccw))
    costsad = fte.swap_axes(mtc.zsad(iml, imr, maxdisp, sadw))
    costsob = fte.swap_axes(mtc.sadsob(mtc.sobel(iml), mtc.sobel(imr), maxdisp, sobelw))
    return tuple(_crop(c, board_h, board_w_left, board_w_right) for c in (costcensus, costncc, costsob, costsad))


def extract_features_left(census, ncc, sobel, sad, cens_sigma=128.0, ncc_sigma=0.02, sad_sigma=20000.0,
                          sobel_sigma=20000.0, disp_image=None):
    """-> [8, ndisp, H', W'] float32.  One C-ABI call (normalisation, likelihood and the [8,D,H,W] transpose all on the
    device; torch's GPU divide-by-scalar is a reciprocal multiply and would not be bit-faithful).  As in the
    reference the Sobel channel's likelihood uses sad_sigma and sobel_sigma is ignored (:298,303)."""
    was_numpy = isinstance(census, np.ndarray)
    if was_numpy:
        census, ncc, sobel, sad = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (census, ncc, sobel, sad))
    h, w, nd = census.shape
    ins = [_lib.require_gpu_f32(a, nm).reshape(h * w, nd) for a, nm in
           ((census, "census"), (ncc, "ncc"), (sobel, "sobel"), (sad, "sad"))]
    out = torch.empty((8, nd, h, w), device=ins[0].device, dtype=torch.float32)
    check(_lib.load().msnet_extract_features_left(ptr(ins[0]), ptr(ins[1]), ptr(ins[2]), ptr(ins[3]), ptr(out), h * w, nd,
                                                  float(cens_sigma), float(ncc_sigma), float(sad_sigma), stream_ptr()),
          "msnet_extract_features_left")
    return out.cpu().numpy() if was_numpy else out


def extract_features_lr(census, ncc, sobel, sad, cens_sigma=128.0, ncc_sigma=0.02, sad_sigma=20000.0, sobel_sigma=20000.0,
                        disp_image=None):
    """cbmv_generator.py:84-254 (the is_left_only=False branch of generate_test_cbmv): -> [16, ndisp, H', W'] float32,
    channels 0-7 as extract_features_left, 8-15 the same eight features of the right costs (get_right_cost of each
    cropped left cost).  Four re-indexing launches + two feature launches; costs stay on the device."""
    was_numpy = isinstance(census, np.ndarray)
    if was_numpy:
        census, ncc, sobel, sad = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (census, ncc, sobel, sad))
    left = extract_features_left(census, ncc, sobel, sad, cens_sigma, ncc_sigma, sad_sigma, sobel_sigma)
    right = extract_features_left(*(fte.get_right_cost(c) for c in (census, ncc, sobel, sad)),
                                  cens_sigma, ncc_sigma, sad_sigma, sobel_sigma)
    out = torch.cat([left, right], dim=0)
    return out.cpu().numpy() if was_numpy else out


class VolumeBuilder:
    """Fused build: owns the workspace for one (Hb, Wb, ndisp) shape so repeated calls allocate nothing.
    layout "ncdhw" (default): out [8, D', H', W'] -- the reference's layout (cbmv_generator.py:307-308), the drop-in.
    layout "ndhwc": out [D', H', W', 8] -- the aggregator kernels' own layout, for GCNet_CostVolumeAggre.forward_ndhwc: the
    802 MB layout pass between build and aggregator disappears.  Same values bit for bit.  Shapes / parameters the
    channels-last kernel does not take (msnet_build_volume_ndhwc_supported) are built NCDHW and converted."""

    def __init__(self, Hb, Wb, ndisp, device="cuda", params=None, layout="ncdhw"):
        if layout not in ("ncdhw", "ndhwc"):
            raise ValueError("layout must be 'ncdhw' or 'ndhwc'")
        self.layout = layout
        lib = _lib.load()
        self.Hb, self.Wb, self.nd = int(Hb), int(Wb), int(ndisp)
        self.params = _lib.VolumeParams()
        lib.msnet_volume_default_params(ctypes.byref(self.params))
        for k, v in (params or {}).items():
            setattr(self.params, k, v)
        self.Hc = self.Hb - 2 * self.params.border_h
        self.Wc = self.Wb - 2 * self.params.border_w
        nbytes = lib.msnet_build_volume_workspace_bytes(self.Hb, self.Wb, self.nd)
        self.workspace = torch.empty(max(1, nbytes), device=device, dtype=torch.uint8)
        self.native_cl = bool(layout == "ndhwc" and lib.msnet_build_volume_ndhwc_supported(self.Hb, self.Wb, self.nd,
                                                                                             ctypes.byref(self.params)))
        self._tmp = None

    @property
    def out_shape(self):
        return (self.nd, self.Hc, self.Wc, 8) if self.layout == "ndhwc" else (8, self.nd, self.Hc, self.Wc)

    def __call__(self, imgl, imgr, out=None):
        imgl = _lib.require_gpu_f32(imgl, "imgl", torch.uint8)
        imgr = _lib.require_gpu_f32(imgr, "imgr", torch.uint8)
        if tuple(imgl.shape) != (self.Hb, self.Wb) or imgl.shape != imgr.shape:
            raise ValueError("expected two [%d,%d] uint8 images" % (self.Hb, self.Wb))
        if out is None:
            out = torch.empty(self.out_shape, device=imgl.device, dtype=torch.float32)
        elif tuple(out.shape) != self.out_shape or not out.is_contiguous() or out.dtype != torch.float32:
            raise ValueError("out must be a contiguous float32 %s tensor" % (self.out_shape,))
        if self.native_cl:
            check(_lib.load().msnet_build_volume_ndhwc(ptr(imgl), ptr(imgr), self.Hb, self.Wb, self.nd, ctypes.byref(self.params),
                                                       ptr(self.workspace), ptr(out), stream_ptr()), "msnet_build_volume_ndhwc")
            return out
        dst = out
        if self.layout == "ndhwc":          # not taken channels-last by the kernel: NCDHW build + one conversion
            if self._tmp is None:
                self._tmp = torch.empty((8, self.nd, self.Hc, self.Wc), device=imgl.device, dtype=torch.float32)
            dst = self._tmp
        check(_lib.load().msnet_build_volume(ptr(imgl), ptr(imgr), self.Hb, self.Wb, self.nd, ctypes.byref(self.params),
                                             ptr(self.workspace), ptr(dst), stream_ptr()), "msnet_build_volume")
        if dst is not out:
            check(_lib.load().msnet_ncdhw_to_ndhwc(ptr(dst), ptr(out), 1, 8, self.nd, self.Hc, self.Wc, stream_ptr()),
                  "msnet_ncdhw_to_ndhwc")
        return out


def build_ms_volume(imgl_board, imgr_board, ndisp, params=None, layout="ncdhw"):
    """Two bordered uint8 images [Hb, Wb] (NumPy or GPU tensors) -> [8, ndisp, Hb-2*border, Wb-2*border]
    (layout="ndhwc": [ndisp, H', W', 8], see VolumeBuilder)."""
    was_numpy = isinstance(imgl_board, np.ndarray)
    if was_numpy:
        if not torch.cuda.is_available():
            raise RuntimeError("build_ms_volume (HIP): no MI355X device visible and there is no CPU fallback")
        imgl_board = torch.from_numpy(np.ascontiguousarray(imgl_board)).cuda()
        imgr_board = torch.from_numpy(np.ascontiguousarray(imgr_board)).cuda()
    vb = VolumeBuilder(imgl_board.shape[0], imgl_board.shape[1], ndisp, imgl_board.device, params, layout=layout)
    out = vb(imgl_board, imgr_board)
    return out.cpu().numpy() if was_numpy else out


# ---- test-time pre-processing on the device -------------------------------------------------------------
def _gaussian_taps(ds):
    """The anti-aliasing taps exactly as scipy.ndimage._gaussian_kernel1d builds them for sigma = (ds-1)/2, truncate 4."""
    sigma = (ds - 1) / 2.0
    radius = int(4.0 * sigma + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return np.ascontiguousarray(phi / phi.sum(), dtype=np.float64)


def prepare_test_image(img, encoder_ds=32, ds=2, board=10):
    """generate_test_cbmv's image preparation (cbmv_generator.py:780-788, 811-812, 819-823) for one grayscale uint8
    image [h, w] (NumPy or GPU tensor): pad top/right to a multiple of `encoder_ds`, rescale by 1/ds with
    anti-aliasing (down_sampling_input), add a `board`-pixel zero border.  Returns a GPU uint8 tensor [Hb, Wb]."""
    lib = _lib.load()
    if isinstance(img, np.ndarray):
        if img.dtype != np.uint8 or img.ndim != 2:
            raise TypeError("img must be a 2-D uint8 array")
        if not torch.cuda.is_available():
            raise RuntimeError("prepare_test_image (HIP): no MI355X device visible and there is no CPU fallback")
        img = torch.from_numpy(np.ascontiguousarray(img)).cuda()
    img = _lib.require_gpu_f32(img, "img", torch.uint8)
    if img.dim() != 2:
        raise ValueError("img must be [h, w]")
    h, w = int(img.shape[0]), int(img.shape[1])
    hb, wb = ctypes.c_int(0), ctypes.c_int(0)
    check(lib.msnet_preprocess_out_shape(h, w, int(encoder_ds), int(ds), int(board), ctypes.byref(hb), ctypes.byref(wb)),
          "msnet_preprocess_out_shape")
    out = torch.empty((hb.value, wb.value), device=img.device, dtype=torch.uint8)
    ws = torch.empty(4, device=img.device, dtype=torch.uint8)
    taps = _gaussian_taps(int(ds)) if int(ds) > 1 else np.ones(1, np.float64)
    check(lib.msnet_preprocess_image(ptr(img), h, w, int(encoder_ds), int(ds), int(board),
                                     taps.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ptr(out), ptr(ws), stream_ptr()),
          "msnet_preprocess_image")
    return out


def down_sampling_input(ds_scale, imgl, imgr, anti_aliasing=True, multichannel=False, preserve_range=True):
    """cbmv_generator.py:465-482 on the device: both images rescaled by `ds_scale` (= 1/integer).  NumPy in, NumPy out,
    like the reference; GPU tensors in, GPU tensors out."""
    if not anti_aliasing or multichannel or not preserve_range:
        raise NotImplementedError("only the reference's own call (anti_aliasing, single channel, preserve_range) is built")
    ds = int(round(1.0 / float(ds_scale)))
    if ds < 1 or abs(ds * float(ds_scale) - 1.0) > 1e-9:
        raise ValueError("ds_scale must be 1/integer")
    outs = []
    for im in (imgl, imgr):
        if im.shape[0] % ds or im.shape[1] % ds:
            raise ValueError("image size %s is not a multiple of %d (the reference pads to encoder_ds first)" % (tuple(im.shape), ds))
        was_numpy = isinstance(im, np.ndarray)
        o = prepare_test_image(im, encoder_ds=ds, ds=ds, board=0)
        outs.append(o.cpu().numpy() if was_numpy else o)
    return outs[0], outs[1]


def generate_test_cbmv_from_images(imgl, imgr, encoder_ds=32, maxdisp=192, args_dict=None):
    """generate_test_cbmv (cbmv_generator.py:726-845) from already-decoded grayscale images, left features only:
    pre-processing and the 8-channel volume on the device.  Returns (features [8, D', H', W'] GPU tensor, (pad_h, pad_w))."""
    args = get_default_args_dict() if args_dict is None else args_dict
    ds = int(args.get("ds_scale", 2))
    h, w = int(imgl.shape[0]), int(imgl.shape[1])
    lb = prepare_test_image(imgl, encoder_ds, ds, 10)
    rb = prepare_test_image(imgr, encoder_ds, ds, 10)
    params = dict(censw=args["censw"], nccw=args["nccw"], sadw=args["sadw"], sobelw=args["sobelw"],
                  cens_sigma=args["cens_sigma"], ncc_sigma=args["ncc_sigma"], sad_sigma=args["sad_sigma"])
    vol = build_ms_volume(lb, rb, maxdisp // ds, params=params)
    return vol, ((encoder_ds - h % encoder_ds) % encoder_ds, (encoder_ds - w % encoder_ds) % encoder_ds)
