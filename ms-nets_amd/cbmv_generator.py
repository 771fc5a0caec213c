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


class VolumeBuilder:
    """Fused build: owns the workspace for one (Hb, Wb, ndisp) shape so repeated calls allocate nothing."""

    def __init__(self, Hb, Wb, ndisp, device="cuda", params=None):
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

    def __call__(self, imgl, imgr, out=None):
        imgl = _lib.require_gpu_f32(imgl, "imgl", torch.uint8)
        imgr = _lib.require_gpu_f32(imgr, "imgr", torch.uint8)
        if tuple(imgl.shape) != (self.Hb, self.Wb) or imgl.shape != imgr.shape:
            raise ValueError("expected two [%d,%d] uint8 images" % (self.Hb, self.Wb))
        if out is None:
            out = torch.empty((8, self.nd, self.Hc, self.Wc), device=imgl.device, dtype=torch.float32)
        check(_lib.load().msnet_build_volume(ptr(imgl), ptr(imgr), self.Hb, self.Wb, self.nd, ctypes.byref(self.params),
                                             ptr(self.workspace), ptr(out), stream_ptr()), "msnet_build_volume")
        return out


def build_ms_volume(imgl_board, imgr_board, ndisp, params=None):
    """Two bordered uint8 images [Hb, Wb] (NumPy or GPU tensors) -> [8, ndisp, Hb-2*border, Wb-2*border]."""
    was_numpy = isinstance(imgl_board, np.ndarray)
    if was_numpy:
        if not torch.cuda.is_available():
            raise RuntimeError("build_ms_volume (HIP): no MI355X device visible and there is no CPU fallback")
        imgl_board = torch.from_numpy(np.ascontiguousarray(imgl_board)).cuda()
        imgr_board = torch.from_numpy(np.ascontiguousarray(imgr_board)).cuda()
    vb = VolumeBuilder(imgl_board.shape[0], imgl_board.shape[1], ndisp, imgl_board.device, params)
    out = vb(imgl_board, imgr_board)
    return out.cpu().numpy() if was_numpy else out
