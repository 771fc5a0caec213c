"""HIP-backed stand-in for the reference extension module ``src.cpp.lib.libmatchers``
(/root/reference/src/cpp/matchers/matchers.cpp:565-580): same function names, argument order and result
layouts.  NumPy in -> NumPy out (host round trip, like the reference), or torch GPU tensors in -> torch
GPU tensors out (device-resident pipelines).  No CPU implementation: without a GPU every call raises."""
import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

THREADS_NUM_USED = 8   # paramSetting.hpp:11; meaningless on the GPU, kept for initthreads()'s return value


def _to_dev(a, dtype, name):
    """-> (device tensor, was_numpy)"""
    if isinstance(a, np.ndarray):
        if a.dtype != dtype[0]:
            raise TypeError("%s must be %s (got %s)" % (name, dtype[0], a.dtype))
        if not torch.cuda.is_available():
            raise RuntimeError("libmatchers (HIP): no MI355X device visible and there is no CPU fallback")
        return torch.from_numpy(np.ascontiguousarray(a)).cuda(), True
    t = _lib.require_gpu_f32(a, name, dtype[1])
    return t, False


_U8 = (np.uint8, torch.uint8)
_F32 = (np.float32, torch.float32)


def _ret(t, was_numpy):
    return t.cpu().numpy() if was_numpy else t


def _pair(left, right, dt):
    l, npy = _to_dev(left, dt, "left")
    r, _ = _to_dev(right, dt, "right")
    if l.dim() != 2 or l.shape != r.shape:
        raise ValueError("left/right must be 2-D arrays of the same shape")
    return l, r, npy


def census(left, right, ndisp, wsize):
    """-> float32 [H, W, ndisp] (matchers.cpp:232-353)."""
    l, r, npy = _pair(left, right, _U8)
    H, W = l.shape
    lib = _lib.load()
    out = torch.empty((H, W, ndisp), device=l.device, dtype=torch.float32)
    ws = torch.empty(max(1, lib.msnet_census_workspace_bytes(H, W, wsize)), device=l.device, dtype=torch.uint8)
    check(lib.msnet_census(ptr(l), ptr(r), ptr(out), ptr(ws), H, W, ndisp, wsize, stream_ptr()), "msnet_census")
    return _ret(out, npy)


def nccNister(left, right, ndisp, wsize):
    """-> float32 [ndisp, H, W] (matchers.cpp:47-228)."""
    l, r, npy = _pair(left, right, _U8)
    H, W = l.shape
    out = torch.empty((ndisp, H, W), device=l.device, dtype=torch.float32)
    check(_lib.load().msnet_ncc(ptr(l), ptr(r), ptr(out), H, W, ndisp, wsize, stream_ptr()), "msnet_ncc")
    return _ret(out, npy)


def zsad(left, right, ndisp, wsize):
    """-> float32 [ndisp, H, W] (matchers.cpp:442-512)."""
    l, r, npy = _pair(left, right, _U8)
    H, W = l.shape
    out = torch.empty((ndisp, H, W), device=l.device, dtype=torch.float32)
    check(_lib.load().msnet_zsad(ptr(l), ptr(r), ptr(out), H, W, ndisp, wsize, stream_ptr()), "msnet_zsad")
    return _ret(out, npy)


def sobel(img):
    """-> float32 [H, W] (matchers.cpp:515-554)."""
    t, npy = _to_dev(img, _U8, "img")
    if t.dim() != 2:
        raise ValueError("img must be 2-D")
    H, W = t.shape
    out = torch.empty((H, W), device=t.device, dtype=torch.float32)
    check(_lib.load().msnet_sobel(ptr(t), ptr(out), H, W, stream_ptr()), "msnet_sobel")
    return _ret(out, npy)


def sadsob(left, right, ndisp, wsize):
    """left/right: float32 Sobel images -> float32 [ndisp, H, W] (matchers.cpp:356-438)."""
    l, r, npy = _pair(left, right, _F32)
    H, W = l.shape
    lib = _lib.load()
    out = torch.empty((ndisp, H, W), device=l.device, dtype=torch.float32)
    ws = torch.empty(max(1, lib.msnet_sadsob_workspace_bytes(H, W, ndisp)), device=l.device, dtype=torch.uint8)
    check(lib.msnet_sadsob(ptr(l), ptr(r), ptr(out), ptr(ws), H, W, ndisp, wsize, stream_ptr()), "msnet_sadsob")
    return _ret(out, npy)


def initthreads():
    """matchers.cpp:556-563 returns the OpenMP team size it could spin up; the HIP library needs no thread
    pool, so this only proves the library loads."""
    _lib.load()
    return THREADS_NUM_USED
