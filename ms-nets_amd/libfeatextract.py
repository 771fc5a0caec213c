"""HIP-backed stand-in for ``src.cpp.lib.libfeatextract`` (featextract.cpp:529-553), restricted to the two
functions on the left-only test path: swap_axes (:49-76) and extract_likelihood(vol, sigma) (:415-462)."""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .libmatchers import _F32, _ret, _to_dev


def swap_axes(cost):
    """float32 [D, H, W] -> [H, W, D]."""
    t, npy = _to_dev(cost, _F32, "cost")
    if t.dim() != 3:
        raise ValueError("cost must be [D,H,W]")
    D, H, W = t.shape
    out = torch.empty((H, W, D), device=t.device, dtype=torch.float32)
    check(_lib.load().msnet_swap_axes(ptr(t), ptr(out), D, H, W, stream_ptr()), "msnet_swap_axes")
    return _ret(out, npy)


def get_right_cost(cost):
    """float32 [H, W, D] left cost -> right cost (featextract.cpp:136-172): res[i,j,d] = cost[i,j+d,d] where j+d < W, else the
    fill value cost[0,0,0]."""
    t, npy = _to_dev(cost, _F32, "cost")
    if t.dim() != 3:
        raise ValueError("cost must be [H,W,D]")
    H, W, D = t.shape
    out = torch.empty_like(t)
    check(_lib.load().msnet_get_right_cost(ptr(t), ptr(out), H, W, D, stream_ptr()), "msnet_get_right_cost")
    return _ret(out, npy)


def extract_likelihood(vol, sigma):
    """float32 [P, D] -> [P, D] per-row likelihood exp(-(c-min)^2/sigma) / sum (the 2-argument overload,
    featextract.cpp:546-549)."""
    t, npy = _to_dev(vol, _F32, "vol")
    if t.dim() != 2:
        raise ValueError("vol must be [P,D]")
    P, D = t.shape
    out = torch.empty((P, D), device=t.device, dtype=torch.float32)
    check(_lib.load().msnet_extract_likelihood(ptr(t), ptr(out), P, D, float(sigma), stream_ptr()),
          "msnet_extract_likelihood")
    return _ret(out, npy)
