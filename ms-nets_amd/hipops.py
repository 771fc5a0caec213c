"""Tensor-level wrappers over the C ABI for the aggregator kernels.  PyTorch is used only for device
memory (torch.empty) and the current HIP stream; all arithmetic happens in libmsnet_hip.so.

Internal activation layout is channels-last fp32 [N, D, H, W, C] ("NDHWC")."""
import torch

from . import _lib
from ._lib import check, ptr, require_gpu_f32, stream_ptr


def ncdhw_to_ndhwc(x):
    x = require_gpu_f32(x, "x")
    n, c, d, h, w = x.shape
    y = torch.empty((n, d, h, w, c), device=x.device, dtype=torch.float32)
    check(_lib.load().msnet_ncdhw_to_ndhwc(ptr(x), ptr(y), n, c, d, h, w, stream_ptr()), "msnet_ncdhw_to_ndhwc")
    return y


def ndhwc_to_ncdhw(x):
    x = require_gpu_f32(x, "x")
    n, d, h, w, c = x.shape
    y = torch.empty((n, c, d, h, w), device=x.device, dtype=torch.float32)
    check(_lib.load().msnet_ndhwc_to_ncdhw(ptr(x), ptr(y), n, c, d, h, w, stream_ptr()), "msnet_ndhwc_to_ncdhw")
    return y


def pack_conv_weight(w, transposed=False, f16s=False, stride=1):
    """nn.Conv3d.weight [Co,Ci,3,3,3] (or ConvTranspose3d.weight [Ci,Co,3,3,3]) -> MFMA-ordered buffer
    (fp32 lane order, or the split-fp16 hi/lo image when f16s)."""
    w = require_gpu_f32(w, "weight")
    if tuple(w.shape[2:]) != (3, 3, 3):
        raise ValueError("only 3x3x3 kernels are built (got %s)" % (tuple(w.shape),))
    ci, co = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
    lib = _lib.load()
    out = torch.empty(lib.msnet_packed_weight_floats(max(ci, 16) if f16s else ci, co), device=w.device,
                      dtype=torch.float32)
    if f16s and transposed:
        check(lib.msnet_pack_deconv_weight_f16s(ptr(w), ptr(out), ci, co, stream_ptr()), "msnet_pack_deconv_weight_f16s")
        return out
    if f16s:
        check(lib.msnet_pack_conv_weight_f16s(ptr(w), ptr(out), ci, co, int(stride), stream_ptr()),
              "msnet_pack_conv_weight_f16s")
        return out
    fn = lib.msnet_pack_deconv_weight if transposed else lib.msnet_pack_conv_weight
    check(fn(ptr(w), ptr(out), ci, co, stream_ptr()), "msnet_pack_weight")
    return out


PRECISIONS = ("fp32", "split-fp16")
_default_precision = "split-fp16"
# Transposed convs with a split-fp16 kernel use it: Ci = 64 on the tiled kernel (1.22 vs 1.99 ms on deconvbn4, 0.28 vs
# 0.84 ms on deconvbn3, profiles/r01l_*), Ci = 32 / 128 and every small layer on the direct kernel (deconvbn1: 0.04 vs 0.14 ms).
USE_F16S_DECONV = True


def set_default_precision(p):
    """'fp32': exact fp32-input MFMA everywhere.  'split-fp16': layers that have a split-fp16 kernel use it
    (3 fp16 MFMAs per product, 22-bit operands), the rest stay on the fp32 MFMA."""
    global _default_precision
    if p not in PRECISIONS:
        raise ValueError("precision must be one of %s" % (PRECISIONS,))
    _default_precision = p


def get_default_precision():
    return _default_precision


def conv3d_k3(x, wpk, scale, shift, co, stride=1, relu=False, residual=None, f16s=False):
    x = require_gpu_f32(x, "x")
    n, d, h, w, ci = x.shape
    od, oh, ow = (d - 1) // stride + 1, (h - 1) // stride + 1, (w - 1) // stride + 1
    y = torch.empty((n, od, oh, ow, co), device=x.device, dtype=torch.float32)
    if residual is not None:
        residual = require_gpu_f32(residual, "residual")
        if residual.shape != y.shape:
            raise ValueError("residual shape %s != output shape %s" % (tuple(residual.shape), tuple(y.shape)))
    fn = _lib.load().msnet_conv3d_k3_f16s if f16s else _lib.load().msnet_conv3d_k3
    check(fn(ptr(x), ptr(wpk), ptr(scale), ptr(shift), ptr(residual), ptr(y), n, d, h, w, ci, co, stride, int(relu),
             stream_ptr()), "msnet_conv3d_k3_f16s" if f16s else "msnet_conv3d_k3")
    return y


def deconv3d_k3s2(x, wpk, scale, shift, co, relu=False, residual=None, f16s=False):
    x = require_gpu_f32(x, "x")
    n, d, h, w, ci = x.shape
    y = torch.empty((n, 2 * d, 2 * h, 2 * w, co), device=x.device, dtype=torch.float32)
    if residual is not None:
        residual = require_gpu_f32(residual, "residual")
        if residual.shape != y.shape:
            raise ValueError("residual shape %s != output shape %s" % (tuple(residual.shape), tuple(y.shape)))
    fn = _lib.load().msnet_deconv3d_k3s2_f16s if f16s else _lib.load().msnet_deconv3d_k3s2
    check(fn(ptr(x), ptr(wpk), ptr(scale), ptr(shift), ptr(residual), ptr(y), n, d, h, w, ci, co, int(relu), stream_ptr()),
          "msnet_deconv3d_k3s2_f16s" if f16s else "msnet_deconv3d_k3s2")
    return y


def conv3d_k3_cout1(x, w, add=None):
    """Conv3d(Ci->1) head.  x NDHWC, w [1,Ci,3,3,3] -> [N,D,H,W]."""
    x = require_gpu_f32(x, "x")
    w = require_gpu_f32(w, "weight")
    n, d, h, wd, ci = x.shape
    y = torch.empty((n, d, h, wd), device=x.device, dtype=torch.float32)
    if add is not None:
        add = require_gpu_f32(add, "add")
        if add.shape != y.shape:
            raise ValueError("add shape mismatch")
    check(_lib.load().msnet_conv3d_k3_cout1(ptr(x), ptr(w), ptr(add), ptr(y), n, d, h, wd, ci, stream_ptr()),
          "msnet_conv3d_k3_cout1")
    return y


def softargmin(logits):
    logits = require_gpu_f32(logits, "logits")
    n, d, h, w = logits.shape
    disp = torch.empty((n, h, w), device=logits.device, dtype=torch.float32)
    check(_lib.load().msnet_softargmin(ptr(logits), ptr(disp), n, d, h, w, stream_ptr()), "msnet_softargmin")
    return disp


def deconv5_softargmin(x, w, bias):
    """Fused ConvTranspose3d(Ci->1,k3,s2,p1,op1,bias) + softmax(D) + sum d*p.  x NDHWC -> [N,2H,2W]."""
    x = require_gpu_f32(x, "x")
    w = require_gpu_f32(w, "weight")
    n, d, h, wd, ci = x.shape
    disp = torch.empty((n, 2 * h, 2 * wd), device=x.device, dtype=torch.float32)
    check(_lib.load().msnet_deconv5_softargmin(ptr(x), ptr(w), float(bias), ptr(disp), n, d, h, wd, ci, stream_ptr()),
          "msnet_deconv5_softargmin")
    return disp


def deconv3d_cout1(x, w, bias, stride=2):
    x = require_gpu_f32(x, "x")
    w = require_gpu_f32(w, "weight")
    n, d, h, wd, ci = x.shape
    y = torch.empty((n, stride * d, stride * h, stride * wd), device=x.device, dtype=torch.float32)
    check(_lib.load().msnet_deconv3d_cout1(ptr(x), ptr(w), float(bias), ptr(y), n, d, h, wd, ci, stride, stream_ptr()),
          "msnet_deconv3d_cout1")
    return y


def trilinear_softargmin(cost, out_dhw):
    cost = require_gpu_f32(cost, "cost")
    n, d, h, w = cost.shape
    D, H, W = out_dhw
    disp = torch.empty((n, H, W), device=cost.device, dtype=torch.float32)
    check(_lib.load().msnet_trilinear_softargmin(ptr(cost), ptr(disp), n, d, h, w, D, H, W, stream_ptr()),
          "msnet_trilinear_softargmin")
    return disp


class ConvBNPlan:
    """Device-side constants of one conv(+BN) layer: MFMA-packed weight and the eval-mode BN affine
    y = x*scale + shift with scale = gamma/sqrt(var+eps), shift = beta - mean*scale."""

    def __init__(self, conv, bn=None, transposed=False, precision=None):
        w = conv.weight.detach()
        self.co = w.shape[1] if transposed else w.shape[0]
        ci = w.shape[0] if transposed else w.shape[1]
        stride = conv.stride[0]
        precision = precision or _default_precision
        lib = _lib.load()
        if precision != "split-fp16":
            self.f16s = False
        elif transposed:
            self.f16s = bool(USE_F16S_DECONV and lib.msnet_deconv3d_k3s2_f16s_supported(ci, self.co))
        else:
            self.f16s = bool(lib.msnet_conv3d_k3_f16s_supported(ci, self.co, stride))
        if bn is not None:
            inv = 1.0 / torch.sqrt(bn.running_var.detach().float() + bn.eps)
            self.scale = (bn.weight.detach().float() * inv).contiguous()
            self.shift = (bn.bias.detach().float() - bn.running_mean.detach().float() * self.scale).contiguous()
        else:
            self.scale = None
            self.shift = None if conv.bias is None else conv.bias.detach().float().contiguous()
        if self.f16s and self.scale is not None:
            # split-fp16 kernels: the BN scale goes into the packed weights (one fp32 multiply per weight before the
            # hi/lo split) and the shift becomes the accumulators' start value, so their epilogue has no constants.
            shape = (1, -1, 1, 1, 1) if transposed else (-1, 1, 1, 1, 1)
            w = w.float() * self.scale.view(shape)
            self.scale = None
        self.wpk = pack_conv_weight(w, transposed, f16s=self.f16s, stride=stride)


def state_key(module):
    """Cheap fingerprint of a module's parameters/buffers: plans are rebuilt when any of them changes
    (load_state_dict, .to(), in-place edits all bump _version or data_ptr)."""
    return tuple((t.data_ptr(), t._version, str(t.device)) for t in module.state_dict(keep_vars=True).values())
