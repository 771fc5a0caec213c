"""Tensor-level wrappers over the C ABI for the aggregator kernels.  PyTorch is used only for device
memory (torch.empty) and the current HIP stream; all arithmetic happens in libmsnet_hip.so.

Internal activation layout is channels-last fp32 [N, D, H, W, C] ("NDHWC")."""
import os
import threading
import warnings

import torch

from . import _lib
from ._lib import check, ptr, require_gpu_f32, stream_ptr


class Arena:
    """Activation buffers of one module, reused from forward to forward: the k-th buffer request of a forward gets the tensor
    the k-th request of the previous forward got (same shape, same device), so a steady-state forward allocates nothing --
    neither hipMalloc nor the caching allocator (whose occasional re-segmentation of the ~6 GB / map it hands out showed up
    as one 40 ms step in ten).  Buffers are internal activations only; what a module returns is always a fresh tensor.
    One forward at a time per module (buffers are stream-ordered on the caller's stream)."""

    def __init__(self):
        self.bufs = []
        self.k = 0

    def empty(self, shape, device):
        shape = tuple(int(v) for v in shape)
        if self.k < len(self.bufs) and tuple(self.bufs[self.k].shape) == shape and self.bufs[self.k].device == device:
            t = self.bufs[self.k]
        else:
            t = torch.empty(shape, device=device, dtype=torch.float32)
            if self.k < len(self.bufs):
                self.bufs[self.k] = t
            else:
                self.bufs.append(t)
        self.k += 1
        return t

    def nbytes(self):
        return sum(t.numel() * 4 for t in self.bufs)


class _ThreadState(threading.local):
    arena = None            # the Arena the wrappers below allocate from ON THIS THREAD (use_arena)


_tls = _ThreadState()


class use_arena:
    """Context: activation outputs of the wrappers below come from `arena` (None = plain torch.empty) -- for the calling
    THREAD only: two modules running forwards on two Python threads (one per replica, as under the reference's
    nn.DataParallel, main_msnet.py:174) each see their own arena."""

    def __init__(self, arena):
        self.arena = arena

    def __enter__(self):
        self.prev, _tls.arena = _tls.arena, self.arena
        if self.arena is not None:
            self.arena.k = 0
        return self.arena

    def __exit__(self, *exc):
        _tls.arena = self.prev
        return False


def _new(shape, device):
    if _tls.arena is not None:
        return _tls.arena.empty(shape, device)
    return torch.empty(shape, device=device, dtype=torch.float32)


class _Slot:
    """Everything a module keeps between forwards FOR ONE DEVICE: activation arena, packed-weight plans, range guard, the
    sticky fp32 fallback, captured graphs -- and the lock that makes a forward exclusive."""

    def __init__(self):
        self.arena = Arena()
        self.plan = None
        self.plan_key = None
        self.guard = None
        self.forced_precision = None
        self.forced_key = None
        self.graphs = {}
        self.lock = threading.RLock()
        self.last_event = None              # recorded behind the last forward on last_stream (guarded_forward)
        self.last_stream = None


class ModuleState:
    """Per-device slots of one aggregator module.  The object is shared BY REFERENCE between a module and any shallow copy of
    it, so two copies can never hand the same activation buffers to two concurrent forwards."""

    def __init__(self):
        self._slots = {}
        self._mu = threading.Lock()

    def slot(self, device):
        key = str(device)
        with self._mu:
            s = self._slots.get(key)
            if s is None:
                s = self._slots[key] = _Slot()
            return s

    def clear(self):
        with self._mu:
            self._slots.clear()

    # copy.deepcopy(model) / pickling a whole module (torch.save(model)): device state is not part of a model's value -- a copy
    # starts with empty slots of its own (locks, arenas and packed weights are neither copyable nor meaningful elsewhere)
    def __deepcopy__(self, memo):
        return ModuleState()

    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self.__init__()


def _slot_property(name):
    def get(self):
        return getattr(self._slot(), name)

    def set_(self, value):
        setattr(self._slot(), name, value)
    return property(get, set_)


class DeviceStateMixin:
    """Mixed into the two aggregator modules: `_arena`, `_plan`, `_plan_key`, `_guard`, `_forced_precision`, `_forced_key`,
    `_graphs` are those of the slot of the device the module's parameters live on."""
    _arena = _slot_property("arena")
    _plan = _slot_property("plan")
    _plan_key = _slot_property("plan_key")
    _guard = _slot_property("guard")
    _forced_precision = _slot_property("forced_precision")
    _forced_key = _slot_property("forced_key")
    _graphs = _slot_property("graphs")

    def _init_device_state(self):
        self.__dict__["_state"] = ModuleState()

    def _device(self):
        for p in self.parameters():
            return p.device
        return torch.device("cpu")

    def _slot(self):
        return self.__dict__["_state"].slot(self._device())

    def _drop_device_state(self):
        self.__dict__["_state"].clear()
        self.__dict__.pop("_state_tensors", None)      # state_key re-walks the module tree


REPLICA_ERROR = ("this module is an nn.DataParallel replica: the HIP aggregators run one process per GPU -- shard the batch with "
                 "ms-nets_amd.dist (init_from_env / shard_indices / gather_disparities) instead of wrapping the model in "
                 "nn.DataParallel (INTEGRATION.md section 3)")


def ncdhw_to_ndhwc(x):
    x = require_gpu_f32(x, "x")
    n, c, d, h, w = x.shape
    y = _new((n, d, h, w, c), x.device)
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


F16S_MAX = 65504.0      # largest finite fp16: a BN-folded WEIGHT of a split-fp16 kernel must stay below it (hi = fp16(w))
# Activations (and the module input) must stay below HALF of it: the Winograd-depth kernel splits sums / differences of two
# activations.  The kernels' epilogues and the input conversion compare against this value (csrc/conv_common.h: kSplitMax).
ACT_MAX = 32752.0
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


USE_WINOGRAD_DEPTH = True      # 32 -> 32 stride-1 layers on the Winograd-depth kernel where the shape is taken (A/B switch)
# Experiment: 64 -> 64 stride-1 layers as FOUR Winograd-depth launches on 32-channel slices
# (two output halves x two input halves, the first input half's partial sum handed over as the second launch's residual).
# Measured slower than the direct kernel at 48x136x240 -- off; bench.py --wd64 / tools_layer_bench.py switch it on.
USE_WD64 = os.environ.get("MSNET_WD64", "0") == "1"


def winograd_depth_weights(w):
    """w f32 [32,32,3,3,3] (BN-folded, pre-scaled) -> packed image of msnet_conv3d_k3_wd_f16s.  The F(2,3) transform of the
    three depth taps is done in fp64 on the host: g0 = w[kd=0], g1 = (w0+w1+w2)/2, g2 = (w0-w1+w2)/2, g3 = w[kd=2]."""
    w = require_gpu_f32(w, "weight")
    if tuple(w.shape) != (32, 32, 3, 3, 3):
        raise ValueError("winograd_depth_weights takes a [32,32,3,3,3] weight (got %s)" % (tuple(w.shape),))
    wd = w.double()
    w0, w1, w2 = wd[:, :, 0], wd[:, :, 1], wd[:, :, 2]                      # [Co,Ci,3,3] each
    g = torch.stack([w0, (w0 + w1 + w2) * 0.5, (w0 - w1 + w2) * 0.5, w2], dim=2)      # [Co,Ci,4,3,3]
    g36 = g.reshape(32, 32, 36).float().contiguous()
    out = torch.empty(36 * 32 * 32, device=w.device, dtype=torch.float32)             # 2 fp16 per weight = one float each
    check(_lib.load().msnet_pack_conv_weight_wd_f16s(ptr(g36), ptr(out), stream_ptr()), "msnet_pack_conv_weight_wd_f16s")
    return out


def conv3d_k3_winograd_depth(x, wpk_wd, scale, shift, relu=False, residual=None):
    x = require_gpu_f32(x, "x")
    n, d, h, w, ci = x.shape
    if ci != 32:
        raise ValueError("conv3d_k3_winograd_depth takes 32 input channels (got %d)" % ci)
    y = _new((n, d, h, w, 32), x.device)
    if residual is not None:
        residual = require_gpu_f32(residual, "residual")
        if residual.shape != y.shape:
            raise ValueError("residual shape %s != output shape %s" % (tuple(residual.shape), tuple(y.shape)))
    check(_lib.load().msnet_conv3d_k3_wd_f16s(ptr(x), ptr(wpk_wd), ptr(scale), ptr(shift), ptr(residual), ptr(y), n, d, h, w,
                                              int(relu), stream_ptr()), "msnet_conv3d_k3_wd_f16s")
    return y


def conv3d_k3_wd64(x, wpk_wd4, scale, shift, relu=False):
    """64 -> 64 stride-1 conv as four Winograd-depth launches (USE_WD64).  wpk_wd4[h][c]: packed image of w[32h:32h+32, 32c:32c+32]."""
    x = require_gpu_f32(x, "x")
    n, d, h, w, ci = x.shape
    y = _new((n, d, h, w, 64), x.device)
    lib = _lib.load()
    for hh in range(2):
        yv, xs = c_void_p_off(y, 32 * hh), None
        sc = c_void_p_off(scale, 32 * hh) if scale is not None else None
        sh = c_void_p_off(shift, 32 * hh) if shift is not None else None
        check(lib.msnet_conv3d_k3_wd_f16s_strided(c_void_p_off(x, 0), ptr(wpk_wd4[hh][0]), sc, None, None, yv, n, d, h, w, 64, 64, 0,
                                                  stream_ptr()), "msnet_conv3d_k3_wd_f16s_strided")
        check(lib.msnet_conv3d_k3_wd_f16s_strided(c_void_p_off(x, 32), ptr(wpk_wd4[hh][1]), sc, sh, yv, yv, n, d, h, w, 64, 64,
                                                  int(relu), stream_ptr()), "msnet_conv3d_k3_wd_f16s_strided")
    return y


def c_void_p_off(t, floats):
    """Device pointer of tensor t advanced by `floats` fp32 elements."""
    import ctypes
    return ctypes.c_void_p(t.data_ptr() + 4 * int(floats))


def conv3d_k3(x, wpk, scale, shift, co, stride=1, relu=False, residual=None, f16s=False, wpk_wd=None, wpk_wd4=None):
    x = require_gpu_f32(x, "x")
    n, d, h, w, ci = x.shape
    if (wpk_wd4 is not None and f16s and USE_WD64 and residual is None and stride == 1 and ci == 64 and co == 64 and
            _lib.load().msnet_conv3d_k3_wd_f16s_supported(d, h, w, 32, 32, 1)):
        return conv3d_k3_wd64(x, wpk_wd4, scale, shift, relu=relu)
    if (wpk_wd is not None and f16s and USE_WINOGRAD_DEPTH and
            _lib.load().msnet_conv3d_k3_wd_f16s_supported(d, h, w, ci, co, stride)):
        return conv3d_k3_winograd_depth(x, wpk_wd, scale, shift, relu=relu, residual=residual)
    od, oh, ow = (d - 1) // stride + 1, (h - 1) // stride + 1, (w - 1) // stride + 1
    y = _new((n, od, oh, ow, co), x.device)
    if residual is not None:
        residual = require_gpu_f32(residual, "residual")
        if residual.shape != y.shape:
            raise ValueError("residual shape %s != output shape %s" % (tuple(residual.shape), tuple(y.shape)))
    fn = _lib.load().msnet_conv3d_k3_f16s if f16s else _lib.load().msnet_conv3d_k3
    check(fn(ptr(x), ptr(wpk), ptr(scale), ptr(shift), ptr(residual), ptr(y), n, d, h, w, ci, co, stride, int(relu),
             stream_ptr()), "msnet_conv3d_k3_f16s" if f16s else "msnet_conv3d_k3")
    return y


def check_input_range(x):
    """fp16-range check of a channels-last module input (RangeGuard.INPUT), one read-only pass; returns x (contiguous)."""
    x = require_gpu_f32(x, "x").contiguous()
    check(_lib.load().msnet_check_input_range(ptr(x), x.numel(), stream_ptr()), "msnet_check_input_range")
    return x


def conv3d_c8_ncdhw(x, wpk, scale, shift, co, relu=False):
    """First layer straight from the NCDHW volume x [N,8,D,H,W] (no layout-conversion pass) -> NDHWC [N,D,H,W,co]; split-fp16."""
    x = require_gpu_f32(x, "x")
    n, c, d, h, w = x.shape
    if c != 8:
        raise ValueError("conv3d_c8_ncdhw takes an 8-channel volume (got %d)" % c)
    y = _new((n, d, h, w, co), x.device)
    check(_lib.load().msnet_conv3d_k3_c8_ncdhw_f16s(ptr(x), ptr(wpk), ptr(scale), ptr(shift), ptr(y), n, d, h, w, co, int(relu),
                                                    stream_ptr()), "msnet_conv3d_k3_c8_ncdhw_f16s")
    return y


def conv3d_c8_in(x, wpk, scale, shift, co, relu=False):
    """First layer on a channels-last MODULE INPUT x [N,D,H,W,8] (VolumeBuilder(layout="ndhwc")) -> NDHWC [N,D,H,W,co]; split-fp16.
    Same kernel as conv3d_k3 with Ci = 8, plus the fp16-range check of the module input (RangeGuard.INPUT)."""
    x = require_gpu_f32(x, "x")
    n, d, h, w, c = x.shape
    if c != 8:
        raise ValueError("conv3d_c8_in takes an 8-channel channels-last volume (got %d channels)" % c)
    y = _new((n, d, h, w, co), x.device)
    check(_lib.load().msnet_conv3d_k3_c8_in_f16s(ptr(x), ptr(wpk), ptr(scale), ptr(shift), ptr(y), n, d, h, w, co, int(relu),
                                                 stream_ptr()), "msnet_conv3d_k3_c8_in_f16s")
    return y


def conv3d_k3_in(x, wpk, scale, shift, co, relu=False):
    """Split-fp16 stride-1 conv on a channels-last MODULE INPUT x [N,D,H,W,Ci] with the fp16-range check of that input
    (RangeGuard.INPUT) -- inside the conv's loaders where the tiled kernel takes the shape, as one read-only pass in front of
    it otherwise (msnet_conv3d_k3_in_f16s).  Same bits as conv3d_k3(..., f16s=True)."""
    x = require_gpu_f32(x, "x").contiguous()
    n, d, h, w, ci = x.shape
    lib = _lib.load()
    # (MSNET_IN_FUSED=0: A/B switch, the round-5 route -- a read-only range pass in front of the plain conv)
    if not hasattr(lib, "msnet_conv3d_k3_in_f16s") or os.environ.get("MSNET_IN_FUSED", "1") == "0":     # or an older variant library
        return conv3d_k3(check_input_range(x), wpk, scale, shift, co, relu=relu, f16s=True)
    y = _new((n, d, h, w, co), x.device)
    check(lib.msnet_conv3d_k3_in_f16s(ptr(x), ptr(wpk), ptr(scale), ptr(shift), ptr(y), n, d, h, w, ci, co, int(relu), stream_ptr()),
          "msnet_conv3d_k3_in_f16s")
    return y


def deconv3d_k3s2(x, wpk, scale, shift, co, relu=False, residual=None, f16s=False):
    x = require_gpu_f32(x, "x")
    n, d, h, w, ci = x.shape
    y = _new((n, 2 * d, 2 * h, 2 * w, co), x.device)
    if residual is not None:
        residual = require_gpu_f32(residual, "residual")
        if residual.shape != y.shape:
            raise ValueError("residual shape %s != output shape %s" % (tuple(residual.shape), tuple(y.shape)))
    fn = _lib.load().msnet_deconv3d_k3s2_f16s if f16s else _lib.load().msnet_deconv3d_k3s2
    check(fn(ptr(x), ptr(wpk), ptr(scale), ptr(shift), ptr(residual), ptr(y), n, d, h, w, ci, co, int(relu), stream_ptr()),
          "msnet_deconv3d_k3s2_f16s" if f16s else "msnet_deconv3d_k3s2")
    return y


def pow2_prescale(w):
    """(w * 2^k, 2^-k) with k chosen so that max|w * 2^k| lies in [2^8, 2^9): the split-fp16 kernels' weights stay fp16-normal
    (22-bit hi + lo) whatever the checkpoint's scale; the kernel multiplies its result by 2^-k (exact).  One host sync."""
    w = w.detach().float().contiguous()
    m = float(w.abs().max())
    if not (m > 0) or m != m or m == float("inf"):
        return w, 1.0
    import math
    k = max(-100, min(100, 8 - math.floor(math.log2(m))))
    return (w * (2.0 ** k)).contiguous(), 2.0 ** -k


def conv3d_k3_cout1(x, w, add=None, wscale=1.0):
    """Conv3d(Ci->1) head.  x NDHWC, w [1,Ci,3,3,3] -> [N,D,H,W].  y = wscale * conv(x, w) (+ add), see pow2_prescale."""
    x = require_gpu_f32(x, "x")
    w = require_gpu_f32(w, "weight")
    n, d, h, wd, ci = x.shape
    y = _new((n, d, h, wd), x.device)
    if add is not None:
        add = require_gpu_f32(add, "add")
        if add.shape != y.shape:
            raise ValueError("add shape mismatch")
    check(_lib.load().msnet_conv3d_k3_cout1(ptr(x), ptr(w), float(wscale), ptr(add), ptr(y), n, d, h, wd, ci, stream_ptr()),
          "msnet_conv3d_k3_cout1")
    return y


def softargmin(logits):
    logits = require_gpu_f32(logits, "logits")
    n, d, h, w = logits.shape
    disp = torch.empty((n, h, w), device=logits.device, dtype=torch.float32)
    check(_lib.load().msnet_softargmin(ptr(logits), ptr(disp), n, d, h, w, stream_ptr()), "msnet_softargmin")
    return disp


def deconv5_softargmin(x, w, bias, wscale=1.0):
    """Fused ConvTranspose3d(Ci->1,k3,s2,p1,op1,bias) + softmax(D) + sum d*p.  x NDHWC -> [N,2H,2W].
    logits = wscale * deconv(x, w) + bias, see pow2_prescale."""
    x = require_gpu_f32(x, "x")
    w = require_gpu_f32(w, "weight")
    n, d, h, wd, ci = x.shape
    disp = torch.empty((n, 2 * h, 2 * wd), device=x.device, dtype=torch.float32)
    lib = _lib.load()
    # (an A/B variant library named through MSNET_HIP_LIB may predate the segmented entry points: _lib.load tolerates that)
    ws_bytes = getattr(lib, "msnet_deconv5_softargmin_workspace_bytes", None)
    nbytes = int(ws_bytes(n, d, h, wd)) if ws_bytes is not None and hasattr(lib, "msnet_deconv5_softargmin_ws") else 0
    if nbytes:                            # depth-segmented tail: partial softmax states of the segments (arena memory)
        ws = _new(((nbytes + 3) // 4,), x.device)
        check(lib.msnet_deconv5_softargmin_ws(ptr(x), ptr(w), float(bias), float(wscale), ptr(disp), n, d, h, wd, ci, ptr(ws), nbytes,
                                              stream_ptr()), "msnet_deconv5_softargmin_ws")
    else:
        check(lib.msnet_deconv5_softargmin(ptr(x), ptr(w), float(bias), float(wscale), ptr(disp), n, d, h, wd, ci, stream_ptr()),
              "msnet_deconv5_softargmin")
    return disp


def deconv3d_cout1(x, w, bias, stride=2):
    x = require_gpu_f32(x, "x")
    w = require_gpu_f32(w, "weight")
    n, d, h, wd, ci = x.shape
    y = _new((n, stride * d, stride * h, stride * wd), x.device)
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


class _ActiveGuard(threading.local):
    guard = None


_active_guard = _ActiveGuard()          # the RangeGuard of the forward running on this thread (the library's flag is per thread too)
EARLY_READBACK = os.environ.get("MSNET_EARLY_READBACK", "1") != "0"      # A/B switch (same result either way)


def guard_checkpoint():
    """Modules call this behind their last launch that can raise the range flag (see RangeGuard.checkpoint)."""
    g = _active_guard.guard
    if g is not None and EARLY_READBACK:
        g.checkpoint()


class RangeGuard:
    """fp16-range guard for one forward on the split-fp16 kernels: registers a device word with the library
    (msnet_set_overflow_flag); every conv epilogue ORs bit 0 into it when it stores a magnitude the split (hi = fp16(x)) cannot
    represent (|x| >= ACT_MAX), the layout conversion of the module INPUT ORs bit 1.  `word()` reads it back: one 4-byte
    device-to-host copy per forward, started by `checkpoint()` behind the last range-checked launch and waited for under the
    tail kernels (without a checkpoint: a plain read-back after the last launch)."""
    ACTIVATION, INPUT = 1, 2

    def __init__(self, device):
        self.flag = torch.zeros(1, dtype=torch.int32, device=device)
        self._host = None                   # pinned word + event of an early read-back (checkpoint)
        self._event = None
        self._pending = False

    def __enter__(self):
        self.flag.zero_()
        self._pending = False
        check(_lib.load().msnet_set_overflow_flag(ptr(self.flag)), "msnet_set_overflow_flag")
        _active_guard.guard = self
        return self

    def __exit__(self, *exc):
        _active_guard.guard = None
        _lib.load().msnet_set_overflow_flag(None)
        return False

    def checkpoint(self):
        """Start the read-back NOW: the caller promises that no launch after this point can raise the flag (the modules call
        it behind their last conv, in front of the tail kernels, which have no range check).  word() then waits for this
        copy only, i.e. the host learns the verdict while the tail is still running and prepares the next forward under it
        instead of behind it (the wait at the end of a forward otherwise exposes ~0.2 ms of host work per forward).
        Not inside a graph capture (the replay path reads the word after the replay)."""
        if torch.cuda.is_current_stream_capturing():
            return
        if self._host is None:
            self._host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._event = torch.cuda.Event()
        self._host.copy_(self.flag, non_blocking=True)
        self._event.record()
        self._pending = True

    def word(self):
        if self._pending:
            self._pending = False
            self._event.synchronize()
            return int(self._host[0])
        return int(self.flag.item())

    def tripped(self):
        return bool(self.word())


class exact_tails:
    """Context: deconv5 / classification-head kernels contract channels with fp32 FMAs instead of the split-fp16 MFMA."""

    def __enter__(self):
        _lib.load().msnet_set_exact_tails(1)

    def __exit__(self, *exc):
        _lib.load().msnet_set_exact_tails(0)
        return False


def _current_precision(module):
    """The module's sticky fp32 fallback belongs to the parameter state it was raised for: load_state_dict / .to() / any
    tracked in-place edit (state_key) lifts it."""
    if module._forced_precision is not None and module._forced_key != state_key(module):
        module._forced_precision = None
    return module._forced_precision or _default_precision


def _range_fallback(module, run, word):
    """A forward on the split-fp16 kernels raised the range flag: repeat it on the exact fp32 kernels.  Raised by an
    ACTIVATION (a property of the weights -- it would recur): the module stays on fp32 until its parameters change or
    invalidate_plans().  Raised only by the module INPUT (one out-of-range / NaN / inf voxel in this sample): this call only,
    so one bad sample in a serving loop does not move the module onto the slower path for good."""
    # (a bad input usually drags activations out of range with it: only a trip WITHOUT the input bit is the weights' doing)
    sticky = not (word & RangeGuard.INPUT)
    warnings.warn("msnet: %s left the fp16 range of the split-fp16 conv kernels (|x| >= %g%s); this forward was repeated on the "
                  "exact fp32 MFMA kernels%s" % ("an activation" if sticky else "the module input", ACT_MAX,
                                                 "" if sticky else ", inf or NaN",
                                                 ", which this module now keeps using" if sticky else " (this call only)"),
                  RuntimeWarning)
    if sticky:
        module._forced_precision = "fp32"
        module._forced_key = state_key(module)
    with exact_tails():
        return run("fp32")


def guarded_forward(module, run, graph_key=None, use_arena_=True):
    """Shared by the two aggregators.  run(precision) -> output.  Exclusive per (module, device): the slot's lock is held for
    the whole forward, so a second thread calling the same module waits instead of sharing its activation buffers; an
    nn.DataParallel replica is refused (REPLICA_ERROR).  On the split-fp16 path the forward runs under a
    RangeGuard; if an activation (or the input) left the fp16 range the result is discarded, a warning is issued and the
    forward is repeated on the exact fp32-input MFMA kernels with fp32 tails (_range_fallback: sticky for activations, per
    call for the input).
    graph_key (input address, shapes): with module.use_graph set, the forward is captured once per key as a HIP graph and
    replayed afterwards (_graphed_forward)."""
    if getattr(module, "_is_replica", False):
        raise RuntimeError(REPLICA_ERROR)
    slot = module._slot()
    with slot.lock:
        # the arena's buffers are ordered by the stream of the forward that used them last: a forward arriving on ANOTHER
        # stream (another thread) first waits for that one's kernels
        stream = torch.cuda.current_stream()
        if slot.last_event is not None and slot.last_stream != stream.cuda_stream:
            stream.wait_event(slot.last_event)
        out = _guarded_forward_locked(module, run, graph_key, use_arena_)
        if slot.last_event is None:
            slot.last_event = torch.cuda.Event()
        slot.last_event.record(stream)
        slot.last_stream = stream.cuda_stream
        return out


def _guarded_forward_locked(module, run_, graph_key, use_arena_):
    precision = _current_precision(module)

    def run(prec):                      # activations come from the module's arena (not when taps are handed out)
        with use_arena(module._arena if use_arena_ else None):
            return run_(prec)
    if graph_key is not None and getattr(module, "use_graph", False) and use_arena_:
        return _graphed_forward(module, run_, graph_key)
    if precision != "split-fp16":
        with exact_tails():
            return run(precision)
    if not module.range_check:
        return run(precision)
    guard = module._guard
    if guard is None or guard.flag.device != module._device():
        guard = module._guard = RangeGuard(module._device())
    with guard:
        out = run(precision)
    word = guard.word()
    if word:
        out = _range_fallback(module, run, word)
    return out


MAX_GRAPHS_PER_MODULE = 4
MAX_GRAPH_ARENA_BYTES = 32 << 30          # activation memory the captured graphs of one module (and device) may pin


def _evict_graphs(graphs, skey):
    """Make room for one more graph: entries captured for another parameter state (their key ends in another state_key) can
    never be replayed again and go first; then oldest first while the cache holds MAX_GRAPHS_PER_MODULE entries or its arenas
    pin more than MAX_GRAPH_ARENA_BYTES."""
    for k in [k for k in graphs if k[-1] != skey]:
        graphs.pop(k)
    while graphs and (len(graphs) >= MAX_GRAPHS_PER_MODULE or
                      sum(v["arena"].nbytes() for v in graphs.values() if v.get("arena")) > MAX_GRAPH_ARENA_BYTES):
        graphs.pop(next(iter(graphs)))


def _graphed_forward(module, run, graph_key):
    """module.use_graph = True: the ~45 launches of a forward (and their host work: ctypes calls, shape checks, buffer
    look-ups) become one hipGraphLaunch.  A graph is tied to the input's ADDRESS and shape, the precision and the parameter
    state (state_key: the check the eager path makes too), so it pays in loops that reuse their input buffer -- a serving
    loop, bench.py --graph; up to MAX_GRAPHS_PER_MODULE keys are kept.  The first call per key runs eagerly (plans, range
    fallback), the second captures.  A graph bakes in the ADDRESSES of its activation buffers, so every graph owns its
    Arena (filled by one eager pass right before the capture, so the capture itself allocates nothing): a later forward of
    another shape re-sizes the module's arena, never a captured graph's buffers.  The range guard stays on: its flag is
    cleared inside the graph and read after the replay; a trip drops the graphs and repeats the forward eagerly on fp32.
    Returns a copy of the graph's output buffer."""
    graphs = module._graphs
    precision = _current_precision(module)
    skey = state_key(module)
    key = tuple(graph_key) + (precision, skey)
    g = graphs.get(key)
    if g is None:
        module.use_graph = False
        try:
            out = guarded_forward(module, run)          # eager: builds plans and arena, settles the precision
        finally:
            module.use_graph = True
        if _current_precision(module) == precision:     # (else: the range guard moved the module to fp32)
            # a graph pins the packed weights and its own arena (~6 GB of activations per map at 960x544x192): graphs of an
            # older parameter state can never be replayed again, so they go first; then the cache is bounded both by count
            # and by the bytes its arenas hold (MAX_GRAPH_ARENA_BYTES), oldest first
            _evict_graphs(graphs, skey)
            graphs[key] = {"graph": None}               # captured on the next call with this key
        return out
    if g["graph"] is None:
        dev = module._device()
        guard = RangeGuard(dev) if precision == "split-fp16" and module.range_check else None
        arena = Arena()

        def body():
            with use_arena(arena):
                if precision != "split-fp16":
                    with exact_tails():
                        return run(precision)
                return run(precision)
        body()                                          # eager pass through the graph's own arena: all its allocations
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            if guard is not None:
                guard.__enter__()
            try:
                out = body()
            finally:
                if guard is not None:
                    guard.__exit__()
        # the graph also bakes in the addresses of the packed weights: pin the plan it was captured with
        g.update(graph=graph, out=out, guard=guard, arena=arena, plan=module._plan.get(precision) if isinstance(module._plan, dict) else None)
    g["graph"].replay()
    word = g["guard"].word() if g["guard"] is not None else 0
    if word:
        if not (word & RangeGuard.INPUT):
            graphs.clear()
        module.use_graph = False
        try:
            return _range_fallback(module, lambda prec: _with_arena(module, run, prec), word)
        finally:
            module.use_graph = True
    out = g["out"]
    return tuple(t.clone() for t in out) if isinstance(out, tuple) else out.clone()


def _with_arena(module, run, prec):
    with use_arena(module._arena):
        return run(prec)


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
        if self.f16s:
            # split-fp16 kernels: the BN scale goes into the packed weights (one fp32 multiply per weight before the
            # hi/lo split) and the shift becomes the accumulators' start value, so their epilogue has no constants.
            shape = (1, -1, 1, 1, 1) if transposed else (-1, 1, 1, 1, 1)
            wf = w.float() * self.scale.view(shape) if self.scale is not None else w.float()
            # fp16 range of the `hi` half: a folded weight beyond it (tiny running_var, huge gamma) would become inf.
            # Such a layer stays on the exact fp32-input MFMA kernel (one host sync per plan build, not per forward).
            wmax = float(wf.abs().max())
            if not wmax < F16S_MAX:
                warnings.warn("msnet: folded conv weight magnitude %.3g exceeds the fp16 range of the split-fp16 kernels; "
                              "this layer runs on the fp32 MFMA kernel" % wmax, RuntimeWarning)
                self.f16s = False
            else:
                # Per-output-channel power-of-two pre-scale: channel c's folded weights are multiplied by 2^k_c so that their
                # largest magnitude lies in [2^8, 2^9) and the epilogue multiplies the accumulator by 2^-k_c (`scale`, exact).
                # hi + lo*2^-11 carries 22 bits only while |w| >= 2^-13 (below that the fp16 halves run out of exponent and
                # the error floor is 2^-35 absolute), so a channel whose folded weights are all tiny (small gamma, large
                # running_var) would otherwise lose precision.  Bit-identical to the unscaled form whenever no half underflows.
                red = (0, 2, 3, 4) if transposed else (1, 2, 3, 4)
                cmax = wf.abs().amax(dim=red)
                k = torch.where(cmax > 0, 8 - torch.floor(torch.log2(cmax.clamp_min(1e-38))), torch.zeros_like(cmax))
                k = k.clamp(-100, 100)
                w = wf * torch.exp2(k).view(shape)
                self.scale = torch.exp2(-k).contiguous()
        self.wpk = pack_conv_weight(w, transposed, f16s=self.f16s, stride=stride)
        # 32 -> 32 stride-1 layers also get the Winograd-depth image (same folded, pre-scaled weights; used when the shape is taken)
        self.wpk_wd = None
        if self.f16s and not transposed and stride == 1 and ci == 32 and self.co == 32:
            self.wpk_wd = winograd_depth_weights(w.float())
        self.wpk_wd4 = None                 # experiment: four 32 x 32 Winograd images of a 64 -> 64 layer (USE_WD64)
        if USE_WD64 and self.f16s and not transposed and stride == 1 and ci == 64 and self.co == 64:
            wf_ = w.float()
            self.wpk_wd4 = [[winograd_depth_weights(wf_[32 * hh:32 * hh + 32, 32 * cc:32 * cc + 32].contiguous()) for cc in range(2)]
                            for hh in range(2)]


def _registered_tensors(module):
    """[(owner dict, name, tensor)] for every parameter / buffer registered in the module tree."""
    out = []
    for mod in module.modules():
        for d in (mod._parameters, mod._buffers):
            for name, t in d.items():
                if t is not None:
                    out.append((d, name, t))
    return out


def state_key(module):
    """Cheap fingerprint of a module's parameters/buffers (data pointer, version counter, device): the packed weights and BN
    plans are rebuilt when it changes.  load_state_dict, .to(), copy_ and every autograd-visible in-place op bump it.
    Edits made THROUGH `.data` (p.data.mul_(2), the idiom of the reference's net_init.py) do NOT bump the version counter;
    after such an edit call `model.invalidate_plans()`.
    Called once or twice per forward, on the host path between two forwards: walking the module tree (140 us for the 110
    tensors of the GCNet aggregator) is done once; afterwards the registered tensors are re-checked by identity (a re-assigned
    parameter or buffer is a different object in its owner's dict) and only pointers and version counters are read (~30 us).
    Submodules added after the first forward are not seen: invalidate_plans() re-walks the tree."""
    cache = module.__dict__.get("_state_tensors")
    if cache is None or not all(d.get(name) is t for d, name, t in cache):
        cache = module.__dict__["_state_tensors"] = _registered_tensors(module)
    ts = [c[2] for c in cache]
    return (tuple(map(torch.Tensor.data_ptr, ts)), tuple(t._version for t in ts), ts[0].device if ts else None)
