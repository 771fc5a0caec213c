"""MS-GCNet cost-volume aggregator, drop-in for the reference class of the same name
(/root/reference/src/models/gcnet_3dcnn.py:57-141) with the forward pass on hand-written HIP kernels.

Contract kept from the reference:
  * constructor signature and defaults (gcnet_3dcnn.py:58-65);
  * parameter / buffer names, so reference checkpoints load unchanged:
      conv3dbn_{1,2}.{0.weight,1.*}, block_3d_{1..4}.convbn_3d_{1..3}.{0,1}.*, deconvbn{1..4}.{0,1}.*,
      deconv5.{weight,bias};
  * forward(cv[N,C,D',H',W'] fp32 on the GPU) -> disp[N,H,W]; AssertionError when the regressed depth
    differs from maxdisp (gcnet_3dcnn.py:135).
Not kept: training-mode BatchNorm / autograd (the path is forward-only) and the CPU code path -- there is
no fallback: without libmsnet_hip.so or a GPU tensor the forward raises.
"""
import os

import torch
import torch.nn as nn

from . import hipops
from .net_init import net_init


# First layer straight from the NCDHW volume (msnet_conv3d_k3_c8_ncdhw_f16s) instead of layout conversion + NDHWC first layer.
# Measured at 960x544x192 (profiles/r03c_*): the conversion pass (0.14 ms, 802 MB) disappears but the first layer goes from
# 0.65 to 0.83 ms -- a 34-voxel tile row is 136 bytes in each of 8 planes (two cache lines each) instead of 1088 contiguous
# bytes, 1.7x the L1 fills and 4.5x the load instructions -- so it is NOT the default; it saves 401 MB of activation memory.
FUSE_INPUT_LAYOUT = os.environ.get("MSNET_FUSE_INPUT_LAYOUT", "0") == "1"


def _convbn(cin, cout, stride):
    return nn.Sequential(nn.Conv3d(cin, cout, 3, stride=stride, padding=1, bias=False), nn.BatchNorm3d(cout))


def _deconvbn(cin, cout):
    return nn.Sequential(nn.ConvTranspose3d(cin, cout, 3, stride=2, padding=1, output_padding=1, bias=False),
                         nn.BatchNorm3d(cout))


class Conv3DBlock(nn.Module):
    """Parameter container for one encoder level (three conv+BN+ReLU, first with the level's stride);
    gcnet_3dcnn.py:30-44.  The arithmetic is issued by GCNet_CostVolumeAggre.forward."""

    def __init__(self, in_planes, planes, stride=1, kernel_size=3):
        super().__init__()
        if kernel_size != 3:
            raise ValueError("only kernel_size=3 is built")
        self.convbn_3d_1 = _convbn(in_planes, planes, stride)
        self.convbn_3d_2 = _convbn(planes, planes, 1)
        self.convbn_3d_3 = _convbn(planes, planes, 1)
        self.stride = stride


class GCNet_CostVolumeAggre(hipops.DeviceStateMixin, nn.Module):
    def __init__(self, maxdisp=192, cbmv_in_planes=8, kernel_size=3, is_quarter_input_size=False):
        super().__init__()
        if kernel_size != 3:
            raise ValueError("only kernel_size=3 is built")
        self.maxdisp = maxdisp
        self.F = 32
        self.kernel_size = kernel_size
        self.is_quarter_input_size = bool(is_quarter_input_size)
        F = self.F
        self.conv3dbn_1 = _convbn(cbmv_in_planes, F, 1)
        self.conv3dbn_2 = _convbn(F, F, 1)
        self.block_3d_1 = Conv3DBlock(F, 2 * F, stride=2)
        self.block_3d_2 = Conv3DBlock(2 * F, 2 * F, stride=2)
        self.block_3d_3 = Conv3DBlock(2 * F, 2 * F, stride=2)
        self.block_3d_4 = Conv3DBlock(2 * F, 4 * F, stride=2)
        self.deconvbn1 = _deconvbn(4 * F, 2 * F)
        self.deconvbn2 = _deconvbn(2 * F, 2 * F)
        self.deconvbn3 = _deconvbn(2 * F, 2 * F)
        self.deconvbn4 = _deconvbn(2 * F, F)
        if self.is_quarter_input_size:   # gcnet_3dcnn.py:88-90
            self.deconv5 = nn.ConvTranspose3d(F, 1, 3, stride=4, padding=1, output_padding=3)
        else:
            self.deconv5 = nn.ConvTranspose3d(F, 1, 3, stride=2, padding=1, output_padding=1)
        net_init(self)
        # per-device state (packed-weight plans, activation arena, range guard, captured graphs): hipops.ModuleState
        self._init_device_state()
        # fp16-range guard of the split-fp16 kernels (hipops.guarded_forward): costs one 4-byte read-back per forward;
        # set range_check = False when the activations are known to stay below 32752 (hipops.ACT_MAX)
        self.range_check = os.environ.get("MSNET_RANGE_CHECK", "1") != "0"
        self.use_graph = False            # True: forwards are captured as HIP graphs per input buffer (hipops._graphed_forward)

    # ---- device constants ------------------------------------------------------------------------
    def invalidate_plans(self):
        """Drop the packed weights / folded BN constants; the next forward rebuilds them from the current parameters.
        Needed only after edits through `.data` (which leave no trace in the tensors' version counters)."""
        self._drop_device_state()          # plans, arenas, range guards, captured graphs (they hold the old packed weights)

    def _plans(self, precision):
        key = hipops.state_key(self)
        if self._plan is None or key != self._plan_key:
            self._plan, self._plan_key = {}, key           # one plan per precision; all dropped when the parameters change
        if precision not in self._plan:
            P = lambda *a, **k: hipops.ConvBNPlan(*a, precision=precision, **k)      # noqa: E731
            plan = {"conv3dbn_1": P(*self.conv3dbn_1), "conv3dbn_2": P(*self.conv3dbn_2)}
            for b in ("block_3d_1", "block_3d_2", "block_3d_3", "block_3d_4"):
                blk = getattr(self, b)
                for c in ("convbn_3d_1", "convbn_3d_2", "convbn_3d_3"):
                    plan[b + "." + c] = P(*getattr(blk, c))
            for dname in ("deconvbn1", "deconvbn2", "deconvbn3", "deconvbn4"):
                plan[dname] = P(*getattr(self, dname), transposed=True)
            plan["deconv5.w"] = self.deconv5.weight.detach().float().contiguous()
            plan["deconv5.wps"], plan["deconv5.wsc"] = hipops.pow2_prescale(self.deconv5.weight)   # fused tail (split-fp16 MFMA)
            plan["deconv5.b"] = float(self.deconv5.bias.detach().float().item())
            self._plan[precision] = plan
        return self._plan[precision]

    # ---- forward -----------------------------------------------------------------------------------
    def forward(self, cv, taps=None):
        """cv [N,C,D',H',W'] -> disparity [N,H,W].  `taps`: optional dict that receives every layer output
        converted back to NCDHW under the reference's names (parity tests only)."""
        if getattr(self, "_is_replica", False):
            raise RuntimeError(hipops.REPLICA_ERROR)
        if self.training:
            raise RuntimeError("GCNet_CostVolumeAggre (HIP) is forward/inference only: call .eval() first")
        cv = hipops.require_gpu_f32(cv, "cv")
        if cv.dim() != 5:
            raise ValueError("cv must be [N,C,D,H,W]")
        # (tapped activations are handed to the caller: fresh tensors instead of the arena's)
        return hipops.guarded_forward(self, lambda precision: self._forward(cv, taps, precision),
                                      graph_key=(cv.data_ptr(), tuple(cv.shape)) if taps is None else None,
                                      use_arena_=taps is None)

    def forward_ndhwc(self, cv_cl):
        """The forward on a CHANNELS-LAST volume cv_cl [N,D',H',W',C] (cbmv_generator.VolumeBuilder(layout="ndhwc") writes it):
        the aggregator kernels' own activation layout, so no layout pass runs between the volume build and conv3dbn_1.  Same
        result, bit for bit, as forward(cv) on the NCDHW volume of the same values.  Not part of the reference's interface
        (its module takes NCDHW, gcnet_3dcnn.py:97-101): forward() stays the drop-in."""
        if getattr(self, "_is_replica", False):
            raise RuntimeError(hipops.REPLICA_ERROR)
        if self.training:
            raise RuntimeError("GCNet_CostVolumeAggre (HIP) is forward/inference only: call .eval() first")
        cv_cl = hipops.require_gpu_f32(cv_cl, "cv_cl")
        if cv_cl.dim() != 5 or cv_cl.shape[4] != self.conv3dbn_1[0].in_channels:
            raise ValueError("cv_cl must be [N,D,H,W,%d] (got %s)" % (self.conv3dbn_1[0].in_channels, tuple(cv_cl.shape)))
        return hipops.guarded_forward(self, lambda precision: self._forward(cv_cl, None, precision, channels_last=True),
                                      graph_key=(cv_cl.data_ptr(), tuple(cv_cl.shape), "ndhwc"))

    def _forward(self, cv, taps, precision, channels_last=False):
        pl = self._plans(precision)
        if taps is not None:
            taps.clear()

        def tap(name, t):
            if taps is not None:
                taps[name] = hipops.ndhwc_to_ncdhw(t)
            return t

        def conv(x, name, stride=1, residual=None):
            p = pl[name]
            return hipops.conv3d_k3(x, p.wpk, p.scale, p.shift, p.co, stride=stride, relu=True, residual=residual,
                                     f16s=p.f16s, wpk_wd=p.wpk_wd, wpk_wd4=p.wpk_wd4)

        def block(x, name, stride):
            x = conv(x, name + ".convbn_3d_1", stride)
            x = conv(x, name + ".convbn_3d_2")
            return tap(name, conv(x, name + ".convbn_3d_3"))

        def deconv(x, name, skip):
            p = pl[name]
            return tap(name, hipops.deconv3d_k3s2(x, p.wpk, p.scale, p.shift, p.co, relu=True, residual=skip, f16s=p.f16s))

        with torch.no_grad():
            p1 = pl["conv3dbn_1"]
            if channels_last:
                if p1.f16s and cv.shape[4] == 8 and p1.co in (32, 64):
                    x = hipops.conv3d_c8_in(cv, p1.wpk, p1.scale, p1.shift, p1.co, relu=True)     # range check of the input inside
                elif not p1.f16s:
                    # fp32 precision: the volume already is the fp32 kernels' input layout and there is no fp16 range to guard
                    # (this branch used to permute + convert -- two passes over the volume -- for nothing)
                    x = conv(cv.contiguous(), "conv3dbn_1")
                else:
                    # (16-channel volumes on split-fp16: no input-checking first-layer kernel -- one read-only range pass in front)
                    x = conv(hipops.check_input_range(cv), "conv3dbn_1")
            elif p1.f16s and cv.shape[1] == 8 and p1.co in (32, 64) and FUSE_INPUT_LAYOUT:
                # the first layer reads the NCDHW volume itself: no layout-conversion pass over the 401 MB
                x = tap("conv3dbn_1", hipops.conv3d_c8_ncdhw(cv, p1.wpk, p1.scale, p1.shift, p1.co, relu=True))
            else:
                x = hipops.ncdhw_to_ndhwc(cv)
                x = tap("conv3dbn_1", conv(x, "conv3dbn_1"))
            res_l20 = x = tap("conv3dbn_2", conv(x, "conv3dbn_2"))
            res_l23 = x = block(x, "block_3d_1", 2)
            res_l26 = x = block(x, "block_3d_2", 2)
            res_l29 = x = block(x, "block_3d_3", 2)
            x = block(x, "block_3d_4", 2)
            x = deconv(x, "deconvbn1", res_l29)
            x = deconv(x, "deconvbn2", res_l26)
            x = deconv(x, "deconvbn3", res_l23)
            x = deconv(x, "deconvbn4", res_l20)
            hipops.guard_checkpoint()                 # the tail kernels below have no range check: read the guard word back under them
            s = 4 if self.is_quarter_input_size else 2
            depth = s * x.shape[1]
            assert depth == self.maxdisp, "%d != %d" % (depth, self.maxdisp)   # gcnet_3dcnn.py:135
            if s == 2 and taps is None:
                return hipops.deconv5_softargmin(x, pl["deconv5.wps"], pl["deconv5.b"], pl["deconv5.wsc"])
            logits = hipops.deconv3d_cout1(x, pl["deconv5.w"], pl["deconv5.b"], stride=s)
            if taps is not None:
                taps["deconv5"] = logits.unsqueeze(1)
            return hipops.softargmin(logits)
