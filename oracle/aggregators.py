"""ORACLE (test infrastructure, not product code) -- CPU fp32 restatement of the two cost-volume
aggregators of ccj5351/MS-Nets as plain torch.nn.functional calls on a state_dict.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.  The
shipped path (ms-nets_amd/) never does: it runs hand-written HIP kernels and raises if they are missing.

Pinning: this restatement is checked against outputs of the *reference itself*, imported from
/root/reference in the build container by tests/golden/make_aggregator_golden.py; the resulting
vectors live in tests/golden/aggregators_*.npz and tests/test_golden_aggregators.py compares them with
this file (CPU, no GPU needed).  A floating-point kernel => a torch fp32 reference is the right oracle.

Reference lines restated (all under /root/reference/src/models/):
  gcnet_3dcnn.py:20-22   convbn_3d      -> _convbn
  gcnet_3dcnn.py:24-27   deconvbn_3d    -> _deconvbn
  gcnet_3dcnn.py:30-44   Conv3DBlock    -> _block
  gcnet_3dcnn.py:97-141  GCNet_CostVolumeAggre.forward + disparityregression -> gcnet_forward
  psmnet_3dcnn.py:47-89  hourglass      -> _hourglass
  psmnet_3dcnn.py:126-179 PSMNet_CostVolumeAggre.forward (eval and train returns) -> psmnet_forward
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm3d default, never overridden by the reference


def _bn(sd, prefix, x):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], False, 0.0, BN_EPS)


def _convbn(sd, prefix, x, stride=1):
    """Conv3d(k3, pad 1, stride, bias=False) + BatchNorm3d(eval).  gcnet_3dcnn.py:20-22."""
    y = F.conv3d(x, sd[prefix + ".0.weight"], None, stride=stride, padding=1)
    return _bn(sd, prefix + ".1", y)


def _deconvbn(sd, prefix, x):
    """ConvTranspose3d(k3, s2, p1, op1, bias=False) + BatchNorm3d(eval).  gcnet_3dcnn.py:24-27."""
    y = F.conv_transpose3d(x, sd[prefix + ".0.weight"], None, stride=2, padding=1, output_padding=1)
    return _bn(sd, prefix + ".1", y)


def _block(sd, prefix, x):
    """Conv3DBlock: three conv+BN+ReLU, the first with stride 2.  gcnet_3dcnn.py:30-44."""
    x = F.relu(_convbn(sd, prefix + ".convbn_3d_1", x, stride=2))
    x = F.relu(_convbn(sd, prefix + ".convbn_3d_2", x))
    x = F.relu(_convbn(sd, prefix + ".convbn_3d_3", x))
    return x


def soft_argmin(logits):
    """squeeze -> softmax over D (no negation) -> sum_d d*p_d.  gcnet_3dcnn.py:126-141,
    psmnet_3dcnn.py:28-37,172-174.  logits [N,D,H,W] -> [N,H,W]."""
    p = F.softmax(logits, 1)
    d = torch.arange(logits.shape[1], dtype=torch.float32).view(1, -1, 1, 1)
    return torch.sum(p * d, 1)


def gcnet_forward(sd, cv, maxdisp, is_quarter_input_size=False, taps=None):
    """GCNet_CostVolumeAggre.forward, gcnet_3dcnn.py:97-130.  cv [N,C,D',H',W'] -> disp [N,H,W].
    `taps`, if a dict, receives every intermediate activation under the reference's layer names."""
    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t

    out = tap("conv3dbn_1", F.relu(_convbn(sd, "conv3dbn_1", cv)))
    out = tap("conv3dbn_2", F.relu(_convbn(sd, "conv3dbn_2", out)))
    res_l20 = out
    out = tap("block_3d_1", _block(sd, "block_3d_1", out)); res_l23 = out
    out = tap("block_3d_2", _block(sd, "block_3d_2", out)); res_l26 = out
    out = tap("block_3d_3", _block(sd, "block_3d_3", out)); res_l29 = out
    out = tap("block_3d_4", _block(sd, "block_3d_4", out))
    out = tap("deconvbn1", F.relu(_deconvbn(sd, "deconvbn1", out) + res_l29))
    out = tap("deconvbn2", F.relu(_deconvbn(sd, "deconvbn2", out) + res_l26))
    out = tap("deconvbn3", F.relu(_deconvbn(sd, "deconvbn3", out) + res_l23))
    out = tap("deconvbn4", F.relu(_deconvbn(sd, "deconvbn4", out) + res_l20))
    if is_quarter_input_size:   # gcnet_3dcnn.py:88-90
        out = F.conv_transpose3d(out, sd["deconv5.weight"], sd["deconv5.bias"], stride=4, padding=1,
                                 output_padding=3)
    else:
        out = F.conv_transpose3d(out, sd["deconv5.weight"], sd["deconv5.bias"], stride=2, padding=1,
                                 output_padding=1)
    out = tap("deconv5", out).squeeze(1)
    assert out.shape[1] == maxdisp, "%d != %d" % (out.shape[1], maxdisp)   # gcnet_3dcnn.py:135
    return soft_argmin(out)


def _hourglass(sd, p, x, presqu, postsqu):
    """hourglass.forward, psmnet_3dcnn.py:69-89."""
    out = F.relu(_convbn(sd, p + ".conv1.0", x, stride=2))
    pre = _convbn(sd, p + ".conv2", out)
    pre = F.relu(pre + postsqu) if postsqu is not None else F.relu(pre)
    out = F.relu(_convbn(sd, p + ".conv3.0", pre, stride=2))
    out = F.relu(_convbn(sd, p + ".conv4.0", out))
    up = _deconvbn(sd, p + ".conv5", out)
    post = F.relu(up + (presqu if presqu is not None else pre))
    out = _deconvbn(sd, p + ".conv6", post)
    return out, pre, post


def _classif(sd, p, x):
    y = F.relu(_convbn(sd, p + ".0", x))
    return F.conv3d(y, sd[p + ".2.weight"], None, stride=1, padding=1)


def psmnet_forward(sd, cost, maxdisp, out_hw, training=False, taps=None):
    """PSMNet_CostVolumeAggre.forward, psmnet_3dcnn.py:126-179.  cost [N,64,D/4,H/4,W/4];
    out_hw = (H, W) = the reference's undefined global ``left.size()[2:4]`` (SURVEY.md defect D2).
    Returns pred3 (eval) or (pred1, pred2, pred3) (training)."""
    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t

    c0 = F.relu(_convbn(sd, "dres0.0", cost))
    c0 = F.relu(_convbn(sd, "dres0.2", c0))
    c1 = F.relu(_convbn(sd, "dres1.0", c0))
    cost0 = tap("cost0", _convbn(sd, "dres1.2", c1) + c0)

    out1, pre1, post1 = _hourglass(sd, "dres2", cost0, None, None)
    out1 = tap("out1", out1 + cost0)
    out2, _, post2 = _hourglass(sd, "dres3", out1, pre1, post1)
    out2 = tap("out2", out2 + cost0)
    out3, _, _ = _hourglass(sd, "dres4", out2, pre1, post2)
    out3 = tap("out3", out3 + cost0)

    cost1 = _classif(sd, "classif1", out1)
    cost2 = _classif(sd, "classif2", out2) + cost1
    cost3 = tap("cost3", _classif(sd, "classif3", out3) + cost2)

    size = [maxdisp, out_hw[0], out_hw[1]]

    def tail(c):
        c = F.interpolate(c, size, mode="trilinear", align_corners=True).squeeze(1)
        return soft_argmin(c)

    if training:
        return tail(cost1), tail(cost2), tail(cost3)
    return tail(cost3)
