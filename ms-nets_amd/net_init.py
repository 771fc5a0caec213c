"""Seeded weight initialisation for the cost-volume aggregators.

Behavioural mirror of the reference initialiser (/root/reference/src/models/net_init.py:26-54),
restricted to the layer kinds the two aggregators actually contain:

* Conv3d / ConvTranspose3d : weight ~ N(0, sqrt(2 / (kd*kh*kw*C_out))), bias = 0
* BatchNorm3d              : gamma = 1, beta = 0

It is needed only so that parity fixtures can be regenerated from a seed (no trained checkpoint is
reachable offline).  The draw order (module registration order, one ``normal_`` per conv
weight) is part of the contract: with the same ``torch.manual_seed`` the resulting state_dict is
bit-identical to the reference's, which tests/test_golden_aggregators.py checks through a sha256.
"""
import math

import torch.nn as nn


def net_init(net: nn.Module) -> None:
    for mod in net.modules():
        if isinstance(mod, (nn.Conv3d, nn.ConvTranspose3d)):
            kd, kh, kw = mod.kernel_size
            std = math.sqrt(2.0 / (kd * kh * kw * mod.out_channels))
            mod.weight.data.normal_(0, std)
            if mod.bias is not None:
                mod.bias.data.zero_()
        elif isinstance(mod, nn.BatchNorm3d):
            mod.weight.data.fill_(1)
            mod.bias.data.zero_()
