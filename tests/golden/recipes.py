"""Seeded recipes shared by the golden-vector generator (tests/golden/make_aggregator_golden.py, which
imports the REFERENCE and therefore only runs in the build container) and by the tests (which never
touch /root/reference).  Nothing in here reads the reference.

A case is fully determined by (seed, shapes): weights come from torch.manual_seed(seed) + the model
constructor (net_init), BatchNorm statistics / affine from `randomize_bn` (default BN would be an
identity and hide BN bugs), the input from `make_input`.
"""
import hashlib

import numpy as np
import torch

AGG_CASES = {
    # name: dict(model, seed, maxdisp, in_shape [N,C,D',H',W'], plus options)
    "gcnet_small": dict(model="gcnet", seed=1, maxdisp=32, in_shape=(1, 8, 16, 16, 32)),
    "gcnet_batch2": dict(model="gcnet", seed=2, maxdisp=64, in_shape=(2, 8, 32, 32, 64)),
    "gcnet_peaky": dict(model="gcnet", seed=3, maxdisp=32, in_shape=(1, 8, 16, 16, 32), peaky=25.0),
    "gcnet_ragged": dict(model="gcnet", seed=4, maxdisp=32, in_shape=(1, 8, 16, 48, 80)),
    "gcnet_quarter": dict(model="gcnet", seed=5, maxdisp=64, in_shape=(1, 8, 16, 16, 32), quarter=True),
    "psmnet_small": dict(model="psmnet", seed=6, maxdisp=32, in_shape=(1, 64, 8, 16, 16)),
    "psmnet_batch2": dict(model="psmnet", seed=7, maxdisp=32, in_shape=(2, 64, 8, 8, 24)),
}
MAX_SAMPLES = 4096


def randomize_bn(model, seed):
    g = torch.Generator().manual_seed(10_000 + seed)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            c = m.num_features
            m.running_mean.copy_(torch.randn(c, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(c, generator=g) * 0.5 + 0.75)
            m.weight.data.copy_(torch.rand(c, generator=g) * 0.5 + 0.75)
            m.bias.data.copy_(torch.randn(c, generator=g) * 0.1)


def make_input(shape, seed):
    g = torch.Generator().manual_seed(20_000 + seed)
    return torch.rand(shape, generator=g)


def apply_options(model, case):
    """Post-construction tweaks of a case; `peaky` scales deconv5 so the softmax is sharply peaked
    (trained-network-like), the regime where reduced-precision convs fail by pixels."""
    if case.get("peaky"):
        model.deconv5.weight.data.mul_(case["peaky"])


def state_sha256(sd):
    h = hashlib.sha256()
    for k in sorted(sd.keys()):
        v = sd[k]
        h.update(k.encode())
        h.update(np.ascontiguousarray(v.detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def sample(t):
    """Deterministic strided sample of a tensor (<= MAX_SAMPLES values) + its flat stride."""
    flat = t.detach().reshape(-1)
    stride = max(1, flat.numel() // MAX_SAMPLES)
    return flat[::stride].cpu().numpy().astype(np.float32), stride


def build_case(case, gcnet_cls, psmnet_cls):
    """Construct the seeded model of a case with the given classes (reference's or ours)."""
    torch.manual_seed(case["seed"])
    if case["model"] == "gcnet":
        m = gcnet_cls(case["maxdisp"], is_quarter_input_size=bool(case.get("quarter", False)))
    else:
        m = psmnet_cls(case["maxdisp"])
    randomize_bn(m, case["seed"])
    apply_options(m, case)
    return m.eval()


def out_hw(case):
    n, c, d, h, w = case["in_shape"]
    if case["model"] == "psmnet":
        return 4 * h, 4 * w
    s = 4 if case.get("quarter") else 2
    return s * h, s * w
