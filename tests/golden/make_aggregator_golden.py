"""Generates tests/golden/aggregators_<case>.npz by running the REFERENCE aggregators
(/root/reference/src/models/{gcnet_3dcnn,psmnet_3dcnn}.py, imported unmodified) on CPU.

Runs only in the build container (the reference is not on the GPU box); the .npz files it writes are
data: inputs are re-derivable from the seed, outputs are the reference's own numbers.

Harness-side shims (the reference files are untouched):
  D1  disparityregression hard-codes .cuda()          -> torch.Tensor.cuda is made a no-op
  D2  PSMNet forward reads an undefined global `left` -> psmnet_3dcnn.left = empty(1,3,H,W)
At generation time the oracle restatement (oracle/aggregators.py) and our own module classes
(seed -> identical state_dict) are cross-checked against the reference as well.

    python tests/golden/make_aggregator_golden.py
"""
import contextlib
import io
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import numpy as np
import torch

import recipes
from oracle import aggregators as oracle

torch.Tensor.cuda = lambda self, *a, **k: self          # D1
import src.models.gcnet_3dcnn as ref_gc                # noqa: E402
import src.models.psmnet_3dcnn as ref_psm              # noqa: E402

import msnets_amd                                      # noqa: E402,F401
from msnets_amd.gcnet_3dcnn import GCNet_CostVolumeAggre as OurGC      # noqa: E402
from msnets_amd.psmnet_3dcnn import PSMNet_CostVolumeAggre as OurPSM   # noqa: E402

GC_TAPS = ["conv3dbn_1", "conv3dbn_2", "block_3d_1", "block_3d_2", "block_3d_3", "block_3d_4", "deconv5"]


def run_case(name, case):
    torch.set_num_threads(8)
    with contextlib.redirect_stdout(io.StringIO()):
        ref = recipes.build_case(case, ref_gc.GCNet_CostVolumeAggre, ref_psm.PSMNet_CostVolumeAggre)
        ours = recipes.build_case(case, OurGC, OurPSM)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    sha = recipes.state_sha256(sd)
    assert sha == recipes.state_sha256(ours.state_dict()), "our seeded init differs from the reference's"
    x = recipes.make_input(case["in_shape"], case["seed"])
    H, W = recipes.out_hw(case)
    taps_ref = {}
    hooks = []
    if case["model"] == "gcnet":
        for t in GC_TAPS:
            mod = getattr(ref, t)
            relu = t.startswith("conv3dbn")   # the reference applies an in-place ReLU outside the Sequential
            hooks.append(mod.register_forward_hook(
                lambda m, i, o, t=t, relu=relu: taps_ref.__setitem__(t, torch.relu(o.detach().clone()) if relu else o.detach().clone())))
    else:
        ref_psm.left = torch.empty(1, 3, H, W)          # D2
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        disp = ref(x.clone())
    for h in hooks:
        h.remove()

    # cross-check the oracle restatement against the reference, layer by layer
    taps_or = {}
    with torch.no_grad():
        if case["model"] == "gcnet":
            d_or = oracle.gcnet_forward(sd, x, case["maxdisp"], bool(case.get("quarter")), taps=taps_or)
        else:
            d_or = oracle.psmnet_forward(sd, x, case["maxdisp"], (H, W), taps=taps_or)
    err = (d_or - disp).abs().max().item()
    assert err < 1e-4, (name, err)
    for t, v in taps_ref.items():
        e = (taps_or[t] - v).abs().max().item()
        assert e < 1e-4 * max(1.0, v.abs().max().item()), (name, t, e)

    out = {"disp": disp.numpy().astype(np.float32), "state_sha256": np.array(sha),
           "oracle_max_abs_err": np.float32(err)}
    for t, v in taps_ref.items():
        s, stride = recipes.sample(v)
        out["tap_" + t] = s
        out["tapstride_" + t] = np.int64(stride)
    path = os.path.join(HERE, "aggregators_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("%-14s disp %s  range [%.3f, %.3f]  oracle-vs-reference %.2e  sha %s.. -> %s (%d B)"
          % (name, tuple(disp.shape), disp.min(), disp.max(), err, sha[:12], os.path.basename(path),
             os.path.getsize(path)))


if __name__ == "__main__":
    for name, case in recipes.AGG_CASES.items():
        run_case(name, case)
