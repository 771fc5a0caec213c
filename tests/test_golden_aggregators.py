"""CPU: the oracle restatement (oracle/aggregators.py) against golden vectors produced by the REFERENCE
modules (tests/golden/make_aggregator_golden.py), and our drop-in classes' seeded state against the
reference's state_dict sha256 (names + values)."""
import os

import numpy as np
import pytest
import torch

import recipes
from oracle import aggregators as oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _our_classes():
    from msnets_amd.gcnet_3dcnn import GCNet_CostVolumeAggre
    from msnets_amd.psmnet_3dcnn import PSMNet_CostVolumeAggre
    return GCNet_CostVolumeAggre, PSMNet_CostVolumeAggre


@pytest.mark.parametrize("name", sorted(recipes.AGG_CASES))
def test_oracle_matches_reference_golden(name):
    case = recipes.AGG_CASES[name]
    gold = np.load(os.path.join(GOLD, "aggregators_%s.npz" % name))
    model = recipes.build_case(case, *_our_classes())
    sd = model.state_dict()
    assert recipes.state_sha256(sd) == str(gold["state_sha256"]), "seeded weights differ from the reference's"
    x = recipes.make_input(case["in_shape"], case["seed"])
    taps = {}
    with torch.no_grad():
        if case["model"] == "gcnet":
            disp = oracle.gcnet_forward(sd, x, case["maxdisp"], bool(case.get("quarter")), taps=taps)
        else:
            disp = oracle.psmnet_forward(sd, x, case["maxdisp"], recipes.out_hw(case), taps=taps)
    assert disp.shape == gold["disp"].shape
    # same torch build, same ops: the restatement reproduces the reference to the last bit here; allow 1e-5
    assert np.abs(disp.numpy() - gold["disp"]).max() <= 1e-5
    for key in gold.files:
        if key.startswith("tap_"):
            t = key[4:]
            s, stride = recipes.sample(taps[t])
            assert stride == int(gold["tapstride_" + t])
            assert np.abs(s - gold[key]).max() <= 1e-5 * max(1.0, float(np.abs(gold[key]).max())), t


def test_state_dict_names_match_reference_contract():
    """checkpoint keys of the reference load into the drop-in modules unchanged."""
    G, P = _our_classes()
    g = set(G(32).state_dict().keys())
    for k in ["conv3dbn_1.0.weight", "conv3dbn_2.1.running_var", "block_3d_4.convbn_3d_3.1.num_batches_tracked",
              "block_3d_1.convbn_3d_1.0.weight", "deconvbn4.0.weight", "deconvbn1.1.bias", "deconv5.weight",
              "deconv5.bias"]:
        assert k in g, k
    assert len(g) == 110
    p = set(P(32).state_dict().keys())
    for k in ["dres0.0.0.weight", "dres0.2.1.running_mean", "dres1.2.0.weight", "dres2.conv1.0.0.weight",
              "dres3.conv2.1.weight", "dres4.conv5.0.weight", "dres4.conv6.1.bias", "classif1.0.0.weight",
              "classif3.2.weight"]:
        assert k in p, k
    assert len(p) == 153
    assert tuple(G(32).deconvbn1[0].weight.shape) == (128, 64, 3, 3, 3)      # ConvTranspose3d [Ci,Co,...]
    assert tuple(G(64, is_quarter_input_size=True).deconv5.stride) == (4, 4, 4)


def test_oracle_psmnet_training_returns_three_heads():
    case = recipes.AGG_CASES["psmnet_small"]
    model = recipes.build_case(case, *_our_classes())
    x = recipes.make_input(case["in_shape"], case["seed"])
    with torch.no_grad():
        p1, p2, p3 = oracle.psmnet_forward(model.state_dict(), x, case["maxdisp"], recipes.out_hw(case), training=True)
        p = oracle.psmnet_forward(model.state_dict(), x, case["maxdisp"], recipes.out_hw(case))
    assert p1.shape == p2.shape == p3.shape == p.shape
    assert torch.equal(p3, p)
