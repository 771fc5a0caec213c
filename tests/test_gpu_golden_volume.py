"""GPU: the HIP volume path through the drop-in mirrors (ms-nets_amd/cbmv_generator.py: get_costs, extract_features_left,
extract_features_lr, build_ms_volume -- every one a C-ABI call) against tests/golden/volume_*.npz, the outputs of the
REFERENCE's own Python glue (cbmv_generator.py:27-79,84-254,258-308, imported unmodified around the oracle's natives by
tests/golden/make_volume_golden.py).  Cost channels bit-exact; likelihood channels 2e-6 (GPU v_exp_f32 vs glibc expf on
values in [0,1]).  The natives themselves stay parity-unpinned (oracle/matchers_oracle.c header)."""
import os

import numpy as np
import pytest
import torch

import recipes

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(recipes.VOLUME_CASES)
COSTS = ("cost_census", "cost_ncc", "cost_sobel", "cost_sad")
AML_TOL = 2e-6


def load(name):
    z = np.load(os.path.join(GOLDEN, "volume_%s.npz" % name))
    return {k: z[k] for k in z.files}


def bitexact(a, b, what):
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
    bad = a.view(np.uint32) != b.view(np.uint32)
    assert not bad.any(), "%s: %d / %d values differ" % (what, int(bad.sum()), bad.size)


@pytest.mark.parametrize("name", CASES)
def test_get_costs_vs_reference_glue(gpu, name):
    from msnets_amd import cbmv_generator as cg
    g = load(name)
    nd, b = int(g["ndisp"]), int(g["board"])
    got = cg.get_costs(g["left"], g["right"], nd, 11, 3, 5, 5, b, b, b)
    for a, key in zip(got, COSTS):
        bitexact(a, g[key], key)
    dflt = cg.get_costs(g["left"], g["right"], maxdisp=nd)
    assert tuple(dflt[0].shape) == tuple(g["dflt_shape"])
    assert recipes.arrays_sha256(dflt) == bytes(g["dflt_sha256"])


@pytest.mark.parametrize("name", CASES)
def test_extract_features_vs_reference_glue(gpu, name):
    from msnets_amd import cbmv_generator as cg
    g = load(name)
    costs = [g[k] for k in COSTS]
    left = cg.extract_features_left(*costs, 128.0, 0.02, 20000.0, 20000.0)
    assert left.dtype == np.float32
    bitexact(left[:4], g["features_left"][:4], "left cost channels")
    assert np.abs(left[4:] - g["features_left"][4:]).max() <= AML_TOL
    lr = cg.extract_features_lr(*costs, 128.0, 0.02, 20000.0, 20000.0)
    assert lr.shape == (16,) + g["features_left"].shape[1:]
    bitexact(lr[:4], g["features_left"][:4], "lr: left cost channels")
    bitexact(lr[8:12], g["features_right"][:4], "lr: right cost channels")
    assert np.abs(lr[4:8] - g["features_left"][4:]).max() <= AML_TOL
    assert np.abs(lr[12:] - g["features_right"][4:]).max() <= AML_TOL


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("layout", ["ncdhw", "ndhwc"])
def test_fused_build_vs_reference_glue(gpu, name, layout):
    """msnet_build_volume / msnet_build_volume_ndhwc from the two images == the reference glue's 8-channel volume."""
    from msnets_amd import cbmv_generator as cg
    g = load(name)
    nd = int(g["ndisp"])
    l, r = torch.from_numpy(g["left"]).to(gpu), torch.from_numpy(g["right"]).to(gpu)
    vol = cg.build_ms_volume(l, r, nd, layout=layout)
    vol = (vol.permute(3, 0, 1, 2) if layout == "ndhwc" else vol).contiguous().cpu().numpy()
    bitexact(vol[:4], g["features_left"][:4], "fused build (%s): cost channels" % layout)
    assert np.abs(vol[4:] - g["features_left"][4:]).max() <= AML_TOL
