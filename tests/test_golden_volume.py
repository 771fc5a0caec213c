"""CPU: the oracle's restated Python glue (oracle/ms_volume.py: get_costs, extract_features_left, extract_features_lr,
build_ms_volume) against tests/golden/volume_*.npz -- outputs of the REFERENCE's own cbmv_generator.py:27-79,84-254,258-308,
imported unmodified with the oracle's C functions served as its `src.cpp.lib.libmatchers / libfeatextract`
(tests/golden/make_volume_golden.py).  Bit for bit: every function between the natives and the volume is NumPy on both sides.

This pins rows a7 / a9 / f2's glue arithmetic to the reference.  It does NOT pin the natives (a1-a6, a8): both sides of the
comparison call oracle/matchers_oracle.c for those -- see that file's header."""
import hashlib
import os

import numpy as np
import pytest

import recipes
from oracle import ms_volume as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(recipes.VOLUME_CASES)


def load(name):
    z = np.load(os.path.join(GOLDEN, "volume_%s.npz" % name))
    return {k: z[k] for k in z.files}


def bitexact(a, b, what):
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
    bad = a.view(np.uint32) != b.view(np.uint32)
    assert not bad.any(), "%s: %d / %d values differ" % (what, int(bad.sum()), bad.size)


@pytest.mark.parametrize("name", CASES)
def test_fixture_images_are_the_recipe(name):
    """The stored images are re-derivable from the seed (the fixture is data, not a black box)."""
    g = load(name)
    l, r = recipes.volume_pair(recipes.VOLUME_CASES[name])
    assert np.array_equal(l, g["left"]) and np.array_equal(r, g["right"])
    assert int(g["ndisp"]) == recipes.VOLUME_CASES[name]["ndisp"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_get_costs_equals_reference_glue(name):
    g = load(name)
    nd, b = int(g["ndisp"]), int(g["board"])
    got = O.get_costs(g["left"], g["right"], nd, 11, 3, 5, 5, b, b, b)
    for a, key in zip(got, ("cost_census", "cost_ncc", "cost_sobel", "cost_sad")):      # the reference's return order
        bitexact(a, g[key], key)
        assert a.flags["C_CONTIGUOUS"]
    # keyword defaults of get_costs (board_w_right = 0 -> no crop on the right)
    dflt = O.get_costs(g["left"], g["right"], maxdisp=nd)
    assert tuple(dflt[0].shape) == tuple(g["dflt_shape"])
    assert recipes.arrays_sha256(dflt) == bytes(g["dflt_sha256"])


@pytest.mark.parametrize("name", CASES)
def test_oracle_features_equal_reference_glue(name):
    g = load(name)
    costs = [g[k] for k in ("cost_census", "cost_ncc", "cost_sobel", "cost_sad")]
    left = O.extract_features_left(*costs, 128.0, 0.02, 20000.0, 20000.0)
    bitexact(left, g["features_left"], "extract_features_left")
    lr = O.extract_features_lr(*costs, 128.0, 0.02, 20000.0, 20000.0)
    assert lr.shape[0] == 16
    bitexact(lr[:8], g["features_left"], "extract_features_lr[:8]")
    bitexact(lr[8:], g["features_right"], "extract_features_lr[8:]")
    # the Sobel channel's likelihood ignores sobel_sigma (cbmv_generator.py:298,303 pass sad_sigma)
    bitexact(O.extract_features_left(*costs, 128.0, 0.02, 20000.0, 1.0), g["features_left"], "sobel_sigma is ignored")
    assert left.min() >= 0.0 and left.max() <= 1.0


@pytest.mark.parametrize("name", CASES)
def test_oracle_build_ms_volume_equals_reference_glue(name):
    g = load(name)
    bitexact(O.build_ms_volume(g["left"], g["right"], int(g["ndisp"]), int(g["board"])), g["features_left"], "build_ms_volume")


def test_fixtures_cover_the_branches():
    """What the three cases are there for: the flat pair drives NCC's non-finite branch (cost 1 where a window is constant),
    every pair has sentinel entries beyond d <= x + const, the shifted pair's census minimum sits at the planted disparity."""
    flat, shifted, rnd = load("flat"), load("shifted"), load("random")
    h, w = flat["cost_ncc"].shape[:2]
    assert (flat["cost_ncc"][h // 2 + 3:-3, 3: w // 3 - 3, 0] == 1.0).all()   # either window constant there
    assert (flat["cost_ncc"][3: h // 3 - 3, 3: w // 3 - 3, 0] == 1.0).all()    # ... and the right one alone
    for g in (flat, shifted, rnd):
        assert (g["cost_census"] == np.float32(2147483648.0)).any()
        assert (g["features_left"][0][g["cost_census"].transpose(2, 0, 1) == np.float32(2147483648.0)] == 1.0).all()
    am = np.argmin(shifted["cost_census"], axis=2)
    hs = shifted["cost_census"].shape[0]
    assert np.median(am[: hs // 2, 20:]) == 5 and np.median(am[hs // 2:, 20:]) == 13
    assert hashlib.sha256(rnd["left"].tobytes()).hexdigest() != hashlib.sha256(rnd["right"].tobytes()).hexdigest()
