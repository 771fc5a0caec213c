"""Generates tests/golden/volume_<case>.npz by running the REFERENCE's own Python glue --
/root/reference/src/dataloader/cbmv_generator.py: get_costs (:27-79), extract_features_left (:258-308) and
extract_features_lr (:84-254), imported UNMODIFIED -- on three tiny bordered pairs.

What this pins and what it does not (oracle/matchers_oracle.c header):
  PINNED   the NumPy glue of rows a7 / a9 / f2 -- which matcher goes where, the (maxdisp, 11, 3, 5, 5) windows handed on
           by the caller, the swap_axes placement, the border crop, clip / normalise, the float64 scratch, sad_sigma on
           the Sobel channel, the right-cost re-indexing order, the transposes and the float32 cast -- GIVEN the natives.
  UNPINNED the natives themselves (rows a1-a6, a8: matchers.cpp / featextract.cpp).  They need Boost.Python, which this
           image lacks; the reference module's `src.cpp.lib.libmatchers / libfeatextract` imports are therefore served by
           oracle/matchers_oracle.c (the restatement) through sys.modules.  A fixture produced here says "the reference's
           glue around the oracle's natives", never "the reference's natives".

Harness-side shims (reference files untouched, nothing of them is written anywhere):
  S1  cv2, skimage, skimage.transform, scipy.misc, matplotlib(.pyplot)  -> empty stand-in modules in sys.modules; none
      of the three functions run here touches them (they serve the data-set readers / plotting helpers of the module)
  S2  src.cpp.lib.libmatchers / libfeatextract                           -> the oracle's C functions

Runs only in the build container (the reference is not on the GPU box); the .npz files are data: the two uint8 images,
the reference glue's four cropped costs, its 8-channel volume and channels 8-15 of its 16-channel volume.

    python tests/golden/make_volume_golden.py
"""
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402

import recipes  # noqa: E402
from oracle import ms_volume as O  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference_glue():
    """-> the reference's cbmv_generator module, its natives served by the oracle (S1, S2)."""
    for name in ("cv2", "skimage", "skimage.transform", "scipy.misc", "matplotlib", "matplotlib.pyplot"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _stub(name)
    if "skimage" in sys.modules and "skimage.transform" in sys.modules:
        sys.modules["skimage"].transform = sys.modules["skimage.transform"]
    mtc = _stub("src.cpp.lib.libmatchers", census=O.census, nccNister=O.nccNister, zsad=O.zsad, sobel=O.sobel,
                sadsob=O.sadsob, initthreads=lambda: 8)
    fte = _stub("src.cpp.lib.libfeatextract", swap_axes=O.swap_axes, extract_likelihood=O.extract_likelihood,
                get_right_cost=O.get_right_cost)
    import src.cpp  # the reference's package (an empty __init__)
    lib = _stub("src.cpp.lib", libmatchers=mtc, libfeatextract=fte)
    lib.__path__ = []
    src.cpp.lib = lib
    import src.dataloader.cbmv_generator as ref
    assert ref.__file__.startswith("/root/reference/"), ref.__file__
    return ref


def main():
    O.build()
    ref = import_reference_glue()
    for name, case in recipes.VOLUME_CASES.items():
        l, r = recipes.volume_pair(case)
        nd, b = case["ndisp"], case["board"]
        # generate_test_cbmv's own call sequence (cbmv_generator.py:826-839): maxdisp // ds windows 11, 3, 5, 5, border on
        # all four sides, then the left features with the reference's default sigmas
        costs = ref.get_costs(l, r, nd, 11, 3, 5, 5, b, b, b)
        left8 = ref.extract_features_left(*costs, 128.0, 0.02, 20000.0, 20000.0)
        lr16 = ref.extract_features_lr(*costs, 128.0, 0.02, 20000.0, 20000.0)
        assert left8.dtype == np.float32 and lr16.dtype == np.float32
        assert left8.shape == (8, nd, l.shape[0] - 2 * b, l.shape[1] - 2 * b)
        assert np.array_equal(lr16[:8].view(np.uint32), left8.view(np.uint32))
        # get_costs' keyword defaults (border_w_right = 0: the training-time crop) as a second glue route
        costs_dflt = ref.get_costs(l, r, maxdisp=nd)
        # cross-check the oracle's restated glue against the reference's, bit for bit, before writing anything
        o_costs = O.get_costs(l, r, nd, 11, 3, 5, 5, b, b, b)
        for a, c in zip(o_costs, costs):
            assert a.dtype == c.dtype and np.array_equal(a.view(np.uint32), c.view(np.uint32))
        assert np.array_equal(O.extract_features_left(*o_costs).view(np.uint32), left8.view(np.uint32))
        assert np.array_equal(O.extract_features_lr(*o_costs).view(np.uint32), lr16.view(np.uint32))
        assert np.array_equal(O.build_ms_volume(l, r, nd, b).view(np.uint32), left8.view(np.uint32))
        for a, c in zip(O.get_costs(l, r, maxdisp=nd), costs_dflt):
            assert np.array_equal(a.view(np.uint32), c.view(np.uint32))
        sent = float((left8[:4] == 1.0).mean())
        out = os.path.join(HERE, "volume_%s.npz" % name)
        np.savez_compressed(out, left=l, right=r, ndisp=np.int32(nd), board=np.int32(b),
                            cost_census=costs[0], cost_ncc=costs[1], cost_sobel=costs[2], cost_sad=costs[3],
                            dflt_shape=np.array(costs_dflt[0].shape, np.int32),
                            dflt_sha256=np.frombuffer(recipes.arrays_sha256(costs_dflt), np.uint8),
                            features_left=left8, features_right=lr16[8:])
        print("%-14s images %s  D'=%d  volume %s  clipped-to-1 fraction of cost channels %.3f  -> %s (%.0f KB)" % (
            name, l.shape, nd, left8.shape, sent, os.path.basename(out), os.path.getsize(out) / 1024))


if __name__ == "__main__":
    main()
