"""Bit-identity of the shipped library against variant builds of the same source (each runs in its own process through
MSNET_HIP_LIB, the A/B hook of ms-nets_amd/_lib.py):

  libx_fullbarrier.so  -DEXP_FULL_GROUP_BARRIER (built by __graft_entry__.build()): the MFMA waves' group barriers drain
                       lgkmcnt(0) again.  The shipped kernels use MSNET_READER_BARRIER there (a bare s_barrier, csrc/conv_common.h)
                       on the invariant that no loader write between two group barriers targets a plane or weight buffer with a
                       prefetch in flight; if a change to the prefetch distances ever broke that invariant, the two builds
                       would stop agreeing -- here, not in a tolerance test.
  libx_r03.so          the round-3 library, when a builder left one in the tree (round 4 pruned the experiment switches and split
                       conv3d_f16s.hip into per-family units: same bits expected).  Skipped when absent.
"""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PKG = os.path.join(ROOT, "ms-nets_amd")

SCRIPT = r'''
import hashlib, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests", "golden"))
import torch
import msnets_amd, recipes
from msnets_amd.gcnet_3dcnn import GCNet_CostVolumeAggre as G
from msnets_amd.psmnet_3dcnn import PSMNet_CostVolumeAggre as P
h = hashlib.sha256()
cases = [dict(model="gcnet", seed=41, maxdisp=64, in_shape=(1, 8, 32, 48, 80)),          # ragged widths: edge tiles of every family
         dict(model="gcnet", seed=42, maxdisp=192, in_shape=(1, 8, 96, 272, 480)),        # config #2: sliding window, Winograd depth
         dict(model="psmnet", seed=43, maxdisp=192, in_shape=(1, 64, 48, 136, 240)),      # config #3: the 16-wide Co = 64 tiles
         dict(model="gcnet", seed=44, maxdisp=64, in_shape=(2, 8, 32, 64, 96))]
for case in cases:
    m = recipes.build_case(case, G, P).cuda()
    x = recipes.make_input(case["in_shape"], case["seed"]).cuda()
    h.update(m(x).cpu().numpy().tobytes())
m16 = G(32, cbmv_in_planes=16).eval().cuda()                                          # 16-channel first layer (ws c16 unit)
torch.manual_seed(5)
h.update(m16(torch.rand(1, 16, 16, 32, 48).cuda()).cpu().numpy().tobytes())
print("SHA", h.hexdigest())
'''


def _sha(lib=None):
    env = dict(os.environ)
    env.pop("MSNET_HIP_LIB", None)
    if lib:
        env["MSNET_HIP_LIB"] = lib
    out = subprocess.run([sys.executable, "-c", SCRIPT % dict(root=ROOT)], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    return [ln for ln in out.stdout.splitlines() if ln.startswith("SHA")][0]


@pytest.fixture(scope="module")
def shipped_sha():
    return _sha()


def test_full_group_barrier_build_is_bit_identical(gpu, shipped_sha):
    lib = os.path.join(PKG, "libx_fullbarrier.so")
    assert os.path.exists(lib), "libx_fullbarrier.so missing: __graft_entry__.build() builds it"
    assert _sha(lib) == shipped_sha


def test_round3_library_is_bit_identical(gpu, shipped_sha):
    lib = os.path.join(PKG, "libx_r03.so")
    if not os.path.exists(lib):
        pytest.skip("no round-3 library in the tree")
    assert _sha(lib) == shipped_sha
