"""Host-side driver pieces: crop, PFM round trip with the reference's byte layout, EPE / bad-x, key fix-up."""
import numpy as np

from msnets_amd import driver_utils as du


def test_crop_keeps_bottom_left():
    disp = np.arange(2 * 6 * 8, dtype=np.float32).reshape(2, 6, 8)
    out = du.crop_disparity(disp, 6, 8, 4, 5)
    assert out.shape == (4, 5) and np.array_equal(out, disp[0, 2:6, 0:5])
    assert du.crop_disparity(disp, 6, 8, 9, 9).shape == (6, 8)


def test_pfm_bytes_and_round_trip(tmp_path):
    img = (np.arange(12, dtype=np.float32).reshape(3, 4) - 3.5)
    p = tmp_path / "d.pfm"
    du.save_pfm(str(p), img)
    raw = p.read_bytes()
    head = b"Pf\n4 3\n-1.000000\n"                       # little-endian host: negative scale
    assert raw.startswith(head) and len(raw) == len(head) + 48
    assert np.array_equal(np.frombuffer(raw[len(head):], "<f4").reshape(3, 4), img[::-1])   # rows bottom-up
    assert np.array_equal(du.read_pfm(str(p)), img)
    rgb = np.random.default_rng(0).random((5, 7, 3), dtype=np.float32)
    du.save_pfm(str(p), rgb)
    assert p.read_bytes().startswith(b"PF\n7 5\n") and np.array_equal(du.read_pfm(str(p)), rgb)


def test_pfm_matches_the_references_writer(tmp_path):
    """tests/golden/pfm_reference.pfm was written by the reference's pfmutil.save (make_pfm_golden.py) for this array."""
    import os
    gold = open(os.path.join(os.path.dirname(__file__), "golden", "pfm_reference.pfm"), "rb").read()
    img = (np.arange(12, dtype=np.float32).reshape(3, 4) - 3.5)
    p = tmp_path / "d.pfm"
    du.save_pfm(str(p), img)
    assert p.read_bytes() == gold
    assert np.array_equal(du.read_pfm(os.path.join(os.path.dirname(__file__), "golden", "pfm_reference.pfm")), img)


def test_epe_and_bad_rate():
    gt = np.array([[0.0, 10.0, 50.0], [200.0, 20.0, 30.0]], np.float32)       # 0 and 200 are outside the mask
    pr = np.array([[5.0, 11.0, 46.0], [0.0, 20.5, 40.0]], np.float32)
    epe, rate = du.get_epe_rate(gt, pr, max_disp=192, threshold=3.0)
    assert abs(epe - (1 + 4 + 0.5 + 10) / 4) < 1e-6 and abs(rate - 0.5) < 1e-9


def test_strip_module_prefix():
    sd = {"module.conv3dbn_1.0.weight": 1, "deconv5.bias": 2}
    assert du.strip_module_prefix(sd) == {"conv3dbn_1.0.weight": 1, "deconv5.bias": 2}
