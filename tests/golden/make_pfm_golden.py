"""Writes tests/golden/pfm_reference.pfm with the REFERENCE's own writer (src/utils/pfmutil.py:86-110, imported from
/root/reference) for a fixed 3x4 float32 array, so that tests/test_driver_utils.py can compare bytes without the reference."""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
path = None
for root, _, files in os.walk("/root/reference"):
    if "pfmutil.py" in files:
        path = os.path.join(root, "pfmutil.py")
        break
spec = importlib.util.spec_from_file_location("ref_pfmutil", path)
mod = importlib.util.module_from_spec(spec)
try:
    spec.loader.exec_module(mod)
except Exception as e:       # matplotlib etc. may be missing: the writer itself needs numpy and sys only
    sys.exit("cannot import the reference pfmutil: %s" % e)
img = (np.arange(12, dtype=np.float32).reshape(3, 4) - 3.5)
mod.save(os.path.join(HERE, "pfm_reference.pfm"), img)
print("wrote", os.path.join(HERE, "pfm_reference.pfm"))
