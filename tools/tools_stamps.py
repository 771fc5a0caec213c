import sys, os, ctypes
os.environ["MSNET_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ms-nets_amd", "lib_EXP_STAMP.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, tools_layer_bench as T
from msnets_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else "s1_32_32"
T.run(name, "split-fp16", reps=1)
lib = ctypes.CDLL(os.environ["MSNET_HIP_LIB"])
buf = (ctypes.c_ulonglong * 512)()
assert lib.msnet_debug_read_stamps(buf) == 0
for role, nm in ((0, "MFMA wave0"), (1, "loader wave4")):
    v = [buf[role * 256 + i] for i in range(256)]
    t0 = v[0]
    print(nm, "deltas (cycles) between consecutive stamps, first 70:")
    print(" ".join(str(v[i + 1] - v[i]) for i in range(70)))
