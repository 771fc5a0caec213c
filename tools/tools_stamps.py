import sys, os, ctypes
os.environ["MSNET_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ms-nets_amd", "lib_EXP_STAMP.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, tools_layer_bench as T
from msnets_amd import _lib
name = sys.argv[1] if len(sys.argv) > 1 else "s1_32_32"
T.run(name, "split-fp16", reps=1)
lib = ctypes.CDLL(os.environ["MSNET_HIP_LIB"])
buf = (ctypes.c_ulonglong * (12 * 128))()
assert lib.msnet_debug_read_stamps(buf) == 0
t0 = min(buf[w * 128] for w in range(12) if buf[w * 128])
for w in range(12):
    if not buf[w * 128]: continue
    v = [buf[w * 128 + i] for i in range(128)]
    lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (40, 84)
    print("wave %d (%s): stamp times relative to the first stamp of the block, stamps %d..%d:" % (w, "MFMA" if w < 4 else "loader", lo, hi - 1))
    print(" ".join(str(v[i] - t0) for i in range(lo, hi)))
