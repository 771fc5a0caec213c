"""Per-wave barrier stamps of one layer on a given -DEXP_STAMP build: prints, per wave, the cycles between consecutive stamps
for one steady-state item.   python tools/tools_stamps2.py <lib.so> <layer> [first_stamp] [count]"""
import sys, os, ctypes
os.environ["MSNET_HIP_LIB"] = os.path.abspath(sys.argv[1])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, tools_layer_bench as T
name = sys.argv[2]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 40
cnt = int(sys.argv[4]) if len(sys.argv) > 4 else 41
T.run(name, "split-fp16", reps=1)
lib = ctypes.CDLL(os.environ["MSNET_HIP_LIB"])
buf = (ctypes.c_ulonglong * (12 * 128))()
assert lib.msnet_debug_read_stamps(buf) == 0
for w in range(12):
    if not buf[w * 128]: continue
    v = [buf[w * 128 + i] for i in range(128)]
    d = [v[i + 1] - v[i] for i in range(lo, lo + cnt - 1)]
    print("wave %2d (%s): item span %6d  deltas: %s" % (w, "MFMA  " if w < 4 else "loader", v[lo + 20] - v[lo] if lo + 20 < 128 else -1, " ".join("%5d" % x for x in d)))
