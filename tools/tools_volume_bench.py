"""Volume build alone, in a loop: per-kernel microseconds per map from the library's own HIP events (msnet_prof_*).
    python tools/tools_volume_bench.py [cfg2|cfg5|cfg1] [reps] [ndhwc|ncdhw]      (MSNET_HIP_LIB=... picks another build of the ABI)
Also prints the likelihood channels' max |difference| against the oracle on a 68x100, D'=16 pair (the AML tolerance gate)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import msnets_amd  # noqa: E402,F401
from msnets_amd import _lib, cbmv_generator, synthetic  # noqa: E402

SHAPES = {"cfg1": (128, 256, 32), "cfg2": (272, 480, 96), "cfg5": (192, 624, 96)}


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    layout = sys.argv[3] if len(sys.argv) > 3 else "ndhwc"
    hh, wh, nd = SHAPES[cfg]
    dev = torch.device("cuda:0")
    l, r, _ = synthetic.stereo_pair(hh, wh, nd, seed=0)
    l, r = torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)
    b = cbmv_generator.VolumeBuilder(hh + 20, wh + 20, nd, dev, layout=layout)
    out = torch.empty(b.out_shape, device=dev, dtype=torch.float32)
    for _ in range(3):
        b(l, r, out=out)
    torch.cuda.synchronize()
    _lib.prof_enable(True, None)
    for _ in range(reps):
        b(l, r, out=out)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    prof = _lib.prof_collect()
    row = {k: 1e3 * v["ms"] / reps for k, v in sorted(prof.items())}
    print("%s %s lib=%s  " % (cfg, layout, os.path.basename(_lib.LIB_PATH)) + "  ".join("%s %.1f us" % kv for kv in row.items()), flush=True)
    # AML deviation against the oracle (CPU libm expf), small pair
    from oracle import ms_volume as O
    ls, rs, _ = synthetic.stereo_pair(48, 80, 16, seed=0)
    got = cbmv_generator.build_ms_volume(torch.from_numpy(ls).to(dev), torch.from_numpy(rs).to(dev), 16).cpu().numpy()
    ref = O.build_ms_volume(ls, rs, 16)
    print("  small pair: cost channels bit-equal %s, likelihood max|diff| %.3e" % (
        bool((got[:4].view(np.uint32) == ref[:4].view(np.uint32)).all()), float(np.abs(got[4:] - ref[4:]).max())), flush=True)


if __name__ == "__main__":
    main()
