"""Round 6, VERDICT r05 #3: does co-scheduling two maps on DISJOINT CU sets use the package's power headroom?

The step of config #2 alternates MFMA-bound kernels (64->64, Winograd-depth: 1.5-1.8 GHz under the 1.4 kW cap) with HBM-shaped
ones (first layer, stride 2, transposed, tail: 2.0-2.25 GHz, power to spare).  Premise check before any per-kernel routing:
TWO host threads, each with its own CU-masked HIP stream (hipExtStreamCreateWithCUMask), its own module (packed weights +
activation arena) and its own volume builder, each running whole forwards of its share of the batch; the two streams drift
against each other, so MFMA phases of one meet HBM phases of the other about half the time.  Compared, interleaved on one box:

    single:B          one stream, all CUs, B maps per step (what bench.py --batch-per-gpu B times)
    two:MASKA/MASKB:b two threads x b maps per step; MASK = number of CUs (low bits of the mask = the same count in every XCD,
                      KFD spreads mask bit i to XCC i % 8) or `all` (no mask: the streams compete for every CU)
                      MASKB = `rest` takes the complement of MASKA

Prints one row per run: maps/s of the whole device, package power, granted clock.  Diagnostic; never the headline.

    python tools/r06_cumask_ab.py --configs "single:4;two:128/rest:2;two:all/all:2" --steps 6 --repeats 3
"""
import argparse
import ctypes
import os
import sys
import threading
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402  (PowerMeter)
import msnets_amd  # noqa: E402,F401
from msnets_amd import _lib, cbmv_generator, synthetic  # noqa: E402
from msnets_amd.gcnet_3dcnn import GCNet_CostVolumeAggre  # noqa: E402

HIP = ctypes.CDLL("libamdhip64.so")
NCU = 256


def masked_stream(bits):
    """bits: iterable of CU indices (0..255) -> raw hipStream_t with that CU mask."""
    words = (ctypes.c_uint32 * (NCU // 32))()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = HIP.hipExtStreamCreateWithCUMask(ctypes.byref(s), NCU // 32, words)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
    return s.value


class Lane:
    """One stream's worth of state: module, builder, pairs, volume."""

    def __init__(self, dev, b, seed0, H=544, W=960, D=192):
        hh, wh, nd = H // 2, W // 2, D // 2
        self.pairs = []
        for i in range(b):
            l, r, _ = synthetic.stereo_pair(hh, wh, nd, seed=seed0 + i)
            self.pairs.append((torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)))
        torch.manual_seed(0)
        m = GCNet_CostVolumeAggre(D).eval()
        synthetic.randomize_bn(m, 0)
        self.model = m.to(dev)
        self.builder = cbmv_generator.VolumeBuilder(hh + 20, wh + 20, nd, dev, layout="ndhwc")
        self.vol = torch.empty((b,) + self.builder.out_shape, device=dev, dtype=torch.float32)
        self.b = b

    def step(self):
        for k, (l, r) in enumerate(self.pairs):
            self.builder(l, r, out=self.vol[k])
        return self.model.forward_ndhwc(self.vol)


def run_single(dev, b, steps, warm, meter):
    lane = Lane(dev, b, 0)
    for _ in range(warm):
        out = lane.step()
    torch.cuda.synchronize()
    meter.start()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = lane.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    p = meter.stop()
    return b * steps / dt, p, out


def run_two(dev, masks, b, steps, warm, meter, offset_ms=0.0):
    lanes = [Lane(dev, b, 0), Lane(dev, b, b)]
    streams = []
    for m in masks:
        streams.append(torch.cuda.Stream(device=dev) if m is None else torch.cuda.ExternalStream(masked_stream(m), device=dev))
    gate = threading.Barrier(3)
    done = [0.0, 0.0]
    outs = [None, None]
    errs = []

    def worker(k):
        try:
            torch.cuda.set_device(dev)
            with torch.cuda.stream(streams[k]):
                for _ in range(warm):
                    outs[k] = lanes[k].step()
                streams[k].synchronize()
                gate.wait()
                if k == 1 and offset_ms > 0:
                    time.sleep(offset_ms * 1e-3)
                for _ in range(steps):
                    outs[k] = lanes[k].step()
                streams[k].synchronize()
                done[k] = time.perf_counter()
        except Exception as e:      # noqa: BLE001
            errs.append(e)
            try:
                gate.abort()
            except Exception:       # noqa: BLE001
                pass
    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    meter.start()
    torch.cuda.synchronize()
    gate.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    if errs:
        raise errs[0]
    torch.cuda.synchronize()
    dt = max(done) - t0
    p = meter.stop()
    return 2 * b * steps / dt, p, outs[0]


def parse_mask(tok, other=None):
    if tok == "all":
        return None
    if tok == "rest":
        used = set(other or [])
        return [i for i in range(NCU) if i not in used]
    return list(range(int(tok)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="single:4;two:128/rest:2;two:all/all:2")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--offset-ms", type=float, default=0.0, help="thread 1 starts this much later (half a network: ~7 x b)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    _lib.load()
    meter = bench.PowerMeter(0, dev)
    ref = None
    print("# config                 maps/s   power_W  sclk_MHz   (steps %d, warmup %d)" % (args.steps, args.warmup), flush=True)
    for rep in range(args.repeats):
        for cfg in args.configs.split(";"):
            parts = cfg.split(":")
            if parts[0] == "single":
                v, p, out = run_single(dev, int(parts[1]), args.steps, args.warmup, meter)
            else:
                a, b_ = parts[1].split("/")
                ma = parse_mask(a)
                mb = parse_mask(b_, ma)
                v, p, out = run_two(dev, [ma, mb], int(parts[2]), args.steps, args.warmup, meter, args.offset_ms)
            o0 = out[0].float().cpu()
            if ref is None:
                ref = o0
            same = bool(torch.equal(ref, o0))            # map 0 is the same pair in every configuration: same bits expected
            print("%-22s %8.2f  %8s  %8s   rep %d  map0 bits %s" % (
                cfg, v, "%.0f" % p["power_w"] if p["power_w"] else "n/a", "%.0f" % p["sclk_mhz"] if p["sclk_mhz"] else "n/a", rep,
                "same" if same else "DIFFER"), flush=True)
            del out
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
