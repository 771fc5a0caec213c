#!/usr/bin/env python3
"""Headline benchmark: disparity maps / second of the MS-GCNet cost-volume forward pass
(BASELINE.json: "disparity maps/sec, 960x540 D=192 MS-GCNet fwd @1/2/4/8 GPU").

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch of synthetic stereo pairs already resident in HBM:
  two bordered uint8 half-res images -> HIP matching-space volume [8,96,272,480] -> HIP MS-GCNet (19 3-D convs; by default
  on the split-fp16 MFMA path: every fp32 operand as fp16 hi + fp16 lo, three fp16 MFMAs per product, fp32 accumulate;
  --precision fp32 runs them on the fp32-input MFMA) -> fused deconv5 + soft-argmin -> disparity [544,960]; for N>1 an
  RCCL all-gather of the per-rank maps closes the step.  Weak scaling: every rank processes --batch-per-gpu pairs per step.

Rank 0 prints ONE JSON line (contract in the task statement) extended with
  roofline        : the kernel FAMILY with the largest total time per step (at config #2 the 64->64 stride-1 convs) -- algorithmic
                    FLOPs / HIP-event time vs the dense fp16 MFMA peak / executed MFMAs per algorithmic product (3 split-fp16, 2 in
                    the Winograd-depth launch), vs the MFMA rate measured on this device, and the top three families (`kernels`);
  roofline_step   : the convs' algorithmic FLOPs per map / wall time per step against the blended MFMA ceiling;
  roofline_volume : the matching-space volume build -- 401.4 MB algorithmic per map / its HIP-event time vs 8 TB/s (vendor) and vs a
                    float4 copy measured on this device, with what the tracked counter pass says limits it (`limiter`);
  power           : package power over the timed region (the SMU's energy counter) and the granted shader clock (per-XCD probe);
  step_ms         : median / p10 / p90 of the K per-step times (HIP events between the steps of the timed region);
  fp32_exact      : the same step timed again with every conv on the exact fp32-input MFMA (the reference's arithmetic);
  dropin_ncdhw    : the same step through the REFERENCE's module contract -- VolumeBuilder(layout="ncdhw") + forward(cv[N,8,D',H',W'])
                    (gcnet_3dcnn.py:97) -- timed behind the fp32 loop, same bits as the headline route (`max_abs_diff_vs_headline_route`
                    = 0); `value` itself times the channels-last hand-over (forward_ndhwc) unless --volume-layout ncdhw.  Compare the
                    two routes with interleaved runs of the two flags, not within one line;
  cpu_baseline    : the CPU oracle (a port, not the reference binary) timed on this host on a bounded sample: `value`, and
                    `volume_s` / `aggregator_s` as numbers;
  config.per_rank : one row per rank (device, PCI bus id, NUMA node bound, pairs per step, local ms per step), gathered through the
                    process group; config.weights says how the timed weights were drawn (seeded net_init + randomised BatchNorm).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

FP32_MATRIX_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md "Peak FP32 (matrix)", spec; 155 measured
FP16_MATRIX_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA", dense
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md "HBM3E peak BW" (spec); 6290 measured there with a float4 copy
SPLIT_MFMAS_PER_PRODUCT = 3         # split-fp16: ah*wh, al*wh, ah*wl  
PMC_FILE = os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")
FACTS_FILE = os.path.join(ROOT, "profiles", "r06_profile_facts.json")     # tools/tools_profile_facts.py (rocprofv3 summaries)
WORKLOADS = {
    # name: (padded H, W, maxdisp, description)
    "cfg2": (544, 960, 192, "MS-GCNet forward (MS volume build + 19-conv aggregator + soft-argmin), Scene-Flow "
                            "960x540 padded to 960x544, D=192"),
    "cfg5": (384, 1248, 192, "MS-GCNet forward, KITTI 1242x375 padded to 1248x384, D=192"),
    "cfg1": (256, 512, 64, "MS-GCNet forward, 256x512, D=64"),
    "cfg3": (544, 960, 192, "PSMNet-style aggregator forward on a random [64, D/4, H/4, W/4] volume (module as released), "
                            "960x540 padded to 960x544, D=192 -- NOT the headline metric"),
}
VOLUME_FAMILIES = ("vol_",)


def gcnet_flops(H, W, D):
    """Algorithmic FLOPs of the 19 convs per map: 2 * 27 * Ci * Co * voxels."""
    d, h, w = D // 2, H // 2, W // 2
    v = [d * h * w // (8 ** k) for k in range(5)]
    mac = 27 * (8 * 32 * v[0] + 32 * 32 * v[0])
    chans = [(32, 64), (64, 64), (64, 64), (64, 128)]
    for k, (ci, co) in enumerate(chans, start=1):
        mac += 27 * (ci * co + 2 * co * co) * v[k]
    for (ci, co, k) in [(128, 64, 4), (64, 64, 3), (64, 64, 2), (64, 32, 1), (32, 1, 0)]:
        mac += 27 * ci * co * v[k]
    return 2.0 * mac


def psmnet_flops(H, W, D):
    """Algorithmic FLOPs of the PSMNet aggregator's 28 convs per map (505 GMAC at cfg#3)."""
    v0 = (D // 4) * (H // 4) * (W // 4)
    mac = 27 * v0 * (64 * 32 + 3 * 32 * 32)                                      # dres0, dres1
    mac += 3 * 27 * v0 * (32 * 64 // 8 + 64 * 64 // 8 + 3 * 64 * 64 // 64 + 64 * 32 // 8)      # three hourglasses
    mac += 3 * 27 * v0 * (32 * 32 + 32)                                          # three classification heads
    return 2.0 * mac


class PowerMeter:
    """Package power and granted shader clock OVER the timed region.
    power_w  : energy counter of the package (rocm_smi_lib rsmi_dev_energy_count_get: the SMU's accumulator, 15.3 uJ units)
               read right before and right after the region -> joules / seconds.  Reads sysfs through the library: no HIP
               call, no child process.  (The instantaneous hwmon files of this pool -- power1_input, freq1_input, pp_dpm_sclk --
               read IDLE values while the GPU is loaded, profiles/r04_power_probe.txt, so they are not used.)
    sclk_mhz : two probe launches (msnet_clock_probe: one thread per XCD records its XCD's counters) on the step's stream, the
               first in front of the first step and the second behind the last: shader-clock ticks (s_memtime) / constant-100-MHz
               ticks (s_memrealtime) between them, per XCD, averaged.
    Every field is None where the box offers no such source."""

    def __init__(self, index, dev):
        import ctypes
        self.ct, self.index, self.rsmi, self.e0, self.t0 = ctypes, index, None, None, None
        # rocm_smi numbers the node's devices; HIP numbers the VISIBLE ones.  Under any visible-device remapping `index` would name
        # another GPU's energy counter: no power figure then (the clock probe runs on the device itself and stays).
        from msnets_amd import dist as _msdist
        remapped = _msdist.devices_remapped()              # ("0" / "0,1,..." re-number nothing: this pool exports ROCR_VISIBLE_DEVICES=0)
        for cand in (() if remapped else ("/opt/rocm/lib/librocm_smi64.so", "librocm_smi64.so")):
            try:
                lib = ctypes.CDLL(cand)
                if lib.rsmi_init(0) == 0:
                    self.rsmi = lib
                    break
            except (OSError, AttributeError):
                pass
        self.clk = torch.zeros(32, dtype=torch.int64, device=dev)          # two probes x 8 XCDs x {shader ticks, 100 MHz ticks}

    def _energy_j(self):
        if self.rsmi is None:
            return None
        ct = self.ct
        e, res, ts = ct.c_uint64(), ct.c_float(), ct.c_uint64()
        if self.rsmi.rsmi_dev_energy_count_get(self.index, ct.byref(e), ct.byref(res), ct.byref(ts)) != 0:
            return None
        return e.value * res.value * 1e-6

    def start_energy(self):
        self.e0, self.t0 = self._energy_j(), time.perf_counter()

    def start_clock(self):
        from msnets_amd import _lib
        _lib.check(_lib.load().msnet_clock_probe(_lib.ptr(self.clk), _lib.stream_ptr()), "msnet_clock_probe")

    def start(self):
        self.start_energy()
        self.start_clock()

    def stop(self):
        """Call behind the region's closing synchronize."""
        from msnets_amd import _lib
        _lib.check(_lib.load().msnet_clock_probe(_lib.ptr(self.clk[16:]), _lib.stream_ptr()), "msnet_clock_probe")
        torch.cuda.synchronize()
        e1, t1 = self._energy_j(), time.perf_counter()
        c = [int(v) for v in self.clk.cpu()]
        per_xcd = []                        # the XCDs' shader-clock counters are separate counters: one ratio per XCD, then the mean
        for i in range(8):
            t0_, r0_, t1_, r1_ = c[2 * i], c[2 * i + 1], c[16 + 2 * i], c[16 + 2 * i + 1]
            if t0_ and t1_ and r1_ > r0_ and t1_ > t0_:
                per_xcd.append((t1_ - t0_) / ((r1_ - r0_) / 1e8) / 1e6)
        ticks, real = (sum(per_xcd) / len(per_xcd), 1e-6) if per_xcd else (0, 0)
        # (a region shorter than the counter's update period reads the same count twice: no power figure then)
        return {"power_w": (e1 - self.e0) / (t1 - self.t0) if self.e0 is not None and e1 is not None and e1 > self.e0 and t1 > self.t0 else None,
                "energy_j_per_map": None,
                "sclk_mhz": ticks if per_xcd else None, "sclk_mhz_per_xcd": [round(v, 1) for v in per_xcd],
                "source": "power: package energy counter (rocm_smi_lib) over the timed region; clock: s_memtime / s_memrealtime "
                          "between two probe launches around it, per XCD, averaged"}


def _source_sha(*names):
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(ROOT, "ms-nets_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(key, workload, batch):
    """HBM bytes per launch / per map from the committed rocprofv3 PMC passes (FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for wide coalesced reads on gfx950, WRITE_SIZE as is) -- measured with tools/tools_pmc_bench.sh on the same
    workload, not by the timed run.  Returned only if it was measured on this workload, this batch size and THIS kernel source
    (sha256 of the .hip files it names); otherwise None."""
    try:
        rec = json.load(open(PMC_FILE))[key]
        if rec.get("workload") != workload or rec.get("batch_per_gpu") != batch:
            return None
        if rec.get("source_sha16") != _source_sha(*rec["sources"]):
            return None
        return rec["hbm_bytes"]
    except Exception:
        return None


def profile_facts(workload, batch):
    """Numbers of the committed rocprofv3 passes (profiles/r05_profile_facts.json: average kernel durations of the
    --kernel-trace --stats run and the clock the chip held under the dominant kernel from the SQ/GRBM counter pass), returned
    only when they were taken on this workload, this batch size and THIS kernel source -- so the live line and the tracked
    profile cannot drift apart unnoticed.  Otherwise None."""
    try:
        rec = json.load(open(FACTS_FILE))
        if rec.get("workload") != workload or rec.get("batch_per_gpu") != batch:
            return None
        if rec.get("source_sha16") != _source_sha(*rec["sources"]):
            return None
        return rec
    except Exception:
        return None


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(seed=0, samples=3):
    """The oracle (CPU port of the same path: C matchers + NumPy glue + torch fp32 aggregator) on a bounded sample, by
    BASELINE.md section 4's protocol: one warm-up on a 32x64 crop (pages in the libraries), ONE untimed full cfg#2 map
    (oneDNN's primitives for the real shapes, the allocator's pool), then `samples` timed full maps (272x480 half-res, D'=96,
    another seeded pair each); `value` is from the MEDIAN map time."""
    from msnets_amd import synthetic
    from msnets_amd.gcnet_3dcnn import GCNet_CostVolumeAggre
    from oracle import aggregators, ms_volume
    cores = min(os.cpu_count() or 1, 64)     # MKL-DNN conv3d stops scaling (and regresses) far below 256 threads
    omp_env = os.environ.get("OMP_NUM_THREADS")
    torch.set_num_threads(cores)
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    torch.manual_seed(0)
    sd = synthetic.randomize_bn(GCNet_CostVolumeAggre(192), 0).eval().state_dict()

    def run(hs, ws, nd, sd_pair):
        left, right, _ = synthetic.stereo_pair(hs, ws, nd, seed=sd_pair)
        t0 = time.time()
        vol = ms_volume.build_ms_volume(left, right, nd)
        t1 = time.time()
        with torch.no_grad():
            aggregators.gcnet_forward(sd, torch.from_numpy(vol).unsqueeze(0), 2 * nd)
        return t1 - t0, time.time() - t1

    run(32, 64, 96, seed)
    run(272, 480, 96, seed)                                  # untimed full-size warm-up
    runs = [run(272, 480, 96, seed + 1 + i) for i in range(max(1, int(samples)))]
    tot = sorted(tv + ta for tv, ta in runs)
    med = tot[len(tot) // 2] if len(tot) % 2 else 0.5 * (tot[len(tot) // 2 - 1] + tot[len(tot) // 2])
    tv, ta = sorted(runs, key=lambda r: abs(r[0] + r[1] - med))[0]
    return {"value": 1.0 / med, "unit": "maps/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": _cpu_model(), "volume_s": tv, "aggregator_s": ta, "samples": len(runs),
            "map_s": [round(t, 3) for t in tot], "torch_num_threads": torch.get_num_threads(),
            "omp_num_threads_env": omp_env, "host_cpus": os.cpu_count(),
            "sample": "median of %d full maps after a 32x64 and one full-size warm-up: oracle volume build %.1fs + torch-CPU "
                      "fp32 GCNet forward %.1fs at 272x480 half-res, D'=96 (times of the median map)" % (len(runs), tv, ta)}


def measure_peaks(dev):
    """Attainable peaks of THIS device: float4 copy (GB/s of read + write) and register-only fp16 MFMA loop (TFLOP/s)."""
    from msnets_amd import _lib
    lib = _lib.load()
    n = 1 << 30
    src = torch.empty(n, device=dev, dtype=torch.uint8)
    dst = torch.empty(n, device=dev, dtype=torch.uint8)
    src.fill_(1)
    ev = lambda: torch.cuda.Event(enable_timing=True)       # noqa: E731

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        best = None
        for _ in range(reps):
            a, b = ev(), ev()
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            t = a.elapsed_time(b)
            best = t if best is None else min(best, t)
        return best

    t_copy = timed(lambda: _lib.check(lib.msnet_peak_copy(_lib.ptr(src), _lib.ptr(dst), n, _lib.stream_ptr()), "msnet_peak_copy"), 5)
    flops = [0.0]

    def mfma():
        flops[0] = lib.msnet_peak_mfma_f16(_lib.ptr(dst), 40000, _lib.stream_ptr())
    t_mfma = timed(mfma, 3)
    f32 = flops[0]

    def mfma16():
        flops[0] = lib.msnet_peak_mfma_f16_16x16(_lib.ptr(dst), 40000, _lib.stream_ptr())
    t_mfma16 = timed(mfma16, 3)
    out = {"hbm_copy_GBs": 2.0 * n / t_copy / 1e6, "mfma_f16_TFLOPs": f32 / t_mfma / 1e9,
           "mfma_f16_16x16x32_TFLOPs": flops[0] / t_mfma16 / 1e9}
    for key, shape in (("mfma_f16_changing_operands_TFLOPs", 0), ("mfma_f16_16x16x32_changing_operands_TFLOPs", 1)):
        def mfma_r(shape=shape):
            flops[0] = lib.msnet_peak_mfma_f16_rand(_lib.ptr(dst), 40000, shape, _lib.stream_ptr())
        t = timed(mfma_r, 3)
        out[key] = flops[0] / t / 1e9
    return out


def self_launch(n):
    """`python bench.py --gpus N` without an external launcher: this process -- which has made NO GPU / HIP call yet and never
    does -- starts `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a CHILD process (it is never
    replaced by another program), relays the child's output (rank 0's JSON line on stdout) and returns its exit code.  The
    child runs in its own process group; if this process is interrupted or terminated the whole group is, so no rank is left
    holding a GPU."""
    import signal
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, MSNET_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs on this pool (a user's own value wins)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, n))))
    argv = [a for a in sys.argv[1:] if a != "--self-launch"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    # rank 0's JSON line goes to stdout alone; anything else the ranks or RCCL print on stdout (RCCL's version banner) is relayed
    # on stderr, so that `python bench.py --gpus N` prints ONE line on stdout as the single-process run does
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)

    def stop_child(sig=signal.SIGTERM):
        if proc.poll() is None:
            try:
                os.killpg(proc.pid, sig)               # the group this Popen created (start_new_session): launcher + ranks, nothing else
            except ProcessLookupError:
                pass

    def on_signal(signum, frame):
        stop_child(signum)
        raise KeyboardInterrupt
    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    try:
        for ln in proc.stdout:
            (sys.stdout if ln.lstrip().startswith("{") else sys.stderr).write(ln)
            sys.stdout.flush()
        return proc.wait()
    except (KeyboardInterrupt, BrokenPipeError):
        return 130
    finally:
        for sg, h in old.items():
            signal.signal(sg, h)
        if proc.poll() is None:
            stop_child()
            try:
                proc.wait(10)
            except subprocess.TimeoutExpired:
                stop_child(signal.SIGKILL)
                proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch-per-gpu", type=int, default=1)
    ap.add_argument("--global-batch", type=int, default=None,
                    help="total pairs per step over all ranks (default: --batch-per-gpu x world); a value that is not a multiple "
                         "of the world size gives the ranks uneven shares (sample i -> rank i mod world) -- diagnostic")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-samples", type=int, default=3, help="timed full maps of the CPU port (median reported)")
    ap.add_argument("--no-extras", action="store_true", help="skip the fp32_exact loop and the peak micro-benchmarks")
    ap.add_argument("--no-volume", action="store_true", help="aggregator only (random volume), not the headline")
    ap.add_argument("--volume-layout", default="ndhwc", choices=["ndhwc", "ncdhw"],
                    help="layout of the volume handed from the build to the aggregator: 'ndhwc' (default) = the kernels' own "
                         "channels-last layout (VolumeBuilder(layout='ndhwc') + forward_ndhwc: no layout pass); 'ncdhw' = the "
                         "reference's layout through the drop-in forward() (one 802 MB conversion pass per map)")
    ap.add_argument("--identity-bn", action="store_true",
                    help="leave BatchNorm at its identity defaults (rounds 1-4); default: seeded non-identity BN statistics and "
                         "affine terms (synthetic.randomize_bn, the recipe of the golden fixtures)")
    ap.add_argument("--precision", default="split-fp16", choices=["split-fp16", "fp32"])
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not record per-launch HIP events (diagnostic)")
    ap.add_argument("--keep-gc", action="store_true", help="leave CPython's cyclic garbage collector enabled inside the timed region")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the aggregator forward from a captured HIP graph (module.use_graph); its launches then carry no "
                         "HIP events, so the roofline of the dominant kernel is not measured in such a run -- diagnostic, not the default")
    ap.add_argument("--dist-backend", default=None, choices=["nccl", "gloo"],
                    help="torch.distributed backend (default: RCCL = 'nccl'; env MSNET_DIST_BACKEND).  'gloo' lets several ranks share "
                         "one GPU (LOCAL_RANK modulo the device count, collective through host memory): a functional run of the "
                         "world > 1 path on a 1-GPU box, not a measurement")
    ap.add_argument("--pipeline", action="store_true",
                    help="serving-loop mode: the volume of step k+1 is built on a side stream while step k is aggregated (two volume "
                         "buffers; every timed step still issues one build and one aggregation).  Steps overlap, so ms_per_step is "
                         "no longer a per-map latency -- diagnostic, not the headline")
    ap.add_argument("--wd64", action="store_true",
                    help="experiment (DESIGN 10): the 64->64 stride-1 layers as four Winograd-depth launches each -- diagnostic")
    ap.add_argument("--self-launch", action="store_true",
                    help="start the ranks as a child `python -m torch.distributed.run` even for --gpus 1 (what --gpus N>1 does by "
                         "itself when no launcher set RANK)")
    args = ap.parse_args()

    if args.dist_backend:
        os.environ["MSNET_DIST_BACKEND"] = args.dist_backend
    if (args.gpus > 1 or args.self_launch) and "RANK" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    import msnets_amd
    from msnets_amd import dist as msdist
    numa_node = None
    if "RANK" in os.environ:        # one process per GPU: keep the launch thread on the GPU's socket (sysfs only, no GPU call yet)
        numa_node = msdist.bind_to_gpu_numa_node(int(os.environ.get("LOCAL_RANK", "0")))
    from msnets_amd import _lib, cbmv_generator, hipops, synthetic
    from msnets_amd.gcnet_3dcnn import GCNet_CostVolumeAggre

    rank, world, local = msdist.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE %d: launch with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the product path")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    _lib.load()
    hipops.set_default_precision(args.precision)
    if args.wd64:
        hipops.USE_WD64 = True

    H, W, D, desc = WORKLOADS[args.workload]
    hh, wh, nd = H // 2, W // 2, D // 2
    n_total = args.global_batch or args.batch_per_gpu * world
    if n_total < world:
        raise SystemExit("--global-batch %d < world size %d" % (n_total, world))
    B = len(msdist.shard_indices(n_total, rank, world))        # this rank's pairs per step (rank 0 holds the largest share)

    # synthetic inputs, resident in HBM before the timed region; sample i -> rank i % world
    pairs = []
    for i in msdist.shard_indices(n_total, rank, world):
        l, r, _ = synthetic.stereo_pair(hh, wh, nd, seed=i)
        pairs.append((torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)))
    torch.manual_seed(0)
    if args.workload == "cfg3":
        from msnets_amd.psmnet_3dcnn import PSMNet_CostVolumeAggre
        model = PSMNet_CostVolumeAggre(D).eval()
        if not args.identity_bn:
            synthetic.randomize_bn(model, 0)
        model = model.to(dev)
        vol = synthetic.random_volume((B, 64, D // 4, H // 4, W // 4), seed=rank).to(dev)
        args.no_volume = True
        # (nothing builds a 64-plane volume on this path -- the reference's PSMNet features are out of scope -- so the synthetic
        # volume is simply handed over in either layout: channels-last through forward_ndhwc, or the reference's NCDHW through forward)
        cl = args.volume_layout == "ndhwc"
        if cl:
            vol = vol.permute(0, 2, 3, 4, 1).contiguous()
    else:
        model = GCNet_CostVolumeAggre(D).eval()
        if not args.identity_bn:
            synthetic.randomize_bn(model, 0)
        model = model.to(dev)
        cl = args.volume_layout == "ndhwc" and not args.no_volume
        builder = cbmv_generator.VolumeBuilder(hh + 20, wh + 20, nd, dev, layout="ndhwc" if cl else "ncdhw")
        vol = torch.empty((B,) + builder.out_shape, device=dev, dtype=torch.float32)
        if args.no_volume:
            vol.copy_(synthetic.random_volume(tuple(vol.shape), seed=rank).to(dev))

    model.use_graph = bool(args.graph)

    def local_step():
        if not args.no_volume:
            for b, (l, r) in enumerate(pairs):
                builder(l, r, out=vol[b])
        return model.forward_ndhwc(vol) if cl else model(vol)

    if args.pipeline and not args.no_volume:
        # software pipeline of a serving loop: build(k+1) on a side stream under aggregate(k); two volume buffers, events both ways
        vols = [vol, torch.empty_like(vol)]
        side = torch.cuda.Stream()
        built = [torch.cuda.Event(), torch.cuda.Event()]
        consumed = [torch.cuda.Event(), torch.cuda.Event()]
        pstate = {"k": 0, "primed": False}

        def enqueue_build(i):
            side.wait_event(consumed[i])                 # the aggregation that last read this buffer (no-op before its first record)
            with torch.cuda.stream(side):
                for b, (l, r) in enumerate(pairs):
                    builder(l, r, out=vols[i][b])
                built[i].record(side)

        def local_step():       # noqa: F811
            i = pstate["k"] & 1
            if not pstate["primed"]:
                enqueue_build(i)
                pstate["primed"] = True
            enqueue_build(i ^ 1)                         # the NEXT step's volume, under this step's aggregator
            torch.cuda.current_stream().wait_event(built[i])
            out_ = model.forward_ndhwc(vols[i]) if cl else model(vols[i])
            consumed[i].record()
            pstate["k"] += 1
            return out_

    def step():
        return msdist.gather_disparities(local_step(), n_total)

    # Setup (untimed, not part of the W warm-up steps): the first forward packs the weights into MFMA order and the next
    # one or two let torch's caching allocator reach its steady-state pool (the path allocates ~6 GB of activations per
    # map; a cold hipMalloc of a 1.6 GB block costs milliseconds).
    # Under RCCL the first barrier creates the communicator and the first step after each of the first barriers is tens
    # of milliseconds slow (lazy RCCL/runtime initialisation), so the setup alternates barriers and steps until that is
    # over -- otherwise it would land in the timed region, whose opening barrier the contract fixes.
    # Everything the timed region needs is created HERE, in front of the first step the device ever runs (round 6): loading
    # rocm_smi_lib and rsmi_init take 0.1-0.5 s, and while that sat between the warm-up and the timed region the GPU idled, dropped
    # its clocks, and the first three timed steps ran 8.4 / 7.5 / 7.1 ms against 6.97 in the steady state
    # (profiles/r06_step_series.txt) -- 1.5-2.5 % of a 20-step headline that measured the idle gap, not the path.  From the first
    # setup step to the closing synchronize the device now only ever waits for the host's short synchronisation points.
    import gc
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    meter = PowerMeter(local, dev) if rank == 0 else None
    if meter:
        meter._energy_j()                               # (first call opens the sysfs file behind the counter)
    if not args.keep_gc:
        gc.collect()                                    # (tens of ms: here, not between the warm-up and the timed region)
    for _ in range(3):
        msdist.barrier()
        out = step()
        torch.cuda.synchronize()

    # Outside the timed region: the gathered batch must be the per-rank maps in sample order (sample i from rank i % world).
    mine = local_step()
    gathered = msdist.gather_disparities(mine, n_total)
    for j, i in enumerate(msdist.shard_indices(n_total, rank, world)):
        assert torch.equal(gathered[i], mine[j]), "all-gather order: sample %d is not rank %d's map %d" % (i, rank, j)
    del mine, gathered

    assert out.shape == (n_total, H, W) and bool(torch.isfinite(out).all())      # (the setup steps' result: checked before the warm-up, not behind it)

    # HIP events inside the timed region: around the launches of the stride-1 conv families (one of them is the step's largest
    # family by total time in every workload) and the volume build only -- the two event records per launch cost 0.18 ms per
    # step (2 %) when all ~45 launches are timed.  Every family is timed in an untimed POST-PASS of three more steps
    # (`roofline.kernels`), which is also what names the dominant family; --verbose times every family inside the timed region.
    f16path = args.precision != "fp32"
    timed_families = ("conv3d_s1_wd_f16s", "conv3d_s1_f16s_co") if f16path else ("conv3d_s1",)
    dom_prefix = None if args.verbose else ",".join(timed_families + VOLUME_FAMILIES)
    # the gc hygiene of the timed region starts in front of the warm-up (nothing but the W steps between here and the barrier)
    if not args.keep_gc:
        gc.freeze()
        gc.disable()
    # exactly W warm-up steps, and nothing between the last of them and the opening barrier + synchronize but two host calls
    for _ in range(args.warmup):
        out = step()
    _lib.prof_enable(not args.no_kernel_timing, dom_prefix)
    if meter:
        meter.start_energy()                            # (a sysfs read: in front of the barrier, not between it and the first step)
    msdist.barrier()
    torch.cuda.synchronize()
    if meter:
        meter.start_clock()
    # host hygiene inside the timed region: CPython's cyclic collector is held off (gc.freeze + disable, re-enabled right after) --
    # a generation-2 pass over torch's module graph takes 10-40 ms and lands in whichever step triggers it (--keep-gc: leave it on).
    gc_log = []
    if args.verbose:
        def _gc_cb(phase, info, _t=[0.0]):
            if phase == "start":
                _t[0] = time.perf_counter()
            else:
                gc_log.append((info.get("generation"), 1e3 * (time.perf_counter() - _t[0])))
        gc.callbacks.append(_gc_cb)
    host_ms = []
    t0 = time.perf_counter()
    step_ev[0].record()
    for k in range(args.steps):
        th = time.perf_counter()
        out = step()
        step_ev[k + 1].record()
        host_ms.append(1e3 * (time.perf_counter() - th))
    torch.cuda.synchronize()
    msdist.barrier()
    dt = time.perf_counter() - t0
    if not args.keep_gc:
        gc.enable()
        gc.unfreeze()
    if args.verbose:
        gc.callbacks.remove(_gc_cb)
    power = meter.stop() if meter else None
    if power and power["power_w"] is not None:
        power["energy_j_per_map"] = power["power_w"] * dt / (args.steps * B)
    _lib.prof_enable(False)
    prof = _lib.prof_collect()
    all_timed = dom_prefix is None and not args.no_kernel_timing
    step_series = [step_ev[k].elapsed_time(step_ev[k + 1]) for k in range(args.steps)]
    per_step = sorted(step_series)
    guard_tripped = model._forced_precision is not None           # the fp16-range guard moved the module to fp32 (hipops)

    dt_local = dt
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if msdist.backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    # one row per rank, through the process group itself: a SCALE run then says on its own line which device, which NUMA node and
    # which local time every rank had ("did RCCL see N ranks on N devices" is answerable from the JSON, no log archaeology)
    props = torch.cuda.get_device_properties(dev)
    mine_row = {"rank": rank, "local_rank": local, "device": props.name, "gcn_arch": getattr(props, "gcnArchName", None),
                "pci_bus_id": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0),
                                                   getattr(props, "pci_device_id", 0)),
                "numa_node_bound": numa_node, "pairs_per_step": B, "ms_per_step_local": 1e3 * dt_local / args.steps}
    if msdist.backend():
        per_rank = [None] * world
        torch.distributed.all_gather_object(per_rank, mine_row)
    else:
        per_rank = [mine_row]

    # post-pass (untimed, every rank: a step contains the collective): all families under HIP events
    POST_STEPS = 3
    if all_timed:
        prof_all, post_steps = prof, args.steps
    elif args.no_kernel_timing:
        prof_all, post_steps = {}, 0
    else:
        _lib.prof_enable(True, None)
        for _ in range(POST_STEPS):
            step()
        torch.cuda.synchronize()
        _lib.prof_enable(False)
        prof_all, post_steps = _lib.prof_collect(), POST_STEPS

    def family_row(name, v, steps, src):
        """One kernel family against the roofline that bounds it.  MFMA families: algorithmic (direct-conv) FLOPs / time vs
        2500 / executed fp16 MFMAs per algorithmic product (3 split-fp16; 2 in the Winograd-depth kernel: 3 x 2/3), or vs the
        fp32-input MFMA peak; HBM families: algorithmic bytes / time vs 8 TB/s."""
        ms = v["ms"]
        row = {"family": name, "ms_per_step": ms / max(1, steps), "launches_per_step": v["calls"] / max(1, steps), "timed_in": src}
        conv = name.startswith(("conv3d_s", "deconv3d")) and v["flops"] > 0
        if conv:
            mf = (2.0 if name == "conv3d_s1_wd_f16s" else 3.0) if "f16s" in name else None
            peak = FP16_MATRIX_PEAK_TFLOPS / mf if mf else FP32_MATRIX_PEAK_TFLOPS
            ach = v["flops"] / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            row.update(bound="mfma", achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak, mfmas_per_algorithmic_product=mf)
        else:
            ach = v["bytes"] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            row.update(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS)
        return row

    if rank == 0:
        maps = n_total * args.steps
        pct = lambda q: float(np.percentile(per_step, q))      # noqa: E731
        # every family of the step, largest first ("volk_*" are the kernels inside "vol_build": not counted twice)
        fam = {k: v for k, v in prof_all.items() if not k.startswith("volk_") and v["ms"] > 0}
        tot_ms = sum(v["ms"] for v in fam.values())
        ranked = sorted(fam, key=lambda k: -fam[k]["ms"])
        kernels = []
        for k in ranked[:3]:
            # a family timed inside the timed region is quoted from there, the others from the post-pass
            row = family_row(k, prof[k], args.steps, "timed region") if k in prof else family_row(k, fam[k], post_steps, "post-pass")
            row["time_share_of_kernels"] = fam[k]["ms"] / tot_ms
            kernels.append(row)
        # dominant = the family with the largest TOTAL time (profiles/*_kernel_stats.csv row 1 names the same kernel)
        if kernels:
            top = kernels[0]
            dom_family = top["family"]
            dom = prof[dom_family] if dom_family in prof else fam[dom_family]
            dom_steps = args.steps if dom_family in prof else post_steps
        else:           # --no-kernel-timing
            dom_family, dom, dom_steps, top = "none", {"ms": 0.0, "flops": 0.0, "calls": 0, "bytes": 0.0}, 1, None
        DESCR = {
            "conv3d_s1_wd_f16s": ("conv3d_wd_f16s_kernel (32->32 stride-1 layers, Winograd F(2,3) along depth, split-fp16 MFMA: 2 MFMAs per "
                                  "algorithmic product)", "fp16 dense MFMA peak 2500 TFLOP/s / 2 executed MFMAs per algorithmic product (3 "
                                  "split-fp16 MFMAs x 2/3 Winograd); `achieved` counts the direct convolution's FLOPs"),
            "conv3d_s1_f16s_co64": ("conv3d_k3s1_f16s_ws, Co = 64 instantiations (the 64->64 stride-1 layers of the encoder; split-fp16 MFMA, "
                                    "3 MFMAs per product)", "fp16 dense MFMA peak 2500 TFLOP/s / 3 MFMAs per algorithmic product"),
            "conv3d_s1_f16s_co32": ("conv3d_k3s1_f16s_ws, Co = 32 (direct split-fp16 MFMA, 3 MFMAs per product)",
                                    "fp16 dense MFMA peak 2500 TFLOP/s / 3 MFMAs per algorithmic product"),
            "conv3d_s1": ("conv3d_k3_mfma_ws (fp32-input MFMA, stride-1 launches)", "fp32-input MFMA peak"),
        }
        dom_name, peak_note = DESCR.get(dom_family, (dom_family, "see roofline.kernels"))
        mfmas = top.get("mfmas_per_algorithmic_product") if top else None
        peak = top["peak"] if top and top["bound"] == "mfma" else (FP16_MATRIX_PEAK_TFLOPS / SPLIT_MFMAS_PER_PRODUCT if f16path
                                                                    else FP32_MATRIX_PEAK_TFLOPS)
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
        conv_ms = sum(v["ms"] for k, v in prof_all.items() if k.startswith(("conv3d", "deconv3d")))
        conv_fl = sum(v["flops"] for k, v in prof_all.items() if k.startswith(("conv3d", "deconv3d")))
        line = {
            "metric": ("disparity maps/sec, 960x540 D=192 MS-GCNet fwd" if args.workload == "cfg2" and not args.pipeline else
                       "disparity maps/sec (%s%s, not the headline)" % (args.workload, ", software-pipelined steps" if args.pipeline else "")),
            "value": maps / dt, "unit": "maps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "f32 (conv operands as split fp16 hi+lo, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": desc + ", batch=%d per GPU" % B, "global_batch": n_total,
                       "parallelism": "dp%d (rank-sharded pairs, %s all-gather of disparity maps)" % (world, msdist.backend() or "no"),
                       "includes_volume_build": not args.no_volume, "hip_graph": bool(args.graph),
                       "pipelined_volume_build": bool(args.pipeline and not args.no_volume),
                       "volume_layout": ("ndhwc (channels-last hand-over, no layout pass)" if cl else
                                         "ncdhw (the reference's layout, one conversion pass)"),
                       "collective": ("%s all_gather_into_tensor" % ("rccl" if msdist.backend() == "nccl" else msdist.backend())
                                      if msdist.backend() else "none (single process)"),
                       "world_size": (torch.distributed.get_world_size() if msdist.backend() else 1),
                       "dist_backend": msdist.backend(),
                       "rccl_version": (".".join(str(v) for v in torch.cuda.nccl.version()) if msdist.backend() == "nccl" else None),
                       "ranks_per_device": msdist.ranks_per_device(),
                       "numa_node_bound": numa_node,
                       "per_rank": per_rank,
                       "distinct_devices": len({r_["pci_bus_id"] for r_ in per_rank}),
                       "launcher": ("bench.py self-launch (child torch.distributed.run)" if os.environ.get("MSNET_BENCH_SELF_LAUNCHED")
                                    else "torch.distributed.run" if "RANK" in os.environ else "single process"),
                       # what the headline depends on besides the code (the step is power-limited, so identical
                       # instruction streams run 5 % apart on different data and 3 % apart on different boxes)
                       "weights": ("net_init (seeded random), " + ("BatchNorm at its identity defaults" if args.identity_bn else
                                   "BatchNorm statistics and affine terms randomised (synthetic.randomize_bn, seed 0: the golden "
                                   "fixtures' recipe)") + " -- no trained checkpoint exists offline"),
                       "range_guard_tripped": bool(guard_tripped)},
            "power": power,
            "step_ms": {"median": pct(50), "p10": pct(10), "p90": pct(90), "source": "HIP events between steps, rank 0",
                        "in_order": [round(t, 3) for t in step_series], "host_issue_ms_in_order": [round(t, 3) for t in host_ms],
                        "python_gc": "enabled" if args.keep_gc else "frozen + disabled over the timed region"},
            "roofline": {"bound": "mfma", "kernel": dom_name, "family": dom_family,
                         "dominant_by": "largest total kernel time per step (all families under HIP events, %s)" % (
                             "timed region" if all_timed else "post-pass of %d untimed steps" % post_steps),
                         "achieved": achieved,
                         "peak": peak, "peak_note": peak_note, "unit": "TFLOP/s", "frac": achieved / peak,
                         "timed_in": top["timed_in"] if top else None,
                         # rounds 1-2 quoted the direct split-fp16 ceiling 2500/3; kept for comparison only
                         "frac_of_direct_split_peak": (achieved / (FP16_MATRIX_PEAK_TFLOPS / SPLIT_MFMAS_PER_PRODUCT)
                                                       if mfmas else None),
                         "mfmas_per_algorithmic_product": mfmas,
                         "time_share_of_step": dom["ms"] / dom_steps / (1e3 * dt / args.steps) if dt > 0 else 0.0,
                         "launches": dom["calls"], "avg_launch_ms": dom["ms"] / max(1, dom["calls"]),
                         "all_conv_tflops": (conv_fl / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else None),
                         "kernels": kernels,
                         "traffic": pmc_traffic(dom_family, args.workload, B)},
        }
        facts = profile_facts(args.workload, B)
        if facts:
            # the tracked rocprofv3 --kernel-trace --stats run of this very source: average duration of the dominant family's launches
            ns = facts.get("family_ns_per_step", {}).get(dom_family)          # all launches of the family in one step (= one map)
            nl = facts.get("family_launches_per_step", {}).get(dom_family)
            if ns and nl and dom["calls"]:
                fl = dom["flops"] / dom_steps / B                            # the family's algorithmic FLOPs per map
                line["roofline"]["avg_launch_ms_rocprof"] = ns * 1e-6 / nl
                line["roofline"]["frac_rocprof"] = fl / (ns * 1e-9) / 1e12 / peak
            else:
                line["roofline"].update(avg_launch_ms_rocprof=None, frac_rocprof=None)
            line["roofline"]["sustained_clock_ghz"] = facts.get("sustained_clock_ghz", {}).get(dom_family)
            line["roofline"]["profile_facts"] = os.path.relpath(FACTS_FILE, ROOT)
        else:
            line["roofline"].update(avg_launch_ms_rocprof=None, frac_rocprof=None, sustained_clock_ghz=None,
                                    profile_facts="none for this kernel source (profiles/ holds an older build's)")
        if f16path and (prof_all or args.workload != "cfg3"):
            # whole-step MFMA roofline: the convs' algorithmic FLOPs per map / the step time (volume build, layout conversion,
            # tail and all gaps included).  Executed fp16 MFMAs per algorithmic product over the whole map: 3 everywhere, 2 in
            # the Winograd-depth launches, whose FLOPs PER MAP come from the events (a launch covers the B maps of the batch).
            fl_map = psmnet_flops(H, W, D) if args.workload == "cfg3" else gcnet_flops(H, W, D)
            step_tf = fl_map * n_total / world / (1e-3 * 1e3 * dt / args.steps) / 1e12
            wd_src, wd_steps = (prof, args.steps) if "conv3d_s1_wd_f16s" in prof else (prof_all, post_steps)
            wd_fl = wd_src.get("conv3d_s1_wd_f16s", {"flops": 0.0})
            wd_fl_map = wd_fl["flops"] / (max(1, wd_steps) * B)
            if not prof_all and args.workload != "cfg3" and hipops.USE_WINOGRAD_DEPTH and \
                    _lib.load().msnet_conv3d_k3_wd_f16s_supported(nd, hh, wh, 32, 32, 1):
                wd_fl_map = 2.0 * 27 * 32 * 32 * nd * hh * wh           # no events at all (--no-kernel-timing): conv3dbn_2, analytically
            step_mfmas = SPLIT_MFMAS_PER_PRODUCT - wd_fl_map / fl_map
            step_peak = FP16_MATRIX_PEAK_TFLOPS / step_mfmas
            line["roofline_step"] = {"bound": "mfma", "achieved": step_tf, "peak": step_peak, "unit": "TFLOP/s",
                                     "frac": step_tf / step_peak,
                                     "frac_of_direct_split_peak": step_tf / (FP16_MATRIX_PEAK_TFLOPS / SPLIT_MFMAS_PER_PRODUCT),
                                     "mfmas_per_algorithmic_product": step_mfmas, "flops_per_map": fl_map,
                                     "winograd_flops_per_map": wd_fl_map,
                                     "note": "per GPU: algorithmic (direct-conv) FLOPs of the maps one rank processes per step / "
                                             "wall time per step; peak = 2500 TFLOP/s fp16 dense / executed MFMAs per algorithmic "
                                             "product averaged over the convs (3, or 2 in the Winograd-depth launches)"}
        volk = {k: v for k, v in prof.items() if k.startswith(VOLUME_FAMILIES)}
        if volk and not args.no_volume:
            vms = sum(v["ms"] for v in volk.values())
            alg_bytes = 4.0 * 8 * nd * hh * wh + 2.0 * (hh + 20) * (wh + 20)        # 401.4 MB at cfg#2
            builds = maps / world
            gbs = alg_bytes * builds / (vms * 1e-3) / 1e9 if vms > 0 else 0.0
            # SURVEY 8(d) prices the build against HBM (its algorithmic bytes / time vs 8 TB/s: `achieved`, `frac`); what LIMITS it
            # is what the tracked counter pass of this very source says (profiles/*_profile_facts.json: "volume_limiter"), null
            # when profiles/ holds no pass of this source
            vfacts = (facts or {}).get("volume_limiter")
            line["roofline_volume"] = {
                "bound": (vfacts or {}).get("bound", "hbm"), "priced_against": "hbm", "limiter": vfacts,
                "kernel": "msnet_build_volume: " + " + ".join(sorted(volk)), "achieved": gbs, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "hbm_frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_map": alg_bytes,
                "us_per_map": 1e3 * vms / builds,
                "us_per_map_by_kernel": {k: 1e3 * v["ms"] / builds for k, v in sorted(volk.items())},
                "traffic": pmc_traffic("volume_build", args.workload, B)}
        if args.verbose:
            tot = sum(v["ms"] for v in prof.values())
            for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
                tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0
                gb = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0
                print("  %-22s calls %4d  %9.3f ms/step (%5.1f%%)  %7.1f TFLOP/s  %8.1f GB/s(alg)" % (
                    k, v["calls"], v["ms"] / args.steps, 100 * v["ms"] / tot, tf, gb), file=sys.stderr)
            print("  kernels %.3f ms/step of %.3f ms/step wall" % (tot / args.steps, 1e3 * dt / args.steps), file=sys.stderr)
            print("  per-step ms (sorted): " + " ".join("%.2f" % t for t in per_step), file=sys.stderr)
            print("  per-step ms (in order): " + " ".join("%.2f" % t for t in step_series), file=sys.stderr)
            print("  host issue ms (in order): " + " ".join("%.2f" % t for t in host_ms), file=sys.stderr)
            print("  python gc passes inside the timed region (generation, ms): %s" % gc_log, file=sys.stderr)
        if world == 1 and not args.no_extras:
            pk = measure_peaks(dev)
            line["peaks_measured"] = dict(pk, note="this device, this run: float4 copy of 1 GiB (read + write bytes); register-only "
                                                   "v_mfma_f32_32x32x16_f16 / 16x16x32 loops on constant small-integer operands and "
                                                   "(changing_operands) on eight pseudo-random full-mantissa fragments cycled per "
                                                   "instruction -- the power-limited rate a real kernel can reach; best of 3-5")
            if mfmas:
                att = pk["mfma_f16_TFLOPs"] / mfmas
                sus = pk["mfma_f16_changing_operands_TFLOPs"] / mfmas
                line["roofline"]["peak_attainable"] = att
                line["roofline"]["frac_attainable"] = achieved / att if att > 0 else None
                # the rate the chip's power management sustains when every MFMA multiplies fresh full-mantissa operands (the conv
                # kernels' case: identical instruction streams run 1.5x faster on all-zero data, tools/tools_power_probe.py)
                line["roofline"]["peak_sustained"] = sus
                line["roofline"]["frac_sustained"] = achieved / sus if sus > 0 else None
                if "roofline_step" in line:
                    rs = line["roofline_step"]
                    att_s = pk["mfma_f16_TFLOPs"] / rs["mfmas_per_algorithmic_product"]
                    sus_s = pk["mfma_f16_changing_operands_TFLOPs"] / rs["mfmas_per_algorithmic_product"]
                    rs["peak_attainable"] = att_s
                    rs["frac_attainable"] = rs["achieved"] / att_s if att_s > 0 else None
                    rs["peak_sustained"] = sus_s
                    rs["frac_sustained"] = rs["achieved"] / sus_s if sus_s > 0 else None
            if "roofline_volume" in line:
                line["roofline_volume"]["peak_attainable"] = pk["hbm_copy_GBs"]
                line["roofline_volume"]["frac_attainable"] = line["roofline_volume"]["achieved"] / pk["hbm_copy_GBs"]
            if args.precision != "fp32" and args.workload != "cfg3":
                # a driver-timed number at the reference's own arithmetic: every conv on the exact fp32-input MFMA
                hipops.set_default_precision("fp32")
                for _ in range(2):
                    step()
                torch.cuda.synchronize()
                k32 = max(3, min(5, args.steps))
                t1 = time.perf_counter()
                for _ in range(k32):
                    step()
                torch.cuda.synchronize()
                d32 = time.perf_counter() - t1
                hipops.set_default_precision(args.precision)
                line["fp32_exact"] = {"value": n_total * k32 / d32, "unit": "maps/s", "ms_per_step": 1e3 * d32 / k32, "steps": k32,
                                      "dtype": "f32 (exact fp32-input MFMA in every conv)"}
        if world == 1 and not args.no_extras and args.workload != "cfg3" and not args.no_volume and not args.pipeline:
            # the reference's module contract on the line: VolumeBuilder(layout="ncdhw") -> model(vol) = forward(cv: f32[N,8,D',H',W'])
            # (gcnet_3dcnn.py:97, cbmv_generator.py:307-308), i.e. the route a maintainer gets by swapping the two imports
            # (INTEGRATION.md); `value` above times the channels-last hand-over unless --volume-layout ncdhw
            b_nc = cbmv_generator.VolumeBuilder(hh + 20, wh + 20, nd, dev, layout="ncdhw")
            v_nc = torch.empty((B,) + b_nc.out_shape, device=dev, dtype=torch.float32)

            def dropin_step():
                for b, (l, r) in enumerate(pairs):
                    b_nc(l, r, out=v_nc[b])
                return msdist.gather_disparities(model(v_nc), n_total)

            o_hl = step().clone()                         # the headline route's maps, same weights, same pairs
            for _ in range(2):
                o_nc = dropin_step()
            torch.cuda.synchronize()
            kd = max(5, min(10, args.steps))
            t1 = time.perf_counter()
            for _ in range(kd):
                o_nc = dropin_step()
            torch.cuda.synchronize()
            dd = time.perf_counter() - t1
            line["dropin_ncdhw"] = {"value": n_total * kd / dd, "unit": "maps/s", "ms_per_step": 1e3 * dd / kd, "steps": kd,
                                    "route": "VolumeBuilder(layout='ncdhw') -> GCNet_CostVolumeAggre.forward(cv[N,8,D',H',W']): the "
                                             "reference's module contract, one NCDHW->NDHWC pass inside forward()",
                                    "max_abs_diff_vs_headline_route": float((o_nc - o_hl).abs().max())}
            del v_nc, b_nc, o_nc, o_hl
        if world == 1 and not args.no_cpu_baseline:
            msdist.restore_affinity()          # the NUMA pinning of a launched rank must not confine the CPU baseline's threads
            line["cpu_baseline"] = cpu_baseline(samples=args.cpu_baseline_samples)
        print(json.dumps(line), flush=True)

    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
