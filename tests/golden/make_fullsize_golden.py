"""Generates tests/golden/fullsize_<case>.npz: the REFERENCE aggregators (/root/reference/src/models/, imported
unmodified, same harness-side shims as make_aggregator_golden.py) run on CPU at the FULL benchmark shapes of
BASELINE.json configs #2 / #5 (MS-GCNet) and #3 (PSMNet aggregator), with random-init and with `peaky`
(trained-network-like softmax) weights -- recipes.FULL_CASES.

Runs only in the build container (needs /root/reference, ~10 GB of RAM, a few minutes); what it writes is data:
the reference's full disparity map, strided samples of its activations / logits, the sha256 of the seeded
state_dict, and softmax statistics of the case (printed and stored, so a reader can see how peaky it is).

    python tests/golden/make_fullsize_golden.py [case ...]
    python tests/golden/make_fullsize_golden.py --alt [case ...]      (round 4)

--alt writes tests/golden/fullsize_<case>_alt.npz for recipes.ALT_CASES: the SAME unmodified reference forward, same seeded
weights, same input, run again under recipes.ALT_VARIANTS -- torch.set_num_threads(1) instead of the 8 threads of the
main fixture, and oneDNN switched off (ATen's vol2col + GEMM convolution; 47 GB peak for MS-GCNet: nothing else may run).  The convolutions and reductions sum in another
order then, so |disp_alt - disp| is the reference's own fp32 summation-order noise floor at this shape: what the GPU tests
hold the HIP-vs-reference error distribution against.

    python tests/golden/make_fullsize_golden.py --f64 [case ...]      (round 6)

--f64 writes tests/golden/fullsize_<case>_f64.npz for recipes.F64_CASES: the SAME unmodified reference module, same seeded weights
and input, moved to float64 (`ref.double()`, input `.double()`): every convolution, BatchNorm, the softmax and the regression in
double precision, i.e. the mathematically exact forward to ~1e-13.  |disp_f64 - disp| is then how far the reference's OWN float32
forward sits from the exact answer, and |HIP - disp_f64| how far the HIP path does: the GPU tests hold the second against the
first (a float32 implementation cannot be asked to be closer to another float32 implementation than both are to the truth).
One harness-side memory shim, M1: ATen's float64 Conv3d on CPU is vol2col + GEMM with a [Ci*27, D*H*W] column buffer (86 GB for
conv3dbn_2), so nn.Conv3d.forward is wrapped to run stride-1 convolutions in depth slabs (each output voxel is still computed
by the same ATen routine from the same 27*Ci products; only the number of voxels per call changes).
"""
import contextlib
import io
import os
import sys
import time

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import numpy as np
import torch
import torch.nn.functional as F

import recipes
from oracle import aggregators as oracle

torch.Tensor.cuda = lambda self, *a, **k: self          # D1
import src.models.gcnet_3dcnn as ref_gc                # noqa: E402
import src.models.psmnet_3dcnn as ref_psm              # noqa: E402

import msnets_amd                                      # noqa: E402,F401
from msnets_amd.gcnet_3dcnn import GCNet_CostVolumeAggre as OurGC      # noqa: E402
from msnets_amd.psmnet_3dcnn import PSMNet_CostVolumeAggre as OurPSM   # noqa: E402

GC_TAPS = ["conv3dbn_1", "conv3dbn_2", "block_3d_1", "block_3d_2", "block_3d_3", "block_3d_4", "deconv5"]
NS = recipes.FULL_MAX_SAMPLES


def case_input(case):
    if case.get("ms_volume"):
        from oracle import ms_volume as O
        from msnets_amd import synthetic
        n, c, d, h, w = case["in_shape"]
        left, right, _ = synthetic.stereo_pair(h, w, d, seed=case["seed"])
        return torch.from_numpy(O.build_ms_volume(left, right, d)).unsqueeze(0)
    return recipes.full_input(case)


def case_rows(case):
    """Planted half-resolution disparity per row of an ms_volume case (synthetic.stereo_pair)."""
    from msnets_amd import synthetic
    n, c, d, h, w = case["in_shape"]
    return synthetic.stereo_pair(h, w, d, seed=case["seed"])[2]


def run_alt(name, case):
    """The reference again under other thread counts -> fullsize_<name>_alt.npz (see the module docstring)."""
    t0 = time.time()
    gold = np.load(os.path.join(HERE, "fullsize_%s.npz" % name))
    with contextlib.redirect_stdout(io.StringIO()):
        ref = recipes.build_case(case, ref_gc.GCNet_CostVolumeAggre, ref_psm.PSMNet_CostVolumeAggre)
    assert recipes.state_sha256(ref.state_dict()) == str(gold["state_sha256"])
    x = case_input(case)
    H, W = recipes.out_hw(case)
    base = torch.from_numpy(gold["disp"])
    out = {"state_sha256": gold["state_sha256"]}
    for variant in recipes.ALT_VARIANTS:
        torch.set_num_threads(int(variant[1:]) if variant[0] == "t" else 8)
        mk = torch.backends.mkldnn.flags(enabled=(variant != "nomkldnn"))
        keep, hooks = {}, []
        if case["model"] == "gcnet":
            hooks.append(ref.deconv5.register_forward_hook(lambda m, i, o: keep.__setitem__("deconv5", o.detach().clone())))
            tapname = "deconv5"
        else:
            ref_psm.left = torch.empty(1, 3, H, W)          # D2
            for t in ("classif1", "classif2", "classif3"):
                hooks.append(getattr(ref, t).register_forward_hook(lambda m, i, o, t=t: keep.__setitem__(t, o.detach().clone())))
            tapname = "cost3"
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()), mk:
            disp = ref(x.clone())
        for h in hooks:
            h.remove()
        if case["model"] == "gcnet":
            tap = keep["deconv5"]
        else:
            tap = keep["classif3"] + (keep["classif2"] + keep["classif1"])      # psmnet_3dcnn.py:140-147
        s, stride = recipes.sample(tap, NS)
        g = gold["tap_" + tapname]
        out["disp_" + variant] = disp.numpy().astype(np.float32)
        out["tap_%s_%s" % (tapname, variant)] = s
        e = (disp - base).abs().flatten()
        q = lambda f: float(e.kthvalue(max(1, int(f * e.numel())))[0])      # noqa: E731
        print("%-24s %-8s: |ref_alt - ref| max %.3e p99.9 %.3e p99 %.3e median %.3e; %.2f%% <= 1e-3; logit samples rel %.2e  (%.0f s)"
              % (name, variant, float(e.max()), q(0.999), q(0.99), q(0.5), 100 * float((e <= 1e-3).double().mean()),
                 float(np.abs(s - g).max() / max(1.0, float(np.abs(g).max()))), time.time() - t0), flush=True)
    torch.set_num_threads(8)
    path = os.path.join(HERE, "fullsize_%s_alt.npz" % name)
    np.savez_compressed(path, **out)
    print("  -> %s (%.2f MB)" % (os.path.basename(path), os.path.getsize(path) / 1e6), flush=True)


F64_COLUMN_BUDGET = 6e9          # bytes of vol2col buffer per Conv3d call (M1)


def _install_conv3d_depth_slabs():
    """M1: nn.Conv3d.forward in depth slabs when ATen's column buffer would not fit (see the module docstring)."""
    if getattr(torch.nn.Conv3d, "_msnet_slabs", False):
        return
    plain = torch.nn.Conv3d.forward

    def forward(self, x):
        n, ci, d, h, w = x.shape
        col = ci * 27 * d * h * w * x.element_size()
        if (x.dtype != torch.float64 or col <= F64_COLUMN_BUDGET or self.stride != (1, 1, 1) or self.kernel_size != (3, 3, 3)
                or self.padding != (1, 1, 1) or self.dilation != (1, 1, 1) or self.groups != 1):
            return plain(self, x)
        step = max(1, int(d * F64_COLUMN_BUDGET / col))
        outs = []
        for d0 in range(0, d, step):
            d1 = min(d, d0 + step)
            lo, hi = max(d0 - 1, 0), min(d1 + 1, d)
            slab = F.pad(x[:, :, lo:hi], (0, 0, 0, 0, 1 if d0 == 0 else 0, 1 if d1 == d else 0))     # zero halo at the ends
            outs.append(F.conv3d(slab, self.weight, self.bias, 1, (0, 1, 1)))
        return torch.cat(outs, 2)
    torch.nn.Conv3d.forward = forward
    torch.nn.Conv3d._msnet_slabs = True


def run_f64(name, case):
    """The reference in float64 -> fullsize_<name>_f64.npz (see the module docstring)."""
    t0 = time.time()
    torch.set_num_threads(8)
    gold = np.load(os.path.join(HERE, "fullsize_%s.npz" % name))
    with contextlib.redirect_stdout(io.StringIO()):
        ref = recipes.build_case(case, ref_gc.GCNet_CostVolumeAggre, ref_psm.PSMNet_CostVolumeAggre)
    assert recipes.state_sha256(ref.state_dict()) == str(gold["state_sha256"])
    x = case_input(case)
    H, W = recipes.out_hw(case)
    _install_conv3d_depth_slabs()
    # self-check of M1 on a small float64 volume: slabs == one call, to the last bit or two of a double
    conv = torch.nn.Conv3d(8, 16, 3, padding=1, bias=False).double()
    xs = torch.rand(1, 8, 12, 10, 14, dtype=torch.float64)
    global F64_COLUMN_BUDGET
    keep_budget, F64_COLUMN_BUDGET = F64_COLUMN_BUDGET, 8 * 27 * 5 * 10 * 14 * 8
    with torch.no_grad():
        a = conv(xs)
        F64_COLUMN_BUDGET = 1e30
        b = conv(xs)
    F64_COLUMN_BUDGET = keep_budget
    assert float((a - b).abs().max()) < 1e-14, float((a - b).abs().max())
    ref = ref.double()
    keep, hooks = {}, []
    if case["model"] == "gcnet":
        hooks.append(ref.deconv5.register_forward_hook(lambda m, i, o: keep.__setitem__("deconv5", o.detach().clone())))
        tapname = "deconv5"
    else:
        ref_psm.left = torch.empty(1, 3, H, W)          # D2
        for t in ("classif1", "classif2", "classif3"):
            hooks.append(getattr(ref, t).register_forward_hook(lambda m, i, o, t=t: keep.__setitem__(t, o.detach().clone())))
        tapname = "cost3"
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        disp = ref(x.double())
    for h in hooks:
        h.remove()
    assert disp.dtype == torch.float64, disp.dtype
    tap = keep["deconv5"] if case["model"] == "gcnet" else keep["classif3"] + (keep["classif2"] + keep["classif1"])
    assert tap.dtype == torch.float64
    flat = tap.reshape(-1)
    stride = max(1, flat.numel() // NS)
    assert stride == int(gold["tapstride_" + tapname])
    ts = flat[::stride].numpy().copy()
    base = torch.from_numpy(gold["disp"]).double()
    e = (disp - base).abs().flatten()
    q = lambda f: float(e.kthvalue(max(1, int(f * e.numel())))[0])      # noqa: E731
    g = gold["tap_" + tapname].astype(np.float64)
    out = {"state_sha256": gold["state_sha256"], "disp_f64": disp.numpy(), "tap_%s_f64" % tapname: ts,
           "ref32_vs_f64": np.array([q(0.5), q(0.99), q(0.999), float(e.max())])}
    path = os.path.join(HERE, "fullsize_%s_f64.npz" % name)
    np.savez_compressed(path, **out)
    print("%-24s float64 reference: |ref_f32 - ref_f64| median %.3e p99 %.3e p99.9 %.3e max %.3e; %.3f%% <= 1e-3; logit samples rel "
          "%.2e  -> %s (%.2f MB, %.0f s)" % (name, q(0.5), q(0.99), q(0.999), float(e.max()), 100 * float((e <= 1e-3).double().mean()),
                                            float(np.abs(ts - g).max() / max(1.0, float(np.abs(ts).max()))), os.path.basename(path),
                                            os.path.getsize(path) / 1e6, time.time() - t0), flush=True)


def softmax_stats(logits):
    """logits [1,D,H,W] fp32 -> dict of how peaky the case is (fp64 softmax)."""
    out = {}
    pmax, kap = [], []
    for h0 in range(0, logits.shape[2], 32):                     # in slabs: the fp64 softmax of the whole map is 800 MB
        l = logits[:, :, h0:h0 + 32].double()
        p = F.softmax(l, 1)
        d = torch.arange(l.shape[1], dtype=torch.float64).view(1, -1, 1, 1)
        disp = (p * d).sum(1, keepdim=True)
        kap.append((p * (d - disp).abs()).sum(1).flatten())
        pmax.append(p.max(1)[0].flatten())
    pmax, kap = torch.cat(pmax), torch.cat(kap)
    out["logit_absmax"] = float(logits.abs().max())
    out["pmax_median"] = float(pmax.median())
    out["pmax_p10"] = float(pmax.kthvalue(max(1, pmax.numel() // 10))[0])
    out["kappa_median"] = float(kap.median())
    out["kappa_max"] = float(kap.max())
    out["_kappa_le1_frac"] = float((kap <= 1).double().mean())
    return out


def run_case(name, case):
    t0 = time.time()
    torch.set_num_threads(8)
    with contextlib.redirect_stdout(io.StringIO()):
        ref = recipes.build_case(case, ref_gc.GCNet_CostVolumeAggre, ref_psm.PSMNet_CostVolumeAggre)
        ours = recipes.build_case(case, OurGC, OurPSM)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    sha = recipes.state_sha256(sd)
    assert sha == recipes.state_sha256(ours.state_dict()), "our seeded init differs from the reference's"
    del ours
    x = case_input(case)
    assert tuple(x.shape) == tuple(case["in_shape"]), x.shape
    H, W = recipes.out_hw(case)
    samples, hooks, keep = {}, [], {}

    def hook(t, relu):
        def f(m, i, o):
            s, stride = recipes.sample(o, NS)
            samples[t] = (np.maximum(s, 0) if relu else s, stride)
            if t == "deconv5":
                keep["logits"] = o.detach().squeeze(1).clone()
            if t.startswith("classif"):
                keep[t] = o.detach().clone()
        return f
    if case["model"] == "gcnet":
        for t in GC_TAPS:
            # conv3dbn_*: the reference applies its (in-place) ReLU outside the Sequential, so the sample gets it here
            hooks.append(getattr(ref, t).register_forward_hook(hook(t, t.startswith("conv3dbn"))))
    else:
        ref_psm.left = torch.empty(1, 3, H, W)          # D2
        for t in ("classif1", "classif2", "classif3"):
            hooks.append(getattr(ref, t).register_forward_hook(hook(t, False)))
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        disp = ref(x.clone())
    for h in hooks:
        h.remove()
    out = {"disp": disp.numpy().astype(np.float32), "state_sha256": np.array(sha)}
    if case["model"] == "gcnet":
        logits = keep.pop("logits")
    else:
        cost2 = keep["classif2"] + keep["classif1"]                 # psmnet_3dcnn.py:140-147, same order => same bits
        cost3 = keep["classif3"] + cost2
        s, stride = recipes.sample(cost3, NS)
        samples = {"cost3": (s, stride)}
        logits = F.interpolate(cost3, [case["maxdisp"], H, W], mode="trilinear", align_corners=True).squeeze(1)
        assert torch.equal(oracle.soft_argmin(logits), disp)
    stats = softmax_stats(logits)
    k1 = stats.pop("_kappa_le1_frac")
    if case.get("unimodal"):                # (only the round-4 case stores these: the older fixtures regenerate unchanged)
        stats["kappa_le1_frac"] = k1
        rows = case_rows(case)
        truth = torch.from_numpy(np.repeat(2.0 * rows, 2)).float().view(1, -1, 1)
        stats["planted_within_half_px_frac"] = float(((disp - truth).abs() <= 0.5).double().mean())
    del logits
    # the oracle restatement must reproduce the reference at this size too (disparity only: no second 8 GB of taps)
    with torch.no_grad():
        if case["model"] == "gcnet":
            d_or = oracle.gcnet_forward(sd, x, case["maxdisp"])
        else:
            d_or = oracle.psmnet_forward(sd, x, case["maxdisp"], (H, W))
    err = float((d_or - disp).abs().max())
    assert err < 1e-4, (name, err)
    out["oracle_max_abs_err"] = np.float32(err)
    for t, (s, stride) in samples.items():
        out["tap_" + t] = s
        out["tapstride_" + t] = np.int64(stride)
    for k, v in stats.items():
        out["stat_" + k] = np.float64(v)
    path = os.path.join(HERE, "fullsize_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("%-20s disp %s range [%.2f, %.2f]  oracle-vs-reference %.1e  max|logit| %.1f  pmax median %.3f p10 %.3f  "
          "kappa median %.2f max %.1f  -> %s (%.2f MB, %.0f s)"
          % (name, tuple(disp.shape), disp.min(), disp.max(), err, stats["logit_absmax"], stats["pmax_median"],
             stats["pmax_p10"], stats["kappa_median"], stats["kappa_max"], os.path.basename(path),
             os.path.getsize(path) / 1e6, time.time() - t0), flush=True)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a not in ("--alt", "--f64")]
    if "--f64" in sys.argv[1:]:
        for name in args or list(recipes.F64_CASES):
            run_f64(name, recipes.FULL_CASES[name])
    elif "--alt" in sys.argv[1:]:
        for name in args or list(recipes.ALT_CASES):
            run_alt(name, recipes.FULL_CASES[name])
    else:
        for name in args or list(recipes.FULL_CASES):
            run_case(name, recipes.FULL_CASES[name])
