"""How far apart do correct fp32 soft-argmin tails land from each other?  CPU emulation on unimodal logits (noise of sigma 2 plus a
tent of gain 26 at a planted disparity, like tests/golden gcnet_cfg2_ms_unimodal): the reference's tail (torch.softmax + sum p*d,
fp32), the single-chain online softmax of deconv5_tail_mfma_kernel (push / push2 in the kernel's order, every operation rounded
to fp32), the same in four depth segments merged as softargmin_merge_kernel does, and the exact value (fp64).
Round 5, DESIGN 4.2: the variants scatter by ~6e-5 around each other on 3000 pixels -- the level at which the full-size
unimodal case moved (2.44e-4 -> 3.05e-4 vs the reference) when the fused tail was cut into segments."""
import numpy as np
import torch

f32 = np.float32
rng = np.random.default_rng(0)
NP, D = 3000, 192
d = np.arange(D)
peak = rng.integers(20, 190, size=NP)
x = (rng.standard_normal((NP, D)) * 2.0).astype(f32)
for i in range(NP):
    x[i] += np.maximum(0, 1 - np.abs(d - peak[i]) / 2.0).astype(f32) * 26
ref = (torch.softmax(torch.from_numpy(x), 1) * torch.arange(D, dtype=torch.float32)).sum(1).numpy()
x64 = x.astype(np.float64)
e = np.exp(x64 - x64.max(1, keepdims=True))
exact = (e * d).sum(1) / e.sum(1)
GROUPS = [[0]] + [[2 * P - 1, 2 * P] for P in range(1, 96)] + [[191]]          # pushes of slice steps P = 0 .. 96


def run(xi, groups):
    m, s, t = f32(-np.inf), f32(0), f32(0)
    for grp in groups:
        mn = f32(max(m, max(xi[g] for g in grp)))
        a = f32(np.exp(f32(m - mn))) if np.isfinite(m) else f32(0)
        es = [f32(np.exp(f32(xi[g] - mn))) for g in grp]
        s, t = f32(s * a), f32(t * a)
        for g, eg in zip(grp, es):
            s = f32(s + eg)
            t = f32(t + f32(f32(g) * eg))
        m = mn
    return m, s, t


def single(xi):
    _, s, t = run(xi, GROUPS)
    return f32(t / s)


def segmented(xi, nseg=4):
    per = 96 // nseg
    st = []
    for k in range(nseg):
        lo, hi = k * per, (k + 1) * per
        st.append(run(xi, [g for P, g in enumerate(GROUPS) if lo <= P < hi or (k == nseg - 1 and P == 96)]))
    m = max(v[0] for v in st)
    s = t = f32(0)
    for mk, sk, tk in st:
        a = f32(np.exp(f32(mk - m)))
        s = f32(s + f32(sk * a))
        t = f32(t + f32(tk * a))
    return f32(t / s)


one = np.array([single(x[i]) for i in range(NP)])
seg = np.array([segmented(x[i]) for i in range(NP)])
for name, a, b in (("reference tail vs exact", ref, exact), ("single chain  vs exact", one, exact), ("4 segments    vs exact", seg, exact),
                   ("single chain  vs reference tail", one, ref), ("4 segments    vs reference tail", seg, ref),
                   ("4 segments    vs single chain", seg, one)):
    dlt = np.abs(a.astype(np.float64) - b)
    print("%-34s max %.2e  p99 %.2e" % (name, dlt.max(), np.percentile(dlt, 99)))
