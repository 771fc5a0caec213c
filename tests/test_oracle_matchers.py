"""CPU: known-answer and property tests of the matcher oracle (oracle/matchers_oracle.c + oracle/ms_volume.py).
The reference's C++ cannot be built here (Boost.Python), so these pin the restatement to the semantics
SURVEY.md section 8a/H2-H4 spells out, with tiny independent pure-Python/NumPy evaluations."""
import numpy as np
import pytest

from oracle import ms_volume as O

SENT = np.float32(2147483648.0)


def _rand_pair(H, W, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (H, W), dtype=np.uint8), rng.integers(0, 256, (H, W), dtype=np.uint8)


def test_sentinel_is_rand_max_as_float():
    l, r = _rand_pair(20, 30, 0)
    c = O.census(l, r, 4, 11)
    assert c[0, 0, 0] == SENT and c.dtype == np.float32


def test_census_small_known_answer():
    """Independent evaluation of the 11x11 census Hamming distance at every valid (i, j, d)."""
    l, r = _rand_pair(16, 19, 1)
    nd, ws, wc = 5, 11, 5
    got = O.census(l, r, nd, ws)
    H, W = l.shape
    exp = np.full((H, W, nd), SENT, np.float32)
    for i in range(H - ws):
        for j in range(W - ws):
            bl = l[i:i + ws, j:j + ws].astype(int) > int(l[i + wc, j + wc])
            for d in range(min(nd, j + 1)):
                br = r[i:i + ws, j - d:j - d + ws].astype(int) > int(r[i + wc, j - d + wc])
                exp[i + wc, j + wc, d] = np.count_nonzero(bl != br)
    assert np.array_equal(got, exp)


def test_loop_bounds_are_strictly_less_than():
    """i < H - wsize (not <=): the last window position is never evaluated."""
    l, r = _rand_pair(14, 15, 2)
    z = O.zsad(l, r, 2, 5)          # [nd, H, W], valid rows 2 .. H-5-1+2 = 10
    assert np.all(z[0, 11] == SENT) and np.any(z[0, 10] != SENT)
    s = O.sobel(l)                  # valid rows 1 .. H-3-1+1 = 11
    assert np.all(s[12:] == 0) and np.all(s[0] == 0) and np.any(s[11] != 0)


def test_sobel_known_answer():
    img = np.zeros((6, 7), np.uint8)
    img[:, 3:] = 10                                  # vertical edge
    s = O.sobel(img)
    assert s[1, 2] == 0 + 4 * 10 * 1.0               # [-1 0 1; -2 0 2; -1 0 1] on columns (1,2,3) -> (0,0,10)
    assert s[1, 3] == 40.0 and s[1, 4] == 0.0 and s.dtype == np.float32


def test_zsad_small_known_answer():
    l, r = _rand_pair(9, 12, 3)
    nd, ws, wc = 3, 5, 2
    got = O.zsad(l, r, nd, ws)
    f = np.float32
    for d in range(nd):
        for i in range(9 - ws):
            for j in range(d, 12 - ws):
                ml = f(0); mr = f(0)
                for wh in range(ws):
                    for ww in range(ws):
                        ml = f(ml + f(l[i + wh, j + ww])); mr = f(mr + f(r[i + wh, j - d + ww]))
                ml = f(ml / f(25)); mr = f(mr / f(25))
                acc = f(0)
                for wh in range(ws):
                    for ww in range(ws):
                        t = f(f(f(f(l[i + wh, j + ww]) - ml) - f(r[i + wh, j - d + ww])) + mr)
                        acc = f(acc + abs(t))
                assert got[d, i + wc, j + wc] == acc
    assert got[1, 2, 2] == SENT                      # j = 0 < d = 1 never written


def test_ncc_identical_windows_give_minus_one_and_flat_gives_one():
    rng = np.random.default_rng(4)
    l = rng.integers(0, 256, (12, 14), dtype=np.uint8)
    n = O.nccNister(l, l.copy(), 2, 3)
    assert np.allclose(n[0, 1:8, 1:10], -1.0, atol=1e-6)        # perfect correlation -> cost -1
    flat = np.full((12, 14), 77, np.uint8)
    n = O.nccNister(flat, flat, 2, 3)
    assert np.all(n[0, 1:8, 1:10] == 1.0)                       # non-finite normaliser branch (matchers.cpp:203-204)


def test_ncc_small_known_answer():
    """Independent evaluation of nccNister at every valid (i, j, d) from DIRECT window sums (no integral images): every sum is an
    exact integer in double, so -(n*Sum(LR) - Sum(L)Sum(R)) * Cl * Cr with C = 1/sqrt(n*Sum(x^2) - Sum(x)^2), evaluated left to
    right in double and cast to float32 (matchers.cpp:146-147,196-204), must reproduce the oracle's integral-image route bit for bit;
    a non-finite C (constant window) gives 1."""
    rng = np.random.default_rng(11)
    l = rng.integers(0, 256, (11, 14), dtype=np.uint8)
    r = rng.integers(0, 256, (11, 14), dtype=np.uint8)
    l[2:6, 3:8] = 77                                   # a constant patch: non-finite C for the windows inside it
    r[5:9, 1:5] = 200
    nd, ws, wc = 4, 3, 1
    got = O.nccNister(l, r, nd, ws)
    H, W = l.shape
    exp = np.full((nd, H, W), SENT, np.float32)
    L, R = l.astype(np.int64), r.astype(np.int64)
    with np.errstate(divide="ignore"):
        for i in range(H - ws):
            for j in range(W - ws):
                wl = L[i:i + ws, j:j + ws]
                al, bl = int(wl.sum()), int((wl * wl).sum())
                cl = np.float64(1.0) / np.sqrt(np.float64(ws * ws * bl) - np.float64(al) * np.float64(al))
                for d in range(min(nd, j + 1)):
                    wr = R[i:i + ws, j - d:j - d + ws]
                    ar, br = int(wr.sum()), int((wr * wr).sum())
                    cr = np.float64(1.0) / np.sqrt(np.float64(ws * ws * br) - np.float64(ar) * np.float64(ar))
                    if np.isfinite(cl) and np.isfinite(cr):
                        num = np.float64(ws * ws * int((wl * wr).sum()) - al * ar)
                        exp[d, i + wc, j + wc] = np.float32(-num * cl * cr)
                    else:
                        exp[d, i + wc, j + wc] = np.float32(1.0)
    assert (exp == np.float32(1.0)).any() and (exp != SENT).sum() > 100
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))


def test_sadsob_matches_sequential_float32_integral():
    rng = np.random.default_rng(5)
    sl = rng.integers(-1020, 1021, (10, 13)).astype(np.float32)
    sr = rng.integers(-1020, 1021, (10, 13)).astype(np.float32)
    got = O.sadsob(sl, sr, 3, 5)
    H, W = sl.shape
    for d in range(3):
        S = np.zeros((H + 1, W + 1), np.float32)
        S[1:, d + 1:] = np.abs(sl[:, d:] - sr[:, :W - d])
        S = np.cumsum(S, axis=0, dtype=np.float32)              # sequential float32 adds, columns then rows
        S[:, d:] = np.cumsum(S[:, d:], axis=1, dtype=np.float32)
        for i in range(H - 5):
            for j in range(d, W - 5):
                e = np.float32(np.float32(np.float32(S[i + 5, j + 5] - S[i + 5, j]) - S[i, j + 5]) + S[i, j])
                assert got[d, i + 2, j + 2] == e


def test_likelihood_properties():
    rng = np.random.default_rng(6)
    vol = (rng.random((50, 24), dtype=np.float32) * 100).astype(np.float32)
    vol[3] = SENT
    vol[7, 5:] = SENT
    out = O.extract_likelihood(vol, 128.0)
    assert not out[3].any()                                      # all-sentinel row -> zeros (featextract.cpp:452)
    assert np.all(out[7, 5:] == 0) and abs(out[7].sum() - 1) < 1e-5
    live = np.delete(np.arange(50), 3)
    assert np.abs(out[live].sum(1) - 1).max() < 1e-5
    assert np.all(out.argmax(1)[live] == vol.argmin(1)[live])    # the cheapest disparity is the most likely


def test_likelihood_small_known_answer():
    """Independent evaluation of extract_likelihood in float32 NumPy, operation by operation (featextract.cpp:415-462): m = min,
    den = sequential float32 sum of exp(-((c - m)^2) / sigma), out = exp(...) / den.  NumPy's float32 exp and glibc's expf are both
    faithfully rounded but not the same function: 2 ulp of the result is allowed, nothing more."""
    f = np.float32
    rng = np.random.default_rng(12)
    for sigma, scale in ((128.0, 120.0), (0.02, 2.0), (20000.0, 8192.0)):
        vol = (rng.random((40, 16), dtype=np.float32) * f(scale)).astype(np.float32)
        vol[rng.random(vol.shape) < 0.15] = SENT
        got = O.extract_likelihood(vol, sigma)
        for i in range(vol.shape[0]):
            m = vol[i].min()
            if m == SENT:
                assert not got[i].any()
                continue
            e = np.array([np.exp(f(-(f(f(c - m) * f(c - m)) / f(sigma)))) for c in vol[i]], dtype=np.float32)
            den = f(0)
            for x in e:
                den = f(den + x)
            exp_row = (e / den).astype(np.float32)
            assert np.abs(got[i] - exp_row).max() <= 2.5e-7, (sigma, i, float(np.abs(got[i] - exp_row).max()))
            assert np.all(got[i][vol[i] == SENT] == 0)


def test_swap_axes_and_volume_layout():
    rng = np.random.default_rng(7)
    c = rng.random((5, 6, 7), dtype=np.float32)
    assert np.array_equal(O.swap_axes(c), c.transpose(1, 2, 0))
    from msnets_amd import synthetic
    l, r, drows = synthetic.stereo_pair(32, 64, 16, seed=1, bands=1)
    vol = O.build_ms_volume(l, r, 16)
    assert vol.shape == (8, 16, 32, 64) and vol.dtype == np.float32
    assert vol.min() >= 0 and vol.max() <= 1
    # channel 2 is the Sobel-SAD cost, channel 3 ZSAD (get_costs returns census, ncc, sobel, sad)
    cs = O.get_costs(l, r, 16, 11, 3, 5, 5, 10, 10, 10)
    assert np.array_equal(vol[2], (np.clip(cs[2], 0, 8192) / np.float32(8192)).transpose(2, 0, 1))
    am = vol[0].argmin(0)
    assert (am[:, 24:] == drows[:, None]).mean() > 0.95          # planted disparity recovered by census


def test_sentinels_become_one_in_costs_and_zero_in_aml():
    """SURVEY H3."""
    from msnets_amd import synthetic
    l, r, _ = synthetic.stereo_pair(32, 48, 16, seed=2)
    vol = O.build_ms_volume(l, r, 16)
    # cropped column x has census costs only for d <= x + 5
    assert np.all(vol[0, 10:, :, 0] == 1.0) and np.all(vol[4, 10:, :, 0] == 0.0)
    assert np.all(vol[3, 12:, :, 2] == 1.0)          # zsad: d <= x + 8


def test_get_right_cost_known_answer():
    """featextract.cpp:136-172: res[i,j,d] = cost[i,j+d,d], the uncovered right margin takes cost[0,0,0]."""
    from oracle import ms_volume as O
    cost = np.arange(2 * 5 * 3, dtype=np.float32).reshape(2, 5, 3) + 100
    res = O.get_right_cost(cost)
    assert res[1, 2, 0] == cost[1, 2, 0] and res[1, 2, 1] == cost[1, 3, 1] and res[0, 2, 2] == cost[0, 4, 2]
    assert res[1, 4, 1] == cost[0, 0, 0] and res[0, 3, 2] == cost[0, 0, 0]
    feats = O.extract_features_lr(cost, cost / 200 - 1, cost, cost)
    assert feats.shape == (16, 3, 2, 5) and feats.dtype == np.float32


def test_three_operation_division_is_the_ieee_quotient():
    """volume_fused.hip divides by constants (120, sigma) and by the per-pixel likelihood sum with q = a*rb, r = fma(-q, b, a),
    q' = fma(r, rb, q), rb = RN(1/b) (Markstein).  Emulated here in float64 (products of two float32 are exact in float64;
    the residual a - q*b cancels to an exactly representable value): the result must equal the IEEE float32 quotient the
    reference computes -- exhaustively for the census costs 0..121 / 120 and on random operands for the other divisors."""
    f32 = np.float32

    def div_rn(a, b):
        a = a.astype(f32); b = f32(b)
        rb = f32(1.0) / b
        q = (a * rb).astype(f32)
        r = (a.astype(np.float64) - q.astype(np.float64) * np.float64(b)).astype(f32)      # fma(-q, b, a): exact, then one rounding
        return (q.astype(np.float64) + r.astype(np.float64) * np.float64(rb)).astype(f32)   # fma(r, rb, q)

    census = np.arange(0, 122, dtype=f32).clip(0, 120)
    assert np.array_equal(div_rn(census, 120.0), census / f32(120.0))
    rng = np.random.default_rng(0)
    for b in (0.02, 20000.0, 128.0):
        a = (rng.random(200000, dtype=f32) * f32(1e4)) ** 2
        a = np.concatenate([a, f32([0.0, 1e-30, 4.6e18, 1.0, 3.0])])
        assert np.array_equal(div_rn(a, b), a / f32(b)), b
    den = (rng.random(200000, dtype=f32) * 95 + 1).astype(f32)
    e = rng.random(200000, dtype=f32)
    rb = f32(1.0) / den
    q = (e * rb).astype(f32)
    r = (e.astype(np.float64) - q.astype(np.float64) * den.astype(np.float64)).astype(f32)
    got = (q.astype(np.float64) + r.astype(np.float64) * rb.astype(np.float64)).astype(f32)
    assert np.array_equal(got, e / den)


def _cpu_has(*flags):
    try:
        words = set(open("/proc/cpuinfo").read().split())
    except OSError:
        return False
    return all(f in words for f in flags)


@pytest.mark.skipif(not _cpu_has("avx2", "fma"), reason="the reference's -march=core-avx2 build needs AVX2 + FMA")
def test_oracle_is_independent_of_the_reference_compile_flags():
    """The one thing a line-by-line restatement can still get wrong is arithmetic that depends on how it was COMPILED.  The
    reference is built with -O3 -msse4.1 -march=core-avx2 -funroll-loops and g++'s default FMA contraction
    (src/cpp/CMakeLists.txt:11); the oracle with -O2 -ffp-contract=off.  Both builds of the restatement must return the same
    bits on every image kind of the GPU parity suite (tests/test_gpu_volume.py CASES, the full 292x500x96 pair included):
    raw costs of all four matchers, Sobel, and all eight volume channels.  (Found by this test: the raw NCC cost's exact
    zeros change SIGN with FMA contraction -- see below; nothing else moves.)  The oracle stays PARITY UNPINNED (no
    reference binary can be built here); this only removes compile-flag dependence from the list of ways it could differ."""
    from test_gpu_volume import CASES, _pair

    def everything(l, r, nd):
        sl, sr = O.sobel(l), O.sobel(r)
        raw = {"census": O.census(l, r, nd, 11), "ncc": O.nccNister(l, r, nd, 3), "zsad": O.zsad(l, r, nd, 5), "sobel_l": sl,
               "sobel_r": sr, "sadsob": O.sadsob(sl, sr, nd, 5)}
        if min(l.shape) > 20:
            vol = O.build_ms_volume(l, r, nd)
            for c in range(8):
                raw["volume_ch%d" % c] = vol[c]
        return raw

    for H, W, nd, seed, kind in CASES:
        l, r = _pair(H, W, nd, seed, kind)
        a = everything(l, r, nd)
        with O.variant("refflags"):
            b = everything(l, r, nd)
        for k in a:
            bad = a[k].view(np.uint32) != b[k].view(np.uint32)
            if k == "ncc":
                # The one flag-dependent expression of the path: matchers.cpp:200 negates (sqwin*lD - Al*Ar) before the two
                # multiplies.  Where that difference is exactly 0 (a window pair with zero covariance) plain evaluation
                # gives -0.0, an FMA-contracting build folds the negation into vfnmadd (Al*Ar - sqwin*lD) and gives +0.0.
                # Which one the reference's own binary returns depends on its compiler; the VALUE is 0 either way and
                # every consumer ((1 + clip)/2, (c - min)^2) maps both to the same bits -- the volume channels below are
                # compared bit for bit with no such exception.
                assert np.array_equal(a[k], b[k]) and not a[k][bad].any(), "ncc differs by more than the sign of a zero"
                continue
            assert not bad.any(), "%s %dx%d D'=%d: %s differs in %d values between the two oracle builds" % (
                kind, H, W, nd, k, int(bad.sum()))
