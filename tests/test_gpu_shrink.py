"""GPU parity of the SureShrink / RelErrorShrink threshold selection (csrc/wx_shrink.hip) against the numpy restatement of
Denoising.jl:146-166, 285-381 in the oracle, and of denoiseall with estnoise = relerrorthreshold (test/denoising.jl:59-83).
The selected threshold is one of the coefficient magnitudes, so the comparison is exact up to the final x/xmax*xmax
rounding; a pick that differs from the oracle's must tie with it to within the rounding of the cumulative sums."""
import numpy as np
import pytest

from helpers import relerr

pytestmark = pytest.mark.gpu


def _decay(rng, n, B, dtype=np.float64):
    """signals whose coefficient magnitudes decay: a relative error curve with a visible elbow"""
    t = np.arange(n)[:, None]
    x = rng.standard_normal((n, B)) * np.exp(-t / (n / 8.0)) + 0.05 * rng.standard_normal((n, B))
    return np.asfortranarray(x.astype(dtype))


@pytest.mark.parametrize("n", [8, 64, 256, 1000, 4096, 8192])
def test_surethreshold_and_relerrorthreshold_one_signal(wx, oracle, n):
    rng = np.random.default_rng(n)
    x = _decay(rng, n, 1)[:, 0].copy()
    assert wx.surethreshold(x, False) == pytest.approx(oracle.surethreshold(x, False), rel=1e-14)
    for elbows in (1, 2, 3):
        assert wx.relerrorthreshold(x, False, None, elbows) == pytest.approx(oracle.relerrorthreshold(x, False, None, elbows), rel=1e-14)


def test_threshold_selection_batched_matches_per_signal(wx, oracle):
    rng = np.random.default_rng(5)
    x = _decay(rng, 512, 33)
    s = wx.surethresholdall(x, False)
    r = wx.relerrorthresholdall(x, False)
    for i in range(x.shape[1]):
        assert s[i] == pytest.approx(oracle.surethreshold(x[:, i], False), rel=1e-14)
        assert r[i] == pytest.approx(oracle.relerrorthreshold(x[:, i], False), rel=1e-14)


def test_threshold_selection_redundant_tables(wx, oracle):
    """sdwt table (all columns), swpd table with a tree (leaf columns only), above the LDS window (global scratch)"""
    rng = np.random.default_rng(6)
    wt = wx.wavelet(wx.WT.db4)
    x = _decay(rng, 256, 3)
    xw = wx.sdwtall(x, wt, 4)                                   # (256, 5, 3)
    for i in range(3):
        assert wx.surethreshold(xw[:, :, i], True) == pytest.approx(oracle.surethreshold(xw[:, :, i], True), rel=1e-14)
        assert wx.relerrorthreshold(xw[:, :, i], True) == pytest.approx(oracle.relerrorthreshold(xw[:, :, i], True), rel=1e-14)
    tree = wx.maketree(256, 4, "dwt")
    xp = wx.swpdall(x, wt)                                      # (256, 511, 3): getleaf(tree) indexes the full-depth table
    got = wx.relerrorthresholdall(xp, True, tree)
    for i in range(3):
        assert got[i] == pytest.approx(oracle.relerrorthreshold(xp[:, :, i], True, tree), rel=1e-14)
        assert wx.surethreshold(xp[:, :, i], True, tree) == pytest.approx(oracle.surethreshold(xp[:, :, i], True, tree), rel=1e-14)
    big = _decay(rng, 4096, 2)
    xb = wx.sdwtall(big, wt, 5)                                 # 6 x 4096 = 24576 coefficients: global scratch window
    got = wx.relerrorthresholdall(xb, True)
    for i in range(2):
        assert got[i] == pytest.approx(oracle.relerrorthreshold(xb[:, :, i], True), rel=1e-13)


def test_threshold_selection_float32(wx, oracle):
    rng = np.random.default_rng(7)
    x = _decay(rng, 1024, 4, np.float32)
    s = wx.surethresholdall(x, False)
    r = wx.relerrorthresholdall(x, False)
    assert s.dtype == np.float32 and r.dtype == np.float32
    mags = np.sort(np.abs(x), axis=0)
    for i in range(4):
        # Float32 sums: the pick may move to a neighbouring magnitude; it must be one of the magnitudes and close in rank
        for got, ref in ((s[i], oracle.surethreshold(x[:, i].astype(np.float64), False)),
                         (r[i], oracle.relerrorthreshold(x[:, i].astype(np.float64), False))):
            k_got = int(np.argmin(np.abs(mags[:, i] - got)))
            k_ref = int(np.argmin(np.abs(mags[:, i].astype(np.float64) - ref)))
            assert abs(float(mags[k_got, i]) - float(got)) <= 1e-6 * float(mags[-1, i])
            assert abs(k_got - k_ref) <= 8, (k_got, k_ref)


def test_sureshrink_constructors_and_denoise(wx, oracle):
    """test/denoising.jl:3-11: the constructors; SureShrink(xw) carries surethreshold(xw)"""
    rng = np.random.default_rng(8)
    wt = wx.wavelet(wx.WT.db4)
    x = _decay(rng, 256, 1)[:, 0].copy()
    xw = wx.dwt(x, wt, 4)
    assert wx.RelErrorShrink().t == 1.0 and isinstance(wx.RelErrorShrink(wx.SoftTH()).th, wx.SoftTH)
    assert wx.SureShrink(wx.HardTH(), 1).t == 1.0
    d = wx.SureShrink(xw)
    assert d.t == pytest.approx(oracle.surethreshold(xw, False), rel=1e-14) and isinstance(d.th, wx.HardTH)
    tree = wx.maketree(256, 4, "full")
    assert wx.SureShrink(wx.wpt(x, wt, tree), False, tree, wx.SoftTH()).t > 0
    y = wx.denoise(xw, "dwt", wt, L=4, dnt=d)
    exp = oracle.denoise(xw, "dwt", wt.qmf, L=4, th="hard", t=d.t)
    assert relerr(y, exp) <= 1e-12


@pytest.mark.parametrize("inputtype", ["wpt", "dwt", "sdwt", "swpd", "acdwt", "acwpd"])
def test_denoiseall_with_relerrorthreshold_as_estnoise(wx, oracle, inputtype):
    """test/denoising.jl:59-83: dnt = RelErrorShrink(HardTH(), 0.3), estnoise = relerrorthreshold, with and without bestTH"""
    rng = np.random.default_rng(9)
    wt = wx.wavelet(wx.WT.db4)
    n, L, B = 128, 4, 5
    x = _decay(rng, n, B)
    tree = wx.maketree(n, L, "full" if inputtype in ("wpt", "swpd", "acwpd") else "dwt")
    fwd = {"wpt": lambda: wx.wptall(x, wt, tree), "dwt": lambda: wx.dwtall(x, wt, L), "sdwt": lambda: wx.sdwtall(x, wt, L),
           "swpd": lambda: wx.swpdall(x, wt), "acdwt": lambda: wx.acdwtall(x, wt, L), "acwpd": lambda: wx.acwpdall(x, wt)}
    xw = fwd[inputtype]()
    dnt = wx.RelErrorShrink(wx.HardTH(), 0.3)
    red = inputtype in ("sdwt", "swpd", "acdwt", "acwpd")
    tr = None if inputtype in ("dwt", "sdwt", "acdwt") else tree
    y = wx.denoiseall(xw, inputtype, wt, L=L, tree=tree, dnt=dnt, estnoise=wx.relerrorthreshold)
    if inputtype == "acdwt":
        # Denoising.jl:683-690 sends :acdwt input with a summary threshold through estnoise(x, true, tree), and
        # relerrorthreshold then indexes the (n, L+1) matrix with the leaves of a heap-ordered tree (Denoising.jl:298-299):
        # a BoundsError in the reference, an assertion here
        with pytest.raises(AssertionError):
            wx.denoiseall(xw, inputtype, wt, L=L, tree=tree, dnt=dnt, estnoise=wx.relerrorthreshold, bestTH=np.mean)
        y2 = None
    else:
        y2 = wx.denoiseall(xw, inputtype, wt, L=L, tree=tree, dnt=dnt, estnoise=wx.relerrorthreshold, bestTH=np.mean)
    sig = [oracle.relerrorthreshold(xw[..., i], red, tr) for i in range(B)]
    for i in range(B):
        exp = oracle.denoise(xw[..., i], inputtype, wt.qmf, L=L, tree=tree, th="hard", t=0.3, estnoise=sig[i])
        assert relerr(y[:, i], exp) <= 1e-10, (inputtype, i)
        if y2 is not None:
            exp2 = oracle.denoise(xw[..., i], inputtype, wt.qmf, L=L, tree=tree, th="hard", t=0.3, estnoise=float(np.mean(sig)))
            assert relerr(y2[:, i], exp2) <= 1e-10, (inputtype, i)
