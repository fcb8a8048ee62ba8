"""GPU parity: denoising core (noisest = MAD/0.6745, threshold!, denoise / denoiseall for the VisuShrink family)
against the CPU oracle (Denoising.jl:214-232, 483-712; Wavelets.jl Threshold restated).  Order statistics and the
hard threshold are exact (==); reconstructed signals within 1e-10 / 1e-5."""
import numpy as np
import pytest

from helpers import TOL, relerr

pytestmark = pytest.mark.gpu

TH = {"hard": "HardTH", "soft": "SoftTH", "semisoft": "SemiSoftTH", "stein": "SteinTH"}


def _noisy(rng, n, B, dtype):
    t = np.linspace(0, 1, n)
    x0 = 4 * np.sin(4 * np.pi * t) - np.sign(t - 0.3) - np.sign(0.72 - t)          # heavisine
    return np.asfortranarray((x0[:, None] + 0.5 * rng.standard_normal((n, B))).astype(dtype)), x0


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_noisest_and_threshold_exact(wx, oracle, dtype):
    rng = np.random.default_rng(5001)
    for n in (2, 8, 64, 128, 256, 1024, 4096):
        v = np.asfortranarray(rng.standard_normal((n, 3)).astype(dtype))
        # noisest on a dwt-shaped signal: MAD of the upper half
        for i in range(3):
            assert wx.noisest(v[:, i], False) == pytest.approx(oracle.noisest(v[:, i], False), rel=0, abs=0)
        tree = wx.maketree(n, wx.maxtransformlevels(n), "dwt") if n > 2 else None
        if tree is not None:
            assert wx.noisest(v[:, 0], False, tree) == oracle.noisest(v[:, 0], False, tree)
        for name, cls in TH.items():
            t = 0.7
            got = wx.threshold(v, getattr(wx, cls)(), t)
            exp = oracle.threshold(v, name, t)
            if name == "hard":
                assert (got == exp).all()
            else:
                assert relerr(got, exp) <= (1e-15 if dtype == np.float64 else 1e-6), name
    red = np.asfortranarray(rng.standard_normal((64, 7)).astype(dtype))
    assert wx.noisest(red, True) == oracle.noisest(red, True)
    tree = wx.maketree(64, 2, "full")
    assert wx.noisest(red, True, tree) == oracle.noisest(red, True, tree)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("inputtype", ["sig", "dwt", "wpt", "sdwt", "swpd", "acdwt", "acwpd"])
def test_denoise_and_denoiseall(wx, oracle, inputtype, dtype):
    if inputtype in ("acdwt", "acwpd") and dtype == np.float32:
        pytest.skip("ACWT is Float64-only in the reference")
    rng = np.random.default_rng(5002)
    n, B = 256, 4
    wt = wx.wavelet(wx.WT.db4)
    x, x0 = _noisy(rng, n, B, dtype)
    tol = TOL[np.dtype(dtype)]
    L = 4
    tree = wx.maketree(n, L, "full") if inputtype in ("wpt", "swpd", "acwpd") else None
    fwd = {"sig": lambda a: a, "dwt": lambda a: wx.dwtall(a, wt, L), "wpt": lambda a: wx.wptall(a, wt, tree),
           "sdwt": lambda a: wx.sdwtall(a, wt, L), "swpd": lambda a: wx.swpdall(a, wt, L),
           "acdwt": lambda a: wx.acdwtall(a, wt, L), "acwpd": lambda a: wx.acwpdall(a, wt, L)}[inputtype]
    X = wx.to_numpy(fwd(x))
    for smooth in ("regular", "undersmooth"):
        for thname in ("hard", "soft"):
            dnt = wx.VisuShrink(n, getattr(wx, TH[thname])())
            kw = dict(L=L, dnt=dnt, smooth=smooth)
            if tree is not None:
                kw["tree"] = tree
            Y = wx.to_numpy(wx.denoiseall(X, inputtype, wt, **kw))
            assert Y.shape == (n, B)
            for i in range(B):
                Xi = np.asfortranarray(X[..., i])
                exp = oracle.denoise(Xi, inputtype, wt.qmf, L=L, tree=tree, th=thname, t=dnt.t, smooth=smooth)
                assert relerr(Y[:, i], exp) <= 20 * tol, (inputtype, smooth, thname, i)
                one = wx.to_numpy(wx.denoise(Xi, inputtype, wt, **kw))
                assert relerr(one, exp) <= 20 * tol
    # summary threshold (bestTH) and precomputed noise
    dnt = wx.VisuShrink(n)
    kw = dict(L=L, dnt=dnt)
    if tree is not None:
        kw["tree"] = tree
    Y = wx.to_numpy(wx.denoiseall(X, inputtype, wt, bestTH=np.mean, **kw))
    sig = []
    for i in range(B):
        Xi = np.asfortranarray(X[..., i])
        if inputtype == "sig":
            Xi = oracle.wpt(Xi, wt.qmf, oracle.maketree1d(n, L, "dwt"))
        red = inputtype in ("sdwt", "swpd", "acdwt", "acwpd")
        # (Denoising.jl:683-690: with a summary threshold :acdwt input takes the `else` branch, estnoise(x, true, tree) with the
        # default :dwt tree -- the finest detail node read as a column of a heap-ordered table)
        tr = tree if inputtype in ("wpt", "swpd", "acwpd") else (oracle.maketree1d(n, L, "dwt") if inputtype == "acdwt" else None)
        sig.append(oracle.noisest(Xi, red, tr))
    sbar = float(np.mean(np.asarray(sig, dtype=np.float64)))
    for i in range(B):
        exp = oracle.denoise(np.asfortranarray(X[..., i]), inputtype, wt.qmf, L=L, tree=tree, t=dnt.t, estnoise=sbar)
        assert relerr(Y[:, i], exp) <= 20 * tol
    if inputtype in ("sdwt", "acdwt", "sig"):
        # denoising helps on the redundant transforms / default pipeline (test/denoising.jl asserts this kind of bound)
        err0 = np.mean([np.linalg.norm(x[:, i] - x0) for i in range(B)])
        err1 = np.mean([np.linalg.norm(Y[:, i] - x0) for i in range(B)])
        assert err1 <= err0


def test_denoise_reference_test_expectations(wx):
    """the bounds test/denoising.jl:14-97 asserts on a noisy heavisine (n = 2^8, Haar, VisuShrink(2, HardTH())):
    denoising never makes the relative error worse than (twice) the noise level, for single signals and groups"""
    rng = np.random.default_rng(5003)
    n = 256
    t = np.linspace(0, 1, n)
    x0 = 4 * np.sin(4 * np.pi * t) - np.sign(t - 0.3) - np.sign(0.72 - t)
    x = x0 + 0.5 * rng.standard_normal(n)
    wt = wx.wavelet(wx.WT.haar)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    err = rel(x, x0)
    dnt = wx.VisuShrink(2, wx.HardTH())
    assert rel(wx.denoise(x, "sig", wt, dnt=dnt), x0) <= err
    assert rel(wx.denoise(wx.dwt(x, wt, 4), "dwt", wt, L=4, dnt=dnt, smooth="undersmooth"), x0) <= 2 * err
    assert rel(wx.denoise(wx.dwt(x, wt), "dwt", wt, dnt=dnt, smooth="undersmooth"), x0) <= 2 * err
    full = wx.maketree(n, 8, "full")
    assert rel(wx.denoise(wx.wpt(x, wt), "wpt", wt, tree=full, dnt=dnt, smooth="undersmooth"), x0) <= 2 * err
    assert rel(wx.denoise(wx.sdwt(x, wt), "sdwt", wt, dnt=dnt, smooth="undersmooth"), x0) <= 2 * err
    assert rel(wx.denoise(wx.swpd(x, wt), "swpd", wt, smooth="undersmooth"), x0) <= 2 * err
    assert rel(wx.denoise(wx.acdwt(x, wt), "acdwt", wt, dnt=dnt, smooth="undersmooth"), x0) <= 2 * err
    assert rel(wx.denoise(wx.acwpd(x, wt), "acwpd", wt, smooth="undersmooth"), x0) <= 2 * err
    # group denoising
    X0 = np.asfortranarray(np.stack([np.roll(x0, 2 * i) for i in range(5)], axis=1))
    X = np.asfortranarray(X0 + 0.5 * rng.standard_normal(X0.shape))
    max_err = max(rel(X[:, i], X0[:, i]) for i in range(5))
    mean_err = lambda Y: np.mean([rel(Y[:, i], X0[:, i]) for i in range(5)])
    assert mean_err(wx.denoiseall(X, "sig", wt, dnt=dnt, bestTH=np.mean)) <= max_err
    assert mean_err(wx.denoiseall(wx.dwtall(X, wt), "dwt", wt, dnt=dnt)) <= max_err
    assert mean_err(wx.denoiseall(wx.sdwtall(X, wt), "sdwt", wt)) <= max_err
    assert mean_err(wx.denoiseall(wx.acdwtall(X, wt), "acdwt", wt)) <= max_err
    with pytest.raises(AssertionError):
        wx.denoise(x, "nope", wt)
    with pytest.raises(AssertionError):
        wx.denoise(x, "sig", wt, smooth="oversmooth")


def test_noisest_degenerate_distributions(wx, oracle):
    """the bucketed selection behind noisest must stay exact when the values do not spread: constants, two values,
    heavy ties, one huge outlier (everything else in one bucket -> the narrowing loop), tiny and odd/even counts"""
    rng = np.random.default_rng(5004)
    cases = []
    for n in (4, 8, 64, 128, 256, 512, 1024, 2048, 4096):      # from 256 on: a wavefront sorts the n / 2 details in its registers (k_mad_sort)
        half = n // 2
        cases += [
            np.zeros(n), np.full(n, -3.5),
            np.where(np.arange(n) % 2 == 0, 1.0, -1.0),
            np.round(rng.standard_normal(n) * 2) / 2,                       # heavy ties
            np.concatenate([rng.standard_normal(n - 1) * 1e-6, [1e12]]),    # one outlier stretches the range
            np.concatenate([np.full(half, 1.0), 1.0 + np.arange(half) * 2.0 ** -40]),   # near-equal cluster
            rng.standard_normal(n) * 10.0 ** rng.integers(-200, 200),
        ]
    for v in cases:
        v = np.ascontiguousarray(rng.permutation(v))
        assert wx.noisest(v, False) == oracle.noisest(v, False), v[:4]
        with np.errstate(over="ignore"):
            v32 = v.astype(np.float32)
        if np.isfinite(v32).all():
            assert wx.noisest(v32, False) == oracle.noisest(v32, False)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_noisest_wave_sort_every_size_class_batches_and_nan(wx, oracle, dtype):
    """k_mad_sort (round 6): detail ranges of 65 ... 512 coefficients, sorted by one wavefront -- every size class with counts at, below
    and between the powers of two (the C ABI takes any row_lo: the slots beyond the count are padded), batches that do not fill the last
    workgroup, bit-equal to the oracle's selection; a NaN among the details gives NaN like Statistics.median, in the short
    kernel (<= 64 values) and in the sorting one"""
    import ctypes
    import sys
    lib = sys.modules[wx.__name__ + "._lib"]
    rng = np.random.default_rng(6006)
    suf = "_f64" if dtype == np.float64 else "_f32"
    fn = getattr(lib.lib(), "wx_noisest" + suf)
    for n, B in ((256, 9), (512, 6), (1024, 5), (2048, 3)):
        X = np.asfortranarray((rng.standard_normal((n, B)) * 10.0 ** rng.integers(-3, 3, B)).astype(dtype))
        X[:, 0] = np.round(X[:, 0] * 4) / 4                               # ties
        for row_lo in (n // 2, n // 2 - 1, n // 2 + 3, n // 4 + 1, n - 65, n - 70):
            if n - row_lo > 512 or n - row_lo <= 64:
                continue
            sig = np.empty(B, dtype=dtype)
            lib.check(fn(X.ctypes.data_as(ctypes.c_void_p), n, 1, B, row_lo, 0, sig.ctypes.data_as(ctypes.c_void_p), None))
            exp = np.array([oracle.noisest_range(X[row_lo:, b]) for b in range(B)], dtype=dtype)
            assert (sig == exp).all(), (n, row_lo, sig, exp)
    for n in (64, 128, 1024, 4096):
        v = rng.standard_normal(n).astype(dtype)
        v[n - 3] = np.nan
        assert np.isnan(wx.noisest(v, False)), n


@pytest.mark.parametrize("inputtype", ["sig", "dwt", "wpt"])
@pytest.mark.parametrize("smooth", ["regular", "undersmooth"])
def test_device_resident_pipeline_equals_the_host_staged_one(wx, oracle, inputtype, smooth):
    """device tensors take the pipeline whose noise estimates stay on the device and whose threshold rides on the loads of the
    inverse (wx_iwpt1d_thresh_*, Denoising.jl:510-533 in one pass): same numbers as the numpy path, which is checked against
    the oracle above; every threshold rule, a tree-driven and a full-tree inverse (the latter falls back to threshold + lattice)"""
    import torch
    rng = np.random.default_rng(77)
    wt = wx.wavelet(wx.WT.db4)
    for n, L, B in ((256, 4, 7), (4096, 6, 3)):
        x = np.asfortranarray(rng.standard_normal((n, B)) + np.sin(np.arange(n) / 9.0)[:, None] * 3)
        tree = wx.maketree(n, L, "full")
        xin = {"sig": x, "dwt": wx.dwtall(x, wt, L), "wpt": wx.wptall(x, wt, tree)}[inputtype]
        for th in (wx.HardTH(), wx.SoftTH(), wx.SemiSoftTH(), wx.SteinTH()):
            dnt = wx.VisuShrink(n, th)
            ref = wx.denoiseall(xin, inputtype, wt, L=L, tree=tree, dnt=dnt, smooth=smooth)
            xd = wx.to_colmajor(torch.from_numpy(np.ascontiguousarray(xin)).cuda())
            got = wx.denoiseall(xd, inputtype, wt, L=L, tree=tree, dnt=dnt, smooth=smooth)
            assert got.is_cuda
            assert relerr(got.cpu().numpy(), ref) <= 1e-13, (inputtype, smooth, type(th).__name__, n)
    # and the numpy path against the oracle at the larger size (tree-driven inverse with the fused threshold)
    x = np.asfortranarray(rng.standard_normal((4096, 2)))
    xw = wx.dwtall(x, wt, 5)
    y = wx.denoiseall(xw, "dwt", wt, L=5, dnt=wx.VisuShrink(4096, wx.SoftTH()), smooth=smooth)
    for i in range(2):
        exp = oracle.denoise(xw[:, i], "dwt", wt.qmf, L=5, th="soft", smooth=smooth)
        assert relerr(y[:, i], exp) <= 1e-10


@pytest.mark.parametrize("smooth", ["regular", "undersmooth"])
def test_denoise_dwt_full_depth_pyramid_tail(wx, oracle, smooth):
    """denoiseall(:dwt) at the default (maximum) depth: the thresholded pyramid's levels below 64 samples are rebuilt lane-locally
    (csrc/wx_dwttail.hip, threshold on its loads) and handed to the fused inverse; Denoising.jl:510-533 against the oracle"""
    rng = np.random.default_rng(99)
    wt = wx.wavelet(wx.WT.db4)
    for n, B in ((1024, 70), (4096, 3)):
        L = int(np.log2(n))
        x = np.asfortranarray(rng.standard_normal((n, B)) + np.sin(np.arange(n) / 11.0)[:, None] * 3)
        xw = wx.dwtall(x, wt)
        for thname, th in (("hard", wx.HardTH()), ("soft", wx.SoftTH())):
            y = wx.denoiseall(xw, "dwt", wt, dnt=wx.VisuShrink(n, th), smooth=smooth)
            for i in (0, B - 1):
                exp = oracle.denoise(xw[:, i], "dwt", wt.qmf, L=L, th=thname, smooth=smooth)
                assert relerr(y[:, i], exp) <= 1e-10, (n, thname, smooth, i)


@pytest.mark.parametrize("n", [64, 128, 256])
def test_noisest_of_a_batch_of_short_signals(wx, oracle, n):
    """denoiseall(:dwt) of many short signals: per-signal noise estimates from one wavefront per signal, four per workgroup (csrc/wx_denoise.hip
    k_mad_wave), a batch that is not a multiple of four, values with ties; the thresholded inverse through the masked lattice kernels"""
    rng = np.random.default_rng(n)
    B = 1031
    L = wx.maxtransformlevels(n)
    wt = wx.wavelet(wx.WT.db4)
    x = np.asfortranarray(np.round(rng.standard_normal((n, B)) * 8) / 8)
    X = wx.to_numpy(wx.dwtall(x, wt, L))
    dnt = wx.VisuShrink(n)
    Y = wx.to_numpy(wx.denoiseall(X, "dwt", wt, L=L, dnt=dnt))
    for b in range(0, B, 103):
        Xb = np.asfortranarray(X[:, b])
        assert wx.noisest(Xb, False) == oracle.noisest(Xb, False)
        exp = oracle.denoise(Xb, "dwt", wt.qmf, L=L, th="hard", t=dnt.t, smooth="regular")
        assert relerr(Y[:, b], exp) <= 1e-10, (n, b)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_noisest_counting_kernels_every_size(wx, oracle, dtype):
    """the counting noise estimate (csrc/wx_select_count.h): four signals per wavefront for 128 ... 512 details (k_mad_count_rows), one wavefront per
    signal for 1024 ... 4096 (k_mad_count), a workgroup of 4 / 8 wavefronts for 8192 ... 32768 (k_mad_count_wg) -- exact against the oracle's sort on ordinary, tied, sparse and wide-range values, NaN -> NaN"""
    rng = np.random.default_rng(20260)
    for n in (256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536):
        B = 5 if n > 1024 else 23                   # four signals per wavefront up to 1024 samples: a ragged last wavefront
        for kind in ("normal", "ties", "sparse", "range"):
            v = rng.standard_normal((n, B))
            if kind == "ties":
                v = np.round(v * 5) / 5
            elif kind == "sparse":
                v = np.where(rng.random((n, B)) < 0.03, v, 0.0)
            elif kind == "range":
                v = v * np.exp(rng.standard_normal((n, B)) * 12)
            v = np.asfortranarray(v.astype(dtype))
            from waveletsext_jl_amd import denoising as dn
            sig = wx.to_numpy(dn._noisest(dn.Arg(v), True, "dwt", None))
            for i in range(B):
                assert sig[i] == oracle.noisest(v[:, i], False), (n, kind, i, dtype)
        v = np.asfortranarray(rng.standard_normal((n, 3)).astype(dtype))
        v[n - 5, 1] = np.nan
        sig = wx.to_numpy(dn._noisest(dn.Arg(v), True, "dwt", None))
        assert np.isnan(sig[1]) and sig[0] == oracle.noisest(v[:, 0], False) and sig[2] == oracle.noisest(v[:, 2], False)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_denoise_long_signals(wx, oracle, dtype):
    """denoiseall(:sig / :dwt) of 8192 ... 65536-sample signals (the counting noise estimate with a workgroup per signal, the thresholded copy, the
    tiled top passes of the long inverse over the lattice): every rule, both smoothings, shallow and full depth, against the oracle"""
    rng = np.random.default_rng(424242)
    wt = wx.wavelet(wx.WT.db4)
    tol = 1e-10 if dtype == np.float64 else 3e-4
    for n, B in ((8192, 5), (16384, 4), (32768, 3), (65536, 2)):
        x = np.asfortranarray((rng.standard_normal((n, B)) + 3 * np.sin(np.arange(n) / 50.0)[:, None]).astype(dtype))
        Lmax = wx.maxtransformlevels(n)
        for L in (Lmax, 6, 3):
            xw = np.asfortranarray(wx.to_numpy(wx.dwtall(x, wt, L)))
            for smooth in ("regular", "undersmooth"):
                for thname in (("hard", "soft", "semisoft", "stein") if L == Lmax else ("hard",)):
                    dnt = wx.VisuShrink(n, getattr(wx, TH[thname])())
                    Y = wx.to_numpy(wx.denoiseall(xw, "dwt", wt, L=L, dnt=dnt, smooth=smooth))
                    for i in (0, B - 1):
                        exp = oracle.denoise(np.asfortranarray(xw[:, i].astype(np.float64)), "dwt", wt.qmf, L=L, th=thname, t=dnt.t, smooth=smooth)
                        assert relerr(Y[:, i], exp) <= tol, (n, L, smooth, thname, i, dtype)
        Y = wx.to_numpy(wx.denoiseall(x, "sig", wt, dnt=wx.VisuShrink(n)))
        exp = oracle.denoise(x[:, 0].astype(np.float64), "sig", wt.qmf, th="hard", t=wx.VisuShrink(n).t)
        assert relerr(Y[:, 0], exp) <= tol
