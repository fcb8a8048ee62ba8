"""GPU parity: denoiseall(x, :sig, wt; L, dnt, smooth) behind ONE entry point (wx_denoiseall_sig_*, Denoising.jl:651-712): Float64 signals of
1024 / 2048 / 4096 samples take one pass -- pyramid analysis, the two exact medians of the noise estimate by counting, threshold, synthesis in the
registers (csrc/wx_lattice_dn.h); everything else runs dwtall -> noisest -> threshold on the loads of idwtall inside the library.  Both against
the CPU oracle (oracle.denoise: dwt -> mad / 0.6745 -> threshold -> idwt): reconstructed signals within 1e-10 (Float32 1e-4), noise estimates
within 1e-12 (the lattice factorisation rounds differently from the direct form; the order statistics themselves are exact)."""
import ctypes

import numpy as np
import pytest

from helpers import relerr

pytestmark = pytest.mark.gpu

TH = {"hard": "HardTH", "soft": "SoftTH", "semisoft": "SemiSoftTH", "stein": "SteinTH"}


def _signals(rng, n, B, kind):
    t = np.linspace(0, 1, n)
    base = 4 * np.sin(4 * np.pi * t) - np.sign(t - 0.3) - np.sign(0.72 - t)
    if kind == "noisy":
        x = base[:, None] + 0.5 * rng.standard_normal((n, B))
    elif kind == "sparse":               # long exactly-constant stretches: with Haar most finest details are exactly 0 (ties at the median)
        x = np.repeat(np.round(rng.standard_normal((n // 64, B)) * 2), 64, axis=0)
        x[::97] += rng.standard_normal((len(x[::97]), B))
    elif kind == "quantised":
        x = np.round((base[:, None] + rng.standard_normal((n, B))) * 4) / 4
    elif kind == "range":                # 17 decades of dynamic range
        x = np.exp(rng.standard_normal((n, B)) * 20) * np.sign(rng.standard_normal((n, B)))
    elif kind == "cauchy":
        x = rng.standard_cauchy((n, B))
    else:
        raise ValueError(kind)
    return np.asfortranarray(x)


def _sigma_c(wx, x, wt, L, th_kind=0, t=1.0, undersmooth=0):
    """the C entry with the optional sigma output (host arrays go through the library's staging)"""
    from waveletsext_jl_amd import _lib
    n, B = x.shape
    q = np.ascontiguousarray(np.asarray(wt.qmf, dtype=np.float64))
    y = np.empty_like(x, order="F")
    sig = np.empty(B, dtype=x.dtype)
    fn = getattr(_lib.lib(), "wx_denoiseall_sig_f64" if x.dtype == np.float64 else "wx_denoiseall_sig_f32")
    _lib.check(fn(ctypes.c_void_p(x.ctypes.data), ctypes.c_void_p(y.ctypes.data), n, L, B, ctypes.c_void_p(q.ctypes.data), len(q), th_kind, float(t),
                  undersmooth, ctypes.c_void_p(sig.ctypes.data), None))
    return y, sig


@pytest.mark.parametrize("n", [4096, 2048, 1024, 512, 256, 128, 64])
@pytest.mark.parametrize("wname", ["haar", "db2", "db4", "db6", "db8", "db10"])
def test_onepass_against_the_oracle(wx, oracle, n, wname):
    rng = np.random.default_rng(n + len(wname))
    wt = wx.wavelet(getattr(wx.WT, wname))
    B = 37 if n >= 1024 else 3 * (4096 // n) + 5           # not a multiple of the signals per wavefront: the tail wavefront re-does signals
    x = _signals(rng, n, B, "noisy")
    Lmax = wx.maxtransformlevels(n)
    for L in (Lmax, min(5, Lmax - 1), 1):
        for smooth in ("regular", "undersmooth"):
            for thname in ("hard", "soft", "semisoft", "stein"):
                if (L, thname) not in ((Lmax, "hard"), (Lmax, "soft"), (min(5, Lmax - 1), "semisoft"), (min(5, Lmax - 1), "hard"), (1, "soft"), (Lmax, "stein")):
                    continue
                dnt = wx.VisuShrink(n, getattr(wx, TH[thname])())
                Y = wx.to_numpy(wx.denoiseall(x, "sig", wt, L=L, dnt=dnt, smooth=smooth))
                assert Y.shape == (n, B)
                for i in sorted(set(range(0, B, max(B // 9, 1))) | {1, 2, 3, B - 2, B - 1}):
                    exp = oracle.denoise(x[:, i], "sig", wt.qmf, L=L, th=thname, t=dnt.t, smooth=smooth)
                    assert relerr(Y[:, i], exp) <= 1e-10, (n, wname, L, smooth, thname, i)


@pytest.mark.parametrize("n", [4096, 2048, 1024, 512, 256, 128, 64])
def test_onepass_noise_estimates(wx, oracle, n):
    """the sigma output of the entry point against oracle.noisest(dwt(x)) for ordinary, tied, heavy-tailed and wide-range data"""
    wt = wx.wavelet(wx.WT.db4)
    haar = wx.wavelet(wx.WT.haar)
    L = wx.maxtransformlevels(n)
    tree = np.asarray(wx.maketree(n, L, "dwt"), dtype=bool)
    for kind, w in (("noisy", wt), ("sparse", haar), ("quantised", haar), ("quantised", wt), ("range", wt), ("cauchy", wt)):
        rng = np.random.default_rng(hash((n, kind)) & 0xffff)
        B = 21 if n >= 1024 else 2 * (4096 // n) + 3
        x = _signals(rng, n, B, kind)
        y, sig = _sigma_c(wx, x, w, L)
        for i in range(B):
            xw = oracle.wpt(x[:, i], w.qmf, tree)
            exp = oracle.noisest(xw, False)
            assert abs(sig[i] - exp) <= 1e-12 * max(abs(exp), 1e-300) + (1e-13 * np.abs(x[:, i]).max() if kind in ("sparse", "quantised") else 0), (n, kind, i, sig[i], exp)
        # the denoised signals with these estimates (hard threshold at sigma * 1.0): every data kind but the wide-range one, where one ulp of a
        # huge coefficient exceeds the small ones
        if kind != "range":
            for i in (0, B - 1):
                exp = oracle.denoise(x[:, i], "sig", w.qmf, L=L, th="hard", t=1.0, smooth="regular")
                assert relerr(y[:, i], exp) <= 1e-9, (n, kind, i)


def test_onepass_constant_nan_and_small_batches(wx, oracle):
    wt = wx.wavelet(wx.WT.db4)
    for n in (4096, 2048, 1024, 512, 64):
        L = wx.maxtransformlevels(n)
        rng = np.random.default_rng(n)
        x = _signals(rng, n, 9 if n >= 1024 else 4096 // n + 9, "noisy")
        x[:, 2] = 3.25                       # constant: every detail 0 up to rounding, sigma ~ 0, the signal comes back
        x[:, 5] = 0.0                        # exactly zero everywhere
        y, sig = _sigma_c(wx, x, wt, L)
        assert sig[5] == 0.0 and np.abs(y[:, 5]).max() == 0.0
        assert abs(sig[2]) <= 1e-12 and relerr(y[:, 2], x[:, 2]) <= 1e-10
        for i in (0, x.shape[1] - 1):
            assert relerr(y[:, i], oracle.denoise(x[:, i], "sig", wt.qmf, L=L, th="hard", t=1.0)) <= 1e-10
        # a NaN sample: its signal's estimate is NaN (Statistics.median), the neighbours are untouched
        xn = x.copy(order="F")
        xn[n // 3, 4] = np.nan
        y2, sig2 = _sigma_c(wx, xn, wt, L)
        assert np.isnan(sig2[4])
        for i in (3, 5, x.shape[1] - 1):
            assert sig2[i] == sig[i] and (y2[:, i] == y[:, i]).all()
        # batches below the signals per wavefront (the separate kernels take them) and a single signal through denoise()
        for B in (1, 2, 3):
            yb, sb = _sigma_c(wx, np.asfortranarray(x[:, :B]), wt, L)
            for i in range(B):
                assert relerr(yb[:, i], y[:, i]) <= 1e-10 and abs(sb[i] - sig[i]) <= 1e-12 * abs(sig[i]) + 1e-300
        one = wx.to_numpy(wx.denoise(x[:, 0], "sig", wt, dnt=wx.VisuShrink(wx.HardTH(), 1.0)))
        assert relerr(one, y[:, 0]) <= 1e-10


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_entry_point_on_every_other_length_and_type(wx, oracle, dtype):
    """lengths and types the one-pass kernel does not take: the same entry point runs the separate steps"""
    wt = wx.wavelet(wx.WT.db4)
    tol = 1e-10 if dtype == np.float64 else 2e-4
    for n in (8, 32, 8192, 16384) + ((64, 256, 1024, 4096) if dtype == np.float32 else ()):
        rng = np.random.default_rng(n)
        x = np.asfortranarray(_signals(rng, n, 5, "noisy").astype(dtype))
        for L in (wx.maxtransformlevels(n), 2, 0):
            for smooth in ("regular", "undersmooth"):
                dnt = wx.VisuShrink(n, wx.SoftTH())
                Y = wx.to_numpy(wx.denoiseall(x, "sig", wt, L=L, dnt=dnt, smooth=smooth))
                for i in (0, 4):
                    exp = oracle.denoise(x[:, i].astype(np.float64), "sig", wt.qmf, L=L, th="soft", t=dnt.t, smooth=smooth)
                    assert relerr(Y[:, i], exp) <= tol, (n, L, smooth, i)


def test_device_arrays_and_errors(wx, oracle):
    import torch
    wt = wx.wavelet(wx.WT.db4)
    rng = np.random.default_rng(12)
    x = _signals(rng, 2048, 130, "noisy")
    ref = wx.to_numpy(wx.denoiseall(x, "sig", wt))
    xd = wx.to_colmajor(torch.from_numpy(np.ascontiguousarray(x)).cuda())
    got = wx.denoiseall(xd, "sig", wt)
    assert got.is_cuda and (got.cpu().numpy() == ref).all()
    from waveletsext_jl_amd import _lib
    with pytest.raises((AssertionError, _lib.WxError)):
        wx.denoiseall(np.asfortranarray(rng.standard_normal((100, 3))), "sig", wt)          # not dyadic (the mirror's @assert)
    with pytest.raises((AssertionError, _lib.WxError)):
        wx.denoiseall(x, "sig", wt, L=12)                                                    # deeper than maxtransformlevels(2048) = 11
    # and the C entry's own checks
    q = np.ascontiguousarray(np.asarray(wt.qmf, dtype=np.float64))
    fn = _lib.lib().wx_denoiseall_sig_f64
    bad = np.asfortranarray(rng.standard_normal((100, 3)))
    out = np.empty_like(bad, order="F")
    assert fn(ctypes.c_void_p(bad.ctypes.data), ctypes.c_void_p(out.ctypes.data), 100, 2, 3, ctypes.c_void_p(q.ctypes.data), len(q), 0, 1.0, 0, None, None) == _lib.WX_EASSERT
    out = np.empty_like(x, order="F")
    assert fn(ctypes.c_void_p(x.ctypes.data), ctypes.c_void_p(out.ctypes.data), 2048, 12, 130, ctypes.c_void_p(q.ctypes.data), len(q), 0, 1.0, 0, None, None) == _lib.WX_EASSERT
    assert fn(ctypes.c_void_p(x.ctypes.data), ctypes.c_void_p(out.ctypes.data), 2048, 3, 130, ctypes.c_void_p(q.ctypes.data), len(q), 7, 1.0, 0, None, None) == _lib.WX_EARG


@pytest.mark.parametrize("n", [4096, 2048, 1024, 512, 256, 128, 64])
@pytest.mark.parametrize("wname", ["haar", "db4", "db8"])
def test_coefficients_in_signals_out(wx, oracle, n, wname):
    """denoiseall(xw, :dwt, wt; L, dnt, smooth) behind wx_denoiseall_dwt_*: the coefficients of the pyramid in, the denoised signals out (one pass:
    the front end of the tree-driven inverse, the noise estimate in the last register layout, threshold, synthesis), against oracle.denoise(:dwt)"""
    rng = np.random.default_rng(7 * n + len(wname))
    wt = wx.wavelet(getattr(wx.WT, wname))
    B = 37 if n >= 1024 else 3 * (4096 // n) + 5
    x = _signals(rng, n, B, "noisy")
    Lmax = wx.maxtransformlevels(n)
    for L in (Lmax, min(5, Lmax - 1), 1):
        xw = wx.to_numpy(wx.dwtall(x, wt, L))
        for smooth in ("regular", "undersmooth"):
            for thname in ("hard", "soft", "semisoft", "stein"):
                if (L, thname) not in ((Lmax, "hard"), (Lmax, "soft"), (min(5, Lmax - 1), "semisoft"), (1, "hard"), (Lmax, "stein")):
                    continue
                dnt = wx.VisuShrink(n, getattr(wx, TH[thname])())
                Y = wx.to_numpy(wx.denoiseall(xw, "dwt", wt, L=L, dnt=dnt, smooth=smooth))
                for i in sorted(set(range(0, B, max(B // 9, 1))) | {1, B - 1}):
                    exp = oracle.denoise(np.asfortranarray(xw[:, i]), "dwt", wt.qmf, L=L, th=thname, t=dnt.t, smooth=smooth)
                    assert relerr(Y[:, i], exp) <= 1e-10, (n, wname, L, smooth, thname, i)
    # the noise estimates are those of noisest on the same coefficients, exactly (no transform in between)
    from waveletsext_jl_amd import _lib
    xw = np.asfortranarray(wx.to_numpy(wx.dwtall(x, wt, Lmax)))
    q = np.ascontiguousarray(np.asarray(wt.qmf, dtype=np.float64))
    y = np.empty_like(xw, order="F")
    sig = np.empty(B)
    _lib.check(_lib.lib().wx_denoiseall_dwt_f64(ctypes.c_void_p(xw.ctypes.data), ctypes.c_void_p(y.ctypes.data), n, Lmax, B, ctypes.c_void_p(q.ctypes.data),
                                                  len(q), 0, 1.0, 0, ctypes.c_void_p(sig.ctypes.data), None))
    for i in range(B):
        assert sig[i] == oracle.noisest(xw[:, i], False), (n, wname, i)
