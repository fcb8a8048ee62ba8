"""Every even filter length up to 20 taps takes the fused / register kernels (round 3 added 14 taps -- db7 -- to the template
lists of wx_dwt1d.hip, wx_dwt2d.hip, wx_dwttail.hip, wx_swt1d.hip): the same calls for db6 (12), db7 (14), db8 (16) against the
oracle, so a length that falls back to another kernel family still has to give the reference's numbers.

Reference: the transforms take any OrthoFilter (dwt/dwt_one_level.jl:79-107, swt/swt_one_level.jl:99-127, DWT.jl:500-548).
"""
import numpy as np
import pytest

from helpers import relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wname", ["db6", "db7", "db8"])
def test_1d_decimated_paths(wx, oracle, wname):
    rng = np.random.default_rng(14)
    wt = wx.wavelet(getattr(wx.WT, wname))
    for n, B, dt, tol in ((512, 70, np.float64, 1e-10), (4096, 9, np.float32, 2e-5), (8192, 5, np.float64, 1e-10)):
        x = np.asfortranarray(rng.standard_normal((n, B)).astype(dt))
        L = wx.maxtransformlevels(n)
        for tree in (None, np.asarray(wx.maketree(n, L, "dwt")), np.asarray(wx.maketree(n, 3, "full"))):
            arg = L if tree is None else tree
            exp = oracle.wptall(x.astype(np.float64), wt.qmf, arg)
            got = wx.wptall(x, wt, arg)
            assert relerr(np.asarray(got, dtype=np.float64), exp) <= tol, (wname, n, dt)
            back = wx.iwptall(exp.astype(dt), wt, arg)
            assert relerr(np.asarray(back, dtype=np.float64), x.astype(np.float64)) <= tol, (wname, n, dt)
        xw = wx.wpdall(x, wt, 4)
        assert relerr(np.asarray(xw, dtype=np.float64), oracle.wpdall(x.astype(np.float64), wt.qmf, 4)) <= tol, (wname, n)


@pytest.mark.parametrize("wname", ["db6", "db7", "db8"])
def test_1d_redundant_paths(wx, oracle, wname):
    rng = np.random.default_rng(15)
    wt = wx.wavelet(getattr(wx.WT, wname))
    for n, B, L in ((1024, 6, 6), (4096, 3, 9)):
        x = np.asfortranarray(rng.standard_normal((n, B)))
        stack = lambda f, *a: np.stack([f(np.asfortranarray(x[:, b]), *a) for b in range(B)], axis=-1)
        y = wx.sdwtall(x, wt, L)
        assert relerr(y, stack(oracle.sdwt, wt.qmf, L)) <= 1e-10, (wname, n)
        assert relerr(wx.isdwtall(y, wt), x) <= 1e-10, (wname, n)
        a = wx.acdwtall(x, wt, L)
        assert relerr(a, stack(oracle.acdwt, wt.qmf, L)) <= 1e-10, (wname, n)
        p = wx.swptall(x, wt, 4)
        assert relerr(p, stack(oracle.swpt, wt.qmf, 4)) <= 1e-10, (wname, n)
        assert relerr(wx.iswptall(p, wt), x) <= 1e-10, (wname, n)


@pytest.mark.parametrize("wname", ["db6", "db7", "coif6"])
def test_2d_paths(wx, oracle, wname):
    rng = np.random.default_rng(16)
    wt = wx.wavelet(getattr(wx.WT, wname))
    for (m, n), B, L, dt, tol in (((64, 64), 5, 3, np.float64, 1e-10), ((128, 64), 3, 2, np.float32, 2e-5), ((256, 256), 2, 4, np.float32, 2e-5)):
        x = np.asfortranarray(rng.standard_normal((m, n, B)).astype(dt))
        exp = oracle.wptall(x.astype(np.float64), wt.qmf, L)
        got = wx.wptall(x, wt, L)
        assert relerr(np.asarray(got, dtype=np.float64), exp) <= tol, (wname, m, n)
        back = wx.iwptall(exp.astype(dt), wt, L)
        assert relerr(np.asarray(back, dtype=np.float64), x.astype(np.float64)) <= tol, (wname, m, n)
        tree = np.asarray(wx.maketree(m, n, L, "dwt"))
        expd = oracle.wptall(x.astype(np.float64), wt.qmf, tree)
        assert relerr(np.asarray(wx.wptall(x, wt, tree), dtype=np.float64), expd) <= tol, (wname, m, n, "dwt")
        assert relerr(np.asarray(wx.iwptall(expd.astype(dt), wt, tree), dtype=np.float64), x.astype(np.float64)) <= tol
