"""The multi-level top pass of long signals (csrc/wx_toptile.h) against the oracle: full trees, pyramids and random trees of
8192 ... 65536 samples (one pass of 1 ... 4 levels) and of 2^17 / 2^18 samples (two passes), every filter length the pass is
instantiated for, Float64 and Float32, forward against the oracle's wpt and inverse against the signal
(dwt/dwt_one_level.jl:79-107, 192-223; Wavelets.jl wpt / iwpt as called by dwt/dwt_all.jl:152-225)."""
import numpy as np
import pytest

from helpers import random_tree_1d, relerr

pytestmark = pytest.mark.gpu

FILTERS = ["haar", "db2", "db3", "db4", "db5", "coif2", "db7", "db8", "coif6", "db10"]      # 2 ... 20 taps


def _wt(wx, name):
    return wx.wavelet(getattr(wx.WT, name))


@pytest.mark.parametrize("n", [8192, 16384, 32768, 65536])
@pytest.mark.parametrize("wname", ["db4", "db8"])
def test_full_trees_of_long_signals(wx, oracle, n, wname):
    wt = _wt(wx, wname)
    rng = np.random.default_rng(n + len(wname))
    x = np.asfortranarray(rng.standard_normal((n, 3)))
    L = wx.maxtransformlevels(n)
    for Lv in (L, L - 5, 3, 1):
        y = wx.wptall(x, wt, Lv)
        assert relerr(y, oracle.wptall(x, wt.qmf, Lv)) <= 1e-10, (n, Lv)
        assert relerr(wx.iwptall(y, wt, Lv), x) <= 1e-10, (n, Lv)


@pytest.mark.parametrize("wname", FILTERS)
def test_every_filter_length(wx, oracle, wname):
    wt = _wt(wx, wname)
    rng = np.random.default_rng(len(wt.qmf))
    n = 32768
    x = np.asfortranarray(rng.standard_normal((n, 2)))
    for fn, ifn, ofn in ((wx.wptall, wx.iwptall, oracle.wptall),):
        y = fn(x, wt, 15)
        assert relerr(y, ofn(x, wt.qmf, 15)) <= 1e-10
        assert relerr(ifn(y, wt, 15), x) <= 1e-10
    yd = wx.dwtall(x, wt)
    assert relerr(yd, oracle.wptall(x, wt.qmf, wx.maketree(n, 15, "dwt"))) <= 1e-10
    assert relerr(wx.idwtall(yd, wt), x) <= 1e-10
    x32 = np.asfortranarray(x.astype(np.float32))
    y32 = wx.wptall(x32, wt, 15)
    assert relerr(y32, oracle.wptall(x32, wt.qmf, 15)) <= 1e-5
    assert relerr(wx.iwptall(y32, wt, 15), x32) <= 1e-5


@pytest.mark.parametrize("n,dt,tol", [(8192, np.float64, 1e-10), (16384, np.float64, 1e-10), (65536, np.float64, 1e-10),
                                      (131072, np.float64, 1e-10), (262144, np.float64, 1e-10), (32768, np.float32, 1e-5),
                                      (131072, np.float32, 1e-5)])
def test_pyramids_every_depth(wx, oracle, n, dt, tol):
    wt = _wt(wx, "db4")
    rng = np.random.default_rng(n)
    x = np.asfortranarray(rng.standard_normal((n, 2)).astype(dt))
    Lmax = wx.maxtransformlevels(n)
    for L in sorted({1, 2, 3, 4, 5, 6, Lmax - 7, Lmax - 1, Lmax}):
        y = wx.dwtall(x, wt, L)
        assert relerr(y, oracle.wptall(x, wt.qmf, wx.maketree(n, L, "dwt"))) <= tol, (n, L)
        assert relerr(wx.idwtall(y, wt, L), x) <= tol, (n, L)


@pytest.mark.parametrize("n,dt,tol", [(8192, np.float64, 1e-10), (16384, np.float64, 1e-10), (65536, np.float64, 1e-10),
                                      (8192, np.float32, 1e-5), (32768, np.float32, 1e-5)])
@pytest.mark.parametrize("wname", ["db2", "db4", "coif6"])
def test_random_trees(wx, oracle, n, dt, tol, wname):
    wt = _wt(wx, wname)
    rng = np.random.default_rng(7 * n + len(wname))
    x = np.asfortranarray(rng.standard_normal((n, 2)).astype(dt))
    for k in range(6):
        tree = random_tree_1d(n, rng, p=(0.9, 0.7, 0.5)[k % 3])
        if k == 4:
            tree[:] = False
            tree[:7] = True                                    # three full levels and nothing below: all leaves in the top pass
        if k == 5:
            tree[:] = False
            tree[0] = tree[2] = tree[6] = True                 # the rightmost path only
        y = wx.wptall(x, wt, tree)
        assert relerr(y, oracle.wptall(x, wt.qmf, tree)) <= tol, (n, k)
        assert relerr(wx.iwptall(y, wt, tree), x) <= tol, (n, k)


def test_device_batch_round_trip_and_ragged_tiles(wx):
    """a batch that fills the chip, every signal checked on the device (a wrong tile or halo anywhere shows)"""
    import torch
    wt = _wt(wx, "db4")
    for n, B in ((16384, 1031), (65536, 257)):
        x = wx.jl_empty((n, B), torch.float64, "cuda")
        x.normal_()
        L = wx.maxtransformlevels(n)
        for fwd, inv in ((lambda a: wx.wptall(a, wt, L), lambda a: wx.iwptall(a, wt, L)), (lambda a: wx.dwtall(a, wt), lambda a: wx.idwtall(a, wt))):
            y = fwd(x)
            assert float((y * y).sum() / (x * x).sum() - 1.0) < 1e-12        # orthonormal
            assert float((inv(y) - x).abs().max() / x.abs().max()) <= 1e-10


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("n", [8192, 16384, 65536])
def test_wpd_of_long_signals_top_slices_in_one_pass(wx, oracle, dt, n):
    """wpdall of long signals (DWT.jl:164-209 via dwt/dwt_all.jl:262-281): the slices of the levels above the longest node a CU's LDS holds
    -- and slice 0, the signal -- come from ONE tiled pass (wx_dev_top_levels_wpd), the deeper ones from the fused kernel: every slice
    against the oracle's table, several depths, two filters"""
    rng = np.random.default_rng(n)
    tol = 1e-10 if dt == np.float64 else 2e-5
    for wname in ("db4", "db2"):
        wt = wx.wavelet(getattr(wx.WT, wname))
        x = np.asfortranarray(rng.standard_normal((n, 3)).astype(dt))
        for L in (1, 2, 3, 4, 6, wx.maxtransformlevels(n)):
            got = wx.wpdall(x, wt, L)
            exp = oracle.wpdall(x.astype(np.float64), wt.qmf, L)
            assert got.shape == exp.shape
            for l in range(L + 1):
                den = np.abs(exp[:, l, :]).max()
                assert np.abs(got[:, l, :] - exp[:, l, :]).max() <= tol * den, (dt, n, wname, L, l)
            # iwpd of the full tree reads slice L of the table: one tiled pass with the table's stride (L <= 4), a strided copy + the dense
            # long-signal kernels (deeper)
            back = wx.iwpdall(exp.astype(dt), wt, L)
            assert np.abs(back - x).max() <= 4 * tol * np.abs(x).max(), (dt, n, wname, L)
