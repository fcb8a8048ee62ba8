"""Short signals (16 ... 512 samples) along any tree through the small-signal kernel (csrc/wx_smalltree.hip): random trees,
pyramids and full trees, Float64 and Float32, every filter length, batches that do not fill a wavefront's slab, against the
oracle (dwt/dwt_one_level.jl:79-107, 192-223; Wavelets.jl wpt / iwpt as called by dwt/dwt_all.jl:152-225)."""
import numpy as np
import pytest

from helpers import random_tree_1d, relerr

pytestmark = pytest.mark.gpu
FILTERS = ["haar", "db2", "db3", "db4", "db5", "coif2", "db7", "db8", "coif6", "db10"]      # 2 ... 20 taps


def _wt(wx, name):
    return wx.wavelet(getattr(wx.WT, name))


@pytest.mark.parametrize("n", [16, 32, 64, 128, 256, 512])
@pytest.mark.parametrize("dt,tol", [(np.float64, 1e-10), (np.float32, 1e-5)])
def test_random_trees_pyramids_full_trees(wx, oracle, n, dt, tol):
    wt = _wt(wx, "db4")
    rng = np.random.default_rng(n)
    Lmax = wx.maxtransformlevels(n)
    for B in (1, 7, 100):
        x = np.asfortranarray(rng.standard_normal((n, B)).astype(dt))
        trees = [random_tree_1d(n, rng, p=p) for p in (0.9, 0.7, 0.5, 0.3)]
        for t in trees:
            t[0] = True
        trees += [wx.maketree(n, L, "dwt") for L in (1, min(4, Lmax), Lmax)]
        for tree in trees:
            y = wx.wptall(x, wt, tree)
            assert relerr(y, oracle.wptall(x, wt.qmf, tree)) <= tol, (n, B)
            assert relerr(wx.iwptall(y, wt, tree), x) <= tol, (n, B)
        for L in (1, min(4, Lmax), Lmax):
            y = wx.wptall(x, wt, L)
            assert relerr(y, oracle.wptall(x, wt.qmf, L)) <= tol, (n, B, L)
            assert relerr(wx.iwptall(y, wt, L), x) <= tol, (n, B, L)
            yd = wx.dwtall(x, wt, L)
            assert relerr(yd, oracle.wptall(x, wt.qmf, wx.maketree(n, L, "dwt"))) <= tol
            assert relerr(wx.idwtall(yd, wt, L), x) <= tol


@pytest.mark.parametrize("wname", FILTERS)
def test_every_filter_length(wx, oracle, wname):
    wt = _wt(wx, wname)
    rng = np.random.default_rng(len(wt.qmf))
    for n in (32, 64):                                       # filters longer than the deep nodes wrap around several times
        x = np.asfortranarray(rng.standard_normal((n, 19)))
        tree = random_tree_1d(n, rng, p=0.8)
        tree[0] = True
        y = wx.wptall(x, wt, tree)
        assert relerr(y, oracle.wptall(x, wt.qmf, tree)) <= 1e-10
        assert relerr(wx.iwptall(y, wt, tree), x) <= 1e-10
        x32 = np.asfortranarray(x.astype(np.float32))
        y32 = wx.wptall(x32, wt, tree)
        assert relerr(y32, oracle.wptall(x32, wt.qmf, tree)) <= 1e-5
        assert relerr(wx.iwptall(y32, wt, tree), x32) <= 1e-5


def test_device_batch_matches_the_generic_kernels(wx):
    """a batch that fills the chip: the small-signal kernel against the one-level-per-launch kernels (force_generic)"""
    import torch
    wt = _wt(wx, "db4")
    rng = np.random.default_rng(5)
    for n, B in ((64, 100003), (128, 33333)):
        x = wx.jl_empty((n, B), torch.float64, "cuda")
        x.normal_()
        tree = random_tree_1d(n, rng, p=0.7)
        tree[0] = True
        y = wx.wptall(x, wt, tree)
        wx.set_force_generic(1)
        try:
            ref = wx.wptall(x, wt, tree)
        finally:
            wx.set_force_generic(0)
        assert float((y - ref).abs().max() / ref.abs().max()) <= 1e-12
        assert float((wx.iwptall(y, wt, tree) - x).abs().max() / x.abs().max()) <= 1e-10
