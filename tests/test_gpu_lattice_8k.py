"""8192-sample Float64 signals, full trees, in ONE pass over the data (csrc/wx_lattice_8k.h): two wavefronts per
signal, the first level in the direct form of dwt_step! / idwt_step! (dwt/dwt_one_level.jl:79-107, 192-223) inside the load / store phase,
the children on the lattice.  Against the oracle for every filter length the lattice factors (2 .. 20 taps: the halo of the staged
chunks), every depth the kernels take (7 .. 13), odd batches, and iwpd reading column L of a table (signal stride 8192 (L + 1))."""
import numpy as np
import pytest

from helpers import relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db4", "db5", "coif2", "db7", "db8", "coif6", "db10"])
def test_every_filter_full_depth(wx, oracle, wname):
    wt = wx.wavelet(getattr(wx.WT, wname))
    rng = np.random.default_rng(len(wt.qmf))
    x = np.asfortranarray(rng.standard_normal((8192, 5)))
    for L in (13, 7):
        exp = oracle.wptall(x, wt.qmf, L)
        y = wx.wptall(x, wt, L)
        assert relerr(y, exp) <= 1e-12, (wname, L)
        assert relerr(wx.iwptall(exp, wt, L), x) <= 1e-12, (wname, L)


@pytest.mark.parametrize("L", [7, 8, 9, 10, 11, 12, 13])
def test_every_depth_and_tables(wx, oracle, L):
    wt = wx.wavelet(wx.WT.db4)
    rng = np.random.default_rng(L)
    for B in (1, 2, 7):
        x = np.asfortranarray(rng.standard_normal((8192, B)))
        exp = oracle.wptall(x, wt.qmf, L)
        assert relerr(wx.wptall(x, wt, L), exp) <= 1e-12, (L, B)
        assert relerr(wx.iwptall(exp, wt, L), x) <= 1e-12, (L, B)
    x = np.asfortranarray(rng.standard_normal((8192, 3)))
    tab = wx.wpdall(x, wt, L)                                          # (8192, L + 1, 3): the inverse reads column L in place
    assert relerr(wx.iwpdall(tab, wt, L), x) <= 1e-12


def test_chip_filling_batch_on_the_device(wx):
    import torch
    wt = wx.wavelet(wx.WT.db4)
    x = wx.jl_empty((8192, 4099), torch.float64, "cuda")
    x.normal_()
    x *= torch.logspace(-3, 3, 4099, device="cuda", dtype=torch.float64)[None, :]
    y = wx.wptall(x, wt, 13)
    ex, ey = (x ** 2).sum(0), (y ** 2).sum(0)
    assert float(((ey - ex).abs() / ex).max()) <= 1e-12              # orthonormal, per signal
    back = wx.iwptall(y, wt, 13)
    assert float(((back - x).abs().amax(0) / x.abs().amax(0)).max()) <= 1e-12


def _trees_8192(wx, rng):
    from helpers import random_tree_1d
    n = 8192
    trees = {}
    t = np.zeros(n - 1, dtype=bool); t[0] = True
    trees["root only (two leaf children)"] = t
    t = np.zeros(n - 1, dtype=bool); t[[0, 1]] = True
    trees["approximation child split once, detail child a leaf"] = t
    t = np.zeros(n - 1, dtype=bool); t[[0, 2, 6]] = True
    trees["detail side only"] = t
    trees["pyramid depth 13"] = wx.maketree(n, 13, "dwt")
    trees["pyramid depth 5"] = wx.maketree(n, 5, "dwt")
    trees["full depth 3"] = wx.maketree(n, 3, "full")
    for i in range(4):
        r = random_tree_1d(n, rng, p=0.8)
        r[0] = True
        trees["random %d" % i] = r
    return trees


@pytest.mark.parametrize("wname", ["haar", "db2", "db4", "db7", "coif6", "db10"])
def test_trees_in_one_pass(wx, oracle, wname):
    """wptall / iwptall of 8192-sample signals along a tree (wx_lattice_8kt.h): every child configuration of the root -- leaf / leaf, subtree /
    leaf, leaf / subtree, subtree / subtree --, pyramids, random trees; forward against the oracle, inverse against the signal"""
    wt = wx.wavelet(getattr(wx.WT, wname))
    rng = np.random.default_rng(8192 + len(wt.qmf))
    x = np.asfortranarray(rng.standard_normal((8192, 3)))
    for name, tree in _trees_8192(wx, rng).items():
        exp = oracle.wptall(x, wt.qmf, tree)
        y = wx.wptall(x, wt, tree)
        assert relerr(y, exp) <= 1e-12, (wname, name)
        assert relerr(wx.iwptall(exp, wt, tree), x) <= 1e-12, (wname, name)


def test_pyramids_through_dwtall(wx, oracle):
    wt = wx.wavelet(wx.WT.db4)
    rng = np.random.default_rng(81)
    x = np.asfortranarray(rng.standard_normal((8192, 5)))
    for L in (1, 2, 6, 13):
        tree = wx.maketree(8192, L, "dwt")
        exp = oracle.wptall(x, wt.qmf, tree)
        assert relerr(wx.dwtall(x, wt, L), exp) <= 1e-12, L
        assert relerr(wx.idwtall(exp, wt, L), x) <= 1e-12, L
