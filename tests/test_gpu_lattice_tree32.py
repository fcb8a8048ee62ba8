"""Float32 signals along a tree on the lattice kernels (csrc/wx_lattice_tree32.h): the reference is generic in the
element type (dwt/dwt_one_level.jl:79-83), wptall / iwptall with a tree (dwt/dwt_all.jl:152-225).  More than 200 fuzzed trees over
the three signal lengths of the tree-driven kernels, pyramids of every depth, every filter the lattice factors, ragged batches, long
signals (top pass + one lattice launch per 4096-sample subtree), all against the oracle within 1e-5, inverse against the signal."""
import numpy as np
import pytest

from helpers import random_tree_1d, relerr

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _wt(wx, name):
    return wx.wavelet(getattr(wx.WT, name))


@pytest.mark.parametrize("n", [4096, 2048, 1024])
def test_fuzzed_trees(wx, oracle, n):
    wt = _wt(wx, "db4")
    rng = np.random.default_rng(n + 32)
    x = np.asfortranarray(rng.standard_normal((n, 5)).astype(np.float32))
    worst = 0.0
    for k in range(70):
        tree = random_tree_1d(n, rng, p=(0.95, 0.8, 0.7, 0.5, 0.3)[k % 5])
        tree[0] = True
        y = wx.wptall(x, wt, tree)
        e1 = relerr(y, oracle.wptall(x, wt.qmf, tree))
        e2 = relerr(wx.iwptall(y, wt, tree), x)
        worst = max(worst, e1, e2)
        assert e1 <= TOL and e2 <= TOL, (n, k, e1, e2)
    assert worst > 0.0                      # Float32 rounding is there: the Float64 kernels did not run on a widened copy


@pytest.mark.parametrize("n", [4096, 1024])
@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db5", "coif2", "db8", "coif6", "db10"])
def test_filters_pyramids_and_one_sided_trees(wx, oracle, n, wname):
    wt = _wt(wx, wname)
    rng = np.random.default_rng(len(wt.qmf) + n)
    x = np.asfortranarray(rng.standard_normal((n, 7)).astype(np.float32))
    Lmax = wx.maxtransformlevels(n)
    trees = [wx.maketree(n, L, "dwt") for L in (1, 3, 6, 7, Lmax - 1, Lmax)]
    right = np.zeros(n - 1, dtype=bool)
    i = 1
    while i <= n - 1 and i < (1 << 9):
        right[i - 1] = True                  # always the detail child
        i = 2 * i + 1
    trees.append(right)
    trees.append(random_tree_1d(n, rng, p=0.75))
    trees[-1][0] = True
    for tree in trees:
        y = wx.wptall(x, wt, tree)
        assert relerr(y, oracle.wptall(x, wt.qmf, tree)) <= TOL
        assert relerr(wx.iwptall(y, wt, tree), x) <= TOL


@pytest.mark.parametrize("n", [8192, 16384, 65536])
def test_long_float32_signals_along_trees(wx, oracle, n):
    wt = _wt(wx, "db4")
    rng = np.random.default_rng(n + 1)
    x = np.asfortranarray(rng.standard_normal((n, 3)).astype(np.float32))
    for k in range(6):
        tree = random_tree_1d(n, rng, p=(0.9, 0.7, 0.5)[k % 3])
        tree[0] = True
        y = wx.wptall(x, wt, tree)
        assert relerr(y, oracle.wptall(x, wt.qmf, tree)) <= TOL, (n, k)
        assert relerr(wx.iwptall(y, wt, tree), x) <= TOL, (n, k)
    Lmax = wx.maxtransformlevels(n)
    for L in (1, 2, 5, Lmax - 6, Lmax):
        y = wx.dwtall(x, wt, L)
        assert relerr(y, oracle.wptall(x, wt.qmf, wx.maketree(n, L, "dwt"))) <= TOL, (n, L)
        assert relerr(wx.idwtall(y, wt, L), x) <= TOL, (n, L)


def test_device_batch_and_denoise(wx):
    """a batch that fills the chip, ragged against the signals a wavefront interleaves, checked on the device; denoiseall of a
    Float32 batch (the threshold does not ride on the Float32 lattice loads: the call still has to be right)"""
    import torch
    wt = _wt(wx, "db4")
    rng = np.random.default_rng(9)
    for n, B in ((4096, 10007), (1024, 40003)):
        x = wx.jl_empty((n, B), torch.float32, "cuda")
        x.normal_()
        tree = random_tree_1d(n, rng, p=0.7)
        tree[0] = True
        y = wx.wptall(x, wt, tree)
        assert abs(float((y.double() ** 2).sum() / (x.double() ** 2).sum()) - 1.0) < 1e-5
        assert float((wx.iwptall(y, wt, tree) - x).abs().max() / x.abs().max()) <= TOL
    xs = wx.jl_empty((4096, 64), torch.float32, "cuda")
    xs.normal_()
    d = wx.denoiseall(xs, "sig", wt)
    assert tuple(d.shape) == (4096, 64) and bool(torch.isfinite(d).all())
