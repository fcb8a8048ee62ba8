"""Large and non-square images through the 2-D paths of round 4 (persistent tile levels, block-sized launches below them, the row pass
with image-width-dependent strips): pyramids of several depths, a random quad tree and a full tree of partial depth, against the oracle
per image and round trip (DWT.jl:440-710, dwt/dwt_all.jl:39-110, 152-225; 2-D dwt_step! / idwt_step! dwt/dwt_one_level.jl:319-354,
401-436).  Shapes no other test reaches: 2048 x 2048, 2048 x 512, 256 x 2048, 64 x 1024."""
import numpy as np
import pytest

from helpers import random_tree_2d, relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,dt,tol", [((2048, 2048), np.float32, 1e-5), ((2048, 512), np.float64, 1e-10),
                                          ((256, 2048), np.float32, 1e-5), ((64, 1024), np.float64, 1e-10)])
def test_large_and_non_square_images(wx, oracle, shape, dt, tol):
    m, n = shape
    rng = np.random.default_rng(m + n)
    wt = wx.wavelet(wx.WT.db4)
    x = np.asfortranarray(rng.standard_normal((m, n, 2)).astype(dt))
    x64 = x.astype(np.float64)
    Lm = wx.maxtransformlevels(min(m, n))
    for L in (1, 3, Lm):
        y = wx.dwtall(x, wt, L)
        assert relerr(y[:, :, 1], oracle.wpt(x64[:, :, 1], wt.qmf, wx.maketree(m, n, L, "dwt"))) <= tol, (shape, L)
        assert relerr(wx.idwtall(y, wt, L), x) <= tol, (shape, L)
    tree = random_tree_2d(m, n, rng, p=0.6)
    tree[0] = True
    y = wx.wptall(x, wt, tree)
    assert relerr(y[:, :, 0], oracle.wpt(x64[:, :, 0], wt.qmf, tree)) <= tol, shape
    assert relerr(wx.iwptall(y, wt, tree), x) <= tol, shape
    y = wx.wptall(x, wt, 3)
    assert relerr(y[:, :, 0], oracle.wpt(x64[:, :, 0], wt.qmf, 3)) <= tol, shape
    assert relerr(wx.iwptall(y, wt, 3), x) <= tol, shape
