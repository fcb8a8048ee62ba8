"""dwt / idwt (pyramids) of small images through the whole-image-in-LDS kernel (csrc/wx_pyr2d.hip) against the oracle: Wavelets.jl's
2-D dwt / idwt as dwtall / idwtall call them (dwt/dwt_all.jl:39-110) = wpt / iwpt along maketree(m, n, L, :dwt); one level is the 2-D
dwt_step! / idwt_step! of dwt/dwt_one_level.jl:319-354, 401-436.  Square and non-square dyadic images from 4 x 4 to 128 x 128, every
depth, filters of 2 ... 20 taps (compile-time lengths up to 8, the loop form above), Float64 (1e-10) and Float32 (1e-5), ragged and
chip-filling batches, and the same call with the kernel switched off (the tile / block path) for comparison."""
import numpy as np
import pytest

from helpers import relerr

pytestmark = pytest.mark.gpu


def _wt(wx, name):
    return wx.wavelet(getattr(wx.WT, name))


@pytest.mark.parametrize("shape", [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (128, 128), (64, 128), (128, 32), (8, 64)])
@pytest.mark.parametrize("dt,tol", [(np.float64, 1e-10), (np.float32, 1e-5)])
def test_every_depth_db4(wx, oracle, shape, dt, tol):
    m, n = shape
    wt = _wt(wx, "db4")
    rng = np.random.default_rng(m * 1000 + n)
    x = np.asfortranarray(rng.standard_normal((m, n, 5)).astype(dt))
    for L in range(1, wx.maxtransformlevels(min(m, n)) + 1):
        tree = wx.maketree(m, n, L, "dwt")
        y = wx.dwtall(x, wt, L)
        assert y.dtype == dt
        exp = oracle.wptall(x.astype(np.float64), wt.qmf, tree)
        assert relerr(y, exp) <= tol, (shape, L)
        assert relerr(wx.idwtall(y, wt, L), x) <= tol, (shape, L)
        assert relerr(wx.iwptall(exp.astype(dt), wt, tree), x) <= tol, (shape, L)


@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db5", "coif2", "db8", "coif6", "db10"])
def test_filters(wx, oracle, wname):
    wt = _wt(wx, wname)
    rng = np.random.default_rng(len(wt.qmf))
    for m, dt, tol in ((64, np.float64, 1e-10), (32, np.float32, 1e-5), (128, np.float32, 1e-5)):
        x = np.asfortranarray(rng.standard_normal((m, m, 3)).astype(dt))
        L = wx.maxtransformlevels(m)
        y = wx.dwtall(x, wt, L)
        assert relerr(y, oracle.wptall(x.astype(np.float64), wt.qmf, wx.maketree(m, m, L, "dwt"))) <= tol, (wname, m)
        assert relerr(wx.idwtall(y, wt, L), x) <= tol, (wname, m)


def test_batches_that_fill_the_chip(wx):
    import torch
    wt = _wt(wx, "db4")
    for m, dt, B, tol in ((64, torch.float32, 40003, 2e-6), (64, torch.float64, 20001, 1e-12), (128, torch.float32, 5001, 2e-6)):
        x = wx.jl_empty((m, m, B), dt, "cuda")
        x.normal_()
        y = wx.dwtall(x, wt)
        ex, ey = (x.double() ** 2).sum((0, 1)), (y.double() ** 2).sum((0, 1))
        assert float(((ey - ex).abs() / ex).max()) <= 100 * tol                # orthonormal, per image
        err = (wx.idwtall(y, wt) - x).abs().amax(dim=(0, 1)) / x.abs().max()
        assert float(err.max()) <= tol, (m, int(err.argmax()))
