"""GPU parity of the 3-D dwt / idwt and dwtall / idwtall on 4-D arrays (csrc/wx_dwt3d.hip; dwt/dwt_all.jl:39-54, 95-110 over
Wavelets.jl's 3-D transform) against the oracle's separable pyramid.  Tolerance 1e-10 (north_star), Float32 1e-5."""
import numpy as np
import pytest

from helpers import relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wname,n", [("haar", 4), ("db2", 8), ("db4", 16), ("coif2", 16), ("db8", 8)])
def test_dwt3d_matches_oracle_every_depth(wx, oracle, wname, n):
    rng = np.random.default_rng(n)
    wt = wx.wavelet(getattr(wx.WT, wname))
    x = np.asfortranarray(rng.standard_normal((n, n, n)))
    for L in [None] + list(range(0, oracle.maxtransformlevels(n) + 1)):
        exp = oracle.dwt3d(x, wt.qmf, L)
        got = wx.dwt(x, wt) if L is None else wx.dwt(x, wt, L)
        assert relerr(got, exp) <= 1e-12, (wname, L)
        back = wx.idwt(exp, wt) if L is None else wx.idwt(exp, wt, L)
        assert relerr(back, x) <= 1e-12, (wname, L)


def test_dwtall_idwtall_on_a_batch_of_cubes(wx, oracle):
    rng = np.random.default_rng(33)
    wt = wx.wavelet(wx.WT.db4)
    x = np.asfortranarray(rng.standard_normal((16, 16, 16, 5)))
    y = wx.dwtall(x, wt, 2)
    for i in range(5):
        assert relerr(y[..., i], oracle.dwt3d(x[..., i], wt.qmf, 2)) <= 1e-12
    assert relerr(wx.idwtall(y, wt, 2), x) <= 1e-12
    # default depth, Float32, a larger cube: reconstruction and energy
    x32 = np.asfortranarray(rng.standard_normal((64, 64, 64, 3)).astype(np.float32))
    y32 = wx.dwtall(x32, wt)
    assert y32.dtype == np.float32
    assert abs(float((y32.astype(np.float64) ** 2).sum()) / float((x32.astype(np.float64) ** 2).sum()) - 1.0) <= 1e-5
    assert relerr(wx.idwtall(y32, wt), x32) <= 1e-5


def test_dwt3d_argument_checks(wx):
    wt = wx.wavelet(wx.WT.haar)
    with pytest.raises(Exception):
        wx.dwt(np.zeros((8, 8, 4), order="F"), wt, 1)                  # not a cube
    with pytest.raises(Exception):
        wx.dwt(np.zeros((8, 8, 8), order="F"), wt, 4)                  # deeper than maxtransformlevels
