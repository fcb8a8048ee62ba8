"""Float32 full trees of 64 ... 2048 samples on the interleaved lattice kernels (csrc/wx_lattice_sg32.h: 2^SH signals per wavefront,
Float32 at both ends, rotations in Float64) and the paths around them: every depth from 1 to the maximum (the kernels take L + SH >= 6,
shallower trees fall to the tree-driven or the fused kernels), filters of 2 ... 8 taps on the lattice and longer ones beside it, ragged
batches (the last wavefront re-does signals), device-resident batches that fill the chip, and the columns of small Float32 images.
Reference: Wavelets.jl wpt / iwpt on an AbstractArray{T} as called by wptall / iwptall (dwt/dwt_all.jl:152-166, 210-225); the
reference is generic in T (dwt/dwt_one_level.jl:79-83).  Against the oracle within 1e-5, inverse against the signal."""
import numpy as np
import pytest

from helpers import relerr

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _wt(wx, name):
    return wx.wavelet(getattr(wx.WT, name))


@pytest.mark.parametrize("n", [64, 128, 256, 512, 1024, 2048])
@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db4", "db5", "coif6"])
def test_every_depth(wx, oracle, n, wname):
    wt = _wt(wx, wname)
    rng = np.random.default_rng(n + len(wt.qmf))
    x = np.asfortranarray(rng.standard_normal((n, 67)).astype(np.float32))          # 67: no multiple of any 2^SH
    for L in range(1, wx.maxtransformlevels(n) + 1):
        y = wx.wptall(x, wt, L)
        assert y.dtype == np.float32
        assert relerr(y, oracle.wptall(x, wt.qmf, L)) <= TOL, (n, wname, L)
        assert relerr(wx.iwptall(y, wt, L), x) <= TOL, (n, wname, L)


@pytest.mark.parametrize("batch", [1, 2, 3, 31, 64, 65, 1000])
def test_ragged_batches(wx, oracle, batch):
    wt = _wt(wx, "db4")
    rng = np.random.default_rng(batch)
    for n in (64, 256, 2048):
        x = np.asfortranarray(rng.standard_normal((n, batch)).astype(np.float32))
        L = wx.maxtransformlevels(n)
        y = wx.wptall(x, wt, L)
        assert relerr(y, oracle.wptall(x, wt.qmf, L)) <= TOL, (n, batch)
        assert relerr(wx.iwptall(y, wt, L), x) <= TOL, (n, batch)


def test_device_batches_that_fill_the_chip(wx):
    """every signal of a large device-resident batch: orthonormality and the round trip on the device (a wavefront that mixed up
    its 2^SH signals would show)"""
    import torch
    wt = _wt(wx, "db4")
    for n, B in ((64, 1 << 18), (512, (1 << 15) + 5), (2048, (1 << 13) + 1)):
        x = wx.jl_empty((n, B), torch.float32, "cuda")
        x.normal_()
        L = wx.maxtransformlevels(n)
        y = wx.wptall(x, wt, L)
        ex, ey = (x.double() ** 2).sum(0), (y.double() ** 2).sum(0)
        assert float(((ey - ex).abs() / ex).max()) <= 1e-5                           # per signal
        assert float((wx.iwptall(y, wt, L) - x).abs().max() / x.abs().max()) <= TOL


@pytest.mark.parametrize("m", [64, 128, 256])
def test_columns_of_small_float32_images(wx, oracle, m):
    wt = _wt(wx, "db4")
    rng = np.random.default_rng(m)
    x = np.asfortranarray(rng.standard_normal((m, m, 5)).astype(np.float32))
    for L in (2, wx.maxtransformlevels(m)):
        y = wx.wptall(x, wt, L)
        assert relerr(y, oracle.wptall(x, wt.qmf, L)) <= TOL, (m, L)
        assert relerr(wx.iwptall(y, wt, L), x) <= TOL, (m, L)
