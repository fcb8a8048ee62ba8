"""dwt / idwt (the pyramid) of long Float64 signals, 16384 .. 65536 samples: tiled top levels on the approximation branch, then
the 4096-sample pyramid on the tree-driven lattice kernels reading / writing with the long signal's stride, the deepest levels
lane-locally (csrc/wx_dwt1d.hip: wx_dev_dwt_long / wx_dev_idwt_long).

Reference: Wavelets.jl dwt / idwt as called by dwtall / idwtall (dwt/dwt_all.jl:33-121) = wpt / iwpt along maketree(n, L, :dwt).
Float64, 1e-10 relative.
"""
import numpy as np
import pytest

from helpers import relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [16384, 32768, 65536])
@pytest.mark.parametrize("wname", ["haar", "db4", "db7", "coif6"])
def test_long_pyramids_match_oracle(wx, oracle, n, wname):
    rng = np.random.default_rng(n + len(wname))
    wt = wx.wavelet(getattr(wx.WT, wname))
    Lmax = wx.maxtransformlevels(n)
    dl = int(np.log2(n // 4096))
    B = 3
    x = np.asfortranarray(rng.standard_normal((n, B)))
    # full depth (lane-local tail), a depth that ends inside the lattice part, one that ends in the tiled top levels, depth 1
    for L in (Lmax, dl + 3, dl, 1):
        tree = np.asarray(wx.maketree(n, L, "dwt"))
        exp = oracle.wptall(x, wt.qmf, tree)
        got = wx.wptall(x, wt, tree)
        assert relerr(got, exp) <= 1e-10, (n, wname, L)
        back = wx.iwptall(exp, wt, tree)
        assert relerr(back, x) <= 1e-10, (n, wname, L)
    # the named entry points
    assert relerr(wx.dwtall(x, wt), oracle.wptall(x, wt.qmf, np.asarray(wx.maketree(n, Lmax, "dwt")))) <= 1e-10
    assert relerr(wx.idwtall(wx.dwtall(x, wt), wt), x) <= 1e-10


def test_long_pyramid_device_batch_round_trips(wx):
    import torch
    wt = wx.wavelet(wx.WT.db4)
    n, B = 16384, 1000
    x = wx.jl_empty((n, B), torch.float64, "cuda")
    x.normal_(generator=torch.Generator(device="cuda").manual_seed(3))
    y = wx.dwtall(x, wt)
    xr = wx.idwtall(y, wt)
    err = (xr - x).abs().amax(dim=0) / x.abs().max()
    assert float(err.max()) <= 1e-12, int(err.argmax())
    # linearity / energy: an orthogonal transform keeps the norm of every signal
    assert float(((y * y).sum(dim=0) / (x * x).sum(dim=0) - 1).abs().max()) <= 1e-12


@pytest.mark.parametrize("n,dt,tol", [(32768, np.float32, 2e-5), (65536, np.float32, 2e-5), (131072, np.float64, 1e-10),
                                      (262144, np.float32, 2e-5)])
@pytest.mark.parametrize("wname", ["haar", "db4", "coif6"])
def test_long_pyramids_other_types_and_lengths(wx, oracle, n, dt, tol, wname):
    """Float32 signals (the fused LDS kernel finishes the pyramid at 16384 samples) and lengths beyond 65536"""
    rng = np.random.default_rng(n)
    wt = wx.wavelet(getattr(wx.WT, wname))
    Lmax = wx.maxtransformlevels(n)
    x = np.asfortranarray(rng.standard_normal((n, 2)).astype(dt))
    for L in (Lmax, 5, 2):
        tree = np.asarray(wx.maketree(n, L, "dwt"))
        exp = oracle.wptall(x.astype(np.float64), wt.qmf, tree)
        got = wx.wptall(x, wt, tree)
        assert got.dtype == dt
        assert relerr(np.asarray(got, dtype=np.float64), exp) <= tol, (n, wname, L)
        back = wx.iwptall(exp.astype(dt), wt, tree)
        assert relerr(np.asarray(back, dtype=np.float64), x.astype(np.float64)) <= tol, (n, wname, L)


@pytest.mark.parametrize("n", [16384, 65536])
@pytest.mark.parametrize("wname", ["db4", "coif6"])
def test_long_signals_along_any_tree(wx, oracle, n, wname):
    """wptall / iwptall along trees other than the pyramid (bestbasistree output) on long Float64 signals: tiled passes on the split
    nodes of the top levels, one lattice launch per 4096-sample node with its own subtree (wx_dev_wpt_long_tree)"""
    from helpers import random_tree_1d
    rng = np.random.default_rng(n + 5)
    wt = wx.wavelet(getattr(wx.WT, wname))
    x = np.asfortranarray(rng.standard_normal((n, 3)))
    Lmax = wx.maxtransformlevels(n)
    trees = []
    for p in (0.4, 0.7, 0.9):
        for _ in range(3):
            tr = random_tree_1d(n, rng, p)
            tr[0] = True
            trees.append(tr)
    t = np.zeros(n - 1, dtype=bool)              # the mirror of the pyramid: always the detail child
    i = 1
    while i <= n - 1:
        t[i - 1] = True
        i = 2 * i + 1
    trees.append(t)
    t = np.array(wx.maketree(n, 3, "full"), dtype=bool).copy()      # full to depth 3, one node opened to the bottom
    i = 8 + 5
    while i <= n - 1:
        t[i - 1] = True
        i = 2 * i
    trees.append(t)
    trees.append(np.array(wx.maketree(n, 2, "full"), dtype=bool))   # ends inside the tiled levels
    for k, tree in enumerate(trees):
        exp = oracle.wptall(x, wt.qmf, tree)
        got = wx.wptall(x, wt, tree)
        assert relerr(got, exp) <= 1e-10, (n, wname, k)
        assert relerr(wx.iwptall(exp, wt, tree), x) <= 1e-10, (n, wname, k)
