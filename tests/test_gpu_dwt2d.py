"""GPU parity: 2-D decimated wavelet packets (quad trees) vs the CPU oracle.  Float64 1e-10,
Float32 1e-5 (config 4 is Float32)."""
import numpy as np
import pytest

from helpers import TOL, random_tree_2d, relerr

pytestmark = pytest.mark.gpu


def _wt(wx, name):
    return wx.wavelet(getattr(wx.WT, name))


def _stack(fn, X, *a):
    return np.asfortranarray(np.stack([fn(np.asfortranarray(X[..., i]), *a) for i in range(X.shape[-1])], axis=-1))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("wname", ["haar", "db4", "coif6"])
def test_wpd2d_and_wpt2d(wx, oracle, wname, dtype):
    rng = np.random.default_rng(3001)
    wt = _wt(wx, wname)
    tol = TOL[np.dtype(dtype)]
    for (m, n), B in (((2, 2), 2), ((8, 8), 3), ((16, 32), 2), ((24, 8), 2), ((64, 64), 2)):
        x = np.asfortranarray(rng.standard_normal((m, n, B)).astype(dtype))
        Lmax = wx.maxtransformlevels(min(m, n))
        for L in sorted({0, 1, Lmax}):
            got = wx.wpdall(x, wt, L)
            assert got.shape == (m, n, L + 1, B)
            assert relerr(got, _stack(oracle.wpd, x, wt.qmf, L)) <= tol, (m, n, L)
        trees = [None, 1, wx.maketree(m, n, Lmax, "dwt"), random_tree_2d(m, n, rng), random_tree_2d(m, n, rng, 0.85)]
        for arg in trees:
            exp = _stack(oracle.wpt, x, wt.qmf, arg)
            got = wx.wptall(x, wt, arg)
            assert relerr(got, exp) <= tol, (m, n)
            back = wx.iwptall(exp, wt, arg)
            assert relerr(back, _stack(oracle.iwpt, exp, wt.qmf, arg)) <= tol
            assert relerr(back, x) <= 20 * tol
        # single-image methods, and wpd slices == wpt by level (test/transforms.jl:36-43)
        z = wx.wpt(x[:, :, 0], wt, 1)
        assert relerr(z, wx.wpd(x[:, :, 0], wt)[:, :, 1] if Lmax >= 1 else x[:, :, 0]) <= tol


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_iwpd2d(wx, oracle, dtype):
    rng = np.random.default_rng(3002)
    wt = _wt(wx, "db4")
    tol = TOL[np.dtype(dtype)]
    for (m, n), B in (((8, 8), 3), ((32, 16), 2)):
        x = np.asfortranarray(rng.standard_normal((m, n, B)).astype(dtype))
        Lmax = wx.maxtransformlevels(min(m, n))
        xw = _stack(oracle.wpd, x, wt.qmf, Lmax)
        for arg in [None, 2, wx.maketree(m, n, Lmax, "dwt"), random_tree_2d(m, n, rng), random_tree_2d(m, n, rng, 0.85)]:
            got = wx.iwpdall(xw, wt, arg)
            assert relerr(got, _stack(oracle.iwpd, xw, wt.qmf, arg)) <= tol
            assert relerr(got, x) <= 20 * tol
        assert relerr(wx.iwpd(xw[..., 0], wt), x[..., 0]) <= 20 * tol
        with pytest.raises(IndexError):
            wx.iwpdall(xw[:, :, :2, :], wt, Lmax)          # needs slice L+1: BoundsError in the reference


def test_config4_shape_float32_properties(wx):
    """BASELINE config 4 geometry (512x512 Float32, db4, L=6) on a reduced batch: energy is
    preserved by the orthogonal transform and iwpt(wpt(x)) == x within Float32 tolerance."""
    import torch
    wt = _wt(wx, "db4")
    B = 64
    g = torch.Generator(device="cuda").manual_seed(1004)
    x = wx.jl_empty((512, 512, B), torch.float32, "cuda")
    x.normal_(generator=g)
    y = wx.wptall(x, wt, 6)
    e0 = (x.double() ** 2).sum(dim=(0, 1))
    e1 = (y.double() ** 2).sum(dim=(0, 1))
    assert float(((e1 - e0).abs() / e0).max()) < 1e-5
    back = wx.iwptall(y, wt, 6)
    assert float((back - x).abs().max() / x.abs().max()) < 1e-5


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("wname", ["haar", "db2", "db4", "db8", "db10"])
def test_wpd2d_tile_level_matches_oracle(wx, oracle, wname, dtype):
    """images whose sides are multiples of the 64 x 64 (Float32) / 64 x 32 (Float64) tile take the one-pass level
    (k_dwt2d_level_tile): nodes larger than the tile (halo + wrap), equal to it, and smaller (wrap inside LDS),
    square and not; the last levels (nodes below 8 samples) fall back to the two-pass level."""
    rng = np.random.default_rng(64)
    wt = wx.wavelet(getattr(wx.WT, wname))
    for (m, n, L, B) in ((64, 64, 6, 3), (128, 64, 5, 2), (64, 256, 4, 2), (256, 128, 3, 1)):
        x = np.asfortranarray(rng.standard_normal((m, n, B)).astype(dtype))
        got = wx.wpdall(x, wt, L)
        exp = oracle.wpdall(x, wt.qmf, L)
        assert got.shape == (m, n, L + 1, B)
        assert relerr(got, exp) <= TOL[np.dtype(dtype)], (m, n, L)
        assert relerr(wx.iwpdall(got, wt, L), x) <= 10 * TOL[np.dtype(dtype)]


def _truncate_tree2d(tree, depth):
    """drop every node of the quad tree at depth >= `depth` (heap order, depth d starts at (4^d - 1) / 3)"""
    t = np.array(tree, dtype=bool)
    t[(4 ** depth - 1) // 3:] = False
    return t


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("wname", ["haar", "db4", "db8"])
def test_wpt2d_tree_driven_tile_levels(wx, oracle, wname, dtype):
    """trees on images whose sides are multiples of the tile: the one-pass tree-driven levels (decomposed nodes
    only; children that are decomposed further go through the scratch images, leaves straight to the output) --
    nodes larger than, equal to and smaller than a tile, dwt trees, sparse and dense random trees."""
    rng = np.random.default_rng(3010)
    wt = _wt(wx, wname)
    tol = TOL[np.dtype(dtype)]
    for (m, n, depth, B) in ((64, 64, 4, 3), (128, 64, 3, 2), (64, 256, 4, 2), (256, 256, 5, 1)):
        x = np.asfortranarray(rng.standard_normal((m, n, B)).astype(dtype))
        full = wx.maketree(m, n, wx.maxtransformlevels(min(m, n)), "dwt")
        trees = [_truncate_tree2d(full, depth), _truncate_tree2d(full, 1),
                 _truncate_tree2d(random_tree_2d(m, n, rng, 0.5), depth),
                 _truncate_tree2d(random_tree_2d(m, n, rng, 0.9), depth)]
        for tree in trees:
            exp = _stack(oracle.wpt, x, wt.qmf, tree)
            got = wx.wptall(x, wt, tree)
            assert relerr(got, exp) <= tol, (m, n, int(tree.sum()))
            back = wx.iwptall(exp, wt, tree)
            assert relerr(back, _stack(oracle.iwpt, exp, wt.qmf, tree)) <= tol, (m, n, int(tree.sum()))
            assert relerr(back, x) <= 20 * tol
        xw = wx.wpdall(x, wt, depth)                      # iwpd gathers the leaves of the tree from the table first
        for tree in trees[2:]:
            assert relerr(wx.iwpdall(xw, wt, tree), x) <= 20 * tol


@pytest.mark.parametrize("wname", ["haar", "db2", "db4"])
def test_full_tree_rows_in_place_kernel(wx, oracle, wname):
    """row counts that select the one-LDS-image row kernel of the full-tree fast path (512 columns Float32, 256
    columns Float64): forward and inverse against the oracle"""
    rng = np.random.default_rng(3020)
    wt = _wt(wx, wname)
    for (m, n, dtype, L) in ((64, 512, np.float32, 6), (32, 512, np.float32, 3), (64, 256, np.float64, 5), (16, 256, np.float64, 4)):
        x = np.asfortranarray(rng.standard_normal((m, n, 2)).astype(dtype))
        exp = _stack(oracle.wpt, x, wt.qmf, L)
        got = wx.wptall(x, wt, L)
        assert relerr(got, exp) <= TOL[np.dtype(dtype)], (m, n, L)
        assert relerr(wx.iwptall(exp, wt, L), x) <= 20 * TOL[np.dtype(dtype)], (m, n, L)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("wname", ["haar", "db4", "coif6"])
def test_pyramid2d_levels_below_8x8_in_the_registers_of_a_lane(wx, oracle, wname, dtype):
    """dwtall of square images to depths where the approximation is smaller than 8 x 8: the tree-driven levels stop at 8 x 8 and
    csrc/wx_dwttail.hip (k_dwt2d_tail) finishes the pyramid, one lane per image (dwt/dwt_all.jl:39-54,
    dwt/dwt_one_level.jl:319-354); more images than one wavefront, ragged last wavefront, idwtall back"""
    rng = np.random.default_rng(919)
    wt = _wt(wx, wname)
    tol = TOL[np.dtype(dtype)]
    for m, B in ((16, 70), (64, 67)):
        x = np.asfortranarray(rng.standard_normal((m, m, B)).astype(dtype))
        Lmax = int(np.log2(m))
        for L in range(Lmax - 3 + 1, Lmax + 1):
            tree = wx.maketree(m, m, L, "dwt")
            got = wx.dwtall(x, wt, L)
            for b in (0, 63, 64, B - 1):
                assert relerr(got[:, :, b], oracle.wpt(x[:, :, b], wt.qmf, tree)) <= tol, (wname, m, L, b)
            assert relerr(wx.idwtall(got, wt, L), x) <= 20 * tol, (wname, m, L)
    # rectangular images keep the tree-driven levels
    xr = np.asfortranarray(rng.standard_normal((32, 64, 3)).astype(dtype))
    tr = wx.maketree(32, 64, 5, "dwt")
    assert relerr(wx.dwtall(xr, wt)[:, :, 2], oracle.wpt(xr[:, :, 2], wt.qmf, tr)) <= tol
