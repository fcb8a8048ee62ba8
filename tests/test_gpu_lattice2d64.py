"""64 x 64 images, full quad trees of depth 1 .. 6 in one pass (csrc/wx_lattice_2d64.h: an image per wavefront, Float32 two images per
wavefront) against the oracle per image, the inverse against the oracle's, round trips, in place, odd batches (the pair tail), every
filter length the kernels are built for, and iwpd of a full tree (the inverse reading a slice of the packet table).  Reference: 2-D wpt /
iwpt by level DWT.jl:500-548, 662-710; wptall / iwptall dwt/dwt_all.jl:152-225; iwpd DWT.jl:383-420."""
import numpy as np
import pytest

from helpers import relerr

pytestmark = pytest.mark.gpu

FILTERS = ["haar", "db2", "db3", "db4", "db5", "coif2", "db7", "db8", "db10"]     # db10: 20 taps, the two-pass path


def _tol(dt):
    return 1e-11 if dt == np.float64 else 2e-5


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("L", [1, 2, 3, 4, 5, 6])
def test_full_quad_tree_every_depth(wx, oracle, dt, L):
    rng = np.random.default_rng(100 + L)
    wt = wx.wavelet(wx.WT.db4)
    for B in (1, 2, 5, 64):
        x = np.asfortranarray(rng.standard_normal((64, 64, B)).astype(dt))
        y = wx.wptall(x, wt, L)
        for b in {0, B // 2, B - 1}:
            ref = oracle.wpt(x[:, :, b].astype(np.float64), wt.qmf, L)
            assert relerr(y[:, :, b], ref) <= _tol(dt), (dt, L, B, b)
        assert relerr(wx.iwptall(y, wt, L), x) <= _tol(dt), (dt, L, B)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("name", FILTERS)
def test_every_filter_length(wx, oracle, dt, name):
    rng = np.random.default_rng(7)
    wt = wx.wavelet(getattr(wx.WT, name))
    x = np.asfortranarray(rng.standard_normal((64, 64, 3)).astype(dt))
    for L in (2, 6):
        y = wx.wptall(x, wt, L)
        ref = oracle.wpt(x[:, :, 2].astype(np.float64), wt.qmf, L)
        assert relerr(y[:, :, 2], ref) <= _tol(dt) * 4, (name, L)
        c = np.asfortranarray(rng.standard_normal((64, 64, 3)).astype(dt))
        xi = wx.iwptall(c, wt, L)
        refi = oracle.iwpt(c[:, :, 1].astype(np.float64), wt.qmf, L)
        assert relerr(xi[:, :, 1], refi) <= _tol(dt) * 4, (name, L)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_single_image_calls(wx, oracle, dt):
    rng = np.random.default_rng(11)
    wt = wx.wavelet(wx.WT.db4)
    x = np.asfortranarray(rng.standard_normal((64, 64)).astype(dt))
    for L in (3, 6):
        y = wx.wpt(x, wt, L)
        assert relerr(y, oracle.wpt(x.astype(np.float64), wt.qmf, L)) <= _tol(dt)
        assert relerr(wx.iwpt(y, wt, L), x) <= _tol(dt)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_iwpd_of_a_full_tree_reads_the_deepest_slice(wx, oracle, dt):
    rng = np.random.default_rng(13)
    wt = wx.wavelet(wx.WT.db4)
    x = np.asfortranarray(rng.standard_normal((64, 64)).astype(dt))
    for L in (2, 4):
        xw = wx.wpd(x, wt, L)
        assert relerr(xw[:, :, L], oracle.wpt(x.astype(np.float64), wt.qmf, L)) <= _tol(dt)
        assert relerr(wx.iwpd(xw, wt, L), x) <= _tol(dt), (dt, L)


# ---- the row pass on the lattice kernels (csrc/wx_lattice_rows.h): images of 128, 256, 512 columns ----
@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(128, 128), (256, 256), (512, 512), (64, 256), (32, 128), (1024, 512), (16, 512), (8, 256)])
def test_row_pass_on_the_lattice_kernels(wx, oracle, dt, shape):
    m, n = shape
    rng = np.random.default_rng(m * 7 + n)
    wt = wx.wavelet(wx.WT.db4)
    B = 3
    x = np.asfortranarray(rng.standard_normal((m, n, B)).astype(dt))
    Lm = wx.maxtransformlevels(min(m, n))
    for L in sorted({1, 2, 3, Lm - 1, Lm} - {0}):
        y = wx.wptall(x, wt, L)
        ref = oracle.wpt(x[:, :, 1].astype(np.float64), wt.qmf, L)
        assert relerr(y[:, :, 1], ref) <= _tol(dt) * 2, (shape, L)
        assert relerr(wx.iwptall(y, wt, L), x) <= _tol(dt) * 2, (shape, L)
        c = np.asfortranarray(rng.standard_normal((m, n, 2)).astype(dt))
        refi = oracle.iwpt(c[:, :, 0].astype(np.float64), wt.qmf, L)
        assert relerr(wx.iwptall(c, wt, L)[:, :, 0], refi) <= _tol(dt) * 2, (shape, L)


@pytest.mark.parametrize("name", ["haar", "db2", "db3", "coif2", "db7", "db8", "db10"])
def test_row_pass_filters(wx, oracle, name):
    rng = np.random.default_rng(5)
    wt = wx.wavelet(getattr(wx.WT, name))
    for dt in (np.float64, np.float32):
        x = np.asfortranarray(rng.standard_normal((256, 256, 2)).astype(dt))
        for L in (4, 8):
            y = wx.wptall(x, wt, L)
            assert relerr(y[:, :, 1], oracle.wpt(x[:, :, 1].astype(np.float64), wt.qmf, L)) <= _tol(dt) * 4, (name, L)
            assert relerr(wx.iwptall(y, wt, L), x) <= _tol(dt) * 4, (name, L)


@pytest.mark.parametrize("minsh", ["2", "6"])
def test_row_pass_every_geometry_and_the_strips(minsh):
    """WX_LATROWS_MINSH (read once per process): 2 = the lattice row pass for every column count it is built for (1024 and 512 columns are
    not taken by default for every type), 6 = none (the LDS strips of k_rows_fused): both against the oracle on the same images"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        "sys.path[:0] = [%r, %r]\n"
        "import waveletsext_jl_amd as wx, wx_oracle as O\n"
        "wt = wx.wavelet(wx.WT.db4); rng = np.random.default_rng(3)\n"
        "for dt, tol in ((np.float64, 1e-11), (np.float32, 4e-5)):\n"
        "    for (m, n) in ((64, 1024), (1024, 1024), (16, 512), (512, 512), (32, 256), (64, 128)):\n"
        "        x = np.asfortranarray(rng.standard_normal((m, n, 3)).astype(dt))\n"
        "        for L in (wx.maxtransformlevels(min(m, n)), 4):\n"
        "            y = wx.wptall(x, wt, L)\n"
        "            exp = O.wpt(x[:, :, 2].astype(np.float64), wt.qmf, L)\n"
        "            e1 = np.abs(y[:, :, 2] - exp).max() / np.abs(exp).max()\n"
        "            e2 = np.abs(wx.iwptall(y, wt, L) - x).max() / np.abs(x).max()\n"
        "            assert e1 <= tol and e2 <= tol, (dt, m, n, L, e1, e2)\n" % (root, os.path.join(root, "oracle")))
    env = dict(os.environ, WX_KNOBS="1", WX_LATROWS_MINSH=minsh)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]


# ---- 64 x 64 images along any quad tree in one pass (csrc/wx_lattice_2d64t.h) ----
def _quad_trees(wx, rng):
    from helpers import random_tree_2d
    out = [np.asarray(wx.maketree(64, 64, L, "dwt"), dtype=bool) for L in (1, 2, 3, 5, 6)]
    t = np.asarray(wx.maketree(64, 64, 2, "full"), dtype=bool).copy()      # depth 2 full, one grandchild opened to the bottom
    i = 6 + 7
    while i <= t.size:
        t[i - 1] = True
        i = 4 * i - 2 + 3
    out.append(t)
    for p in (0.3, 0.5, 0.7, 0.9):
        for _ in range(3):
            tr = random_tree_2d(64, 64, rng, p=p)
            tr[0] = True
            out.append(tr)
    return out


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("name", ["haar", "db2", "db3", "db4", "coif2", "db8"])
def test_quad_trees_on_64x64_images(wx, oracle, dt, name):
    rng = np.random.default_rng(64 + len(name))
    wt = wx.wavelet(getattr(wx.WT, name))
    for B in (1, 2, 5):
        x = np.asfortranarray(rng.standard_normal((64, 64, B)).astype(dt))
        for tree in _quad_trees(wx, rng):
            y = wx.wptall(x, wt, tree)
            for b in {0, B - 1}:
                ref = oracle.wpt(x[:, :, b].astype(np.float64), wt.qmf, tree)
                assert relerr(y[:, :, b], ref) <= _tol(dt) * 2, (name, B, b, int(tree.sum()))
            c = np.asfortranarray(rng.standard_normal((64, 64, B)).astype(dt))
            xi = wx.iwptall(c, wt, tree)
            refi = oracle.iwpt(c[:, :, B - 1].astype(np.float64), wt.qmf, tree)
            assert relerr(xi[:, :, B - 1], refi) <= _tol(dt) * 2, (name, B, int(tree.sum()))
            assert relerr(wx.iwptall(y, wt, tree), x) <= _tol(dt) * 2, (name, B, int(tree.sum()))


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_pyramids_of_64x64_images(wx, oracle, dt):
    rng = np.random.default_rng(3)
    wt = wx.wavelet(wx.WT.db4)
    x = np.asfortranarray(rng.standard_normal((64, 64, 7)).astype(dt))
    for L in (1, 2, 3, 4, 5, 6):
        y = wx.dwtall(x, wt, L)
        ref = oracle.wpt(x[:, :, 3].astype(np.float64), wt.qmf, wx.maketree(64, 64, L, "dwt"))
        assert relerr(y[:, :, 3], ref) <= _tol(dt) * 2, (dt, L)
        assert relerr(wx.idwtall(y, wt, L), x) <= _tol(dt) * 2, (dt, L)
    y1 = wx.dwt(x[:, :, 0].copy(order="F"), wt, 4)
    assert relerr(y1, oracle.wpt(x[:, :, 0].astype(np.float64), wt.qmf, wx.maketree(64, 64, 4, "dwt"))) <= _tol(dt) * 2


@pytest.mark.parametrize("dt", ["float64", "float32"])
def test_64x64_in_place_and_a_chip_filling_batch(wx, oracle, dt):
    """the one-pass kernels of 64 x 64 images in place (x is y: a wavefront's loads precede its stores; Float32 pairs: the lone last image
    of an odd batch is both halves of its pair) and on a batch that fills the chip several times over (device arrays)"""
    import torch
    from waveletsext_jl_amd.dwt import Arg, _wpt_batched
    tdt = getattr(torch, dt)
    tol = 1e-11 if dt == "float64" else 2e-5
    wt = wx.wavelet(wx.WT.db4)
    B = 20001                                            # odd: the pair tail; 20001 images = 78 (156) wavefronts per CU
    x = wx.jl_empty((64, 64, B), tdt, "cuda")
    x.normal_()
    x0 = x.clone()
    for L in (2, 6):
        y = wx.wptall(x, wt, L)
        z = x.clone()
        _wpt_batched("wx_wpt", Arg(z), Arg(z), 2, wt, L, None)                  # in place
        assert torch.equal(z, y), (dt, L)
        _wpt_batched("wx_iwpt", Arg(z), Arg(z), 2, wt, L, None)
        assert float((z - x0).abs().max() / x0.abs().max()) <= tol, (dt, L)
        for b in (0, B // 2, B - 1):
            ref = oracle.wpt(x0[:, :, b].cpu().numpy().astype(np.float64), wt.qmf, L)
            assert relerr(y[:, :, b].cpu().numpy(), ref) <= tol, (dt, L, b)
    tree = np.asarray(wx.maketree(64, 64, 3, "dwt"), dtype=bool)
    y = wx.wptall(x, wt, tree)
    for b in (0, B - 1):
        assert relerr(y[:, :, b].cpu().numpy(), oracle.wpt(x0[:, :, b].cpu().numpy().astype(np.float64), wt.qmf, tree)) <= tol, (dt, b)
    assert float((wx.iwptall(y, wt, tree) - x0).abs().max() / x0.abs().max()) <= tol


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("name", ["haar", "db2", "db4"])
def test_wpd_of_64x64_images_in_one_pass(wx, oracle, dt, name):
    """2-D wpd (DWT.jl:164-209: slice 0 = the image, slice l = the full quad tree of depth l) of 64 x 64 images: the image read once, every
    slice written once (csrc/wx_lattice_2d64w.hip); every depth, odd batches, against the oracle's table slice by slice; iwpd of the result"""
    rng = np.random.default_rng(17 + len(name))
    wt = wx.wavelet(getattr(wx.WT, name))
    for B in (1, 2, 5):
        x = np.asfortranarray(rng.standard_normal((64, 64, B)).astype(dt))
        for L in (1, 2, 3, 6):
            tab = wx.wpdall(x, wt, L)
            assert tab.shape == (64, 64, L + 1, B)
            ref = oracle.wpd(x[:, :, B - 1].astype(np.float64), wt.qmf, L)
            for l in range(L + 1):
                assert relerr(tab[:, :, l, B - 1], ref[:, :, l]) <= _tol(dt) * 2, (name, B, L, l)
            assert relerr(wx.iwpdall(tab, wt, L), x) <= _tol(dt) * 2, (name, B, L)
