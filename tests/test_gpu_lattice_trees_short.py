"""Tree-driven wpt / iwpt / dwt / idwt on SHORT signals (64 ... 512 samples) through the masked lattice kernels
(csrc/wx_lattice_tree_s.h over wx_lattice_tree_sc.h: 8 ... 64 signals per wavefront) against the oracle: pyramids of every depth, one-sided
and zig-zag paths, a full tree with one leaf opened, fuzzed random trees; Float64 and Float32; batches below, at and off the multiples of
the signals per wavefront (the tail wavefront re-does signals; fewer signals than one wavefront take the older kernels).

Reference behaviour: Wavelets.jl's wpt / iwpt with a tree::BitVector as called at dwt/dwt_all.jl:152-166, 210-225; dwtall / idwtall
(dwt/dwt_all.jl:39-110) are the tree of maketree(:dwt)."""
import numpy as np
import pytest

from helpers import random_tree_1d, relerr

pytestmark = pytest.mark.gpu


def _tol(dt):
    return 1e-10 if dt == np.float64 else 1e-5


def _trees(wx, n, rng, count):
    Lmax = wx.maxtransformlevels(n)
    out = [wx.maketree(n, L, "dwt") for L in sorted({1, 2, 3, Lmax - 1, Lmax})]
    t = np.zeros(n - 1, dtype=bool)                    # the detail-side mirror of the pyramid
    i = 1
    while i <= n - 1:
        t[i - 1] = True
        i = 2 * i + 1
    out.append(t)
    t = np.zeros(n - 1, dtype=bool)                    # a zig-zag path to the bottom
    i, k = 1, 0
    while i <= n - 1:
        t[i - 1] = True
        i = 2 * i + (k & 1)
        k += 1
    out.append(t)
    t = np.array(wx.maketree(n, 3, "full"), dtype=bool).copy()   # a full tree of depth 3 with one leaf opened to the bottom
    i = 8 + 5
    while i <= n - 1:
        t[i - 1] = True
        i = 2 * i
    out.append(t)
    for p in (0.3, 0.5, 0.7, 0.85, 0.95):
        for _ in range(max(1, count // 5)):
            tr = random_tree_1d(n, rng, p)
            tr[0] = True
            out.append(tr)
    return out


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("n", [64, 128, 256, 512])
@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db4"])
def test_short_trees_match_the_oracle(wx, oracle, dt, n, wname):
    rng = np.random.default_rng(n + len(wname))
    wt = wx.wavelet(getattr(wx.WT, wname))
    per = 4096 // n
    for B in (per, 2 * per + 3, 5 * per - 1):
        x = np.asfortranarray(rng.standard_normal((n, B)).astype(dt))
        x64 = x.astype(np.float64)
        for tree in _trees(wx, n, rng, 5):
            got = wx.wptall(x, wt, tree)
            exp = oracle.wptall(x64, wt.qmf, tree)
            assert relerr(got, exp) <= _tol(dt), (n, wname, B, int(tree.sum()))
            back = wx.iwptall(exp.astype(dt), wt, tree)
            assert relerr(back, oracle.iwptall(exp.astype(dt).astype(np.float64), wt.qmf, tree)) <= _tol(dt), (n, wname, B, int(tree.sum()))
            assert relerr(wx.iwptall(got, wt, tree), x) <= _tol(dt) * 2, (n, wname, B, int(tree.sum()))


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("n,count", [(64, 100), (256, 100)])
def test_short_tree_fuzz_db4(wx, oracle, dt, n, count):
    rng = np.random.default_rng(n * 3 + 1)
    wt = wx.wavelet(wx.WT.db4)
    B = 3 * (4096 // n) + 1
    x = np.asfortranarray(rng.standard_normal((n, B)).astype(dt))
    xw_full = oracle.wpdall(x.astype(np.float64), wt.qmf)
    worst = 0.0
    for tree in _trees(wx, n, rng, count):
        exp = np.stack([oracle.getbasiscoef(xw_full[:, :, b], tree) for b in (0, B // 2, B - 1)], axis=1)
        got = wx.wptall(x, wt, tree)
        e = relerr(got[:, [0, B // 2, B - 1]], exp)
        worst = max(worst, e)
        assert e <= _tol(dt), int(tree.sum())
        assert relerr(wx.iwptall(got, wt, tree), x) <= _tol(dt) * 2
    assert worst > 0.0


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("n", [64, 128, 256, 512])
def test_short_pyramids_dwtall_idwtall(wx, oracle, dt, n):
    rng = np.random.default_rng(n + 9)
    wt = wx.wavelet(wx.WT.db4)
    B = 200
    x = np.asfortranarray(rng.standard_normal((n, B)).astype(dt))
    for L in sorted({1, 4, wx.maxtransformlevels(n)}):
        y = wx.dwtall(x, wt, L)
        exp = oracle.wptall(x.astype(np.float64), wt.qmf, wx.maketree(n, L, "dwt"))
        assert relerr(y, exp) <= _tol(dt), (n, L)
        assert relerr(wx.idwtall(y, wt, L), x) <= _tol(dt) * 2, (n, L)


def test_few_signals_and_long_filters_fall_back(wx, oracle):
    """fewer signals than one wavefront holds, and filters beyond 16 taps: the older kernels, same answers"""
    rng = np.random.default_rng(77)
    for n, B, wname in ((64, 5, "db4"), (256, 3, "db2"), (128, 40, "db10"), (512, 16, "db9")):
        wt = wx.wavelet(getattr(wx.WT, wname))
        x = np.asfortranarray(rng.standard_normal((n, B)))
        tree = random_tree_1d(n, rng, 0.7)
        tree[0] = True
        got = wx.wptall(x, wt, tree)
        assert relerr(got, oracle.wptall(x, wt.qmf, tree)) <= 1e-10, (n, B, wname)
        assert relerr(wx.iwptall(got, wt, tree), x) <= 1e-10, (n, B, wname)


@pytest.mark.parametrize("n", [64, 128, 256, 512])
def test_short_tree_iwpd_reads_the_packet_table(wx, oracle, n):
    """iwpd by tree (DWT.jl:340-351): the leaves of depth l sit in slice l of the (n, L+1, B) table"""
    rng = np.random.default_rng(n + 2)
    wt = wx.wavelet(wx.WT.db4)
    B = 2 * (4096 // n) + 1
    x = np.asfortranarray(rng.standard_normal((n, B)))
    xw = oracle.wpdall(x, wt.qmf)
    for tree in _trees(wx, n, rng, 5):
        got = wx.iwpdall(xw, wt, tree)
        assert relerr(got, x) <= 1e-10, int(tree.sum())


@pytest.mark.parametrize("n", [64, 128, 256])
@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db4", "coif2", "db8"])
def test_short_float32_wpd_on_the_lattice_kernel(wx, oracle, n, wname):
    """wpdall of short Float32 signals (csrc/wx_lattice_sgw.hip: the interleaved lattice wpd kernel with Float32 at the two ends) against the
    oracle's packet table (DWT.jl:164-209 via dwt/dwt_all.jl:262-281), every depth, batches at and off the multiples of a wavefront's signals"""
    rng = np.random.default_rng(n + len(wname))
    wt = wx.wavelet(getattr(wx.WT, wname))
    per = 4096 // n
    for B in (per, 3 * per + 5):
        x = np.asfortranarray(rng.standard_normal((n, B)).astype(np.float32))
        for L in sorted({1, 3, wx.maxtransformlevels(n)}):
            got = wx.wpdall(x, wt, L)
            exp = oracle.wpdall(x.astype(np.float64), wt.qmf, L)
            assert got.dtype == np.float32 and got.shape == exp.shape
            assert relerr(got, exp) <= 1e-5, (n, wname, B, L)
            assert relerr(wx.iwpdall(got, wt, L), x) <= 2e-5, (n, wname, B, L)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("wname", ["db5", "coif2", "db7", "db8"])
def test_short_trees_longer_filters(wx, oracle, dt, wname):
    """10 ... 16 taps (5 ... 8 rotation stages) on the same kernels"""
    rng = np.random.default_rng(len(wname) + 40)
    wt = wx.wavelet(getattr(wx.WT, wname))
    for n in (64, 128, 256, 512):
        B = 2 * (4096 // n) + 1
        x = np.asfortranarray(rng.standard_normal((n, B)).astype(dt))
        for tree in _trees(wx, n, rng, 5)[::2]:
            got = wx.wptall(x, wt, tree)
            exp = oracle.wptall(x.astype(np.float64), wt.qmf, tree)
            assert relerr(got, exp) <= _tol(dt) * 2, (n, wname, int(tree.sum()))
            assert relerr(wx.iwptall(got, wt, tree), x) <= _tol(dt) * 4, (n, wname, int(tree.sum()))
        for L in (1, 3, wx.maxtransformlevels(n)):              # FULL trees with these filters take the same kernels
            got = wx.wptall(x, wt, L)
            exp = oracle.wptall(x.astype(np.float64), wt.qmf, L)
            assert relerr(got, exp) <= _tol(dt) * 2, (n, wname, L)
            assert relerr(wx.iwptall(got, wt, L), x) <= _tol(dt) * 4, (n, wname, L)


@pytest.mark.parametrize("wname", ["db5", "coif2", "db7", "db8"])
def test_short_float64_wpd_and_full_trees_longer_filters(wx, oracle, wname):
    """10 ... 16 taps on the interleaved lattice kernels of 64 ... 512-sample Float64 signals (csrc/wx_lattice_sgw_b.hip; full trees through the masked tree kernels as a tree of ones):
    wpdall against the oracle's table, iwpdall of the full tree (the deepest slice), wptall / iwptall of full trees"""
    rng = np.random.default_rng(len(wname) + 7)
    wt = wx.wavelet(getattr(wx.WT, wname))
    for n in (64, 128, 256, 512):
        B = 2 * (4096 // n) + 3
        x = np.asfortranarray(rng.standard_normal((n, B)))
        for L in sorted({1, 3, wx.maxtransformlevels(n)}):
            tab = wx.wpdall(x, wt, L)
            exp = oracle.wpdall(x, wt.qmf, L)
            assert relerr(tab, exp) <= 1e-10, (n, wname, L)
            assert relerr(wx.iwpdall(exp, wt, L), x) <= 1e-10, (n, wname, L)
            y = wx.wptall(x, wt, L)
            assert relerr(y, exp[:, L, :]) <= 1e-10, (n, wname, L)
            assert relerr(wx.iwptall(y, wt, L), x) <= 1e-10, (n, wname, L)
