"""Tree-driven wpt / iwpt / iwpd on the lattice kernels (csrc/wx_lattice_tree.h) against the oracle.

Reference behaviour: Wavelets.jl's wpt / iwpt with a tree::BitVector as called at dwt/dwt_all.jl:152-166, 210-225,
DWT.jl:340-351 (iwpd by tree = getbasiscoef + iwpt, Utils.jl:101-134).  Float64, tolerance 1e-10 relative; the routing of
every coefficient is exact, so wpt(x, tree) must equal getbasiscoef(wpd(x), tree) of the device's own packet table to
rounding (1e-13).
"""
import numpy as np
import pytest

from helpers import random_tree_1d, relerr

pytestmark = pytest.mark.gpu

TOL = 1e-10


def _wt(wx, name):
    return wx.wavelet(getattr(wx.WT, name))


def _depth(tree):
    nz = np.flatnonzero(tree)
    return 0 if nz.size == 0 else int(np.floor(np.log2(nz.max() + 1))) + 1


def _trees(wx, n, rng, count):
    Lmax = wx.maxtransformlevels(n)
    out = [wx.maketree(n, Lmax, "dwt"), wx.maketree(n, 3, "dwt"), wx.maketree(n, 7, "dwt")]
    # one leaf at every depth on the detail side (mirror of the pyramid), a tree with a single deep path in the middle
    t = np.zeros(n - 1, dtype=bool)
    i = 1
    while i <= n - 1:
        t[i - 1] = True
        i = 2 * i + 1
    out.append(t)
    t = np.zeros(n - 1, dtype=bool)
    i, k = 1, 0
    while i <= n - 1:
        t[i - 1] = True
        i = 2 * i + (k & 1)
        k += 1
    out.append(t)
    # a full tree of depth 4 with one leaf opened to the bottom
    t = np.array(wx.maketree(n, 4, "full"), dtype=bool).copy()
    i = 16 + 5
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


@pytest.mark.parametrize("n", [4096, 2048, 1024])
@pytest.mark.parametrize("wname", ["haar", "db2", "db4", "db8", "coif6"])
def test_tree_wpt_iwpt_match_oracle(wx, oracle, n, wname):
    rng = np.random.default_rng(4096 + n + len(wname))
    wt = _wt(wx, wname)
    B = 5                                             # not a multiple of the signals per wavefront (tail wavefront)
    x = np.asfortranarray(rng.standard_normal((n, B)))
    for tree in _trees(wx, n, rng, 10):
        got = wx.wptall(x, wt, tree)
        exp = oracle.wptall(x, wt.qmf, tree)
        assert relerr(got, exp) <= TOL, (n, wname, int(tree.sum()))
        back = wx.iwptall(exp, wt, tree)
        assert relerr(back, x) <= TOL, (n, wname, int(tree.sum()))


@pytest.mark.parametrize("n,count", [(4096, 200), (1024, 200)])
def test_tree_fuzz_db4(wx, oracle, n, count):
    """>= 200 fuzzed trees per length: forward against the oracle, inverse against the signal"""
    rng = np.random.default_rng(n)
    wt = _wt(wx, "db4")
    x = np.asfortranarray(rng.standard_normal((n, 4)))
    xw_full = oracle.wpdall(x, wt.qmf)                 # (n, L+1, B): every leaf of every tree is an entry of this table
    worst = 0.0
    for tree in _trees(wx, n, rng, count):
        exp = np.stack([oracle.getbasiscoef(xw_full[:, :, b], tree) for b in range(x.shape[1])], axis=1)
        got = wx.wptall(x, wt, tree)
        e = relerr(got, exp)
        worst = max(worst, e)
        assert e <= TOL, int(tree.sum())
        assert relerr(wx.iwptall(got, wt, tree), x) <= TOL
    assert worst > 0.0                                # the comparison is not vacuous


@pytest.mark.parametrize("n", [4096, 2048, 1024])
def test_tree_wpt_equals_gather_of_the_device_table(wx, n):
    """the tree only routes: the leaves written by the tree-driven kernel are the entries of the device's own wpd table"""
    rng = np.random.default_rng(n + 1)
    wt = _wt(wx, "db4")
    x = np.asfortranarray(rng.standard_normal((n, 8)))
    xw = wx.wpdall(x, wt)
    for k, tree in enumerate(_trees(wx, n, rng, 10)):
        got = wx.wptall(x, wt, tree)
        gath = wx.getbasiscoefall(xw, tree)
        # same values to rounding, not the same bits: the table's kernel defers every gain to its emission, the tree-driven
        # one normalises the levels below depth 6 - SH one by one (and the deep levels of the pyramids run in the direct form
        # of wx_dwttail.hip); what must be exact is the ROUTING -- a misplaced coefficient is an O(1) error
        assert relerr(got, gath) <= 1e-13, (k, int(tree.sum()))


@pytest.mark.parametrize("n", [4096, 1024])
def test_tree_iwpd_reads_the_packet_table(wx, oracle, n):
    rng = np.random.default_rng(n + 2)
    wt = _wt(wx, "db4")
    x = np.asfortranarray(rng.standard_normal((n, 6)))
    xw = oracle.wpdall(x, wt.qmf)
    for tree in _trees(wx, n, rng, 10):
        got = wx.iwpdall(xw, wt, tree)
        assert relerr(got, x) <= TOL, int(tree.sum())
        assert relerr(got, oracle.iwpdall(xw, wt.qmf, tree)) <= TOL


def test_tree_large_batch_and_device_arrays(wx, oracle):
    """device-resident batch of the benchmark's shape (fewer signals): every signal is transformed, not just the first ones"""
    import torch
    rng = np.random.default_rng(5)
    n, B = 4096, 3000
    wt = _wt(wx, "db4")
    tree = random_tree_1d(n, rng, 0.7)
    tree[0] = True
    x = wx.jl_empty((n, B), torch.float64, "cuda")
    x.normal_()
    y = wx.wptall(x, wt, tree)
    xr = wx.iwptall(y, wt, tree)
    assert float((xr - x).abs().max() / x.abs().max()) <= TOL
    idx = [0, 1, B // 2, B - 2, B - 1]
    exp = oracle.wptall(np.asfortranarray(x[:, idx].cpu().numpy()), wt.qmf, tree)
    assert relerr(y[:, idx].cpu().numpy(), exp) <= TOL


@pytest.mark.parametrize("n", [4096, 2048, 1024])
@pytest.mark.parametrize("wname", ["haar", "db4"])
def test_tree_denoise_threshold_rides_on_the_absorbed_leaves(wx, oracle, n, wname):
    """denoiseall(x, :wpt, wt; tree) (Denoising.jl:527, 651-712): per-signal MAD, threshold, iwpt along the tree -- the threshold
    is applied to the leaves as the tree-driven lattice inverse takes them in (one threshold per signal, several signals per
    wavefront below 4096 samples); hard and soft rules, regular and undersmooth, device-resident and host arrays"""
    rng = np.random.default_rng(n + 7)
    wt = _wt(wx, wname)
    B = 7
    t = np.linspace(0, 1, n)
    x0 = np.asfortranarray(np.stack([np.sin(2 * np.pi * (3 + b) * t) * (1 + b) for b in range(B)], axis=1))
    x = x0 + 0.2 * rng.standard_normal((n, B)) * (1 + np.arange(B))[None, :]          # a different noise level per signal
    for tree in (_trees(wx, n, rng, 5)[3:6] + _trees(wx, n, rng, 5)[-2:]):
        xw = oracle.wptall(x, wt.qmf, tree)
        for thname, TH in (("hard", wx.HardTH), ("soft", wx.SoftTH)):
            dnt = wx.VisuShrink(n, TH())
            for smooth in ("regular", "undersmooth"):
                exp = np.stack([oracle.denoise(np.asfortranarray(xw[:, i]), "wpt", wt.qmf, tree=tree, th=thname, t=dnt.t, smooth=smooth)
                                for i in range(B)], axis=1)
                got = wx.to_numpy(wx.denoiseall(wx.to_device(xw), "wpt", wt, tree=tree, dnt=dnt, smooth=smooth))
                assert relerr(got, exp) <= 1e-9, (n, wname, thname, smooth, int(tree.sum()))
                got_h = wx.denoiseall(xw, "wpt", wt, tree=tree, dnt=dnt, smooth=smooth)
                assert relerr(got_h, exp) <= 1e-9, (n, wname, thname, smooth)


@pytest.mark.parametrize("wname", ["haar", "db2", "db4", "db8", "coif6"])
def test_pyramid_idwt_4096_takes_the_rebuilt_head(wx, oracle, wname):
    """idwt / denoise(:dwt) of 4096-sample signals: the levels below 64 samples run lane-locally (wx_dwttail.hip) and the
    tree-driven lattice inverse reads the 64 rebuilt samples in place of positions 0 .. 63 (wx_lattice_tree_sc.h); every
    pyramid depth that has a tail (L = 7 .. 12), batch sizes around a wavefront's worth of tail signals"""
    rng = np.random.default_rng(77)
    wt = _wt(wx, wname)
    n = 4096
    for L, B in ((12, 5), (9, 64), (7, 67)):
        x = np.asfortranarray(rng.standard_normal((n, B)))
        tree = np.asarray(wx.maketree(n, L, "dwt"))
        xw = oracle.wptall(x, wt.qmf, tree)
        assert relerr(wx.wptall(x, wt, tree), xw) <= TOL, (wname, L)
        assert relerr(wx.iwptall(xw, wt, tree), x) <= TOL, (wname, L)
        assert relerr(wx.to_numpy(wx.iwptall(wx.to_device(xw), wt, tree)), x) <= TOL, (wname, L)
    # denoise(:dwt): the threshold on the deep coefficients is applied by the tail, on the others by the lattice inverse's loads
    B = 6
    t = np.linspace(0, 1, n)
    x = np.asfortranarray(np.stack([np.sin(2 * np.pi * (2 + b) * t) for b in range(B)], axis=1) + 0.3 * rng.standard_normal((n, B)))
    for L in (12, 8):
        tree = np.asarray(wx.maketree(n, L, "dwt"))
        xw = oracle.wptall(x, wt.qmf, tree)
        for thname, TH in (("hard", wx.HardTH), ("soft", wx.SoftTH)):
            dnt = wx.VisuShrink(n, TH())
            exp = np.stack([oracle.denoise(np.asfortranarray(xw[:, i]), "dwt", wt.qmf, L=L, th=thname, t=dnt.t, smooth="regular")
                            for i in range(B)], axis=1)
            got = wx.denoiseall(xw, "dwt", wt, L=L, dnt=dnt, smooth="regular")
            assert relerr(got, exp) <= 1e-9, (wname, L, thname)
