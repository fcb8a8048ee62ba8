"""GPU parity: 2-D redundant transforms (SWT / ACWT on images), 2-D JBB and 2-D getbasiscoef vs the
CPU oracle (restating SWT.jl:132-158,286-358,474-513,648-758,870-902,1095-1199; ACWT.jl 2-D methods;
bestbasis_tree.jl:182-207; BestBasis.jl:85-110; Utils.jl:127-130)."""
import numpy as np
import pytest

from helpers import TOL, random_tree_2d, relerr

pytestmark = pytest.mark.gpu


def _wt(wx, name):
    return wx.wavelet(getattr(wx.WT, name))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("wname", ["haar", "db4"])
def test_swt2d_families(wx, oracle, wname, dtype):
    rng = np.random.default_rng(4001)
    wt = _wt(wx, wname)
    tol = TOL[np.dtype(dtype)]
    for (n, m) in ((8, 8), (16, 8), (24, 8)):     # both sides divisible by 2^L
        x = np.asfortranarray(rng.standard_normal((n, m)).astype(dtype))
        Lmax = wx.maxtransformlevels(min(n, m))
        for L in sorted({1, Lmax}):
            sd = wx.sdwt(x, wt, L); sp = wx.swpt(x, wt, L); sw = wx.swpd(x, wt, L)
            assert relerr(sd, oracle.red2d_fwd("dwt", x, wt.qmf, L)) <= tol
            assert relerr(sp, oracle.red2d_fwd("wpt", x, wt.qmf, L)) <= tol
            assert relerr(sw, oracle.red2d_fwd("wpd", x, wt.qmf, L)) <= tol
            nl = 1 << (2 * L)
            assert (sp == sw[:, :, sw.shape[2] - nl:]).all()                 # test/transforms.jl:111-112
            sms = [None, 1] + ([3] if L >= 2 else [])
            for sm in sms:
                e_sd = oracle.red2d_fwd("dwt", x, wt.qmf, L)
                got = wx.isdwt(e_sd, wt, sm)
                assert relerr(got, oracle.red2d_inv("dwt", e_sd, wt.qmf, sm=sm)) <= tol, (n, m, L, sm)
                assert relerr(got, x) <= 20 * tol
            for sm in [None, 0, (1 << L) - 1]:
                e_sp = oracle.red2d_fwd("wpt", x, wt.qmf, L)
                got = wx.iswpt(e_sp, wt, sm)
                assert relerr(got, oracle.red2d_inv("wpt", e_sp, wt.qmf, sm=sm)) <= tol, (n, m, L, sm)
                assert relerr(got, x) <= 20 * tol
        if n == m or True:
            e_sw = oracle.red2d_fwd("wpd", x, wt.qmf, Lmax)
            trees = [None, 1, wx.maketree(n, m, Lmax, "dwt"), random_tree_2d(n, m, rng), random_tree_2d(n, m, rng, 0.9)]
            for arg in trees:
                for sm in (None, 1 if Lmax >= 1 else None):
                    got = wx.iswpd(e_sw, wt, arg, sm)
                    assert relerr(got, oracle.red2d_inv("wpd", e_sw, wt.qmf, arg, sm)) <= tol
                    assert relerr(got, x) <= 20 * tol
    # batch drivers (swt_all.jl) == per-image results
    X = np.asfortranarray(rng.standard_normal((8, 8, 3)).astype(dtype))
    got = wx.swpdall(X, wt, 2)
    for i in range(3):
        assert relerr(got[..., i], oracle.red2d_fwd("wpd", X[..., i], wt.qmf, 2)) <= tol
    assert relerr(wx.iswpdall(got, wt, 2), X) <= 20 * tol
    assert relerr(wx.isdwtall(wx.sdwtall(X, wt, 2), wt, 3), X) <= 20 * tol


@pytest.mark.parametrize("wname", ["haar", "db4", "coif6"])
def test_acwt2d_families(wx, oracle, wname):
    rng = np.random.default_rng(4002)
    wt = _wt(wx, wname)
    for (n, m) in ((8, 8), (16, 8)):
        x = np.asfortranarray(rng.standard_normal((n, m)))
        Lmax = wx.maxtransformlevels(min(n, m))
        for L in sorted({1, Lmax}):
            ad = wx.acdwt(x, wt, L); ap = wx.acwpt(x, wt, L); aw = wx.acwpd(x, wt, L)
            assert relerr(ad, oracle.red2d_fwd("dwt", x, wt.qmf, L, ac=True)) <= 1e-10
            assert relerr(ap, oracle.red2d_fwd("wpt", x, wt.qmf, L, ac=True)) <= 1e-10
            assert relerr(aw, oracle.red2d_fwd("wpd", x, wt.qmf, L, ac=True)) <= 1e-10
            assert (ap == aw[:, :, aw.shape[2] - (1 << (2 * L)):]).all()     # test/transforms.jl:169-170
            assert relerr(wx.iacdwt(ad), x) <= 1e-10
            assert relerr(wx.iacwpt(ap, wt), x) <= 1e-10
            assert relerr(wx.iacwpd(aw, L), x) <= 1e-10
            assert (wx.iacdwt(ad) == oracle.red2d_inv("dwt", ad, ac=True)).all()
            assert (wx.iacwpt(ap) == oracle.red2d_inv("wpt", ap, ac=True)).all()
        aw = wx.acwpd(x, wt)
        for tree in (wx.maketree(n, m, Lmax, "dwt"), random_tree_2d(n, m, rng), random_tree_2d(n, m, rng, 0.9)):
            got = wx.iacwpd(aw, wt, tree)
            assert (got == oracle.red2d_inv("wpd", aw, None, tree, ac=True)).all()
            assert relerr(got, x) <= 1e-10


def test_jbb2d_and_getbasiscoef2d(wx, oracle):
    rng = np.random.default_rng(4003)
    wt = _wt(wx, "db4")
    n = m = 16
    N = 12
    base = np.outer(np.sin(np.arange(n) * 0.7), np.cos(np.arange(m) * 0.3))
    Y = np.asfortranarray(base[:, :, None] * (1 + 0.3 * rng.standard_normal((1, 1, N))) + 0.2 * rng.standard_normal((n, m, N)))
    yw = wx.wpdall(Y, wt)                                            # (16,16,5,N)
    got = wx.tree_costs(yw, wx.JBB())
    assert relerr(got, oracle.tree_costs_jbb2d(yw)) <= 1e-9
    assert (wx.bestbasistree(yw) == oracle.bestbasistree_jbb2d(yw)).all()
    assert (wx.bestbasistree(yw, wx.JBB(wx.NormCost(1), False)) == oracle.bestbasistree_jbb2d(yw, cost="norm")).all()
    ysw = wx.swpdall(Y, wt, 2)                                       # (16,16,21,N)
    got = wx.tree_costs(ysw, wx.JBB(redundant=True))
    assert relerr(got, oracle.tree_costs_jbb2d(ysw, redundant=True)) <= 1e-9
    tree = wx.bestbasistree(ysw, wx.JBB(redundant=True))
    assert (tree == oracle.bestbasistree_jbb2d(ysw, redundant=True)).all()
    assert wx.isvalidtree(np.zeros((n, m)), tree)
    # getbasiscoef / getbasiscoefall for 2-D tables
    t2 = random_tree_2d(n, m, rng)
    gb = wx.getbasiscoefall(yw, t2)
    for i in range(N):
        assert (gb[..., i] == oracle.getbasiscoef2d(yw[..., i], t2)).all()
    assert (wx.getbasiscoef(yw[..., 0], t2) == gb[..., 0]).all()
    assert relerr(wx.wptall(Y, wt, t2), gb) <= 1e-10                  # wpt by tree == gathered leaves


@pytest.mark.parametrize("wname", ["haar", "db4", "db8"])
def test_one_pass_forward_levels_whole_rows_and_column_tiles(wx, oracle, wname):
    """the one-pass forward level (k_red2d_fwd_fused): strips holding whole rows (up to 256 columns), column tiles
    with halo for wider images, rows not divisible by the preferred strip height, all three containers, both
    families, Float64 and Float32 -- against the oracle (same arithmetic as the two-pass level: exact order of sums)"""
    rng = np.random.default_rng(4100)
    wt = _wt(wx, wname)
    for (m, n, L, dtype) in ((64, 256, 3, np.float64), (32, 512, 2, np.float64), (16, 1024, 3, np.float32),
                             (48, 640, 2, np.float64), (128, 128, 3, np.float32)):
        tol = TOL[np.dtype(dtype)]
        x = np.asfortranarray(rng.standard_normal((m, n)).astype(dtype))
        for kind, fn in (("dwt", wx.sdwt), ("wpt", wx.swpt), ("wpd", wx.swpd)):
            assert relerr(fn(x, wt, L), oracle.red2d_fwd(kind, x, wt.qmf, L)) <= tol, (m, n, L, kind)
        if dtype == np.float64:
            for kind, fn in (("dwt", wx.acdwt), ("wpt", wx.acwpt), ("wpd", wx.acwpd)):
                assert relerr(fn(x, wt, L), oracle.red2d_fwd(kind, x, wt.qmf, L, ac=True)) <= tol, (m, n, L, kind, "ac")
        xb = np.asfortranarray(rng.standard_normal((m, n, 3)).astype(dtype))
        got = wx.swptall(xb, wt, L)
        assert relerr(got[..., 2], oracle.red2d_fwd("wpt", np.asfortranarray(xb[..., 2]), wt.qmf, L)) <= tol
        assert relerr(wx.iswptall(got, wt), xb) <= 50 * tol
