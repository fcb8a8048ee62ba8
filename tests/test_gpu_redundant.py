"""GPU parity: stationary (SWT) and autocorrelation (ACWT) families and the JBB reduction vs the
CPU oracle.  Float64 tolerance 1e-10 relative, Float32 1e-5; tree bits compared with ==."""
import numpy as np
import pytest

from helpers import TOL, random_tree_1d, relerr

pytestmark = pytest.mark.gpu


def _wt(wx, name):
    return wx.wavelet(getattr(wx.WT, name))


def _stack(fn, X, *a):
    return np.asfortranarray(np.stack([fn(np.asfortranarray(X[..., i]), *a) for i in range(X.shape[-1])], axis=-1))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("wname", ["haar", "db2", "db4", "coif6"])
def test_swt_forward_families(wx, oracle, wname, dtype):
    rng = np.random.default_rng(2003)
    wt = _wt(wx, wname)
    tol = TOL[np.dtype(dtype)]
    for n, B in ((4, 2), (16, 3), (64, 5), (24, 2), (512, 2)):
        x = np.asfortranarray(rng.standard_normal((n, B)).astype(dtype))
        Lmax = wx.maxtransformlevels(n)
        for L in sorted({1, Lmax}):
            assert relerr(wx.sdwtall(x, wt, L), _stack(oracle.sdwt, x, wt.qmf, L)) <= tol
            assert relerr(wx.swptall(x, wt, L), _stack(oracle.swpt, x, wt.qmf, L)) <= tol
            assert relerr(wx.swpdall(x, wt, L), _stack(oracle.swpd, x, wt.qmf, L)) <= tol
        # swpt == leaves of swpd (test/transforms.jl:95-96).  The reference gets bit equality because both
        # run the same per-level loop; here swpt fuses levels with composite taps, so the identity holds to
        # rounding (and both match the oracle within the parity tolerance above)
        L = Lmax
        assert relerr(wx.swpt(x[:, 0], wt), wx.swpd(x[:, 0], wt)[:, (1 << L) - 1:]) <= (1e-13 if dtype == np.float64 else 1e-5)
        wx.set_force_generic(1)
        try:                                    # the per-level path keeps the identity exact
            assert (wx.swpt(x[:, 0], wt) == wx.swpd(x[:, 0], wt)[:, (1 << L) - 1:]).all()
        finally:
            wx.set_force_generic(0)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("wname", ["haar", "db4", "db8"])
def test_swt_inverse_families(wx, oracle, wname, dtype):
    rng = np.random.default_rng(2004)
    wt = _wt(wx, wname)
    tol = TOL[np.dtype(dtype)]
    for n, B in ((8, 3), (64, 4), (256, 2)):
        x = np.asfortranarray(rng.standard_normal((n, B)).astype(dtype))
        Lmax = wx.maxtransformlevels(n)
        for L in sorted({1, 3, Lmax}):
            sd = _stack(oracle.sdwt, x, wt.qmf, L)
            sp = _stack(oracle.swpt, x, wt.qmf, L)
            for sm in [None, 1, (1 << L) - 1] + ([5] if L >= 3 else []):
                got = wx.isdwtall(sd, wt, sm)
                assert relerr(got, _stack(oracle.isdwt, sd, wt.qmf, sm)) <= tol, (n, L, sm)
                assert relerr(got, x) <= 20 * tol
            for sm in [None, 0, 1, (1 << L) - 1]:
                got = wx.iswptall(sp, wt, sm)
                assert relerr(got, _stack(oracle.iswpt, sp, wt.qmf, sm)) <= tol, (n, L, sm)
                assert relerr(got, x) <= 20 * tol
        sw = _stack(oracle.swpd, x, wt.qmf, Lmax)
        trees = [None, 2, wx.maketree(n, Lmax, "dwt"), random_tree_1d(n, rng), random_tree_1d(n, rng, 0.5)]
        for arg in trees:
            for sm in (None, 0, 3):
                got = wx.iswpdall(sw, wt, arg, sm)
                exp = np.asfortranarray(np.stack([oracle.iswpd(sw[:, :, i], wt.qmf, arg, sm) for i in range(B)], axis=-1))
                assert relerr(got, exp) <= tol, (n, arg if not isinstance(arg, np.ndarray) else "tree", sm)
                assert relerr(got, x) <= 20 * tol
        assert relerr(wx.iswpd(sw[:, :, 0], wt, trees[3], 3), x[:, 0]) <= 20 * tol
    with pytest.raises(AssertionError):
        wx.isdwt(np.zeros((8, 4)), wt, 0)                    # SWT.jl:266: log2(0) = -Inf
    with pytest.raises(AssertionError):
        wx.isdwt(np.zeros((8, 4)), wt, 8)
    with pytest.raises(AssertionError):
        wx.iswpt(np.zeros((8, 8)), wt, 8)                    # main2depthshift assert


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("wname", ["haar", "db4", "db8"])
def test_iswpt_fused_passes(wx, oracle, wname, dtype):
    """average-based iswpt runs two (three for F <= 4) levels per pass where the residue-class tiles fit the
    LDS: arbitrary (non-transform) coefficient tables, non-dyadic lengths, mixed fused/per-level schedules vs
    the oracle and vs the per-level device path"""
    rng = np.random.default_rng(2010)
    wt = _wt(wx, wname)
    tol = TOL[np.dtype(dtype)]
    for n, L, B in ((384, 7, 2), (1024, 10, 2), (2048, 6, 3), (640, 5, 2)):
        sp = np.asfortranarray(rng.standard_normal((n, 1 << L, B)).astype(dtype))
        got = wx.iswptall(sp, wt)
        assert relerr(got, _stack(oracle.iswpt, sp, wt.qmf, None)) <= tol, (n, L)
        wx.set_force_generic(1)
        try:
            ref = wx.iswptall(sp, wt)
        finally:
            wx.set_force_generic(0)
        assert relerr(got, ref) <= (1e-13 if dtype == np.float64 else 1e-5)


@pytest.mark.parametrize("wname", ["haar", "db4", "coif6"])
def test_acwt_families(wx, oracle, wname):
    rng = np.random.default_rng(2005)
    wt = _wt(wx, wname)
    for n, B in ((8, 3), (64, 4), (24, 2), (256, 2)):
        x = np.asfortranarray(rng.standard_normal((n, B)))
        Lmax = wx.maxtransformlevels(n)
        for L in sorted({1, 2, Lmax}):
            ad = wx.acdwtall(x, wt, L)
            ap = wx.acwptall(x, wt, L)
            aw = wx.acwpdall(x, wt, L)
            assert relerr(ad, _stack(oracle.acdwt, x, wt.qmf, L)) <= 1e-10
            assert relerr(ap, _stack(oracle.acwpt, x, wt.qmf, L)) <= 1e-10
            assert relerr(aw, _stack(oracle.acwpd, x, wt.qmf, L)) <= 1e-10
            assert (ap == aw[:, (1 << L) - 1:(1 << (L + 1)) - 1, :]).all()       # test/transforms.jl:153-154
            assert relerr(wx.iacdwtall(ad), x) <= 1e-10
            assert relerr(wx.iacwptall(ap), x) <= 1e-10
            if wx.isdyadic(n):
                assert relerr(wx.iacwpdall(aw, L), x) <= 1e-10
                assert relerr(wx.iacwpdall(aw, wt, L), x) <= 1e-10
            else:                                   # maketree(n, L, :full) asserts isdyadic(n) (Wavelets.jl)
                with pytest.raises(AssertionError):
                    wx.iacwpdall(aw, L)
            # inverses are bit-compatible with the oracle (same pairwise order)
            assert (wx.iacdwtall(ad) == _stack(oracle.iacdwt, ad)).all()
            assert (wx.iacwptall(ap) == _stack(oracle.iacwpt, ap)).all()
        if wx.isdyadic(n):
            aw = wx.acwpdall(x, wt)
            for tree in (wx.maketree(n, Lmax, "dwt"), random_tree_1d(n, rng), random_tree_1d(n, rng, 0.5)):
                got = wx.iacwpdall(aw, tree)
                exp = np.asfortranarray(np.stack([oracle.iacwpd(aw[:, :, i], tree) for i in range(B)], axis=-1))
                assert (got == exp).all()
                assert relerr(got, x) <= 1e-10
            y = np.empty(n)
            assert wx.iacwpd_(y, aw[:, :, 0], wt, tree) is y
            with pytest.raises(AssertionError):                                   # test/transforms.jl:162
                wx.iacwpd_(np.zeros(n // 2), aw[:, :, 0], wt, tree)
    with pytest.raises(TypeError):
        wx.acwpt(np.zeros(8, dtype=np.float32), wt)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_jbb_costs_and_trees(wx, oracle, dtype):
    rng = np.random.default_rng(2006)
    wt = _wt(wx, "db4")
    n, N = 64, 48
    t = np.arange(n) / n
    base = np.sin(2 * np.pi * 5 * t) + 0.5 * np.sign(np.sin(2 * np.pi * 2 * t))
    X = np.asfortranarray((base[:, None] * (1 + 0.3 * rng.standard_normal((1, N))) + 0.2 * rng.standard_normal((n, N))).astype(dtype))
    xw = oracle.wpdall(X, wt.qmf)
    ctol = 1e-9 if dtype == np.float64 else 2e-3
    for method, okw in ((wx.JBB(), dict()), (wx.JBB(wx.NormCost(1), False), dict(cost="norm")),
                        (wx.JBB(wx.LoglpCost(1)), dict(cost="loglp", p=1.0))):
        got = wx.tree_costs(xw, method)
        exp = oracle.tree_costs_jbb(xw, **okw)
        assert relerr(got, exp) <= ctol
        tree = wx.bestbasistree(xw, method)
        assert wx.isvalidtree(np.zeros(n), tree)
        if dtype == np.float64:
            assert (tree == oracle.bestbasistree_jbb(xw, **okw)).all()           # bit-exact tree indices
    if dtype == np.float64:
        xsw = _stack(oracle.swpd, X, wt.qmf, 4)
        got = wx.tree_costs(xsw, wx.JBB(redundant=True))
        assert relerr(got, oracle.tree_costs_jbb(xsw, redundant=True)) <= ctol
        assert (wx.bestbasistree(xsw, wx.JBB(redundant=True)) == oracle.bestbasistree_jbb(xsw, redundant=True)).all()
        xacw = _stack(oracle.acwpd, X, wt.qmf, 5)
        assert (wx.bestbasistree(xacw, wx.JBB(redundant=True)) == oracle.bestbasistree_jbb(xacw, redundant=True)).all()
        # moments are exact sequential sums -> identical to the oracle's accumulation order
        s, q = wx.jbb_moments(xacw)
        assert (s == xacw.sum(axis=2, dtype=np.float64)).all() or relerr(s, xacw.sum(axis=2)) < 1e-14
        # fused acwpd + moments (config 5 path) and shard accumulation (multi-GPU partials)
        s2, q2 = wx.acwpd_jbb_moments(X, wt, 5)
        assert relerr(s2, s) <= 1e-12 and relerr(q2, q) <= 1e-12
        sa, qa = wx.acwpd_jbb_moments(X[:, :20], wt, 5)
        sa, qa = wx.acwpd_jbb_moments(X[:, 20:], wt, 5, accumulate_into=(sa, qa))
        assert relerr(sa, s) <= 1e-12 and relerr(qa, q) <= 1e-12
        costs = wx.costs_from_moments(sa, qa, N, wx.JBB(redundant=True))
        assert (wx.bestbasis_treeselection(costs, n) == oracle.bestbasistree_jbb(xacw, redundant=True)).all()


def test_jbb_many_signals_chunked_reduction(wx, oracle):
    """few coefficients, many signals: the split-axis moments path must agree with the oracle"""
    rng = np.random.default_rng(2007)
    wt = _wt(wx, "haar")
    X = np.asfortranarray(rng.standard_normal((16, 4096)) * (1 + np.arange(16))[:, None])
    xw = wx.wpdall(X, wt)
    got = wx.tree_costs(xw)
    exp = oracle.tree_costs_jbb(oracle.wpdall(X, wt.qmf))
    assert relerr(got, exp) <= 1e-10
    assert (wx.bestbasistree(xw) == oracle.bestbasistree_jbb(oracle.wpdall(X, wt.qmf))).all()


@pytest.mark.parametrize("n,L,wname", [(256, 8, "coif6"), (128, 7, "db4"), (512, 5, "haar")])
def test_fused_acwpd_moments_deep_trees(wx, oracle, n, L, wname):
    """config-5 path: residue-class subtree kernel (several top-table depths) and the materialising
    fallback must both reproduce the oracle's moments, costs and tree."""
    rng = np.random.default_rng(2008)
    wt = _wt(wx, wname)
    N = 21
    t = np.arange(n) / n
    X = np.asfortranarray(np.sin(2 * np.pi * 7 * t)[:, None] * (1 + 0.4 * rng.standard_normal((1, N))) + 0.3 * rng.standard_normal((n, N)))
    xacw = _stack(oracle.acwpd, X, wt.qmf, L)
    s_ref, q_ref = xacw.sum(axis=2), (xacw ** 2).sum(axis=2)
    for force in (0, 1):
        wx.set_force_generic(force)
        try:
            s, q = wx.acwpd_jbb_moments(X, wt, L)
            assert relerr(s, s_ref) <= 1e-12 and relerr(q, q_ref) <= 1e-12, force
            costs = wx.costs_from_moments(s, q, N, wx.JBB(redundant=True))
            assert (wx.bestbasis_treeselection(costs, n) == oracle.bestbasistree_jbb(xacw, redundant=True)).all()
        finally:
            wx.set_force_generic(0)


def test_jbb_identical_signals_give_zero_sigma(wx, oracle):
    """a batch of identical signals: x^2 and the sums are rounded separately in the reference
    (bestbasis_tree.jl:153-156), so for B = 1, 2, 4 the variance is exactly 0 (sigma = 0, costs -Inf: not NaN, not
    log(1 ulp)); for other B the mean is inexact and the reference's own `@assert all(sigma .>= 0)` / sqrt may
    fail -- whatever the oracle does, the device path must do the same, through the table and the fused path"""
    rng = np.random.default_rng(2011)
    wt = _wt(wx, "db4")
    x1 = rng.standard_normal(64)
    for B in (1, 2, 3, 4, 6):
        x = np.asfortranarray(np.stack([x1] * B, axis=1))
        for X, red in ((wx.wpdall(x, wt), False), (wx.acwpdall(x, wt), True)):
            try:
                exp = oracle.tree_costs_jbb(X, red)
            except Exception:
                exp = None
            if exp is None or np.isnan(exp).any():
                with pytest.raises(AssertionError):
                    wx.tree_costs(X, wx.JBB(redundant=red))
                continue
            c = wx.to_numpy(wx.tree_costs(X, wx.JBB(redundant=red)))
            assert (np.isneginf(c) == np.isneginf(exp)).all(), (B, red)
            if B in (1, 2, 4):
                assert np.isneginf(c).all(), (B, red)
            assert (wx.bestbasistree(X, wx.JBB(redundant=red)) == oracle.bestbasistree_jbb(X, redundant=red)).all()
        if B in (1, 2, 4):
            s_, q_ = wx.acwpd_jbb_moments(x, wt)
            cf = wx.to_numpy(wx.costs_from_moments(s_, q_, B, wx.JBB(redundant=True)))
            assert np.isneginf(cf).all(), B


@pytest.mark.parametrize("wname", ["haar", "db2", "db4", "db6", "coif6", "db10"])
def test_deep_levels_of_swpt_and_acwpt_in_registers(wx, oracle, wname):
    """swpt / acwpt of signals of 1024 samples and more: the levels from depth log2(n) - 4 on run lane-locally in registers
    (csrc/wx_swtdeep.hip); depths with 0 .. 4 such levels, swt/swt_one_level.jl:99-127, acwt/acwt_one_level.jl"""
    rng = np.random.default_rng(77)
    wt = wx.wavelet(getattr(wx.WT, wname))
    for n in (1024, 2048):
        x = np.asfortranarray(rng.standard_normal((n, 2)))
        D0 = int(np.log2(n)) - 4
        for L in (D0, D0 + 1, D0 + 2, D0 + 3, D0 + 4):
            exp = np.stack([oracle.swpt(x[:, b], wt.qmf, L) for b in range(2)], axis=-1)
            assert relerr(wx.swptall(x, wt, L), exp) <= 1e-12, (n, wname, L)
            assert relerr(wx.iswptall(exp, wt), x) <= 1e-10, (n, wname, L)
            expa = np.stack([oracle.acwpt(x[:, b], wt.qmf, L) for b in range(2)], axis=-1)
            assert relerr(wx.acwptall(x, wt, L), expa) <= 1e-12, (n, wname, L)
    # one signal through the single-signal entry points, device memory
    xd = wx.to_device(np.asfortranarray(rng.standard_normal((1024, 3))))
    got = wx.swptall(xd, wt, 10)
    exp = np.stack([oracle.swpt(xd.cpu().numpy()[:, b], wt.qmf, 10) for b in range(3)], axis=-1)
    assert relerr(got.cpu().numpy(), exp) <= 1e-12


def test_deep_levels_of_swpd_and_acwpd_tables(wx, oracle):
    """the heap-ordered tables (every node kept): the same lane-local walk stores both children at every level"""
    rng = np.random.default_rng(78)
    for wname in ("db2", "db4", "coif6"):
        wt = wx.wavelet(getattr(wx.WT, wname))
        x = np.asfortranarray(rng.standard_normal((1024, 2)))
        for L in (6, 7, 9, 10):
            exp = np.stack([oracle.swpd(x[:, b], wt.qmf, L) for b in range(2)], axis=-1)
            assert relerr(wx.swpdall(x, wt, L), exp) <= 1e-12, (wname, L)
            expa = np.stack([oracle.acwpd(x[:, b], wt.qmf, L) for b in range(2)], axis=-1)
            assert relerr(wx.acwpdall(x, wt, L), expa) <= 1e-12, (wname, L)


@pytest.mark.parametrize("wname", ["haar", "db4", "db6", "db8", "coif6", "db10"])
def test_fused_sdwt_isdwt_at_one_workgroup_per_cu(wx, oracle, wname):
    """sdwt / average isdwt of 4096-sample Float64 signals: the all-levels kernels (csrc/wx_swt1d.hip) with register windows at
    every dilation, compile-time taps up to 20, and -- three 32 KiB arrays, one workgroup per CU -- the inverse that fetches
    the next detail column under the current level; more signals than workgroups so that a workgroup walks several.
    swt/swt_one_level.jl:99-127, 257-318, SWT.jl:86-110, 286-325"""
    rng = np.random.default_rng(2026)
    wt = wx.wavelet(getattr(wx.WT, wname))
    n, B = 4096, 300
    x = np.asfortranarray(rng.standard_normal((n, B)))
    for L in (2, 7, 12):
        sd = wx.to_numpy(wx.sdwtall(x, wt, L))
        for b in (0, 1, 255, 256, 299):
            assert relerr(sd[:, :, b], oracle.sdwt(x[:, b], wt.qmf, L)) <= 1e-12, (wname, L, b)
        xr = wx.to_numpy(wx.isdwtall(sd, wt))
        assert relerr(xr, x) <= 1e-10, (wname, L)
        for b in (0, 256, 299):
            assert relerr(xr[:, b], oracle.isdwt(np.asfortranarray(sd[:, :, b]), wt.qmf, None)) <= 1e-12, (wname, L, b)
    # Float32 signals of 8192 samples: 96 KiB of LDS, the same pipeline with eight values per thread
    x32 = np.asfortranarray(rng.standard_normal((8192, 260)).astype(np.float32))
    sd32 = wx.to_numpy(wx.sdwtall(x32, wt, 9))
    assert relerr(sd32[:, :, 259], oracle.sdwt(x32[:, 259], wt.qmf, 9)) <= 1e-5, wname
    assert relerr(wx.to_numpy(wx.isdwtall(sd32, wt)), x32) <= 5e-4, wname
