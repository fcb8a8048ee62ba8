"""GPU parity, randomized differential sweep: seeded random geometry (lengths incl. non-dyadic multiples of
2^L, depths incl. 0, batches incl. 1, every filter in the table, both element types, random trees, host and
device pointers) for every 1-D family against the CPU oracle.  Complements the hand-picked cases of the other
test files; tolerances as everywhere (1e-10 Float64 / 1e-5 Float32, trees and gathers exact)."""
import numpy as np
import pytest

from helpers import TOL, random_tree_1d, relerr

pytestmark = pytest.mark.gpu

import os

SCALE = int(os.environ.get("WX_FUZZ_SCALE", "1"))      # soak runs: WX_FUZZ_SCALE=10 multiplies the case counts

FILTERS = ["haar", "db2", "db3", "db4", "db5", "db6", "db8", "db10", "coif2", "coif4", "coif6"]


def _stack(fn, X, *a):
    return np.asfortranarray(np.stack([fn(np.asfortranarray(X[..., i]), *a) for i in range(X.shape[-1])], axis=-1))


def _cases(seed, count):
    rng = np.random.default_rng(seed)
    for _ in range(count):
        L = int(rng.integers(0, 8))
        odd = int(rng.choice([1, 1, 1, 3, 5]))
        n = odd << int(rng.integers(L, 11)) if odd == 1 else odd << L << int(rng.integers(0, 3))
        n = max(n, 2)
        while L > 0 and (n % (1 << L) != 0):
            L -= 1
        yield dict(n=n, L=L, B=int(rng.choice([1, 2, 3, 7])), wname=str(rng.choice(FILTERS)),
                   dtype=np.dtype(rng.choice([np.float64, np.float32])), dev=bool(rng.random() < 0.3), rng=rng)


def _put(wx, x, dev):
    return wx.to_device(x) if dev else x


def test_fuzz_decimated(wx, oracle):
    for c in _cases(4001, 60 * SCALE):
        n, L, B, dt, rng = c["n"], c["L"], c["B"], c["dtype"], c["rng"]
        wt = wx.wavelet(getattr(wx.WT, c["wname"]))
        tol = TOL[dt]
        x = np.asfortranarray(rng.standard_normal((n, B)).astype(dt))
        tag = (n, L, B, c["wname"], str(dt), c["dev"])
        xw = wx.to_numpy(wx.wpdall(_put(wx, x, c["dev"]), wt, L))
        assert relerr(xw, _stack(oracle.wpd, x, wt.qmf, L)) <= tol, tag
        if n & (n - 1) == 0:                                            # maketree (iwpd by level, trees) needs dyadic lengths
            assert relerr(wx.to_numpy(wx.iwpdall(_put(wx, xw, c["dev"]), wt, L)), x) <= 50 * tol, tag
            tl = wx.to_numpy(wx.wptall(_put(wx, x, c["dev"]), wt, L))
            assert relerr(tl, _stack(oracle.wpt, x, wt.qmf, L)) <= tol, tag
            assert relerr(wx.to_numpy(wx.iwptall(_put(wx, tl, c["dev"]), wt, L)), x) <= 50 * tol, tag
            if n >= 4:
                tree = random_tree_1d(n, rng)
                tt = wx.to_numpy(wx.wptall(_put(wx, x, c["dev"]), wt, tree))
                assert relerr(tt, _stack(oracle.wpt, x, wt.qmf, tree)) <= tol, tag
                assert relerr(wx.to_numpy(wx.iwptall(_put(wx, tt, c["dev"]), wt, tree)), x) <= 50 * tol, tag
                full = wx.to_numpy(wx.wpdall(x, wt))
                assert (wx.to_numpy(wx.getbasiscoefall(_put(wx, full, c["dev"]), tree)) ==
                        _stack(oracle.getbasiscoef, full, tree)).all(), tag


def test_fuzz_redundant(wx, oracle):
    for c in _cases(4002, 40 * SCALE):
        n, L, B, dt, rng = c["n"], min(c["L"], 5), c["B"], c["dtype"], c["rng"]
        if L == 0:
            continue
        wt = wx.wavelet(getattr(wx.WT, c["wname"]))
        tol = TOL[dt]
        x = np.asfortranarray(rng.standard_normal((n, B)).astype(dt))
        tag = (n, L, B, c["wname"], str(dt), c["dev"])
        sp = wx.to_numpy(wx.swptall(_put(wx, x, c["dev"]), wt, L))
        assert relerr(sp, _stack(oracle.swpt, x, wt.qmf, L)) <= tol, tag
        sm = [None, int(rng.integers(0, 1 << L))][int(rng.integers(0, 2))]
        got = wx.to_numpy(wx.iswptall(_put(wx, sp, c["dev"]), wt, sm))
        assert relerr(got, _stack(oracle.iswpt, sp, wt.qmf, sm)) <= tol, tag
        assert relerr(got, x) <= 50 * tol, tag
        sd = wx.to_numpy(wx.sdwtall(_put(wx, x, c["dev"]), wt, L))
        assert relerr(sd, _stack(oracle.sdwt, x, wt.qmf, L)) <= tol, tag
        assert relerr(wx.to_numpy(wx.isdwtall(_put(wx, sd, c["dev"]), wt)), x) <= 50 * tol, tag
        if n & (n - 1) == 0 and dt == np.float64:
            aw = wx.to_numpy(wx.acwpdall(_put(wx, x, c["dev"]), wt, L))
            assert relerr(aw, _stack(oracle.acwpd, x, wt.qmf, L)) <= tol, tag
            assert relerr(wx.to_numpy(wx.iacwpdall(_put(wx, aw, c["dev"]), L)), x) <= 50 * tol, tag
            tree = wx.bestbasistree(_put(wx, aw, c["dev"]), wx.JBB(redundant=True))
            assert (tree == oracle.bestbasistree_jbb(aw, redundant=True)).all(), tag


def _cases2d(seed, count):
    rng = np.random.default_rng(seed)
    for _ in range(count):
        L = int(rng.integers(0, 5))
        m = int(rng.choice([1, 1, 3])) << L << int(rng.integers(0, 3))
        n = int(rng.choice([1, 1, 5])) << L << int(rng.integers(0, 3))
        m, n = max(m, 2), max(n, 2)
        while L > 0 and (m % (1 << L) or n % (1 << L)):
            L -= 1
        yield dict(m=m, n=n, L=L, B=int(rng.choice([1, 2, 5])), wname=str(rng.choice(FILTERS[:8])),
                   dtype=np.dtype(rng.choice([np.float64, np.float32])), dev=bool(rng.random() < 0.3), rng=rng)


def test_fuzz_2d(wx, oracle):
    from helpers import random_tree_2d
    for c in _cases2d(4003, 40 * SCALE):
        m, n, L, B, dt, rng = c["m"], c["n"], c["L"], c["B"], c["dtype"], c["rng"]
        wt = wx.wavelet(getattr(wx.WT, c["wname"]))
        tol = TOL[dt]
        x = np.asfortranarray(rng.standard_normal((m, n, B)).astype(dt))
        tag = (m, n, L, B, c["wname"], str(dt), c["dev"])
        xw = wx.to_numpy(wx.wpdall(_put(wx, x, c["dev"]), wt, L))
        assert relerr(xw, _stack(oracle.wpd, x, wt.qmf, L)) <= tol, tag
        y = wx.to_numpy(wx.wptall(_put(wx, x, c["dev"]), wt, L))
        assert relerr(y, _stack(oracle.wpt, x, wt.qmf, L)) <= tol, tag
        assert relerr(wx.to_numpy(wx.iwptall(_put(wx, y, c["dev"]), wt, L)), x) <= 50 * tol, tag
        assert relerr(wx.to_numpy(wx.iwpdall(_put(wx, xw, c["dev"]), wt, L)), x) <= 50 * tol, tag
        if wx.maxtransformlevels(min(m, n)) >= 1:
            tree = random_tree_2d(m, n, rng)
            Lside = min(wx.maxtransformlevels(m), wx.maxtransformlevels(n))
            if wx.maxtransformlevels(min(m, n)) > Lside and tree[(4 ** Lside - 1) // 3:].any():
                with pytest.raises(AssertionError):                      # a side not divisible by 2^depth
                    wx.wptall(x, wt, tree)
                tree[(4 ** Lside - 1) // 3:] = False
            yt = wx.to_numpy(wx.wptall(_put(wx, x, c["dev"]), wt, tree))
            assert relerr(yt, _stack(oracle.wpt, x, wt.qmf, tree)) <= tol, tag
            assert relerr(wx.to_numpy(wx.iwptall(_put(wx, yt, c["dev"]), wt, tree)), x) <= 50 * tol, tag
        if L >= 1 and m * n <= 4096:
            Lr = min(L, 2)
            sp = wx.to_numpy(wx.swptall(_put(wx, x, c["dev"]), wt, Lr))
            assert relerr(sp, np.asfortranarray(np.stack([oracle.red2d_fwd("wpt", x[:, :, i], wt.qmf, Lr) for i in range(B)], axis=-1))) <= tol, tag
            assert relerr(wx.to_numpy(wx.iswptall(_put(wx, sp, c["dev"]), wt)), x) <= 50 * tol, tag


def test_fuzz_best_basis_degenerate_inputs(wx, oracle):
    """zeros, constants, impulses, identical and single signals through JBB and BB: whatever the oracle does
    (including failing the reference's asserts) the device path does too"""
    rng = np.random.default_rng(4004)
    wt = wx.wavelet(wx.WT.db2)
    n = 32
    base = {
        "zeros": np.zeros(n), "const": np.full(n, 3.25), "impulse": np.eye(1, n, 5).ravel(),
        "ramp": np.arange(n, dtype=float), "noise": rng.standard_normal(n),
    }
    for name, v in base.items():
        for B in (1, 2, 5):
            x = np.asfortranarray(np.stack([v] * B, axis=1))
            if name == "noise":
                x = np.asfortranarray(x + 1e-3 * rng.standard_normal(x.shape))
            X = wx.wpdall(x, wt)
            # BB: one tree per signal
            assert (wx.bestbasistreeall(X, wx.BB()) == oracle.bestbasistreeall_bb(X)).all(), (name, B)
            assert (wx.bestbasistreeall(X, wx.BB(cost=wx.LogEnergyEntropyCost())) ==
                    oracle.bestbasistreeall_bb(X, cost="logenergy")).all(), (name, B)
            # JBB
            try:
                exp = oracle.bestbasistree_jbb(X)
            except Exception:
                exp = None
            if exp is None:
                with pytest.raises(AssertionError):
                    wx.bestbasistree(X, wx.JBB())
            else:
                assert (wx.bestbasistree(X, wx.JBB()) == exp).all(), (name, B)


def test_fuzz_redundant_trees_and_2d_ac(wx, oracle):
    from helpers import random_tree_2d
    rng = np.random.default_rng(4005)
    for it in range(24 * SCALE):
        wt = wx.wavelet(getattr(wx.WT, str(rng.choice(FILTERS[:6]))))
        dt = np.dtype(rng.choice([np.float64, np.float32]))
        tol = TOL[dt]
        n = 1 << int(rng.integers(2, 8))
        B = int(rng.choice([1, 3]))
        x = np.asfortranarray(rng.standard_normal((n, B)).astype(dt))
        Lmax = wx.maxtransformlevels(n)
        L = int(rng.integers(1, min(Lmax, 5) + 1))
        # swpd + iswpd by random tree and random shift
        sw = wx.swpdall(x, wt, L)
        assert relerr(sw, _stack(oracle.swpd, x, wt.qmf, L)) <= tol, (n, L)
        tree = random_tree_1d(n, rng)
        tree[(1 << L) - 1:] = False                                   # the table only holds depth <= L
        sm = None if rng.random() < 0.5 else int(rng.integers(0, 1 << L))
        got = wx.iswpdall(sw, wt, tree, sm)
        exp = np.asfortranarray(np.stack([oracle.iswpd(sw[:, :, i], wt.qmf, tree, sm) for i in range(B)], axis=-1))
        assert relerr(got, exp) <= tol, (n, L, sm)
        assert relerr(got, x) <= 50 * tol, (n, L, sm)
        # isdwt with a random shift
        sd = wx.sdwtall(x, wt, L)
        s2 = int(rng.integers(1, 1 << L)) if L >= 1 else None
        assert relerr(wx.isdwtall(sd, wt, s2), _stack(oracle.isdwt, sd, wt.qmf, s2)) <= tol, (n, L, s2)
        if dt == np.float64:
            # ACWT by tree (1-D) and the 2-D autocorrelation families
            aw = wx.acwpdall(x, wt, L)
            got = wx.iacwpdall(aw, tree)
            exp = np.asfortranarray(np.stack([oracle.iacwpd(aw[:, :, i], tree) for i in range(B)], axis=-1))
            assert (got == exp).all(), (n, L)                        # pairwise sums: bit exact
            m2, n2 = 1 << int(rng.integers(2, 5)), 1 << int(rng.integers(2, 5))
            img = np.asfortranarray(rng.standard_normal((m2, n2, B)))
            L2 = int(rng.integers(1, min(wx.maxtransformlevels(min(m2, n2)), 2) + 1))
            a2 = wx.acwpdall(img, wt, L2)
            e2 = np.asfortranarray(np.stack([oracle.red2d_fwd("wpd", img[:, :, i], wt.qmf, L2, ac=True) for i in range(B)], axis=-1))
            assert relerr(a2, e2) <= tol, (m2, n2, L2)
            assert relerr(wx.iacwpdall(a2, L2), img) <= 50 * tol, (m2, n2, L2)
            t2 = random_tree_2d(m2, n2, rng)
            t2[(4 ** L2 - 1) // 3:] = False
            g2 = wx.iacwpdall(a2, t2)
            x2 = np.asfortranarray(np.stack([oracle.red2d_inv("wpd", a2[:, :, :, i], None, t2, ac=True) for i in range(B)], axis=-1))
            assert relerr(g2, x2) <= tol, (m2, n2, L2)


def test_fuzz_denoise_random_trees(wx, oracle):
    """denoise through random best-basis-like trees (wpt leaves, swpd / acwpd heap columns), both smoothing modes,
    every threshold function, random noise estimates"""
    rng = np.random.default_rng(4006)
    ths = {"hard": wx.HardTH, "soft": wx.SoftTH, "semisoft": wx.SemiSoftTH, "stein": wx.SteinTH}
    for it in range(18 * SCALE):
        n = 1 << int(rng.integers(4, 9))
        wt = wx.wavelet(getattr(wx.WT, str(rng.choice(FILTERS[:6]))))
        x = np.asfortranarray(rng.standard_normal((n, 3)) + np.sin(np.arange(n) / 5.0)[:, None])
        Lmax = wx.maxtransformlevels(n)
        tree = random_tree_1d(n, rng, 0.8)
        tree[0] = True
        thname = str(rng.choice(list(ths)))
        smooth = str(rng.choice(["regular", "undersmooth"]))
        dnt = wx.VisuShrink(ths[thname](), float(rng.uniform(0.5, 3.0)))
        est = None if rng.random() < 0.6 else float(rng.uniform(0.1, 1.0))
        kw = dict(L=Lmax, tree=tree, dnt=dnt, smooth=smooth, estnoise=est)
        inputs = {"wpt": wx.to_numpy(wx.wptall(x, wt, tree)), "swpd": wx.to_numpy(wx.swpdall(x, wt, Lmax)),
                  "acwpd": wx.to_numpy(wx.acwpdall(x, wt, Lmax))}
        for inputtype, X in inputs.items():
            Y = wx.to_numpy(wx.denoiseall(X, inputtype, wt, **kw))
            for i in range(3):
                exp = oracle.denoise(np.asfortranarray(X[..., i]), inputtype, wt.qmf, L=Lmax, tree=tree, th=thname,
                                     t=dnt.t, estnoise=est, smooth=smooth)
                assert relerr(Y[:, i], exp) <= 1e-9, (n, inputtype, thname, smooth, est)


def test_fuzz_long_signals_tiles_and_siwt(wx, oracle):
    """the paths added after the first fuzz rounds: 1-D signals beyond the LDS of a CU (per-level launches, then the
    fused kernel on the nodes), 2-D images made of whole tiles (one-pass levels, full and tree-driven, forward and
    inverse), and the shift-invariant decomposition"""
    from helpers import random_tree_2d
    rng = np.random.default_rng(4010)
    for it in range(6 * SCALE):
        dt = np.dtype(rng.choice([np.float64, np.float32]))
        wt = wx.wavelet(getattr(wx.WT, str(rng.choice(FILTERS[:8]))))
        n = 1 << int(rng.integers(13, 17))
        B = int(rng.choice([1, 2, 5]))
        L = int(rng.integers(1, wx.maxtransformlevels(n) - 2))
        x = np.asfortranarray(rng.standard_normal((n, B)).astype(dt))
        tag = ("1d", n, L, B, wt.name, str(dt))
        exp = _stack(oracle.wpd, x, wt.qmf, L)
        assert relerr(wx.wpdall(x, wt, L), exp) <= TOL[dt], tag
        leaves = np.asfortranarray(exp[:, L, :])
        assert relerr(wx.wptall(x, wt, L), leaves) <= TOL[dt], tag
        assert relerr(wx.iwptall(leaves, wt, L), x) <= 50 * TOL[dt], tag
        assert relerr(wx.iwpdall(exp, wt, L), x) <= 50 * TOL[dt], tag
    for it in range(8 * SCALE):
        dt = np.dtype(rng.choice([np.float64, np.float32]))
        wt = wx.wavelet(getattr(wx.WT, str(rng.choice(["haar", "db2", "db3", "db4", "db5", "db6", "db8", "db10"]))))
        m, n = 64 << int(rng.integers(0, 3)), 64 << int(rng.integers(0, 3))
        B = int(rng.choice([1, 3]))
        depth = int(rng.integers(1, wx.maxtransformlevels(min(m, n)) - 1))           # nodes of >= 8 samples... or fewer
        x = np.asfortranarray(rng.standard_normal((m, n, B)).astype(dt))
        tag = ("2d", m, n, depth, B, wt.name, str(dt))
        xw = wx.wpdall(x, wt, depth)
        assert relerr(xw, _stack(oracle.wpd, x, wt.qmf, depth)) <= TOL[dt], tag
        tree = random_tree_2d(m, n, rng, float(rng.choice([0.4, 0.7, 0.95])))
        tree[(4 ** depth - 1) // 3:] = False
        yt = wx.wptall(x, wt, tree)
        assert relerr(yt, _stack(oracle.wpt, x, wt.qmf, tree)) <= TOL[dt], tag
        assert relerr(wx.iwptall(yt, wt, tree), x) <= 50 * TOL[dt], tag
        assert relerr(wx.iwpdall(xw, wt, tree), x) <= 50 * TOL[dt], tag
    for it in range(10 * SCALE):
        dt = np.dtype(rng.choice([np.float64, np.float32]))
        wt = wx.wavelet(getattr(wx.WT, str(rng.choice(FILTERS[:8]))))
        k = int(rng.integers(2, 8))
        n = int(rng.choice([1, 1, 3])) << k
        L = int(rng.integers(1, k + 1))
        d = int(rng.integers(1, L + 1))
        B = int(rng.choice([1, 4]))
        X = np.asfortranarray(rng.standard_normal((n, B)).astype(dt))
        tag = ("siwt", n, L, d, B, wt.name, str(dt))
        batch = wx.siwpdall(X, wt, L, d)
        ref = oracle.siwpd(X[:, B - 1], wt.qmf, L, d)
        obj = batch[B - 1]
        assert obj.BestTree == ref.BestTree, tag
        for key in list(ref.Nodes)[:: max(1, len(ref.Nodes) // 40)]:
            v = np.asarray(obj.Nodes[key].Value)
            scale = max(1.0, float(np.abs(ref.Nodes[key]["Value"]).max()))
            assert float(np.abs(v - ref.Nodes[key]["Value"]).max()) <= (0.0 if dt == np.float64 else 1e-6) * scale, (tag, key)
            assert abs(obj.Nodes[key].Cost - ref.Nodes[key]["Cost"]) <= (1e-9 if dt == np.float64 else 2e-4) * max(1.0, abs(ref.Nodes[key]["Cost"])), (tag, key)
        wx.bestbasistreeall_(batch)
        oracle.siwt_bestbasistree(ref)
        assert wx.isvalidtree(obj), tag
        assert abs(obj.MinCost - ref.MinCost) <= (1e-9 if dt == np.float64 else 2e-4) * max(1.0, abs(ref.MinCost)), tag
        assert relerr(wx.isiwpdall(batch), X) <= 50 * TOL[dt], tag


def test_fuzz_register_kernels(wx, oracle):
    """the round-2 kernels that only exist for one geometry: lattice wpt / iwpt / wpd (n = 4096 Float64), the 2-D lattice
    (512 x 512 Float32, L = 6), the matrix-pipe subtree moments (n / 2^D0 = 32, full depth) -- random depth, filter, batch
    (ragged tails of the 8-signal blocks, batches larger than one wavefront round), host and device pointers"""
    rng = np.random.default_rng(4242)
    lat = ["db2", "db3", "db4", "db5", "db6", "db7", "db8", "db9", "coif6", "db10"]
    for _ in range(10 * SCALE):
        wt = wx.wavelet(getattr(wx.WT, str(rng.choice(lat))))
        B = int(rng.choice([1, 2, 3, 5, 9]))
        dev = bool(rng.random() < 0.5)
        x = np.asfortranarray(rng.standard_normal((4096, B)))
        L = int(rng.integers(1, 13))
        tab = _stack(oracle.wpd, x, wt.qmf, L)
        assert relerr(wx.to_numpy(wx.wpdall(_put(wx, x, dev), wt, L)), tab) <= 1e-11, (wt.name if hasattr(wt, "name") else "", L, B)
        assert relerr(wx.to_numpy(wx.iwpdall(_put(wx, tab, dev), wt, L)), x) <= 1e-11
        ns = int(rng.choice([64, 128, 256, 512, 1024, 2048]))          # interleaved short signals
        xs = np.asfortranarray(rng.standard_normal((ns, B + int(rng.integers(0, 70)))))
        Ls = int(rng.integers(1, int(np.log2(ns)) + 1))
        es = _stack(oracle.wpt, xs, wt.qmf, Ls)
        assert relerr(wx.to_numpy(wx.wptall(_put(wx, xs, dev), wt, Ls)), es) <= 1e-11, (ns, Ls, B)
        assert relerr(wx.to_numpy(wx.iwptall(_put(wx, es, dev), wt, Ls)), xs) <= 1e-11, (ns, Ls, B)
        ts = _stack(oracle.wpd, xs, wt.qmf, Ls)
        assert relerr(wx.to_numpy(wx.wpdall(_put(wx, xs, dev), wt, Ls)), ts) <= 1e-11, (ns, Ls, B)
        assert relerr(wx.to_numpy(wx.iwpdall(_put(wx, ts, dev), wt, Ls)), xs) <= 1e-11, (ns, Ls, B)
        Lt = int(rng.integers(6, 13))
        exp = _stack(oracle.wpt, x, wt.qmf, Lt)
        assert relerr(wx.to_numpy(wx.wptall(_put(wx, x, dev), wt, Lt)), exp) <= 1e-11
        assert relerr(wx.to_numpy(wx.iwptall(_put(wx, exp, dev), wt, Lt)), x) <= 1e-11
    # subtree moments: other lengths (D0 = log2(n) - 5), filters of every length class, batches that are not multiples of 8
    for _ in range(6 * SCALE):
        n = int(rng.choice([64, 128, 256, 1024]))
        L = int(np.log2(n))
        wt = wx.wavelet(getattr(wx.WT, str(rng.choice(["haar", "db2", "db4", "db7", "db8", "coif2", "coif6"]))))
        B = int(rng.choice([1, 3, 8, 13, 21]))
        x = np.asfortranarray(rng.standard_normal((n, B)))
        X = np.asfortranarray(np.stack([oracle.acwpd(x[:, b], wt.qmf, L) for b in range(B)], axis=-1))
        s, q = wx.acwpd_jbb_moments(x, wt, L)
        assert relerr(s, X.sum(axis=2)) <= 1e-12 and relerr(q, (X ** 2).sum(axis=2)) <= 1e-12, (n, B)
        # accumulate_into over two ragged pieces == one call
        if B > 1:
            k = int(rng.integers(1, B))
            xd = wx.to_device(x)
            sa, qa = wx.acwpd_jbb_moments(xd[:, :k], wt, L)
            wx.acwpd_jbb_moments(xd[:, k:], wt, L, accumulate_into=(sa, qa))
            assert relerr(sa.cpu().numpy(), s) <= 1e-13 and relerr(qa.cpu().numpy(), q) <= 1e-13
    # 2-D lattice: random batch sizes, both filters
    import torch
    for _ in range(3 * SCALE):
        wt = wx.wavelet(getattr(wx.WT, str(rng.choice(["db2", "db4"]))))
        B = int(rng.choice([1, 2, 5]))
        x = np.asfortranarray(rng.standard_normal((512, 512, B)).astype(np.float32))
        got = wx.wptall(x, wt, 6)
        exp = _stack(oracle.wpt, x.astype(np.float64), wt.qmf, 6)
        assert relerr(got.astype(np.float64), exp) <= 2e-6
        assert relerr(wx.iwptall(exp.astype(np.float32), wt, 6).astype(np.float64), x.astype(np.float64)) <= 2e-6
