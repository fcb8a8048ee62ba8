"""GPU parity: Local Discriminant Basis with the TimeFrequency energy map (LDB.jl:139-448) against the loop
restatement in the oracle.  Energy maps / measures / powers within 1e-10 (Float64) / 1e-5 (Float32) relative,
trees and feature orders compared with ==."""
import numpy as np
import pytest

from helpers import TOL, relerr

pytestmark = pytest.mark.gpu


def _classdata(rng, n, per, dtype, two_d=False):
    """three classes of noisy shapes (like generateclassdata(:tri), three classes)"""
    t = np.arange(n)
    X, y = [], []
    for c, lab in enumerate(["a", "b", "c"]):
        for _ in range(per):
            a, b = rng.integers(n // 8, n // 2), rng.integers(n // 8, n // 2)
            base = np.maximum(0, 6 - np.abs(t - (a if c != 1 else a + b)) / (1 + c)) + (c == 2) * np.sin(8 * np.pi * t / n)
            s = base + rng.standard_normal(n)
            if two_d:
                s = np.outer(s, np.roll(base, c * 3)) / 4 + rng.standard_normal((n, n))
            X.append(s); y.append(lab)
    perm = rng.permutation(len(y))
    X = np.asfortranarray(np.stack([X[i] for i in perm], axis=-1).astype(dtype))
    return X, [y[i] for i in perm]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_energy_map_and_measures(wx, oracle, dtype):
    rng = np.random.default_rng(6001)
    tol = TOL[np.dtype(dtype)]
    X, y = _classdata(rng, 64, 7, dtype)
    Xw = wx.wpdall(X, wx.wavelet(wx.WT.db2))
    G = wx.energy_map(Xw, y)
    assert relerr(G, oracle.ldb_energy_map(Xw, y)) <= tol
    assert relerr(wx.to_numpy(wx.energy_map(wx.to_device(Xw), y)), G) <= tol                   # device-resident table
    for dm, name in ((wx.AsymmetricRelativeEntropy(), "are"), (wx.SymmetricRelativeEntropy(), "sre"),
                     (wx.HellingerDistance(), "hellinger"), (wx.LpDistance(2), "lp")):
        assert relerr(wx.discriminant_measure(G, dm), oracle.ldb_discriminant_measure(wx.to_numpy(G), name)) <= 10 * tol, name
    X2, y2 = _classdata(rng, 16, 4, dtype, two_d=True)
    Xw2 = wx.wpdall(X2, wx.wavelet(wx.WT.haar))
    assert relerr(wx.energy_map(Xw2, y2), oracle.ldb_energy_map(Xw2, y2)) <= tol
    with pytest.raises(AssertionError):
        wx.energy_map(Xw, ["a"] * len(y))                                        # @assert nc > 1


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("dp", ["basis", "fisher"])
def test_ldb_fit_transform(wx, oracle, dtype, dp):
    rng = np.random.default_rng(6002)
    tol = TOL[np.dtype(dtype)]
    n = 64
    X, y = _classdata(rng, n, 9, dtype)
    for dm, name, top_k in ((wx.AsymmetricRelativeEntropy(), "are", None), (wx.HellingerDistance(), "hellinger", 5)):
        f = wx.LocalDiscriminantBasis(wt=wx.wavelet(wx.WT.coif2), dm=dm, top_k=top_k, n_features=10,
                                      dp=wx.BasisDiscriminantMeasure() if dp == "basis" else wx.FishersClassSeparability())
        Xc = wx.fit_transform(f, X, y)
        Xw = wx.wpdall(X, f.wt)
        exp = oracle.ldb_fitdec(Xw, y, dm=name, top_k=top_k, dp=dp)
        assert relerr(f.cost, exp["cost"]) <= 50 * tol
        assert (f.tree == exp["tree"]).all()
        assert relerr(f.DP, exp["DP"]) <= 200 * tol
        assert (f.order[:10] == exp["order"][:10]).all()
        leaves = np.asfortranarray(np.stack([oracle.getbasiscoef(np.asfortranarray(Xw[:, :, i]), exp["tree"]) for i in range(X.shape[1])], axis=-1))
        assert relerr(Xc, leaves[exp["order"][:10] - 1, :]) <= tol
        # transform of new data, inverse transform, change of the feature count
        Xn, _ = _classdata(rng, n, 2, dtype)
        Xt = wx.to_numpy(wx.transform(f, Xn))
        assert Xt.shape == (10, Xn.shape[1])
        full = wx.to_numpy(wx.wptall(Xn, f.wt, f.tree))
        assert relerr(Xt, full[f.order[:10] - 1, :]) <= tol
        back = wx.to_numpy(wx.inverse_transform(f, Xt))
        keep = np.zeros_like(full); keep[f.order[:10] - 1, :] = full[f.order[:10] - 1, :]
        assert relerr(back, wx.to_numpy(wx.iwptall(keep, f.wt, f.tree))) <= 50 * tol
        assert (wx.change_nfeatures(f, Xt, 4) == Xt[:4]).all() and f.n_features == 4
    # device-resident training set
    f2 = wx.LocalDiscriminantBasis(wt=wx.wavelet(wx.WT.coif2), n_features=6)
    Xd = wx.fit_transform(f2, wx.to_device(X), y)
    f3 = wx.LocalDiscriminantBasis(wt=wx.wavelet(wx.WT.coif2), n_features=6)
    assert relerr(wx.to_numpy(Xd), wx.fit_transform(f3, X, y)) <= tol and (f2.tree == f3.tree).all()


def test_ldb_2d(wx, oracle):
    rng = np.random.default_rng(6003)
    X, y = _classdata(rng, 16, 5, np.float64, two_d=True)
    f = wx.LocalDiscriminantBasis(n_features=12)
    Xc = wx.fit_transform(f, X, y)
    Xw = wx.wpdall(X, f.wt)
    exp = oracle.ldb_fitdec(Xw, y)
    assert relerr(f.cost, exp["cost"]) <= 1e-9
    assert (f.tree == exp["tree"]).all() and (f.order[:12] == exp["order"][:12]).all()
    assert Xc.shape == (12, X.shape[-1])


def test_energy_map_shards_combine(wx, oracle):
    """the multi-GPU formula on one GPU: maps + norm sums of two shards (one of them missing a class) combine to the
    map of the whole batch"""
    rng = np.random.default_rng(6004)
    X, y = _classdata(rng, 32, 4, np.float64)
    order = np.argsort([{"a": 0, "b": 1, "c": 2}[v] for v in y], kind="stable")    # shard 0 = classes a, b only
    X, y = np.asfortranarray(X[:, order]), [y[i] for i in order]
    Xw = wx.wpdall(X, wx.wavelet(wx.WT.db2))
    classes = ["a", "b", "c"]
    cut = 8
    G0, n0 = wx.energy_map(Xw[:, :, :cut], y[:cut], classes=classes, return_norm_sum=True)
    G1, n1 = wx.energy_map(Xw[:, :, cut:], y[cut:], classes=classes, return_norm_sum=True)
    assert n0[2] == 0 and np.isnan(G0[:, :, 2]).all()
    comb = (np.nan_to_num(G0) * n0 + np.nan_to_num(G1) * n1) / (n0 + n1)
    assert relerr(comb, wx.energy_map(Xw, y)) <= 1e-12


def test_ldb_reference_test_expectations(wx):
    """the behaviours test/ldb.jl asserts: shapes of fit_transform / transform / inverse_transform and
    change_nfeatures (:83-87: shrinking works, growing goes through inverse + transform, after which the old
    feature matrix no longer matches f.n_features -> ArgumentError)"""
    rng = np.random.default_rng(6005)
    X, y = _classdata(rng, 32, 5, np.float64)                     # (32, 15), three classes like the reference's set
    for dm in (wx.AsymmetricRelativeEntropy(), wx.SymmetricRelativeEntropy(), wx.LpDistance(2), wx.HellingerDistance()):
        f = wx.LocalDiscriminantBasis(wt=wx.wavelet(wx.WT.coif6), max_dec_level=3, dm=dm, top_k=5, n_features=5)
        Xc = wx.fit_transform(f, X, y)
        assert Xc.shape == (5, 15)
        wx.fit_(f, X, y)
        assert wx.transform(f, X).shape == (5, 15)
        assert wx.inverse_transform(f, Xc).shape == (32, 15)
    x = wx.change_nfeatures(f, Xc, 5)
    assert x.shape == (5, 15)
    grown = wx.change_nfeatures(f, Xc, 10)
    assert grown.shape == (10, 15) and f.n_features == 10
    with pytest.raises(wx.ArgumentError):
        wx.change_nfeatures(f, Xc, 10)
    X2, y2 = _classdata(rng, 8, 5, np.float64, two_d=True)       # (8, 8, 15)
    f = wx.LocalDiscriminantBasis(max_dec_level=2, n_features=5)
    Xc = wx.fit_transform(f, X2, y2)
    assert Xc.shape == (5, 15) and wx.inverse_transform(f, Xc).shape == (8, 8, 15)
