"""GPU parity of the LDB order statistics (csrc/wx_ldbstat.hip): the robust Fisher power (ldb_measures.jl:481-519) and the
earth mover's distance between class signatures with equal weights (ldb_energymap.jl:186-238, ldb_measures.jl:185-201,
254-360) against the oracle's loop restatements.  Medians / MADs are exact order statistics (bit-equal); the EMD sums
its gaps in a different order (1e-12)."""
import numpy as np
import pytest

from helpers import relerr

pytestmark = pytest.mark.gpu


def _data(rng, shape, N, nc, dtype=np.float64):
    y = rng.integers(0, nc, size=N)
    y[:nc] = np.arange(nc)                                              # every class present
    X = rng.standard_normal(shape + (N,)) + 0.3 * y                     # class-dependent shift
    return np.asfortranarray(X.astype(dtype)), [("c%d" % v) for v in y]


@pytest.mark.parametrize("N,nc", [(7, 2), (40, 3), (128, 4), (301, 2)])
def test_robust_fishers_power_matches_oracle(wx, oracle, N, nc):
    rng = np.random.default_rng(N)
    X, y = _data(rng, (64,), N, nc)
    power, order = wx.discriminant_power(X, y, wx.RobustFishersClassSeparability())
    ep, eo = oracle.ldb_robust_fishers(X, y)
    assert relerr(power, ep) <= 1e-13
    assert np.array_equal(order, eo)
    # 2-D coefficients (sz = (8, 8)), Float32
    X2, y2 = _data(rng, (8, 8), N, nc, np.float32)
    p2, o2 = wx.discriminant_power(X2, y2, wx.RobustFishersClassSeparability())
    e2, _ = oracle.ldb_robust_fishers(X2, y2)
    assert p2.shape == (8, 8) and relerr(p2.astype(np.float64), e2.astype(np.float64)) <= 1e-5


def test_class_medians_are_exact_with_ties(wx, oracle):
    rng = np.random.default_rng(1)
    X = np.asfortranarray(rng.integers(-3, 4, size=(33, 50)).astype(np.float64))     # many ties
    y = [i % 3 for i in range(50)]
    power, _ = wx.discriminant_power(X, y, wx.RobustFishersClassSeparability())
    ep, _ = oracle.ldb_robust_fishers(X, y)
    both = np.isfinite(ep)
    assert np.array_equal(np.isfinite(power), both)
    assert relerr(power[both], ep[both]) <= 1e-13


@pytest.mark.parametrize("N,nc", [(9, 2), (60, 3), (200, 4)])
def test_earth_mover_distance_of_signatures_matches_oracle(wx, oracle, N, nc):
    rng = np.random.default_rng(N)
    wt = wx.wavelet(wx.WT.db2)
    x = np.asfortranarray(rng.standard_normal((16, N)))
    yv = rng.integers(0, nc, size=N); yv[:nc] = np.arange(nc)
    x += 0.5 * yv                                                       # shift per class
    y = list(yv)
    Xw = wx.wpdall(x, wt, 3)                                            # (16, 4, N)
    G = wx.energy_map(Xw, y, wx.Signatures())
    assert len(G) == nc and G[0].weight == 1.0 / (yv == 0).sum()
    assert np.array_equal(G[1].coef, Xw[..., np.flatnonzero(yv == 1)])
    D = wx.discriminant_measure(G, wx.EarthMoverDistance())
    exp = oracle.ldb_emd_measure(Xw, y)
    assert D.shape == (16, 4)
    assert relerr(D, exp) <= 1e-12


def test_emd_with_ties_and_float32(wx, oracle):
    rng = np.random.default_rng(2)
    Xw = np.asfortranarray(rng.integers(-2, 3, size=(8, 3, 30)).astype(np.float64))
    y = [i % 2 for i in range(30)]
    D = wx.discriminant_measure(wx.energy_map(Xw, y, wx.Signatures()), wx.EarthMoverDistance())
    assert relerr(D, oracle.ldb_emd_measure(Xw, y)) <= 1e-12
    X32 = np.asfortranarray(rng.standard_normal((8, 3, 30)).astype(np.float32))
    D32 = wx.discriminant_measure(wx.energy_map(X32, y, wx.Signatures()), wx.EarthMoverDistance())
    assert D32.dtype == np.float32
    assert relerr(D32.astype(np.float64), oracle.ldb_emd_measure(X32.astype(np.float64), y)) <= 1e-5


def test_signatures_weight_types(wx):
    with pytest.raises(ValueError):
        wx.Signatures("other")


def test_probability_density_energy_map_matches_oracle(wx, oracle):
    """energy_map(Xw, y, ProbabilityDensity()) ldb_energymap.jl:143-184 and its discriminant measures"""
    rng = np.random.default_rng(11)
    wt = wx.wavelet(wx.WT.haar)
    N, nc = 50, 3
    yv = rng.integers(0, nc, size=N); yv[:nc] = np.arange(nc)
    x = np.asfortranarray(rng.standard_normal((8, N)) + 0.7 * yv)
    Xw = wx.wpdall(x, wt, 2)                                            # (8, 3, N)
    G = wx.energy_map(Xw, list(yv), wx.ProbabilityDensity())
    exp = oracle.ldb_pdf_energy_map(Xw, list(yv))
    assert G.shape == exp.shape and G.shape[2] >= 100 and G.dtype == np.float64
    assert relerr(G, exp) <= 1e-12
    # every density integrates to one over its own grid: sum(density) * step = 1
    flat = Xw.reshape(-1, N, order="F")
    sd = flat.std(axis=1, ddof=1)
    step = (flat.max(axis=1) - flat.min(axis=1) + sd) / (G.shape[2] - 1)
    integ = G.reshape(-1, G.shape[2], nc, order="F").sum(axis=1) * step[:, None]
    assert np.abs(integ - 1.0).max() <= 1e-12
    for dm, name in ((wx.AsymmetricRelativeEntropy(), "are"), (wx.LpDistance(), "lp"), (wx.HellingerDistance(), "hellinger"),
                     (wx.SymmetricRelativeEntropy(), "sre")):
        D = wx.discriminant_measure(G, dm)
        assert D.shape == (8, 3)
        assert relerr(D, oracle.ldb_pdf_discriminant(exp, name)) <= 1e-11, name


def test_signatures_with_pdf_weights_matches_oracle(wx, oracle):
    rng = np.random.default_rng(12)
    N, nc = 40, 2
    yv = np.array([i % nc for i in range(N)])
    Xw = np.asfortranarray(rng.standard_normal((8, 3, N)) + 0.4 * yv)
    y = list(yv)
    G = wx.energy_map(Xw, y, wx.Signatures("pdf"))
    W = oracle.ldb_signature_weights(Xw, y)
    assert relerr(G.W, W) <= 1e-11
    assert G[1].weight.shape == (8, 3, N // 2) and (G.W >= 0).all()
    D = wx.discriminant_measure(G, wx.EarthMoverDistance())
    assert relerr(D, oracle.ldb_emd_measure_weighted(Xw, W, y)) <= 1e-11


@pytest.mark.parametrize("en_name", ["signatures", "density"])
def test_local_discriminant_basis_with_the_other_energy_maps(wx, oracle, en_name):
    """fitdec! (LDB.jl:186-251) through the Signatures / ProbabilityDensity maps and the robust Fisher power"""
    rng = np.random.default_rng(21)
    wt = wx.wavelet(wx.WT.db2)
    N, nc, n = 36, 3, 16
    yv = np.array([i % nc for i in range(N)])
    t = np.arange(n)[:, None]
    x = np.asfortranarray(rng.standard_normal((n, N)) * 0.3 + np.sin(2 * np.pi * (yv + 1) * t / n))
    y = list(yv)
    if en_name == "signatures":
        f = wx.LocalDiscriminantBasis(wt=wt, max_dec_level=3, dm=wx.EarthMoverDistance(), en=wx.Signatures(),
                                      dp=wx.RobustFishersClassSeparability(), n_features=5)
    else:
        f = wx.LocalDiscriminantBasis(wt=wt, max_dec_level=3, dm=wx.LpDistance(), en=wx.ProbabilityDensity(),
                                      dp=wx.RobustFishersClassSeparability(), n_features=5)
    feats = wx.fit_transform(f, x, y)
    assert feats.shape == (5, N)
    Xw = wx.wpdall(x, wt, 3)
    if en_name == "signatures":
        assert relerr(f.DM, oracle.ldb_emd_measure(Xw, y)) <= 1e-12
    else:
        assert relerr(f.DM, oracle.ldb_pdf_discriminant(oracle.ldb_pdf_energy_map(Xw, y), "lp")) <= 1e-11
    assert wx.isvalidtree(np.empty(n), f.tree)
    coefs = wx.getbasiscoefall(Xw, f.tree)
    ep, eo = oracle.ldb_robust_fishers(coefs, y)
    assert relerr(f.DP, ep) <= 1e-12 and np.array_equal(f.order, eo)
    assert relerr(feats, coefs[np.asarray(f.order[:5]) - 1, :]) == 0.0
