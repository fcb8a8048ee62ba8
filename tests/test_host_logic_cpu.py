"""Host-side logic of the Python mirror that needs no GPU, against the oracle: LDB discriminant measures and
node costs (pure numpy on the small class maps), class indexing, the detail ranges noisest uses, VisuShrink."""
import numpy as np
import pytest


def test_ldb_measures_and_costs_match_the_loop_restatement(wx, oracle):
    from waveletsext_jl_amd import ldb
    rng = np.random.default_rng(7001)
    n, N = 32, 12
    y = ["b", "a", "c"] * 4
    X = np.asfortranarray(rng.standard_normal((n, N)) + np.array([{"a": 0, "b": 1, "c": 2}[v] for v in y])[None, :])
    q = wx.wavelet(wx.WT.db2).qmf
    Xw = oracle.wpdall(X, q, 4)
    G = oracle.ldb_energy_map(Xw, y)
    for dm, name in ((ldb.AsymmetricRelativeEntropy(), "are"), (ldb.SymmetricRelativeEntropy(), "sre"),
                     (ldb.HellingerDistance(), "hellinger"), (ldb.LpDistance(2), "lp")):
        D = ldb.discriminant_measure(G, dm)
        assert np.allclose(D, oracle.ldb_discriminant_measure(G, name), rtol=1e-12, atol=1e-14), name
        for top_k in (n, 3):
            cost = ldb._node_costs(D, (n,), Xw.shape[1], top_k)
            exp = oracle.ldb_fitdec(Xw, y, dm=name, top_k=top_k)["cost"]
            assert np.allclose(cost, exp, rtol=1e-12, atol=1e-14), (name, top_k)
    # zero energies are skipped by the relative entropy (ldb_measures.jl:305)
    Gz = G.copy(); Gz[0, 0, 0] = 0.0
    assert np.isfinite(ldb.discriminant_measure(Gz, ldb.AsymmetricRelativeEntropy())).all()
    # 2-D maps
    img = np.asfortranarray(rng.standard_normal((8, 8, 6)))
    Xw2 = np.asfortranarray(np.stack([oracle.wpd(np.asfortranarray(img[:, :, i]), q, 2) for i in range(6)], axis=-1))
    y2 = [0, 1, 0, 1, 2, 2]
    G2 = oracle.ldb_energy_map(Xw2, y2)
    D2 = ldb.discriminant_measure(G2)
    assert np.allclose(D2, oracle.ldb_discriminant_measure(G2, "are"), rtol=1e-12, atol=1e-14)
    assert np.allclose(ldb._node_costs(D2, (8, 8), 3, 64), oracle.ldb_fitdec(Xw2, y2)["cost"], rtol=1e-12, atol=1e-14)


def test_class_indexing_follows_first_occurrence(wx):
    from waveletsext_jl_amd import ldb
    classes, idx = ldb._classes(["b", "a", "b", "c", "a"])
    assert classes == ["b", "a", "c"] and idx.tolist() == [0, 1, 0, 2, 1] and idx.dtype == np.int32
    classes, idx = ldb._classes(np.array([3, 1, 3, 2]))
    assert classes == [3, 1, 2] and idx.tolist() == [0, 1, 0, 2]


def test_noisest_ranges_and_visushrink(wx, oracle):
    from waveletsext_jl_amd import denoising as dn
    n = 64
    assert dn._detail_range(n, 1, "dwt", None) == (n // 2, 0)
    assert dn._detail_range(n, 7, "sdwt", None) == (0, 6) and dn._detail_range(n, 7, "acdwt", None) == (0, 6)
    rng = np.random.default_rng(7002)
    for _ in range(20):
        tree = np.zeros(n - 1, dtype=bool)
        tree[0] = True
        for i in range(2, n):
            tree[i - 1] = tree[i // 2 - 1] and rng.random() < 0.7
        lo, col = dn._detail_range(n, 1, "wpt", tree)
        assert (lo + 1, col) == (oracle.finestdetailrange(n, tree), 0)            # 1-based lo of (lo:n)
        lo, col = dn._detail_range(n, 2 * n - 1, "swpd", tree)
        assert (lo, col + 1) == (0, oracle.finestdetailrange(n, tree, True))      # 1-based heap column
    assert wx.VisuShrink(128).t == pytest.approx(np.sqrt(2 * np.log(128)))
    assert isinstance(wx.VisuShrink(128).th, wx.HardTH) and isinstance(wx.VisuShrink(128, wx.SoftTH()).th, wx.SoftTH)
    assert wx.VisuShrink(wx.SteinTH(), 1.5).t == 1.5
