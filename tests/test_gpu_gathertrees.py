"""getbasiscoefall(Xw, tree::BitArray{2}) (Utils.jl:199-225): one tree per signal, ONE launch (csrc/wx_gathertrees.hip,
VERDICT r04 item 7iii) -- bit-exact against the oracle's per-signal getbasiscoef (Utils.jl:101-134), 1-D and 2-D, host and
device tables, both element types, trees staged in LDS and (long signals) walked in global memory; the reference's
assertions for a bad tree in the middle of the matrix and for a table with too few levels."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _random_tree1d(rng, n, p, maxdepth):
    t = np.zeros(n - 1, dtype=bool)
    t[0] = rng.random() < 0.95
    for i in range(1, n):
        d = int(np.floor(np.log2(i)))
        if i > 1:
            t[i - 1] = t[i // 2 - 1] and d < maxdepth and rng.random() < p
    return t


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,k,N", [(64, 7, 37), (1024, 11, 19), (4096, 6, 9), (8, 4, 300)])
def test_one_tree_per_signal_1d(wx, oracle, dtype, n, k, N):
    rng = np.random.default_rng(n + N)
    Xw = np.asfortranarray(rng.standard_normal((n, k, N)).astype(dtype))
    trees = np.asfortranarray(np.stack([_random_tree1d(rng, n, 0.7, k - 1) for _ in range(N)], axis=1))
    trees[:, 0] = wx.maketree(n, k - 1, "full")                        # the deepest tree the table holds
    trees[:, 1] = False                                              # the root alone
    exp = np.stack([oracle.getbasiscoef(Xw[:, :, i], trees[:, i]) for i in range(N)], axis=1)
    got = wx.getbasiscoefall(Xw, trees)
    assert got.dtype == dtype and (got == exp).all()
    gd = wx.getbasiscoefall(wx.to_device(Xw), trees)                   # device table, host trees
    assert (gd.cpu().numpy() == exp).all()


def test_long_signals_walk_the_tree_in_global_memory(wx, oracle):
    n, k, N = 131072, 4, 3                                             # 131071 tree bytes: beyond the LDS staging limit
    rng = np.random.default_rng(17)
    Xw = np.asfortranarray(rng.standard_normal((n, k, N)))
    trees = np.asfortranarray(np.stack([_random_tree1d(rng, n, 0.8, k - 1) for _ in range(N)], axis=1))
    exp = np.stack([oracle.getbasiscoef(Xw[:, :, i], trees[:, i]) for i in range(N)], axis=1)
    assert (wx.getbasiscoefall(Xw, trees) == exp).all()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_one_tree_per_image_2d(wx, oracle, dtype):
    rng = np.random.default_rng(22)
    m = n = 32
    k, N = 4, 11
    Xw = np.asfortranarray(rng.standard_normal((m, n, k, N)).astype(dtype))
    nt = wx.gettreelength(m, n)
    trees = np.zeros((nt, N), dtype=bool, order="F")
    for i in range(N):
        t = trees[:, i]
        t[0] = i != 1
        for node in range(2, nt + 1):
            parent = (node + 2) // 4
            d = wx.getdepth(node, "quad")
            t[node - 1] = t[parent - 1] and d < k - 1 and rng.random() < 0.6
    exp = np.stack([oracle.getbasiscoef2d(Xw[:, :, :, i], trees[:, i]) for i in range(N)], axis=2)
    got = wx.getbasiscoefall(Xw, trees)
    assert got.dtype == dtype and (got == exp).all()
    assert (wx.getbasiscoefall(wx.to_device(Xw), trees).cpu().numpy() == exp).all()


def test_reference_assertions(wx):
    rng = np.random.default_rng(5)
    n, k, N = 16, 3, 4
    Xw = np.asfortranarray(rng.standard_normal((n, k, N)))
    trees = np.zeros((n - 1, N), dtype=bool, order="F")
    trees[0, :] = True
    bad = trees.copy()
    bad[4, 2] = True                                                   # node 5 without its parent (node 2): Utils.jl:209
    with pytest.raises(AssertionError):
        wx.getbasiscoefall(Xw, bad)
    deep = trees.copy()
    deep[:7, 3] = True                                                 # depth 3 leaves need a fourth column: Utils.jl:120
    with pytest.raises(wx.ArgumentError):
        wx.getbasiscoefall(Xw, deep)
    with pytest.raises(AssertionError):                                # m == m_t (Utils.jl:210)
        wx.getbasiscoefall(Xw, trees[:, :3])
