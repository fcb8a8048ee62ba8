"""GPU parity: standard (per-signal) best basis BB -- costs, batched tree selection, bestbasistreeall --
against the CPU oracle (bestbasis_tree.jl:210-258, BestBasis.jl:59-110,203-262).  Costs: relative 1e-10
(Float64) / 1e-4 (Float32: the oracle accumulates sequentially in Float32 like the reference, the device in
Float64 partials); trees compared with ==."""
import numpy as np
import pytest

from helpers import relerr

pytestmark = pytest.mark.gpu

CTOL = {np.dtype(np.float64): 1e-10, np.dtype(np.float32): 1e-4}


def _wt(wx, name):
    return wx.wavelet(getattr(wx.WT, name))


def _cost(wx, name):
    return wx.ShannonEntropyCost() if name == "shannon" else wx.LogEnergyEntropyCost()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("cost", ["shannon", "logenergy"])
def test_bb_1d(wx, oracle, dtype, cost):
    rng = np.random.default_rng(3001)
    wt = _wt(wx, "db4")
    for n, B in ((64, 5), (128, 7), (32, 9), (256, 3), (1024, 2)):      # up to 256 samples: k_bb_costs1d_short (rows of several signals per workgroup)
        x = np.asfortranarray(rng.standard_normal((n, B)).astype(dtype))
        x[:, 0] = np.cumsum(x[:, 0])                                     # a smooth-ish signal: non-trivial tree
        Xw = wx.wpdall(x, wt)
        m = wx.BB(cost=_cost(wx, cost))
        for i in range(B):
            c = wx.tree_costs(Xw[:, :, i], m)
            assert relerr(c, oracle.tree_costs_bb(Xw[:, :, i], False, cost)) <= CTOL[np.dtype(dtype)]
        trees = wx.bestbasistreeall(Xw, m)
        exp = oracle.bestbasistreeall_bb(Xw, False, cost)
        assert trees.shape == exp.shape == (n - 1, B)
        assert (trees == exp).all()
        for i in range(B):
            assert wx.isvalidtree(x[:, i], trees[:, i])
            assert (wx.bestbasistree(Xw[:, :, i], m) == exp[:, i]).all()
        # redundant tables (swpd / acwpd heap layout): cost of node i divided by 2^depth
        Xs = wx.swpdall(x, wt, 4)
        mr = wx.BB(cost=_cost(wx, cost), redundant=True)
        assert relerr(wx.tree_costs(Xs[:, :, 0], mr), oracle.tree_costs_bb(Xs[:, :, 0], True, cost)) <= CTOL[np.dtype(dtype)]
        assert (wx.bestbasistreeall(Xs, mr) == oracle.bestbasistreeall_bb(Xs, True, cost)).all()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_bb_2d(wx, oracle, dtype):
    rng = np.random.default_rng(3002)
    wt = _wt(wx, "haar")
    for (n, m), B in (((16, 16), 3), ((32, 16), 2)):
        x = np.asfortranarray(rng.standard_normal((n, m, B)).astype(dtype))
        x[:, :, 0] = np.cumsum(np.cumsum(x[:, :, 0], axis=0), axis=1)
        Xw = wx.wpdall(x, wt)
        for cost in ("shannon", "logenergy"):
            mth = wx.BB(cost=_cost(wx, cost))
            c = wx.tree_costs(Xw[:, :, :, 0], mth)
            assert relerr(c, oracle.tree_costs_bb(Xw[:, :, :, 0], False, cost)) <= CTOL[np.dtype(dtype)]
            trees = wx.bestbasistreeall(Xw, mth)
            assert (trees == oracle.bestbasistreeall_bb(Xw, False, cost)).all()
            assert wx.isvalidtree(x[:, :, 0], trees[:, 0])
        Xs = wx.swpdall(x, wt, 2)
        mr = wx.BB(redundant=True)
        assert relerr(wx.tree_costs(Xs[:, :, :, 1], mr), oracle.tree_costs_bb(Xs[:, :, :, 1], True)) <= CTOL[np.dtype(dtype)]
        assert (wx.bestbasistreeall(Xs, mr) == oracle.bestbasistreeall_bb(Xs, True)).all()


def test_bb_device_tensors_and_edge_cases(wx, oracle):
    import torch
    rng = np.random.default_rng(3003)
    wt = _wt(wx, "db2")
    x = np.asfortranarray(rng.standard_normal((512, 40)))
    x[:, 7] = 0.0                                                        # nrm == 0 -> all costs 0 -> empty tree
    Xd = wx.wpdall(wx.to_device(x), wt)                                  # stays on the device
    trees = wx.bestbasistreeall(Xd, wx.BB())
    exp = oracle.bestbasistreeall_bb(wx.to_numpy(Xd))
    assert (trees == exp).all() and not trees[:, 7].any()
    # selection on its own: batch of cost vectors, min / max, costs mutated like the reference
    from waveletsext_jl_amd import bestbasis as bb
    costs = np.asfortranarray(rng.random((63, 6)))
    for kind in ("min", "max"):
        carg = bb.Arg(costs.copy(order="F"))
        t = bb._bb_trees(carg, (32,), 6, kind)
        for i in range(6):
            cref = costs[:, i].copy()
            assert (t[i].astype(bool) == wx.bestbasis_treeselection(cref, 32, kind)).all()
    with pytest.raises(AssertionError):
        wx.bestbasistreeall(np.zeros((8, 4)), wx.BB())                   # needs a batch (BestBasis.jl:254)
    with pytest.raises(TypeError):
        wx.bestbasistreeall(np.zeros((8, 4, 2)), wx.JBB())


def test_bb_many_short_signals(wx, oracle):
    """per-signal best bases of a batch of short signals whose table is not a multiple of a workgroup's 256 elements (k_bb_costs1d_short)"""
    rng = np.random.default_rng(3003)
    wt = _wt(wx, "db4")
    for n, B in ((64, 301), (16, 77)):
        x = np.asfortranarray(np.cumsum(rng.standard_normal((n, B)), axis=0))
        Xw = wx.wpdall(x, wt)
        m = wx.BB()
        trees = wx.bestbasistreeall(Xw, m)
        for b in range(0, B, 37):
            assert relerr(wx.tree_costs(Xw[:, :, b], m), oracle.tree_costs_bb(Xw[:, :, b], False, "shannon")) <= 1e-11
        assert (trees[:, ::37] == oracle.bestbasistreeall_bb(Xw[:, :, ::37], False, "shannon")).all()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("cost", ["shannon", "logenergy"])
def test_bb_wave_per_signal_kernels(wx, oracle, dtype, cost):
    """64 ... 512-sample signals: one wavefront per signal computes the norm and every node's cost (k_bb_costs1d_wave) and, up to 256 samples, selects
    the tree (k_bb_treeselect_w) -- batches off the multiples of four, tables of fewer levels than the length admits, an all-zero signal, both costs,
    both types; costs within tolerance of the oracle, trees equal"""
    rng = np.random.default_rng(3100)
    wt = _wt(wx, "db4")
    for n in (64, 128, 256, 512):
        Lmax = wx.maxtransformlevels(n)
        for B, L in ((1, Lmax), (7, Lmax), (61, Lmax - 2), (130, 3)):
            x = np.asfortranarray(np.cumsum(rng.standard_normal((n, B)), axis=0).astype(dtype))
            if B > 3:
                x[:, 3] = 0.0
            Xw = wx.wpdall(x, wt, L)
            m = wx.BB(cost=_cost(wx, cost))
            trees = wx.bestbasistreeall(Xw, m)
            exp = oracle.bestbasistreeall_bb(Xw, False, cost)
            assert trees.shape == exp.shape == (n - 1, B)
            assert (trees == exp).all(), (n, B, L, dtype, cost)
            for i in sorted({0, B // 2, B - 1}):
                c = wx.tree_costs(Xw[:, :, i], m)
                assert relerr(c, oracle.tree_costs_bb(Xw[:, :, i], False, cost)) <= CTOL[np.dtype(dtype)], (n, B, L, i)
