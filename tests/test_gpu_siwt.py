"""GPU parity: shift-invariant wavelet packet decomposition through the C ABI vs the CPU oracle
(oracle.siwpd / siwt_bestbasistree / isiwpd, the Dict-and-recursion restatement of SIWT.jl).  Node values:
bit-identical for Float64 (same multiply / add / round sequence), <= 1e-6 relative for Float32; costs <= 1e-10
(Float64) / 1e-4 (Float32: sums of up to n rounded terms); best trees identical whenever the oracle's decisions
are not ties at rounding level (checked through MinCost otherwise); reconstruction <= 1e-10 / 1e-5."""
import numpy as np
import pytest

from helpers import relerr

pytestmark = pytest.mark.gpu

CASES = [("haar", 4, 2, 2), ("haar", 32, 5, 5), ("db2", 32, 5, 5), ("db4", 64, 6, 3), ("coif6", 64, 4, 1),
         ("db4", 48, 4, 4), ("db10", 128, 5, 2), ("db2", 512, 6, 6), ("haar", 24, 3, 2), ("db4", 256, 5, 5), ("db3", 128, 4, 3)]


def _wt(wx, name):
    return wx.wavelet(getattr(wx.WT, name))


def _nodes_equal(obj, ref, dtype):
    assert set(obj.BestTree) == set(ref.Nodes.keys())
    tolv = 0.0 if dtype == np.float64 else 1e-6
    tolc = 1e-10 if dtype == np.float64 else 1e-4
    for key, nd in ref.Nodes.items():
        got = obj.Nodes[key]
        v = np.asarray(got.Value.cpu().numpy() if hasattr(got.Value, "cpu") else got.Value)
        scale = max(1.0, float(np.abs(nd["Value"]).max()))
        assert float(np.abs(v - nd["Value"]).max()) <= tolv * scale, key
        assert abs(got.Cost - nd["Cost"]) <= tolc * max(1.0, abs(nd["Cost"])), key


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("wname,n,L,d", CASES)
def test_siwpd_nodes_costs_and_order(wx, oracle, wname, n, L, d, dtype):
    rng = np.random.default_rng(n * 31 + L)
    wt = _wt(wx, wname)
    x = rng.standard_normal(n).astype(dtype)
    obj = wx.siwpd(x, wt, L, d)
    ref = oracle.siwpd(x, wt.qmf, L, d)
    assert obj.BestTree == ref.BestTree                                # the reference's push order
    assert (obj.SignalSize, obj.MaxTransformLevel, obj.MaxShiftedTransformLevels) == (n, L, d)
    _nodes_equal(obj, ref, dtype)
    assert abs(obj.MinCost - ref.MinCost) <= 1e-4


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("wname,n,L,d", CASES)
def test_bestbasis_and_reconstruction(wx, oracle, wname, n, L, d, dtype):
    rng = np.random.default_rng(n * 17 + d)
    wt = _wt(wx, wname)
    tol = 1e-10 if dtype == np.float64 else 1e-5
    x = rng.standard_normal(n).astype(dtype)
    obj = wx.siwpd(x, wt, L, d)
    ref = oracle.siwpd(x, wt.qmf, L, d)
    tree = wx.bestbasistree_(obj)
    rtree = oracle.siwt_bestbasistree(ref)
    assert wx.isvalidtree(obj)
    assert abs(obj.MinCost - ref.MinCost) <= (1e-9 if dtype == np.float64 else 1e-4) * max(1.0, abs(ref.MinCost))
    # nodes of one or two samples tie structurally between the two kinds of children (sum of the even taps = sum of
    # the odd taps), haar ties at every scale: there the choice hangs on the last bit of the cost sums
    if dtype == np.float64 and wname != "haar" and (n >> L) >= 4:
        assert tree == rtree
        _nodes_equal(obj, ref, dtype)
    xr = wx.isiwpd(obj)
    assert relerr(xr, x) <= tol
    assert obj.BestTree == [(0, 0, 0)]                                 # children merged and deleted (SIWT.jl:223-226)
    assert relerr(oracle.isiwpd(ref), x) <= tol


def test_reference_known_answers(wx):
    """test/transforms.jl:177-267: signal [2,3,-4,5], haar"""
    wt = _wt(wx, "haar")
    signal = np.array([2, 3, -4, 5.0])
    root = wx.ShiftInvariantWaveletTransformObject(signal, wt)
    assert root.SignalSize == 4 and root.MaxTransformLevel == 0 and root.MaxShiftedTransformLevels == 0
    assert root.BestTree == [(0, 0, 0)] and abs(root.MinCost - 1.208) <= 1e-3
    for bad in ((3, 0), (-1, 0), (0, 4), (0, -1)):
        with pytest.raises(wx.ArgumentError):
            wx.ShiftInvariantWaveletTransformObject(signal, wt, *bad)
    assert wx.bestbasistree_(root) == [(0, 0, 0)] and wx.isvalidtree(root)
    with pytest.raises(wx.ArgumentError):
        wx.ShiftInvariantWaveletTransformNode(2, 4, 0, 0.0, signal)
    with pytest.raises(wx.ArgumentError):
        wx.ShiftInvariantWaveletTransformNode(2, 0, 4, 0.0, signal)
    with pytest.raises(wx.ArgumentError):
        wx.ShiftInvariantWaveletTransformNode.from_data(np.zeros((4, 4)), 0, 0, 0)
    nd = wx.ShiftInvariantWaveletTransformNode.from_data(signal, 0, 0, 0)
    assert abs(nd.Cost - 1.208) <= 1e-3
    obj = wx.siwpd(signal, wt, 1)
    exp = {(0, 0, 0): 1.208, (1, 0, 0): 0.382, (1, 0, 1): 0.402, (1, 1, 0): 0.259, (1, 1, 1): 0.566}
    for k, c in exp.items():
        assert abs(obj.Nodes[k].Cost - c) <= 1e-3
    s2 = np.sqrt(2.0)
    assert np.allclose(obj.Nodes[(1, 0, 1)].Value, np.array([7, -1]) / s2)        # dwt(circshift(signal, 1))[1:2]
    wx.bestbasistree_(obj)
    exp = {(0, 0, 0): 0.641, (1, 0, 0): 0.382, (1, 1, 0): 0.259}
    assert set(obj.BestTree) == set(exp) == set(obj.Nodes.keys())
    for k, c in exp.items():
        assert abs(obj.Nodes[k].Cost - c) <= 1e-3
    assert abs(obj.MinCost - 0.641) <= 1e-3 and wx.isvalidtree(obj)
    obj = wx.siwpd(signal, wt)
    wx.bestbasistree_(obj)
    assert np.allclose(wx.isiwpd(obj), signal, atol=1e-12)
    with pytest.raises(AssertionError):
        wx.siwpd(signal, wt, 3)
    with pytest.raises(AssertionError):
        wx.siwpd(signal, wt, 2, 3)


def test_batch_on_device_matches_single_signals(wx, oracle):
    import torch
    wt = _wt(wx, "db4")
    rng = np.random.default_rng(4)
    n, L, d, B = 256, 6, 4, 37
    X = np.asfortranarray(rng.standard_normal((n, B)))
    batch = wx.siwpdall(wx.to_device(X), wt, L, d)
    assert isinstance(batch.Table, torch.Tensor) and tuple(batch.Table.shape) == (n, sum(1 << min(j, d) for j in range(L + 1)), B)
    st = wx.bestbasistreeall_(batch)
    for b in (0, 5, B - 1):
        ref = oracle.siwpd(X[:, b], wt.qmf, L, d)
        rtree = oracle.siwt_bestbasistree(ref)
        assert batch[b].BestTree == rtree
        assert abs(batch.MinCost[b] - ref.MinCost) <= 1e-9
    assert st.shape[1] == B and set(np.unique(st)) <= {0, 1, 2, 3}
    xr = wx.isiwpdall(batch)
    assert relerr(wx.to_numpy(xr), X) <= 1e-10
    # shift invariance (the point of the transform): the best-basis cost does not depend on the rotation
    Xs = np.asfortranarray(np.stack([np.roll(X[:, 0], k) for k in range(8)], axis=1))
    bs = wx.siwpdall(wx.to_device(Xs), wt, L, L)
    wx.bestbasistreeall_(bs)
    assert float(np.ptp(bs.MinCost)) <= 1e-9


def test_literal_readings(wx, oracle):
    """literal=True reproduces the reference's code where it disagrees with its own tests"""
    wt = _wt(wx, "haar")
    signal = np.array([2, 3, -4, 5.0])
    obj = wx.siwpd(signal, wt)
    wx.bestbasistree_(obj)
    assert np.allclose(wx.isiwpd(obj, literal=True), np.roll(signal, -1), atol=1e-12)
    rng = np.random.default_rng(9)
    wt = _wt(wx, "db4")
    X = np.asfortranarray(rng.standard_normal((64, 3)))
    batch = wx.siwpdall(X, wt, 4, 4)
    wx.bestbasistreeall_(batch)
    refs = []
    for b in range(3):
        ref = oracle.siwpd(X[:, b], wt.qmf, 4, 4)
        oracle.siwt_bestbasistree(ref)
        assert wx.isvalidtree(batch[b]) and wx.isvalidtree(batch[b], literal=True) == oracle.siwt_isvalidtree(ref, literal=True)
        refs.append(oracle.isiwpd(ref, literal=True))
    got = wx.isiwpdall(batch, literal=True)
    assert relerr(got, np.asfortranarray(np.stack(refs, axis=1))) <= 1e-10
