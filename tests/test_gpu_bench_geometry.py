"""GPU parity at the EXACT signal geometry of the benchmarked configurations (size-dependent kernel dispatch: K-level
fused passes, residue-class tiles, the D0 = 6 subtree kernel, chunk accumulation), batch reduced so that the oracle
finishes in seconds.  Tolerances: 1e-10 relative (north_star), JBB trees bit-exact."""
import numpy as np
import pytest

from helpers import relerr

pytestmark = pytest.mark.gpu


def test_config3_geometry_swpt_iswpt(wx, oracle):
    """BASELINE config 3: n = 16384, haar, L = 12 -- k_swt_fwd_multi / k_swt_fwd_multi_rc<double,8,8> forward passes and
    k_swt_inv_multi<double,8,8> average-based inverse (SWT.jl:439-472, 613-712; swt/swt_one_level.jl:99-127, 257-318)"""
    import torch
    rng = np.random.default_rng(3)
    wt = wx.wavelet(wx.WT.haar)
    n, L, B = 16384, 12, 2
    x = np.asfortranarray(rng.standard_normal((n, B)))
    xd = wx.to_device(x)
    xw = wx.swptall(xd, wt, L)                                        # (n, 4096, B): 1 GiB on the device
    assert tuple(xw.shape) == (n, 1 << L, B)
    for b in range(B):
        ref = oracle.swpt(x[:, b], wt.qmf, L)
        got = xw[:, :, b].cpu().numpy()
        assert relerr(got, ref) <= 1e-10, b
        if b == 0:
            back_ref = oracle.iswpt(ref, wt.qmf)
            assert np.abs(back_ref - x[:, 0]).max() <= 1e-10
            refd = wx.to_device(np.asfortranarray(ref[:, :, None]))
            back = wx.iswptall(refd, wt)
            assert relerr(back[:, 0].cpu().numpy(), back_ref) <= 1e-10
    xr = wx.iswptall(xw, wt)
    assert float((xr - xd).abs().max()) <= 1e-10
    del xw, xr
    torch.cuda.empty_cache()


def test_config3_chunk_properties(wx):
    """one resident chunk of the bench (64 signals, 32 GiB of leaves): reconstruction, and the energy identity of the
    undecimated packets (|G|^2 + |H|^2 = 2 for an orthonormal QMF pair: every level doubles the total energy)"""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 40 * 2 ** 30:
        pytest.skip("needs 40 GiB of free HBM")
    wt = wx.wavelet(wx.WT.haar)
    n, L, B = 16384, 12, 64
    x = wx.jl_empty((n, B), torch.float64, "cuda")
    x.normal_()
    xw = wx.swptall(x, wt, L)
    e_in = (x * x).sum(dim=0)
    e_out = torch.zeros_like(e_in)
    for c0 in range(0, 1 << L, 256):
        e_out += (xw[:, c0:c0 + 256, :] ** 2).sum(dim=(0, 1))
    assert float(((e_out / (1 << L) - e_in).abs() / e_in).max()) <= 1e-10
    xr = wx.iswptall(xw, wt)
    assert float((xr - x).abs().max() / x.abs().max()) <= 1e-10
    del xw, xr, x
    torch.cuda.empty_cache()


@pytest.mark.parametrize("mode", [0, 1])
def test_config5_geometry_moments_and_tree(wx, oracle, mode):
    """BASELINE config 5: n = 2048, coif6, L = 11 -- mode 0: the D0 = 6 residue-class subtree kernel
    (k_acwpd_subtree_moments<5,4,9>), mode 1: the materialised table + k_jbb_moments; moments <= 1e-12 relative,
    tree identical to the oracle's (ACWT.jl:733-759, bestbasis/bestbasis_tree.jl:150-180)"""
    rng = np.random.default_rng(5)
    wt = wx.wavelet(wx.WT.coif6)
    n, L, B = 2048, 11, 16
    x = np.asfortranarray(rng.standard_normal((n, B)))
    X = np.asfortranarray(np.stack([oracle.acwpd(x[:, b], wt.qmf, L) for b in range(B)], axis=-1))
    s_ref, q_ref = X.sum(axis=2), (X ** 2).sum(axis=2)
    tree_ref = oracle.bestbasistree_jbb(X, redundant=True)
    wx.set_force_generic(mode)
    try:
        s, q = wx.acwpd_jbb_moments(x, wt, L)
        assert relerr(s, s_ref) <= 1e-12 and relerr(q, q_ref) <= 1e-12
        costs = wx.costs_from_moments(s, q, B, wx.JBB(redundant=True))
        assert (wx.bestbasis_treeselection(costs, n) == tree_ref).all()
        # the bench's chunk loop: the same moments accumulated over 16 chunks (here: of one signal each)
        xd = wx.to_device(x)
        sa = qa = None
        for b in range(B):
            if sa is None:
                sa, qa = wx.acwpd_jbb_moments(xd[:, b:b + 1], wt, L)
            else:
                wx.acwpd_jbb_moments(xd[:, b:b + 1], wt, L, accumulate_into=(sa, qa))
        assert relerr(sa.cpu().numpy(), s_ref) <= 1e-12 and relerr(qa.cpu().numpy(), q_ref) <= 1e-12
        costs2 = wx.costs_from_moments(sa, qa, B, wx.JBB(redundant=True))
        assert (wx.bestbasis_treeselection(costs2, n) == tree_ref).all()
    finally:
        wx.set_force_generic(0)
