"""Float32 ARITHMETIC on pairs of signals (round 5: lat_f2v in csrc/wx_lattice_dev.h, launchers wx_lattice_sg32.h / wx_lattice_tree32.h):
a wavefront takes two sets of 2^SH signals.  What is specific to the pairing: every batch remainder (the lone last signal of an odd batch is
both halves of its pair; remainders of at least 2^SH signals overlap inside the last wavefront; smaller ones re-do signals out of place),
in-place calls (x === y through the `!` forms), one signal, strided inputs (iwpd reads column L of the table), and the Float32 semantics of
the reference (rounding at every accumulate, dwt/dwt_one_level.jl:97-103): <= 1e-5 against the oracle's Float32 instantiation."""
import numpy as np
import pytest

from helpers import relerr, random_tree_1d

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _inplace(wx, name, xd, wt, L_or_tree):
    """the batched entry with y === x (what `wpt!(x, x, ...)` over a batch does through the C ABI): (n, B) is a batch of 1-D signals"""
    import importlib
    dwt = importlib.import_module("waveletsext_jl_amd.dwt")          # (the package attribute `dwt` is the function)
    from waveletsext_jl_amd._arrays import Arg
    L, tree = dwt._split_Ltree(L_or_tree, wx.maxtransformlevels(xd.shape[0]))
    a = Arg(xd)
    dwt._wpt_batched(name, a, a, 1, wt, L, tree)


@pytest.mark.parametrize("n", [4096, 2048, 1024, 256, 64])
def test_every_remainder_full_trees(wx, oracle, n):
    wt = wx.wavelet(wx.WT.db4)
    L = wx.maxtransformlevels(n)
    half = 4096 // n
    rng = np.random.default_rng(n)
    for B in sorted({1, 2, 3, half, half + 1, 2 * half - 1, 2 * half, 2 * half + 1, 3 * half - 1, 3 * half, 5 * half + half // 2 + 1}):
        x = np.asfortranarray(rng.standard_normal((n, B)).astype(np.float32))
        exp = oracle.wptall(x, wt.qmf, L)
        y = wx.wptall(x, wt, L)
        assert y.dtype == np.float32 and relerr(y, exp) <= TOL, (n, B)
        assert relerr(wx.iwptall(exp, wt, L), x) <= TOL, (n, B)
        # in place on the device: the `!` forms with y === x (declined by the pair kernels only for remainders below 2^SH signals,
        # which keep the round-4 path)
        xd = wx.to_device(x)
        _inplace(wx, "wx_wpt", xd, wt, L)
        assert relerr(xd.cpu().numpy(), exp) <= TOL, (n, B, "in place")
        _inplace(wx, "wx_iwpt", xd, wt, L)
        assert relerr(xd.cpu().numpy(), x) <= TOL, (n, B, "in place inverse")


@pytest.mark.parametrize("wname", ["db3", "db8"])
@pytest.mark.parametrize("n", [4096, 2048, 1024])
def test_every_remainder_trees(wx, oracle, n, wname):
    wt = wx.wavelet(getattr(wx.WT, wname))
    half = 4096 // n
    rng = np.random.default_rng(n + 1)
    tree = random_tree_1d(n, rng, p=0.75)
    while not (tree[0] and tree[1:3].all() and tree[3:7].any()):
        tree = random_tree_1d(n, rng, p=0.75)
    shallow = wx.maketree(n, 1, "full")                               # depth 1: the tree-driven kernels' shortest run
    for B in sorted({1, 2, 3, half + 1, 2 * half + 1, 3 * half, 4 * half + half // 2 + 1, 67}):
        x = np.asfortranarray(rng.standard_normal((n, B)).astype(np.float32))
        e1 = oracle.wptall(x, wt.qmf, shallow)
        assert relerr(wx.wptall(x, wt, shallow), e1) <= TOL, (n, B, "depth 1")
        assert relerr(wx.iwptall(e1, wt, shallow), x) <= TOL, (n, B, "depth 1 inverse")
        exp = oracle.wptall(x, wt.qmf, tree)
        y = wx.wptall(x, wt, tree)
        assert relerr(y, exp) <= TOL, (n, B)
        assert relerr(wx.iwptall(exp, wt, tree), x) <= TOL, (n, B)
        xd = wx.to_device(x)
        _inplace(wx, "wx_wpt", xd, wt, tree)
        assert relerr(xd.cpu().numpy(), exp) <= TOL, (n, B, "in place")
        _inplace(wx, "wx_iwpt", xd, wt, tree)
        assert relerr(xd.cpu().numpy(), x) <= TOL, (n, B, "in place inverse")


@pytest.mark.parametrize("wname", ["haar", "db2", "db5", "db8", "coif6", "db10"])
def test_filters_and_strided_columns(wx, oracle, wname):
    """every filter length the 4096-sample pair kernels take (2 .. 20 taps), and iwpd reading column L of a Float32 table (signal stride
    n (L + 1): the second signal of a pair sits one table further)"""
    wt = wx.wavelet(getattr(wx.WT, wname))
    rng = np.random.default_rng(len(wt.qmf))
    for n, L, B in ((4096, 12, 5), (4096, 7, 4), (1024, 10, 9)):
        x = np.asfortranarray(rng.standard_normal((n, B)).astype(np.float32))
        exp = oracle.wptall(x, wt.qmf, L)
        assert relerr(wx.wptall(x, wt, L), exp) <= TOL, (wname, n, L)
        tab = np.asfortranarray(np.stack([oracle.wpd(x[:, b], wt.qmf, L) for b in range(B)], axis=-1))
        assert tab.dtype == np.float32
        assert relerr(wx.iwpdall(tab, wt, L), x) <= TOL, (wname, n, L, "iwpd")


def test_pairs_do_not_mix_signals(wx):
    """a large device batch where every signal has its own scale: energies per signal (orthonormality) and the round trip; a pair kernel
    that swapped or blended the two halves of a pair would show at 1e-1, not 1e-5"""
    import torch
    wt = wx.wavelet(wx.WT.db4)
    for n, B in ((4096, 4097), (1024, 8191), (64, (1 << 16) + 33)):
        x = wx.jl_empty((n, B), torch.float32, "cuda")
        x.normal_()
        x *= torch.logspace(-2, 2, B, device="cuda", dtype=torch.float32)[None, :]
        L = wx.maxtransformlevels(n)
        y = wx.wptall(x, wt, L)
        ex, ey = (x.double() ** 2).sum(0), (y.double() ** 2).sum(0)
        assert float(((ey - ex).abs() / ex).max()) <= 2e-5
        back = wx.iwptall(y, wt, L)
        assert float(((back - x).abs().amax(0) / x.abs().amax(0)).max()) <= TOL
