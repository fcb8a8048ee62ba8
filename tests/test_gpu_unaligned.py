"""GPU parity for inputs and outputs the fast kernels' preconditions do not like: device pointers that
are offset by ONE element (8 / 4 bytes: what a Julia `@view x[2:end]` of a column or a sub-array of a batch is -- the reference takes
any view, dwt/dwt_all.jl:277), in place where the C ABI allows it.  Every 1-D / 2-D family goes through its normal dispatch; a launcher
that declines such a pointer must leave the call on a slower path, never on an error (`WX_EHIP`, "did not take a subtree").
Outputs are made unaligned by patching the package's allocator for the duration of a test, so the `*all` drivers -- which allocate
their own results -- are covered too.  Tolerances as everywhere: 1e-10 Float64, 1e-5 Float32 against the oracle."""
import numpy as np
import pytest

from helpers import TOL, random_tree_1d, random_tree_2d, relerr

pytestmark = pytest.mark.gpu


def _off(a):
    """numpy (Julia shape) -> column-major device tensor whose first element sits one element past an aligned allocation"""
    import torch
    a = np.asfortranarray(a)
    flat = torch.empty(a.size + 1, dtype=torch.float64 if a.dtype == np.float64 else torch.float32, device="cuda")
    flat[1:].copy_(torch.from_numpy(np.ascontiguousarray(a.T).reshape(-1)))
    t = flat[1:].view(tuple(reversed(a.shape)))
    t = t.permute(*reversed(range(a.ndim))) if a.ndim > 1 else t
    assert t.data_ptr() % 16 in (4, 8)
    return t


@pytest.fixture()
def unaligned_outputs(wx, monkeypatch):
    """every result tensor the package allocates starts one element past an aligned address"""
    import torch
    import sys
    arrays = sys.modules[wx.__name__ + "._arrays"]

    def jl_empty_off(shape, dtype, device):
        shape = tuple(int(s) for s in shape)
        n = int(np.prod(shape)) if shape else 1
        flat = torch.empty(n + 1, dtype=dtype, device=device)
        t = flat[1:].view(tuple(reversed(shape)))
        return t.permute(*reversed(range(len(shape)))) if len(shape) > 1 else t

    monkeypatch.setattr(arrays, "jl_empty", jl_empty_off)
    yield
    return


def _np(t):
    return np.asfortranarray(t.detach().cpu().numpy())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("wname", ["haar", "db4", "db8"])
def test_1d_packets_of_offset_pointers(wx, oracle, unaligned_outputs, wname, dtype):
    rng = np.random.default_rng(61)
    wt = wx.wavelet(getattr(wx.WT, wname))
    tol = TOL[np.dtype(dtype)]
    for n, B in ((64, 130), (256, 67), (1024, 9), (2048, 5), (4096, 4), (8192, 3), (16384, 2), (96, 5)):
        x = np.asfortranarray(rng.standard_normal((n, B)).astype(dtype))
        xd = _off(x)
        Lmax = wx.maxtransformlevels(n)
        for L in sorted({1, min(4, Lmax), Lmax}):
            # full trees: wptall / iwptall (dyadic lengths: Wavelets.jl's maketree), wpdall / iwpdall
            if n & (n - 1) == 0:
                exp = oracle.wptall(x, wt.qmf, L)
                got = wx.wptall(xd, wt, L)
                assert got.data_ptr() % 16 in (4, 8)
                assert relerr(_np(got), exp) <= tol, ("wptall", n, L)
                back = wx.iwptall(_off(exp.astype(dtype)), wt, L)
                assert relerr(_np(back), x) <= 10 * tol, ("iwptall", n, L)
            tab = wx.wpdall(xd, wt, L)
            expt = oracle.wpdall(x, wt.qmf, L)
            assert relerr(_np(tab), expt) <= tol, ("wpdall", n, L)
            if n & (n - 1) == 0:
                back = wx.iwpdall(_off(expt.astype(dtype)), wt, L)
                assert relerr(_np(back), x) <= 10 * tol, ("iwpdall", n, L)
        if n & (n - 1):
            continue
        # pyramids and random trees
        exp = oracle.wptall(x, wt.qmf, oracle.maketree1d(n, Lmax, "dwt"))
        got = wx.dwtall(xd, wt, Lmax)
        assert relerr(_np(got), exp) <= tol, ("dwtall", n)
        back = wx.idwtall(_off(exp.astype(dtype)), wt, Lmax)
        assert relerr(_np(back), x) <= 10 * tol, ("idwtall", n)
        for _ in range(2):
            tree = random_tree_1d(n, rng)
            exp = oracle.wptall(x, wt.qmf, tree)
            got = wx.wptall(xd, wt, tree)
            assert relerr(_np(got), exp) <= tol, ("wptall tree", n)
            back = wx.iwptall(_off(exp.astype(dtype)), wt, tree)
            assert relerr(_np(back), x) <= 10 * tol, ("iwptall tree", n)
            back = wx.iwpdall(_off(oracle.wpdall(x, wt.qmf, Lmax).astype(dtype)), wt, tree)
            assert relerr(_np(back), x) <= 10 * tol, ("iwpdall tree", n)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_1d_in_place_on_offset_pointers(wx, oracle, dtype):
    """wpt! / iwpt! with y === x (the C ABI allows it) on an offset pointer"""
    rng = np.random.default_rng(62)
    wt = wx.wavelet(wx.WT.db4)
    tol = TOL[np.dtype(dtype)]
    for n in (256, 1024, 4096, 8192, 16384):
        x = rng.standard_normal(n).astype(dtype)
        for tree in (oracle.maketree1d(n, wx.maxtransformlevels(n), "full"), random_tree_1d(n, rng)):
            xd = _off(x)
            exp = oracle.wpt(x, wt.qmf, tree)
            wx.wpt_(xd, xd, wt, tree)
            assert relerr(_np(xd), exp) <= tol, ("wpt! in place", n)
            wx.iwpt_(xd, xd, wt, tree)
            assert relerr(_np(xd), x) <= 10 * tol, ("iwpt! in place", n)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_redundant_transforms_of_offset_pointers(wx, oracle, unaligned_outputs, dtype):
    rng = np.random.default_rng(63)
    tol = TOL[np.dtype(dtype)]
    for wname in ("haar", "db4"):
        wt = wx.wavelet(getattr(wx.WT, wname))
        for n, B, L in ((256, 5, 4), (1024, 3, 10), (4096, 2, 6), (16384, 2, 4)):
            x = np.asfortranarray(rng.standard_normal((n, B)).astype(dtype))
            xd = _off(x)
            for fwd, inv, ofwd, oinv in ((wx.sdwtall, wx.isdwtall, oracle.sdwt, oracle.isdwt), (wx.swptall, wx.iswptall, oracle.swpt, oracle.iswpt),
                                         (wx.swpdall, wx.iswpdall, oracle.swpd, oracle.iswpd)):
                if fwd is wx.swpdall and L > 6:
                    continue
                exp = np.stack([ofwd(x[:, b], wt.qmf, L) for b in range(B)], axis=2)
                got = fwd(xd, wt, L)
                assert relerr(_np(got), exp) <= tol, (fwd.__name__, wname, n, L)
                back = inv(_off(exp.astype(dtype)), wt, L) if inv is wx.iswpdall else inv(_off(exp.astype(dtype)), wt)
                assert relerr(_np(back), x) <= 10 * tol, (inv.__name__, wname, n, L)
            if dtype == np.float64:
                for fwd, inv, ofwd in ((wx.acdwtall, wx.iacdwtall, oracle.acdwt), (wx.acwptall, wx.iacwptall, oracle.acwpt)):
                    exp = np.stack([ofwd(x[:, b], wt.qmf, L) for b in range(B)], axis=2)
                    got = fwd(xd, wt, L)
                    assert relerr(_np(got), exp) <= tol, (fwd.__name__, wname, n, L)
                    back = inv(_off(exp))
                    assert relerr(_np(back), x) <= 10 * tol, (inv.__name__, wname, n, L)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("wname", ["haar", "db4", "db8"])
def test_2d_packets_of_offset_pointers(wx, oracle, unaligned_outputs, wname, dtype):
    rng = np.random.default_rng(64)
    wt = wx.wavelet(getattr(wx.WT, wname))
    tol = 2e-6 if dtype == np.float32 else 1e-10
    for m, B in ((64, 5), (128, 3), (256, 4), (512, 2), (1024, 1)):
        if m == 1024 and dtype == np.float64:
            continue
        x = np.asfortranarray(rng.standard_normal((m, m, B)).astype(dtype))
        xd = _off(x)
        Lmax = wx.maxtransformlevels(m)
        lattice_depth = {256: 5, 512: 6, 1024: 7}.get(m, 3)
        for L in sorted({2, lattice_depth, Lmax}):
            exp = oracle.wptall(x.astype(np.float64), wt.qmf, L)
            got = wx.wptall(xd, wt, L)
            assert relerr(_np(got), exp) <= tol, ("wptall 2-D", m, L)
            back = wx.iwptall(_off(exp.astype(dtype)), wt, L)
            assert relerr(_np(back), x) <= 10 * tol, ("iwptall 2-D", m, L)
        if m <= 256:
            tree = random_tree_2d(m, m, rng)
            exp = oracle.wptall(x.astype(np.float64), wt.qmf, tree)
            got = wx.wptall(xd, wt, tree)
            assert relerr(_np(got), exp) <= tol, ("wptall quad tree", m)
            back = wx.iwptall(_off(exp.astype(dtype)), wt, tree)
            assert relerr(_np(back), x) <= 10 * tol, ("iwptall quad tree", m)
            L = 3
            expt = oracle.wpdall(x.astype(np.float64), wt.qmf, L)
            got = wx.wpdall(xd, wt, L)
            assert relerr(_np(got), expt) <= tol, ("wpdall 2-D", m)
            back = wx.iwpdall(_off(expt.astype(dtype)), wt, L)
            assert relerr(_np(back), x) <= 10 * tol, ("iwpdall 2-D", m)


@pytest.mark.parametrize("inputtype", ["sig", "dwt", "sdwt"])
def test_in_out_arrays_on_offset_pointers(wx, oracle, unaligned_outputs, inputtype):
    """the callers' side mutates arrays in place (threshold!, the costs of the tree selection, the moment sums that accumulate over
    chunks): with every array of the pipeline at an offset pointer the in / out ones pass through the aligned copy BOTH ways.  denoiseall
    on device tensors == the same call on host arrays (checked against the oracle in test_gpu_denoise.py); bestbasistree(JBB) and
    bestbasistreeall(BB) of an offset table == the trees of the aligned one"""
    rng = np.random.default_rng(65)
    wt = wx.wavelet(wx.WT.db4)
    for n, B in ((256, 9), (1024, 4), (4096, 3)):
        x = np.asfortranarray(rng.standard_normal((n, B)) + 3 * np.sin(np.arange(n) / 7.0)[:, None])
        xin = {"sig": x, "dwt": np.asarray(wx.dwtall(x, wt)), "sdwt": np.asarray(wx.sdwtall(x, wt, 4))}[inputtype]
        ref = wx.denoiseall(xin, inputtype, wt)
        got = wx.denoiseall(_off(xin), inputtype, wt)
        assert got.data_ptr() % 16 in (4, 8)
        assert relerr(_np(got), ref) <= 1e-12, (inputtype, n)
    n, B = 512, 6
    x = np.asfortranarray(rng.standard_normal((n, B)))
    tab = wx.wpdall(x, wt)
    assert (np.asarray(wx.bestbasistree(_off(np.asarray(tab)), wx.JBB())) == np.asarray(wx.bestbasistree(tab, wx.JBB()))).all()
    assert (np.asarray(wx.to_numpy(wx.bestbasistreeall(_off(np.asarray(tab)), wx.BB()))) == np.asarray(wx.bestbasistreeall(tab, wx.BB()))).all()


def test_no_hard_error_string_left_reachable():
    """the messages of the old hard errors are gone from the library's sources"""
    import os
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "waveletsext.jl_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")):
            txt = open(os.path.join(csrc, f)).read()
            assert "did not take a subtree" not in txt, f
