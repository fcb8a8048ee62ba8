"""GPU parity: 1-D decimated wavelet packets (wpd / wpt / iwpt / iwpd / getbasiscoef) through the
C ABI vs the CPU oracle on identical seeded inputs.  Tolerance: 1e-10 relative for Float64
(BASELINE.json north_star), 1e-5 for Float32; asserted as max|gpu-oracle|/max|oracle|."""
import numpy as np
import pytest

from helpers import TOL, random_tree_1d, relerr

pytestmark = pytest.mark.gpu

WAVELETS = ["haar", "db2", "db3", "db4", "db8", "coif6", "db10"]


def _wt(wx, name):
    return wx.wavelet(getattr(wx.WT, name))


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


@pytest.mark.parametrize("force_generic", [0, 1])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("wname", WAVELETS)
def test_wpd_matches_oracle(wx, oracle, wname, dtype, force_generic):
    rng = np.random.default_rng(1002)
    wt = _wt(wx, wname)
    wx.set_force_generic(force_generic)
    try:
        for n, B in ((2, 3), (4, 5), (8, 1), (64, 7), (256, 3), (1024, 2)):
            x = np.asfortranarray(rng.standard_normal((n, B)).astype(dtype))
            for L in sorted({0, 1, wx.maxtransformlevels(n)}):
                got = wx.wpdall(x, wt, L)
                exp = oracle.wpdall(x, wt.qmf, L)
                assert got.shape == (n, L + 1, B) and got.dtype == dtype
                assert relerr(got, exp) <= TOL[np.dtype(dtype)], (n, B, L)
    finally:
        wx.set_force_generic(0)


def test_wpd_non_dyadic_lengths(wx, oracle):
    """wpd!/wpdall accept n = odd * 2^k with L <= k (DWT.jl:137; only `wpd` asserts isdyadic)."""
    rng = np.random.default_rng(5)
    wt = _wt(wx, "db4")
    for n, L in ((12, 2), (24, 3), (96, 5), (6, 1), (10, 1)):
        x = np.asfortranarray(rng.standard_normal((n, 4)))
        assert relerr(wx.wpdall(x, wt, L), oracle.wpdall(x, wt.qmf, L)) <= 1e-10
    with pytest.raises(AssertionError):
        wx.wpd(rng.standard_normal(12), wt, 2)                 # DWT.jl:64 isdyadic


@pytest.mark.parametrize("force_generic", [0, 1])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("wname", ["haar", "db4", "db8", "coif6"])
def test_wpt_iwpt_by_level_and_tree(wx, oracle, wname, dtype, force_generic):
    rng = np.random.default_rng(77)
    wt = _wt(wx, wname)
    tol = TOL[np.dtype(dtype)]
    wx.set_force_generic(force_generic)
    try:
        for n, B in ((2, 2), (8, 3), (64, 5), (512, 2)):
            x = np.asfortranarray(rng.standard_normal((n, B)).astype(dtype))
            Lmax = wx.maxtransformlevels(n)
            trees = [wx.maketree(n, Lmax, "dwt"), random_tree_1d(n, rng), random_tree_1d(n, rng, 0.5),
                     np.zeros(n - 1, dtype=bool)]
            for arg in [None, 0, 1, Lmax] + trees:
                got = wx.wptall(x, wt, arg)
                exp = oracle.wptall(x, wt.qmf, arg)
                assert relerr(got, exp) <= tol, (n, arg)
                back = wx.iwptall(exp, wt, arg)
                assert relerr(back, oracle.iwptall(exp, wt.qmf, arg)) <= tol
                assert relerr(back, x) <= 10 * tol
            # single-signal methods
            assert relerr(wx.wpt(x[:, 0], wt, trees[1]), oracle.wpt(x[:, 0], wt.qmf, trees[1])) <= tol
            y = np.empty(n, dtype=dtype)
            assert wx.wpt_(y, x[:, 0], wt, 1) is y
            assert relerr(y, oracle.wpt(x[:, 0], wt.qmf, 1)) <= tol
    finally:
        wx.set_force_generic(0)


@pytest.mark.parametrize("force_generic", [0, 1])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_iwpd_and_getbasiscoef(wx, oracle, dtype, force_generic):
    rng = np.random.default_rng(31)
    wt = _wt(wx, "db4")
    tol = TOL[np.dtype(dtype)]
    wx.set_force_generic(force_generic)
    try:
        for n, B in ((8, 3), (64, 4), (256, 2)):
            x = np.asfortranarray(rng.standard_normal((n, B)).astype(dtype))
            Lmax = wx.maxtransformlevels(n)
            xw = oracle.wpdall(x, wt.qmf)
            for arg in [None, 2, wx.maketree(n, Lmax, "dwt"), random_tree_1d(n, rng), random_tree_1d(n, rng, 0.4)]:
                got = wx.iwpdall(xw, wt, arg)
                assert relerr(got, oracle.iwpdall(xw, wt.qmf, arg)) <= tol
                assert relerr(got, x) <= 10 * tol
            assert relerr(wx.iwpd(xw[:, :, 0], wt), x[:, 0]) <= 10 * tol
            tree = random_tree_1d(n, rng)
            gb = wx.getbasiscoefall(xw, tree)
            for i in range(B):
                assert (gb[:, i] == oracle.getbasiscoef(xw[:, :, i], tree)).all()
            assert (wx.getbasiscoef(xw[:, :, 0], tree) == gb[:, 0]).all()
            # wpt by tree == getbasiscoef(wpd) (test/transforms.jl:25-30 generalised)
            assert relerr(wx.wptall(x, wt, tree), gb) <= tol
    finally:
        wx.set_force_generic(0)


def test_device_pointer_path_is_async_and_matches(wx, oracle, torch_mod):
    torch = torch_mod
    rng = np.random.default_rng(3)
    wt = _wt(wx, "db8")
    x = np.asfortranarray(rng.standard_normal((4096, 6)))
    xd = wx.to_device(x)
    yd = wx.wpdall(xd, wt)
    assert isinstance(yd, torch.Tensor) and yd.is_cuda and tuple(yd.shape) == (4096, 13, 6)
    y = wx.to_numpy(yd)
    assert relerr(y, oracle.wpdall(x, wt.qmf)) <= 1e-10
    back = wx.iwpdall(yd, wt)
    assert relerr(wx.to_numpy(back), x) <= 1e-10
    leaves = wx.wptall(xd, wt, 10)
    assert relerr(wx.to_numpy(leaves), y[:, 10, :]) <= 1e-12
    # a side stream: results must be ordered on torch's current stream
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        y2 = wx.wpdall(xd, wt)
    s.synchronize()
    assert torch.equal(y2, yd)


def test_full_size_config2_properties(wx, torch_mod):
    """BASELINE config 2 at full size (65536 x 4096 Float64, db8, L=12): size-independent
    properties -- every level of an orthogonal packet table carries the signal energy, wpt(L) is
    column L of wpd, and iwpd(wpd(x)) == x."""
    torch = torch_mod
    n, B, L = 4096, 65536, 12
    free, _ = torch.cuda.mem_get_info()
    need = 8 * n * B * (L + 1 + 3)
    if free < need * 1.05:
        pytest.skip("not enough free HBM for the full-size check")
    wt = _wt(wx, "db8")
    g = torch.Generator(device="cuda").manual_seed(1002)
    x = wx.jl_empty((n, B), torch.float64, "cuda")
    x.normal_(generator=g)
    y = wx.wpdall(x, wt, L)
    assert tuple(y.shape) == (n, L + 1, B)
    e0 = (x * x).sum(dim=0)
    for d in (1, 5, L):
        ed = (y[:, d, :] * y[:, d, :]).sum(dim=0)
        assert float(((ed - e0).abs() / e0).max()) < 1e-11
    assert torch.equal(y[:, 0, :], x)
    leaves = wx.wptall(x, wt, L)
    assert float((leaves - y[:, L, :]).abs().max()) < 1e-12
    back = wx.iwpdall(y, wt, L)
    assert float((back - x).abs().max() / x.abs().max()) < 1e-10
    back2 = wx.iwptall(leaves, wt, L)
    assert float((back2 - x).abs().max() / x.abs().max()) < 1e-10


def test_dwtall_is_the_dwt_tree_packet_transform(wx, oracle):
    """dwt/dwt_all.jl:39-110 through the packet kernels"""
    rng = np.random.default_rng(55)
    wt = _wt(wx, "db4")
    x = np.asfortranarray(rng.standard_normal((64, 5)))
    for L in (None, 0, 1, 3):
        Lv = 6 if L is None else L
        tree = wx.maketree(64, Lv, "dwt")
        y = wx.dwtall(x, wt, L)
        assert relerr(y, oracle.wptall(x, wt.qmf, tree)) <= 1e-10
        assert relerr(wx.idwtall(y, wt, L), x) <= 1e-10
    # pyramid layout: [s_L | d_L | ... | d_1]; one level == one dwt_step
    g, h = oracle.makereverseqmfpair(wt.qmf)
    w1, w2 = oracle.dwt_step(x[:, 0], h, g)
    assert relerr(wx.dwt(x[:, 0], wt, 1), np.concatenate([w1, w2])) <= 1e-12
    img = np.asfortranarray(rng.standard_normal((16, 16, 3)))
    yi = wx.dwtall(img, wt)
    assert relerr(yi[..., 1], oracle.wpt(img[..., 1], wt.qmf, oracle.maketree2d(16, 16, 4, "dwt"))) <= 1e-10
    assert relerr(wx.idwtall(yi, wt), img) <= 1e-10


@pytest.mark.parametrize("wname", ["haar", "db2", "db4", "db8", "coif6", "db10"])
def test_pyramid_deep_levels_in_the_registers_of_a_lane(wx, oracle, wname):
    """dwtall to depths where the approximation has fewer than 64 samples: the tree-driven kernel stops at 64 samples and
    csrc/wx_dwttail.hip finishes the pyramid, one lane per signal (dwt/dwt_all.jl:39-54, dwt/dwt_one_level.jl:79-107);
    more signals than one wavefront, a ragged last wavefront, Float64 and Float32, wpt with the same tree, idwtall back"""
    rng = np.random.default_rng(808)
    wt = _wt(wx, wname)
    for n, B in ((128, 70), (1024, 130), (4096, 67)):
        Lmax = int(np.log2(n))
        x = np.asfortranarray(rng.standard_normal((n, B)))
        for L in range(Lmax - 6 + 1, Lmax + 1):
            exp = oracle.wptall(x, wt.qmf, wx.maketree(n, L, "dwt"))
            y = wx.dwtall(x, wt, L)
            assert relerr(y, exp) <= 1e-12, (wname, n, L)
            assert relerr(wx.to_numpy(wx.wptall(wx.to_device(x), wt, wx.maketree(n, L, "dwt"))), exp) <= 1e-12, (wname, n, L)
            assert relerr(wx.idwtall(y, wt, L), x) <= 1e-10, (wname, n, L)
        x32 = x.astype(np.float32)
        y32 = wx.dwtall(x32, wt)
        assert y32.dtype == np.float32
        assert relerr(y32, oracle.wptall(x32, wt.qmf, wx.maketree(n, Lmax, "dwt"))) <= 1e-5, (wname, n)
    # a tree that is not the pyramid keeps the tree-driven kernel for every level
    tree = wx.maketree(1024, 10, "dwt").copy()
    tree[2] = True                                         # node 3 (the first detail) decomposed once
    x = np.asfortranarray(rng.standard_normal((1024, 9)))
    assert relerr(wx.wptall(x, wt, tree), oracle.wptall(x, wt.qmf, tree)) <= 1e-12


def test_native_rccl_exchange_single_rank(wx):
    """the library's own RCCL entry points (C1 all-gather, C2 all-reduce) with a one-rank communicator:
    plumbing, dtype dispatch and stream ordering; multi-rank behaviour is RCCL's"""
    import torch
    from waveletsext_jl_amd import distributed as D
    uid = D.NativeComm.unique_id()
    assert len(uid) == 128
    comm = D.NativeComm(1, 0, uid)
    try:
        for dt in (torch.float64, torch.float32):
            x = wx.jl_empty((64, 5), dt, "cuda")
            x.normal_()
            full = comm.allgather_batch(x, 5)
            torch.cuda.synchronize()
            assert full.shape == x.shape and torch.equal(full, x)
            fullv = comm.allgatherv_batch(x, 5)                       # the ragged form: counts = [64 * 5]
            torch.cuda.synchronize()
            assert torch.equal(fullv, x)
            s = wx.jl_empty((64, 3), dt, "cuda"); s.normal_()
            q = wx.jl_empty((64, 3), dt, "cuda"); q.normal_()
            s2, q2 = comm.allreduce_moments(s, q)
            torch.cuda.synchronize()
            assert torch.equal(s2, s) and torch.equal(q2, q)
        with pytest.raises(wx.ArgumentError):
            import ctypes
            from waveletsext_jl_amd import _lib
            host = np.zeros(4)
            _lib.check(_lib.lib().wx_allreduce_moments_f64(host.ctypes.data, 4, comm.handle, None))
        with pytest.raises(wx.ArgumentError):                         # one count per rank of the communicator
            two = (ctypes.c_int64 * 2)(3, 4)
            _lib.check(_lib.lib().wx_allgatherv_out_f64(x.data_ptr(), x.data_ptr(), ctypes.cast(two, ctypes.c_void_p), 2, comm.handle, None))
    finally:
        comm.close()
    wx.shutdown()


def test_null_pointer_is_an_argument_error(wx):
    """a NULL array with non-zero extent is rejected with WX_EARG before anything touches the device"""
    import ctypes
    from waveletsext_jl_amd import _lib
    q = np.array([1.0, 1.0]) / np.sqrt(2.0)
    y = np.zeros((8, 4, 2), order="F")
    rc = _lib.lib().wx_wpd1d_f64(None, y.ctypes.data, 8, 3, 2, q.ctypes.data, 2, None)
    assert rc == _lib.WX_EARG and b"NULL" in _lib.lib().wx_last_error()
    x = np.zeros((8, 2), order="F")
    rc = _lib.lib().wx_wpd1d_f64(x.ctypes.data, None, 8, 3, 2, q.ctypes.data, 2, None)
    assert rc == _lib.WX_EARG
    assert _lib.lib().wx_wpd1d_f64(None, None, 8, 3, 0, q.ctypes.data, 2, None) == 0      # empty batch: nothing to read


def test_c_host_example_runs(tmp_path, wx):
    """examples/roundtrip.c: a plain C host drives wpdall -> JBB tree -> iwpdall through the C ABI"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "waveletsext.jl_amd", "csrc")
    exe = str(tmp_path / "roundtrip")
    subprocess.run(["gcc", "-std=c99", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "roundtrip.c"),
                    "-o", exe, "-L" + lib, "-lwaveletsext_hip", "-lm", "-Wl,-rpath," + lib], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "round trip max error" in r.stdout


@pytest.mark.parametrize("dtype,n", [(np.float64, 16384), (np.float64, 32768), (np.float32, 32768), (np.float32, 131072)])
@pytest.mark.parametrize("wname", ["haar", "db4", "db10"])
def test_long_signals_split_path_matches_oracle(wx, oracle, wname, dtype, n):
    """Signals beyond the LDS of one CU: per-level launches down to the first depth whose nodes fit, then the
    fused kernel on the nodes (wx_dev_wpt1d / wx_dev_wpd1d / wx_dev_iwpt1d).  Same oracle, same tolerance."""
    rng = np.random.default_rng(n + 3)
    wt = _wt(wx, wname)
    tol = TOL[np.dtype(dtype)]
    x = np.asfortranarray(rng.standard_normal((n, 3)).astype(dtype))
    for L in (wx.maxtransformlevels(n) - 4, 2, 1):
        exp = oracle.wpdall(x, wt.qmf, L)
        got = wx.wpdall(x, wt, L)
        assert relerr(got, exp) <= tol, ("wpd", L)
        leaves = np.asfortranarray(exp[:, L, :])
        assert relerr(wx.wptall(x, wt, L), leaves) <= tol, ("wpt", L)
        assert relerr(wx.iwptall(leaves, wt, L), x) <= 10 * tol, ("iwpt", L)
        assert relerr(wx.iwpdall(exp, wt, L), x) <= 10 * tol, ("iwpd", L)


@pytest.mark.parametrize("n", [1024, 2048, 4096, 8192])
def test_haar_walsh_hadamard_path(wx, oracle, n):
    """Haar, full tree, Float64, 1024 <= n <= 8192, L <= 10: the Walsh-Hadamard kernels of wx_haar.hip (registers +
    cross-lane butterflies, one pass through LDS for the bit-reversed packet order) against the oracle, for every
    depth, host and device arrays; a two-tap filter with unequal taps and the forced generic path give the same
    coefficients through the general kernels."""
    rng = np.random.default_rng(n)
    wt = _wt(wx, "haar")
    x = np.asfortranarray(rng.standard_normal((n, 5)))
    for L in range(1, min(10, wx.maxtransformlevels(n)) + 1):
        exp = _stack_cols(oracle, x, wt.qmf, L)
        got = wx.wptall(x, wt, L)
        assert relerr(got, exp) <= 1e-13, (n, L)
        assert relerr(wx.iwptall(exp, wt, L), x) <= 1e-13, (n, L)
    L = 7
    exp = _stack_cols(oracle, x, wt.qmf, L)
    gd = wx.wptall(wx.to_device(x), wt, L)
    assert relerr(wx.to_numpy(gd), exp) <= 1e-13
    assert relerr(wx.to_numpy(wx.iwptall(gd, wt, L)), x) <= 1e-13
    wx.set_force_generic(1)
    try:
        assert relerr(wx.wptall(x, wt, L), exp) <= 1e-13
    finally:
        wx.set_force_generic(0)
    skew = wx.OrthoFilter([0.6, 0.8], "skew2")                          # q0 != q1: not a Walsh-Hadamard butterfly
    e2 = _stack_cols(oracle, x, skew.qmf, 3)
    assert relerr(wx.wptall(x, skew, 3), e2) <= 1e-13


def _stack_cols(oracle, x, qmf, L):
    return np.asfortranarray(np.stack([oracle.wpt(np.ascontiguousarray(x[:, i]), qmf, L) for i in range(x.shape[1])], axis=1))
