"""GPU parity of the register-resident 1-D packet kernels (csrc/wx_lattice.hip, csrc/wx_haar.hip) against the CPU oracle
at the geometries the bench times: n = 4096 Float64, full trees of depth 6..12.  Tolerance 1e-10 relative
(BASELINE.json north_star); the lattice reassociates the reference's tap sums, observed difference <= 4e-15."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import relerr

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _wt(wx, name):
    return wx.wavelet(getattr(wx.WT, name))


@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db4", "db5", "db6", "db7", "db8", "coif6", "db10"])
def test_lattice_every_depth_matches_oracle(wx, oracle, wname):
    """dwt/dwt_all.jl:152-166, 210-225 over Wavelets.jl's wpt / iwpt by level, every depth the lattice kernels take
    (depths 1 .. 5: csrc/wx_lattice_lo.hip for filters of up to eight taps, the fused LDS kernels for the longer ones)"""
    rng = np.random.default_rng(4096)
    wt = _wt(wx, wname)
    n, B = 4096, 3
    for L in range(1, 13):
        x = np.asfortranarray(rng.standard_normal((n, B)))
        exp = oracle.wptall(x, wt.qmf, L)
        got = wx.wptall(x, wt, L)
        assert relerr(got, exp) <= 1e-12, (wname, L)
        back = wx.iwptall(exp, wt, L)
        assert relerr(back, x) <= 1e-12, (wname, L)


@pytest.mark.parametrize("mode", [0, 2, 1])
def test_target_geometry_every_kernel_family(wx, oracle, mode):
    """BASELINE row T: wptall / iwptall, n = 4096, db4, L = 10 -- through the lattice kernels (mode 0), the fused LDS
    kernels (mode 2) and the one-level-per-launch kernels (mode 1)"""
    rng = np.random.default_rng(1010)
    wt = _wt(wx, "db4")
    x = np.asfortranarray(rng.standard_normal((4096, 5)))
    exp = oracle.wptall(x, wt.qmf, 10)
    wx.set_force_generic(mode)
    try:
        assert relerr(wx.wptall(x, wt, 10), exp) <= 1e-10
        assert relerr(wx.iwptall(exp, wt, 10), x) <= 1e-10
    finally:
        wx.set_force_generic(0)


@pytest.mark.parametrize("fwd_mode", ["0", "1"])
def test_target_geometry_fused_forward_modes(fwd_mode):
    """the two variants of the fused LDS forward kernel (WX_FWD_MODE is read once per process) at the target geometry"""
    code = (
        "import sys, numpy as np\n"
        "sys.path[:0] = [%r, %r]\n"
        "import waveletsext_jl_amd as wx, wx_oracle as O\n"
        "wt = wx.wavelet(wx.WT.db4); rng = np.random.default_rng(3)\n"
        "x = np.asfortranarray(rng.standard_normal((4096, 5)))\n"
        "exp = O.wptall(x, wt.qmf, 10)\n"
        "wx.set_force_generic(2)\n"
        "e1 = np.abs(wx.wptall(x, wt, 10) - exp).max() / np.abs(exp).max()\n"
        "e2 = np.abs(wx.iwptall(exp, wt, 10) - x).max() / np.abs(x).max()\n"
        "assert e1 <= 1e-10 and e2 <= 1e-10, (e1, e2)\n" % (ROOT, os.path.join(ROOT, "oracle")))
    env = dict(os.environ, WX_KNOBS="1", WX_FWD_MODE=fwd_mode)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]


def test_config2_inverse_reads_the_packet_table(wx, oracle):
    """BASELINE config 2, second leg: iwpdall of the full tree = the lattice inverse reading column L of the
    (n, L+1, B) table in place (DWT.jl:340-401, dwt/dwt_all.jl:323-342)"""
    rng = np.random.default_rng(2)
    wt = _wt(wx, "db8")
    n, L, B = 4096, 12, 3
    x = np.asfortranarray(rng.standard_normal((n, B)))
    tab = oracle.wpdall(x, wt.qmf, L)
    assert relerr(wx.wpdall(x, wt, L), tab) <= 1e-10
    assert relerr(wx.iwpdall(tab, wt, L), x) <= 1e-10
    assert relerr(wx.iwpdall(tab, wt), x) <= 1e-10
    for Lp in (6, 9):                                     # shallower full trees of the same table
        exp = oracle.iwpdall(tab, wt.qmf, Lp)
        assert relerr(wx.iwpdall(tab, wt, Lp), exp) <= 1e-10
    wt4 = _wt(wx, "db4")                                  # depths 1 .. 5: the shallow lattice inverse on row L of the table
    tab4 = oracle.wpdall(x, wt4.qmf, 7)
    for Lp in (1, 2, 3, 4, 5):
        assert relerr(wx.iwpdall(tab4, wt4, Lp), oracle.iwpdall(tab4, wt4.qmf, Lp)) <= 1e-10, Lp


def test_lattice_batches_larger_than_one_wave_of_workgroups(wx, oracle):
    """one workgroup per signal: more signals than resident workgroups, and an odd count"""
    import torch
    rng = np.random.default_rng(8)
    wt = _wt(wx, "db4")
    n, B, L = 4096, 3 * 256 * 12 + 5, 10
    x = wx.to_device(np.asfortranarray(rng.standard_normal((n, B))))
    y = wx.wptall(x, wt, L)
    xr = wx.iwptall(y, wt, L)
    assert float((xr - x).abs().max()) <= 1e-11
    idx = [0, 1, 255, 256, 3071, 3072, B - 2, B - 1]
    xs = np.asfortranarray(x[:, idx].cpu().numpy())
    assert relerr(y[:, idx].cpu().numpy(), oracle.wptall(xs, wt.qmf, L)) <= 1e-12
    # energy: orthonormal transform
    assert abs(float((y * y).sum() / (x * x).sum()) - 1.0) <= 1e-12
    del x, y, xr
    torch.cuda.empty_cache()


def test_haar_register_path_batches_larger_than_the_grid(wx, oracle):
    """the Walsh-Hadamard kernels are persistent (grid <= 5 workgroups per CU): every signal of a longer batch against
    the oracle, odd and even depths (the gain of a pair of levels is the exact 1/2)"""
    rng = np.random.default_rng(11)
    wt = _wt(wx, "haar")
    for n, L, B in ((1024, 10, 256 * 8 * 2 + 3), (4096, 9, 256 * 5 + 7)):
        x = np.asfortranarray(rng.standard_normal((n, B)))
        exp = oracle.wptall(x, wt.qmf, L)
        assert relerr(wx.wptall(x, wt, L), exp) <= 1e-13
        assert relerr(wx.iwptall(exp, wt, L), x) <= 1e-13


def test_filter_without_a_lattice_falls_back(wx, oracle):
    """a QMF that is not orthonormal has no rotation lattice: the call must take the general kernels and still match
    the direct-form oracle"""
    rng = np.random.default_rng(5)
    q = np.array([0.5, 0.9, 0.3, -0.1, 0.05, 0.02, 0.0, 0.01])
    wt = wx.OrthoFilter(q)
    x = np.asfortranarray(rng.standard_normal((4096, 2)))
    assert relerr(wx.wptall(x, wt, 8), oracle.wptall(x, q, 8)) <= 1e-10


@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db4", "db7", "db8", "coif6", "db10"])
def test_lattice_wpd_every_depth_matches_oracle(wx, oracle, wname):
    """wpdall of 4096-sample Float64 signals (DWT.jl:131-161, dwt/dwt_all.jl:260-282) through k_lat_wpd_f64: every level
    leaves the registers through its own LDS transposition; every depth 1..12, every column of the table"""
    rng = np.random.default_rng(12)
    wt = _wt(wx, wname)
    for L in range(1, 13):
        x = np.asfortranarray(rng.standard_normal((4096, 2)))
        exp = oracle.wpdall(x, wt.qmf, L)
        got = wx.wpdall(x, wt, L)
        assert got.shape == (4096, L + 1, 2)
        assert np.array_equal(got[:, 0, :], x)                       # column 0 is the signal itself
        assert relerr(got, exp) <= 1e-12, (wname, L)
    # and the kernels it replaces still agree (mode 2: fused LDS kernel)
    wx.set_force_generic(2)
    try:
        assert relerr(wx.wpdall(x, wt, 12), exp) <= 1e-10
    finally:
        wx.set_force_generic(0)


@pytest.mark.parametrize("n", [2048, 1024])
@pytest.mark.parametrize("wname", ["db2", "db4", "db7", "coif6"])
def test_interleaved_short_signals_match_oracle(wx, oracle, n, wname):
    """2048- and 1024-sample signals: 2 / 4 signals interleaved in one wavefront (k_lat_wpt_sh_f64 / k_lat_iwpt_sh_f64), every
    depth the kernels take (L + log2(4096 / n) >= 6) and the depths below it (other kernels), batches that are and are not
    multiples of the signals per wavefront (the last wavefront re-does signals), host and device pointers"""
    rng = np.random.default_rng(n)
    wt = _wt(wx, wname)
    for B in (1, 2, 3, 4, 5, 9):
        x = np.asfortranarray(rng.standard_normal((n, B)))
        for L in range(2, int(np.log2(n)) + 1):
            exp = oracle.wptall(x, wt.qmf, L)
            assert relerr(wx.wptall(x, wt, L), exp) <= 1e-12, (n, wname, B, L)
            assert relerr(wx.iwptall(exp, wt, L), x) <= 1e-12, (n, wname, B, L)
    # wpd / iwpd: every level leaves through the routed stores, the inverse reads the last column of the tables in place
    for B in (2, 3, 4, 6):
        x = np.asfortranarray(rng.standard_normal((n, B)))
        for L in (1, 2, 5, int(np.log2(n)) - 1, int(np.log2(n))):
            tab = np.asfortranarray(np.stack([oracle.wpd(x[:, b], wt.qmf, L) for b in range(B)], axis=-1))
            assert relerr(wx.wpdall(x, wt, L), tab) <= 1e-12, (n, wname, B, L)
            assert relerr(wx.iwpdall(tab, wt, L), x) <= 1e-12, (n, wname, B, L)
    xd = wx.to_device(np.asfortranarray(rng.standard_normal((n, 7))))
    L = int(np.log2(n)) - 1
    yd = wx.wptall(xd, wt, L)
    assert relerr(yd.cpu().numpy(), oracle.wptall(xd.cpu().numpy(), wt.qmf, L)) <= 1e-12
    assert relerr(wx.iwptall(yd, wt, L).cpu().numpy(), xd.cpu().numpy()) <= 1e-12


def test_interleaved_short_signals_large_batch_properties(wx):
    """131072 x 2048 and 262144 x 1024 (the target's byte count): reconstruction, energy, agreement with the LDS kernels"""
    import torch
    wt = _wt(wx, "db4")
    for n, B, L in ((2048, 131072, 10), (1024, 262144, 9)):
        x = wx.jl_empty((n, B), torch.float64, "cuda")
        x.normal_(generator=torch.Generator(device="cuda").manual_seed(n))
        y = wx.wptall(x, wt, L)
        e0, e1 = (x * x).sum(dim=0), (y * y).sum(dim=0)
        assert float(((e1 - e0).abs() / e0).max()) <= 1e-12
        assert float((wx.iwptall(y, wt, L) - x).abs().max()) <= 1e-12
        wx.set_force_generic(2)                                       # the fused LDS kernels
        try:
            y2 = wx.wptall(x[:, :4096], wt, L)
        finally:
            wx.set_force_generic(0)
        assert float((y2 - y[:, :4096]).abs().max()) <= 1e-12
        del x, y, y2


@pytest.mark.parametrize("n", [8192, 16384, 32768])
def test_long_signals_top_levels_then_lattice(wx, oracle, n):
    """signals longer than 4096: log2(n / 4096) top levels of one pass each, then the lattice kernels on the 4096-sample nodes
    (wx_dev_wpt1d / wx_dev_iwpt1d); depths on both sides of the switch-over (L - log2(n / 4096) >= 6)"""
    rng = np.random.default_rng(n)
    for wname in ("db2", "db4", "coif6"):
        wt = _wt(wx, wname)
        x = np.asfortranarray(rng.standard_normal((n, 3)))
        dl = int(np.log2(n)) - 12
        for L in (dl + 5, dl + 6, dl + 8, int(np.log2(n))):
            exp = oracle.wptall(x, wt.qmf, L)
            assert relerr(wx.wptall(x, wt, L), exp) <= 1e-12, (n, wname, L)
            assert relerr(wx.iwptall(exp, wt, L), x) <= 1e-12, (n, wname, L)


@pytest.mark.parametrize("n", [512, 256, 128, 64])
def test_short_signals_8_to_64_per_wavefront(wx, oracle, n):
    """512 .. 64 samples: 8 .. 64 signals interleaved in one wavefront (k_lat_wpt_g_f64 / k_lat_iwpt_g_f64 / k_lat_wpd_g_f64,
    filters of 4, 6, 8 taps), every depth, batches around the signals-per-wavefront count (tail wavefront), a longer filter
    that keeps the LDS kernels, device pointers"""
    rng = np.random.default_rng(n)
    per = 4096 // n
    for wname in ("haar", "db2", "db3", "db4", "db8"):
        wt = _wt(wx, wname)
        for B in (1, per - 1, per, per + 1, 2 * per + 3):
            x = np.asfortranarray(rng.standard_normal((n, B)))
            for L in range(1, int(np.log2(n)) + 1):
                exp = oracle.wptall(x, wt.qmf, L)
                assert relerr(wx.wptall(x, wt, L), exp) <= 1e-12, (n, wname, B, L)
                assert relerr(wx.iwptall(exp, wt, L), x) <= 1e-12, (n, wname, B, L)
                tab = np.asfortranarray(np.stack([oracle.wpd(x[:, b], wt.qmf, L) for b in range(B)], axis=-1))
                assert relerr(wx.wpdall(x, wt, L), tab) <= 1e-12, (n, wname, B, L)
                assert relerr(wx.iwpdall(tab, wt, L), x) <= 1e-12, (n, wname, B, L)
    wt = _wt(wx, "db4")
    xd = wx.to_device(np.asfortranarray(rng.standard_normal((n, 3 * per + 1))))
    L = int(np.log2(n))
    yd = wx.wptall(xd, wt, L)
    assert relerr(yd.cpu().numpy(), oracle.wptall(xd.cpu().numpy(), wt.qmf, L)) <= 1e-12
    assert relerr(wx.iwptall(yd, wt, L).cpu().numpy(), xd.cpu().numpy()) <= 1e-12


def test_float32_signals_through_the_lattice(wx, oracle):
    """Float32 signals of 4096 samples (wx_lattice_f32.hip).  Round 5: Float32 ARITHMETIC on pairs of signals (the reference itself
    rounds to Float32 at every accumulate, dwt/dwt_one_level.jl:97-103): against the reference's Float32 semantics (the oracle's
    Float32 instantiation) within the 1e-5 budget, and against the exact Float64 transform within 4e-6 (observed 1.1e-6 at L = 12;
    the Float64-register kernels of round 4 rounded once: < 1e-6).  A batch of 3: one full pair + the lone last signal."""
    rng = np.random.default_rng(32)
    for wname in ("db2", "db4", "db7", "coif6"):
        wt = _wt(wx, wname)
        x = np.asfortranarray(rng.standard_normal((4096, 3)).astype(np.float32))
        for L in (5, 6, 9, 12):
            exp = oracle.wptall(x.astype(np.float64), wt.qmf, L)
            got = wx.wptall(x, wt, L)
            assert got.dtype == np.float32 and relerr(got.astype(np.float64), exp) <= 4e-6, (wname, L)
            assert relerr(got, oracle.wptall(x, wt.qmf, L)) <= 1e-5, (wname, L)
            assert relerr(wx.iwptall(exp.astype(np.float32), wt, L).astype(np.float64), x.astype(np.float64)) <= 4e-6
            tab = np.stack([oracle.wpd(x[:, b].astype(np.float64), wt.qmf, L) for b in range(3)], axis=-1)
            assert relerr(wx.iwpdall(np.asfortranarray(tab.astype(np.float32)), wt, L).astype(np.float64), x.astype(np.float64)) <= 4e-6
