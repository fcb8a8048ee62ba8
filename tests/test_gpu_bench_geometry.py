"""GPU parity at the EXACT signal geometry of the benchmarked configurations (size-dependent kernel dispatch: K-level
fused passes, residue-class tiles, the D0 = 6 subtree kernel, chunk accumulation), batch reduced so that the oracle
finishes in seconds.  Tolerances: 1e-10 relative (north_star), JBB trees bit-exact."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import relerr

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db4", "db5", "coif2", "db7", "db8", "coif6", "db10"])
def test_config4_geometry_2d_lattice_matches_oracle(wx, oracle, wname):
    """BASELINE config 4: 2-D wptall / iwptall of 512 x 512 Float32 images, L = 6 (dwt/dwt_all.jl:152-166, 210-225 over
    Wavelets.jl's 2-D wpt by level) through the transposing lattice column kernels (csrc/wx_lattice2d.hip).  Float32:
    tolerance 2e-6 x the largest coefficient (observed 5e-7; the reference's own Float32 rounding is of that size)."""
    rng = np.random.default_rng(512)
    wt = wx.wavelet(getattr(wx.WT, wname))
    x = np.asfortranarray(rng.standard_normal((512, 512, 3)).astype(np.float32))
    exp = oracle.wptall(x.astype(np.float64), wt.qmf, 6)
    got = wx.wptall(x, wt, 6)
    assert got.dtype == np.float32
    assert relerr(got.astype(np.float64), exp) <= 2e-6, wname
    back = wx.iwptall(exp.astype(np.float32), wt, 6)
    assert relerr(back.astype(np.float64), x.astype(np.float64)) <= 2e-6, wname


@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db4", "db5", "coif2", "db7", "db8", "coif6", "db10"])
def test_2d_lattice_256x256_matches_oracle(wx, oracle, wname):
    """256 x 256 Float32 images, full depth L = 5, through the same transposing lattice kernels (two images per register
    column, csrc/wx_lattice2d.hip HB = 1): odd and even batches (the last workgroup re-does the last two images), forward
    against the oracle, inverse against the oracle's input (DWT.jl:500-548, 662-710)"""
    rng = np.random.default_rng(256)
    wt = wx.wavelet(getattr(wx.WT, wname))
    for B in (5, 2, 8):
        x = np.asfortranarray(rng.standard_normal((256, 256, B)).astype(np.float32))
        exp = oracle.wptall(x.astype(np.float64), wt.qmf, 5)
        got = wx.wptall(x, wt, 5)
        assert got.dtype == np.float32
        for b in range(B):
            assert relerr(got[:, :, b].astype(np.float64), exp[:, :, b]) <= 2e-6, (wname, B, b)
        back = wx.iwptall(exp.astype(np.float32), wt, 5)
        for b in range(B):
            assert relerr(back[:, :, b].astype(np.float64), x[:, :, b].astype(np.float64)) <= 2e-6, (wname, B, b)


@pytest.mark.parametrize("wname", ["haar", "db2", "db3", "db4", "db5", "coif2", "db7", "db8", "coif6", "db10"])
def test_2d_lattice_1024x1024_matches_oracle(wx, oracle, wname):
    """1024 x 1024 Float32 images, full depth L = 7 (8 columns of 1024 rows per wavefront, csrc/wx_lattice2d.hip HB = 2),
    forward against the oracle, inverse against the oracle's input"""
    rng = np.random.default_rng(1024)
    wt = wx.wavelet(getattr(wx.WT, wname))
    x = np.asfortranarray(rng.standard_normal((1024, 1024, 2)).astype(np.float32))
    exp = oracle.wptall(x.astype(np.float64), wt.qmf, 7)
    got = wx.wptall(x, wt, 7)
    assert got.dtype == np.float32
    for b in range(2):
        assert relerr(got[:, :, b].astype(np.float64), exp[:, :, b]) <= 3e-6, (wname, b)
    back = wx.iwptall(exp.astype(np.float32), wt, 7)
    for b in range(2):
        assert relerr(back[:, :, b].astype(np.float64), x[:, :, b].astype(np.float64)) <= 3e-6, (wname, b)


def test_2d_lattice_1024x1024_batch_round_trips(wx):
    import torch
    wt = wx.wavelet(wx.WT.db4)
    x = wx.jl_empty((1024, 1024, 129), torch.float32, "cuda")
    x.normal_(generator=torch.Generator(device="cuda").manual_seed(11))
    y0 = wx.wptall(x, wt, 7)
    for _ in range(3):
        y = wx.wptall(x, wt, 7)
        assert torch.equal(y, y0)
        xr = wx.iwptall(y, wt, 7)
        err = (xr - x).abs().amax(dim=(0, 1)) / x.abs().max()
        assert float(err.max()) <= 3e-6, (int(err.argmax()), float(err.max()))


def test_2d_lattice_256x256_large_batch_round_trips(wx):
    import torch
    wt = wx.wavelet(wx.WT.db4)
    x = wx.jl_empty((256, 256, 1001), torch.float32, "cuda")
    x.normal_(generator=torch.Generator(device="cuda").manual_seed(7))
    y0 = wx.wptall(x, wt, 5)
    for _ in range(3):
        y = wx.wptall(x, wt, 5)
        assert torch.equal(y, y0)
        xr = wx.iwptall(y, wt, 5)
        err = (xr - x).abs().amax(dim=(0, 1)) / x.abs().max()
        assert float(err.max()) <= 2e-6, (int(err.argmax()), float(err.max()))


def test_config4_every_image_of_a_large_batch_round_trips(wx):
    """the two wavefronts of a 2-D lattice workgroup exchange columns through LDS; a missing wait before the workgroup
    barrier once corrupted one image in a few hundred, so every image of several large batches is checked, repeatedly,
    and the forward result must be identical from run to run"""
    import torch
    wt = wx.wavelet(wx.WT.db4)
    for B in (256, 1024):
        x = wx.jl_empty((512, 512, B), torch.float32, "cuda")
        x.normal_(generator=torch.Generator(device="cuda").manual_seed(B))
        y0 = wx.wptall(x, wt, 6)
        for _ in range(4):
            y = wx.wptall(x, wt, 6)
            assert torch.equal(y, y0)
            xr = wx.iwptall(y, wt, 6)
            err = (xr - x).abs().amax(dim=(0, 1)) / x.abs().max()
            assert float(err.max()) <= 2e-6, (B, int(err.argmax()), float(err.max()))
        # orthonormal transform: the energy of every image is preserved
        e0 = (x.double() ** 2).sum(dim=(0, 1))
        e1 = (y0.double() ** 2).sum(dim=(0, 1))
        assert float(((e1 - e0).abs() / e0).max()) <= 1e-5
        del x, y, y0, xr


def test_config4_lattice_and_lds_kernels_agree():
    """the 2-D lattice path against the previous fused LDS kernels (WX_LATTICE2D=0, read once per process)"""
    code = (
        "import sys, numpy as np\n"
        "sys.path[:0] = [%r]\n"
        "import waveletsext_jl_amd as wx\n"
        "wt = wx.wavelet(wx.WT.db4); rng = np.random.default_rng(7)\n"
        "x = np.asfortranarray(rng.standard_normal((512, 512, 4)).astype(np.float32))\n"
        "y = wx.wptall(x, wt, 6); np.save(sys.argv[1], y); np.save(sys.argv[2], wx.iwptall(y, wt, 6))\n" % ROOT)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        outs = []
        for flag in ("1", "0"):
            f1, f2 = os.path.join(d, "y%s.npy" % flag), os.path.join(d, "x%s.npy" % flag)
            r = subprocess.run([sys.executable, "-c", code, f1, f2], env=dict(os.environ, WX_KNOBS="1", WX_LATTICE2D=flag),
                               capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append((np.load(f1), np.load(f2)))
        assert relerr(outs[0][0], outs[1][0]) <= 2e-6
        assert relerr(outs[0][1], outs[1][1]) <= 2e-6


@pytest.mark.parametrize("m,LD,B", [(256, 5, 3), (256, 5, 4), (512, 6, 2), (1024, 7, 1)])
@pytest.mark.parametrize("wname", ["haar", "db2", "db4", "coif2", "db8", "db10"])
def test_2d_lattice_deeper_than_the_lattice_levels(wx, oracle, m, LD, B, wname):
    """Depths LD + 1 ... LD + 3 on the three lattice geometries -- the FULL depth, the reference's default L = maxtransformlevels
    (DWT.jl:500-548, 662-710), included: the lattice rotates down to nodes of 8 x 8 and the remaining levels are one orthogonal 8 x 8
    matrix per node, applied in the store phase of the forward pass and (transposed) in the load phase of the inverse pass
    (csrc/wx_lattice2d.h, WxL2M; round 4).  Forward against the oracle per image, inverse of the oracle's coefficients against its input;
    odd and even batches for the two-images-per-column geometry."""
    rng = np.random.default_rng(m + len(wname))
    wt = wx.wavelet(getattr(wx.WT, wname))
    x = np.asfortranarray(rng.standard_normal((m, m, B)).astype(np.float32))
    for L in ((LD + 1, LD + 2, LD + 3) if m < 1024 else (LD + 3,)):
        exp = oracle.wptall(x.astype(np.float64), wt.qmf, L)
        got = wx.wptall(x, wt, L)
        assert got.dtype == np.float32
        for b in range(B):
            assert relerr(got[:, :, b].astype(np.float64), exp[:, :, b]) <= 3e-6, (m, wname, L, b)
        back = wx.iwptall(exp.astype(np.float32), wt, L)
        for b in range(B):
            assert relerr(back[:, :, b].astype(np.float64), x[:, :, b].astype(np.float64)) <= 3e-6, (m, wname, L, b)


def test_2d_lattice_full_depth_is_the_lattice_path_and_deterministic(wx):
    """full-depth transforms of a chip-filling batch: bit-identical across launches, every image round-trips, and the deep path agrees
    with the generic two-pass path (WX_LATTICE2D_DEEP=0 in a child process) to Float32 rounding"""
    import os
    import subprocess
    import sys
    import torch
    wt = wx.wavelet(wx.WT.db4)
    for m, L, B in ((512, 9, 257), (256, 8, 1001)):
        x = wx.jl_empty((m, m, B), torch.float32, "cuda")
        x.normal_(generator=torch.Generator(device="cuda").manual_seed(m))
        y0 = wx.wptall(x, wt, L)
        y = wx.wptall(x, wt, L)
        assert torch.equal(y, y0)
        xr = wx.iwptall(y, wt, L)
        err = (xr - x).abs().amax(dim=(0, 1)) / x.abs().max()
        assert float(err.max()) <= 3e-6, (m, int(err.argmax()), float(err.max()))
    code = ("import numpy as np, waveletsext_jl_amd as wx\n"
            "rng = np.random.default_rng(3); x = np.asfortranarray(rng.standard_normal((512, 512, 2)).astype(np.float32))\n"
            "y = wx.wptall(x, wx.wavelet(wx.WT.db4), 9); np.save(%r, y)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for deep in ("1", "0"):
        f = os.path.join(root, "gpurun_out", "deep_%s.npy" % deep)
        os.makedirs(os.path.dirname(f), exist_ok=True)
        subprocess.check_call([sys.executable, "-c", code % f], env=dict(os.environ, WX_KNOBS="1", WX_LATTICE2D_DEEP=deep), cwd=root)
        outs.append(np.load(f).astype(np.float64))
    assert relerr(outs[0], outs[1]) <= 3e-6 and not np.array_equal(outs[0], outs[1])
