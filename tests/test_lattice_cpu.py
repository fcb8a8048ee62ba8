"""CPU checks of the lattice formulation used by csrc/wx_lattice.hip: the rotation factorisation against the oracle,
the LDS exchange maps against the bank rules, and the lane-level emulation of the kernel's data movement."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_lds_exchange_maps_are_conflict_free_bijections():
    from tools import lattice_lds_maps as M
    for name, fn in (("T1", M.t1), ("T2", M.t2), ("T3", M.t3), ("T4", M.t4), ("T4i", M.t4i), ("T3i", M.t3i),
                     ("T2i", M.t2i), ("T1i", M.t1i)):
        for r in range(4):
            assert M.check(name, *fn(r)) <= 1104


@pytest.mark.parametrize("wname", ["haar", "db2", "db4", "db8", "coif6", "db10"])
def test_rotation_lattice_equals_the_direct_form(wx, oracle, wname):
    from tools.lattice_proto import wpt_lattice, iwpt_lattice
    q = np.asarray(wx.wavelet(getattr(wx.WT, wname)).qmf, dtype=np.float64)
    rng = np.random.default_rng(3)
    for n, L in ((64, 6), (256, 5), (1024, 10)):
        x = rng.standard_normal(n)
        ref = oracle.wpt(x, q, L)
        assert np.abs(wpt_lattice(x, q, L) - ref).max() <= 1e-13 * np.abs(ref).max()
        assert np.abs(iwpt_lattice(ref, q, L) - x).max() <= 1e-13 * np.abs(x).max()


@pytest.mark.parametrize("wname,L", [("db2", 6), ("db4", 10), ("db8", 12), ("db4", 7)])
def test_kernel_emulation_matches_oracle(wx, oracle, wname, L):
    """tools/lattice_emu.py follows the kernel statement by statement (layouts, LDS slots, cross-lane rotations, shear
    form of the rotations, path-dependent gain)"""
    from tools.lattice_emu import wpt_emu, iwpt_emu, shear_coefs
    from tools.lattice_proto import lattice_factor
    q = np.asarray(wx.wavelet(getattr(wx.WT, wname)).qmf, dtype=np.float64)
    t, g1 = lattice_factor(q)
    sh = shear_coefs(t)
    rng = np.random.default_rng(9)
    x = rng.standard_normal(4096)
    ref = oracle.wpt(x, q, L)
    assert np.abs(wpt_emu(x, sh, g1, L) - ref).max() <= 1e-13 * np.abs(ref).max()
    assert np.abs(iwpt_emu(ref, sh, g1, L) - x).max() <= 1e-13 * np.abs(x).max()


@pytest.mark.parametrize("wname,L", [("db4", 12), ("db8", 5), ("db2", 9)])
def test_wpd_emission_emulation_matches_oracle(wx, oracle, wname, L):
    """the per-level output routing of k_lat_wpd_f64 (emit_plan: round bits, line-address order, gains) in numpy"""
    from tools.lattice_emu import wpd_emu, shear_coefs, emit_plan, LANES
    from tools.lattice_proto import lattice_factor
    q = np.asarray(wx.wavelet(getattr(wx.WT, wname)).qmf, dtype=np.float64)
    t, g1 = lattice_factor(q)
    x = np.random.default_rng(4).standard_normal(4096)
    ref = oracle.wpd(x, q, L)
    got = wpd_emu(x, shear_coefs(t), g1, L)
    assert not np.isnan(got).any()
    assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()
    # every level's LDS writes are conflict-free: the 16 lanes of a ds_write_b64 group land on 16 different bank pairs
    for lay, levels in ((0, (1, 2)), (2, (3, 4, 5, 6)), (6, (7, 8, 9, 10, 11, 12))):
        for l in levels:
            P = emit_plan(lay, l)
            hi = np.zeros(64, dtype=np.int64)
            pos = np.zeros(64, dtype=np.int64)
            for k in range(6):
                if P["lane_o"][k][0] < 4:
                    pos += ((LANES >> k) & 1) << P["lane_o"][k][0]
            for qq, (kind, b, ob) in enumerate(P["line"]):
                if kind == "lane":
                    hi += ((LANES >> b) & 1) << qq
            slot = 17 * hi + pos
            for g in range(4):
                assert len(set(slot[16 * g:16 * g + 16] % 16)) == 16, (lay, l)
