"""GPU parity against the COMMITTED golden vectors (tests/golden/oracle_outputs.json): the HIP path through the C ABI on
the stored inputs vs the stored oracle outputs.  1e-10 relative for Float64, 1e-5 for Float32 (BASELINE north_star /
SURVEY 8d), trees bit-exact."""
import json
import os
import sys

import numpy as np
import pytest

from helpers import relerr

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def fx():
    from tools import gen_golden as G
    raw = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_outputs.json")))["cases"]
    out = {}
    for name, c in raw.items():
        out[name] = {"in": {k: (G.dec(v) if isinstance(v, dict) else v) for k, v in c["inputs"].items()},
                     "out": {k: G.dec(v) for k, v in c["outputs"].items()}}
    return out


def _wt(wx, c):
    return wx.wavelet(getattr(wx.WT, c["in"]["wavelet"]))


@pytest.mark.parametrize("mode", [0, 1])
def test_config1_wpt_iwpt(wx, fx, mode):
    c = fx["cfg1_wpt"]
    wx.set_force_generic(mode)
    try:
        y = wx.wpt(c["in"]["x"], _wt(wx, c), c["in"]["L"])
        assert relerr(y, c["out"]["wpt"]) <= 1e-10
        assert relerr(wx.iwpt(c["out"]["wpt"], _wt(wx, c), c["in"]["L"]), c["out"]["iwpt_of_wpt"]) <= 1e-10
    finally:
        wx.set_force_generic(0)


@pytest.mark.parametrize("mode", [0, 1])
def test_config2_wpdall_iwpdall(wx, fx, mode):
    c = fx["cfg2_wpdall"]
    wx.set_force_generic(mode)
    try:
        assert relerr(wx.wpdall(c["in"]["x"], _wt(wx, c), c["in"]["L"]), c["out"]["wpd"]) <= 1e-10
        assert relerr(wx.iwpdall(c["out"]["wpd"], _wt(wx, c), c["in"]["L"]), c["out"]["iwpd_of_wpd"]) <= 1e-10
    finally:
        wx.set_force_generic(0)


@pytest.mark.parametrize("mode", [0, 2, 1])
def test_target_wptall_real_length(wx, fx, mode):
    """n = 4096, db4, L = 10: lattice kernels (0), fused LDS kernels (2), one level per launch (1)"""
    c = fx["target_wptall"]
    wx.set_force_generic(mode)
    try:
        assert relerr(wx.wptall(c["in"]["x"], _wt(wx, c), 10), c["out"]["wpt"]) <= 1e-10
        assert relerr(wx.iwptall(c["out"]["wpt"], _wt(wx, c), 10), c["in"]["x"]) <= 1e-10
    finally:
        wx.set_force_generic(0)


def test_config3_swptall_iswptall(wx, fx):
    c = fx["cfg3_swptall"]
    wt = _wt(wx, c)
    got = wx.swptall(c["in"]["x"], wt, c["in"]["L"])
    assert relerr(got, c["out"]["swpt"]) <= 1e-10
    assert relerr(wx.iswptall(c["out"]["swpt"], wt), c["out"]["iswpt_of_swpt"]) <= 1e-10


def test_config4_wpt2d_float32(wx, fx):
    c = fx["cfg4_wpt2d"]
    got = wx.wptall(c["in"]["x"], _wt(wx, c), c["in"]["L"])
    assert got.dtype == np.float32 and relerr(got, c["out"]["wpt"]) <= 1e-5


def test_config5_acwpd_jbb_tree(wx, fx):
    c = fx["cfg5_acwpd_jbb"]
    wt, L, x = _wt(wx, c), c["in"]["L"], c["in"]["x"]
    assert relerr(wx.acwpd(x[:, 0], wt, L), c["out"]["acwpd_signal0"]) <= 1e-10
    for mode in (0, 1):                                  # fused subtree kernel / materialised table
        wx.set_force_generic(mode)
        try:
            s, q = wx.acwpd_jbb_moments(x, wt, L)
        finally:
            wx.set_force_generic(0)
        assert relerr(s, c["out"]["sum"]) <= 1e-10 and relerr(q, c["out"]["sumsq"]) <= 1e-10
        costs = wx.costs_from_moments(s, q, x.shape[1], wx.JBB(redundant=True))
        assert np.allclose(costs, c["out"]["costs"], rtol=1e-9, atol=1e-9)
        assert (wx.bestbasis_treeselection(costs, x.shape[0]) == c["out"]["tree"]).all()
