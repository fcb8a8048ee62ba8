"""The committed oracle outputs (tests/golden/oracle_outputs.json, written by tools/gen_golden.py) pin the oracle: today's
oracle must reproduce them BIT FOR BIT from the stored inputs.  A deliberate oracle change regenerates the fixture."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_oracle_reproduces_the_golden_outputs_bit_for_bit(oracle, wx):
    from tools import gen_golden as G
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_outputs.json")))["cases"]
    assert set(fx) == {"cfg1_wpt", "cfg2_wpdall", "target_wptall", "cfg3_swptall", "cfg4_wpt2d", "cfg5_acwpd_jbb"}
    now = G.recompute_from(fx)
    for name, c in fx.items():
        for key, enc in c["outputs"].items():
            want = G.dec(enc)
            got = np.asarray(now[name][key])
            assert got.shape == want.shape and got.dtype == want.dtype, (name, key)
            if want.dtype == np.bool_:
                assert (got == want).all(), (name, key)
            else:
                assert got.tobytes(order="F") == want.tobytes(order="F"), (name, key)


def test_golden_round_trips_are_reconstructions(wx):
    """sanity of the fixture itself: the stored inverse outputs are the stored inputs to rounding"""
    from tools import gen_golden as G
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_outputs.json")))["cases"]
    for name, key in (("cfg1_wpt", "iwpt_of_wpt"), ("cfg2_wpdall", "iwpd_of_wpd"), ("cfg3_swptall", "iswpt_of_swpt")):
        x = G.dec(fx[name]["inputs"]["x"])
        assert np.abs(G.dec(fx[name]["outputs"][key]) - x).max() <= 1e-12
