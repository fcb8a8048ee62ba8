"""Index arithmetic of the multi-level top pass (csrc/wx_toptile.h) on the CPU: tools/toptile_emu.py restates the kernel's plan,
windows and halos in numpy; here it is compared with the oracle's wpt along full trees, pyramids and random top trees, so
that a wrong offset is found without a GPU (the GPU parity tests are tests/test_gpu_toptile.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import toptile_emu as emu  # noqa: E402


def _tree_from_mask(n, NL, split):
    tree = np.zeros(n - 1, dtype=bool)
    ex = {1}
    for i in range(1, 1 << NL):
        if i in ex and (split >> (i - 1)) & 1:
            tree[i - 1] = True
            ex |= {2 * i, 2 * i + 1}
    return tree


@pytest.mark.parametrize("F,NL", [(2, 1), (4, 2), (8, 1), (8, 3), (8, 4), (6, 4), (12, 3), (20, 4), (16, 2)])
def test_emulated_pass_matches_the_oracle(wx, oracle, F, NL):
    name = {2: "haar", 4: "db2", 6: "db3", 8: "db4", 12: "coif4", 16: "db8", 20: "db10"}[F]
    q = np.asarray(wx.wavelet(getattr(wx.WT, name)).qmf, dtype=np.float64)
    assert q.size == F
    n = 512
    rng = np.random.default_rng(F * 10 + NL)
    x = rng.standard_normal(n)
    masks = [(1 << ((1 << NL) - 1)) - 1,                                   # full tree
             sum(1 << ((1 << l) - 1) for l in range(NL))]                  # pyramid
    masks += [int(rng.integers(1, 1 << ((1 << NL) - 1))) | 1 for _ in range(3)]
    for split in masks:
        tree = _tree_from_mask(n, NL, split)
        ref = oracle.wptall(np.asfortranarray(x[:, None]), q, tree)[:, 0]
        for deep in (0, 0x5555 & ((1 << (1 << NL)) - 1)):
            dst, dp = emu.fwd(x, q, NL, split, deep, TS=64)
            got = np.where(np.isnan(dst), dp, dst)
            assert not np.isnan(got).any()
            assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max(), (F, NL, bin(split), deep)
            # a node lives in exactly one of the two arrays
            assert (np.isnan(dst) != np.isnan(dp)).all()
            back = emu.inv(dst, dp, q, NL, split, deep, TS=64)
            assert np.abs(back - x).max() <= 1e-12 * np.abs(x).max(), (F, NL, bin(split), deep)
