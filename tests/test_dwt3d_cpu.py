"""The oracle's 3-D dwt (separable pyramid on cubes out of the 1-D one-level step; Wavelets.jl's 3-D transform that
dwtall / idwtall call on 4-D arrays, dwt/dwt_all.jl:8-9, 39-54) checked through what its definition implies.  No GPU."""
import numpy as np


def _q(wx, name):
    return np.asarray(wx.wavelet(getattr(wx.WT, name)).qmf, dtype=np.float64)


def test_one_level_of_a_product_is_the_product_of_one_level_steps(oracle, wx):
    rng = np.random.default_rng(3)
    q = _q(wx, "db2")
    g, h = oracle.makereverseqmfpair(q)
    u, v, w = (rng.standard_normal(8) for _ in range(3))
    x = np.asfortranarray(np.einsum("i,j,k->ijk", u, v, w))
    y = oracle.dwt3d(x, q, 1)
    step = lambda s: np.concatenate(oracle.dwt_step(s, h, g))
    exp = np.einsum("i,j,k->ijk", step(u), step(v), step(w))
    assert np.abs(y - exp).max() <= 1e-14 * np.abs(exp).max()


def test_reconstruction_energy_and_depth(oracle, wx):
    rng = np.random.default_rng(4)
    for name, n in (("haar", 4), ("db4", 8), ("coif2", 16)):
        q = _q(wx, name)
        x = np.asfortranarray(rng.standard_normal((n, n, n)))
        for L in range(0, oracle.maxtransformlevels(n) + 1):
            y = oracle.dwt3d(x, q, L)
            assert abs((y ** 2).sum() - (x ** 2).sum()) <= 1e-12 * (x ** 2).sum()      # orthonormal
            assert np.abs(oracle.idwt3d(y, q, L) - x).max() <= 1e-12
            if L == 0:
                assert np.array_equal(y, x)
    # a constant cube: everything ends in the single coarsest scaling coefficient, (sqrt 2)^(3 L) times the constant
    q = _q(wx, "db4")
    y = oracle.dwt3d(np.full((8, 8, 8), 0.5, order="F"), q, 3)
    assert abs(y[0, 0, 0] - 0.5 * 2.0 ** 4.5) <= 1e-12
    y[0, 0, 0] = 0.0
    assert np.abs(y).max() <= 1e-12
