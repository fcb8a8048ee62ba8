"""The oracle's numpy restatement of the SureShrink / RelErrorShrink threshold selection (Denoising.jl:146-166, 285-381):
a hand-worked case, the invariants the definitions imply, and findelbow on a curve whose elbow is known.  No GPU."""
import numpy as np


def test_surethreshold_hand_worked(oracle):
    # sorted squares a = [1, 4, 9], cumsum b = [1, 5, 14], c = [2, 1, 0]: risk = (3 - 2 i + b + c a) / 3 = [4/3, 8/3, 11/3]
    assert oracle.surethreshold(np.array([3.0, -1.0, 2.0]), False) == 1.0
    # one large coefficient among small ones: the risk is smallest at the largest of the small ones
    y = np.array([0.1, -0.2, 0.15, 10.0, 0.05])
    assert oracle.surethreshold(y, False) == 0.2


def test_selected_thresholds_are_coefficient_magnitudes_and_scale(oracle):
    rng = np.random.default_rng(1)
    c = rng.standard_normal(300) * np.exp(-np.arange(300) / 40.0)
    mags = np.abs(c)
    t = oracle.surethreshold(c, False)                                 # (the SURE risk assumes unit noise: no scaling law)
    assert np.min(np.abs(mags - t)) <= 2e-16 * mags.max()
    for f in (lambda v: oracle.relerrorthreshold(v, False), lambda v: oracle.relerrorthreshold(v, False, None, 1),
              lambda v: oracle.relerrorthreshold(v, False, None, 3)):
        t = f(c)
        assert np.min(np.abs(mags - t)) <= 2e-16 * mags.max() or t == 0.0
        assert f(4.0 * c) == 4.0 * t                                   # powers of two commute with every rounding
    # more elbows move towards smaller thresholds
    assert oracle.relerrorthreshold(c, False, None, 1) >= oracle.relerrorthreshold(c, False, None, 2) >= oracle.relerrorthreshold(c, False, None, 3)


def test_redundant_selection_uses_all_columns_or_the_leaves(oracle):
    rng = np.random.default_rng(2)
    tab = np.asfortranarray(rng.standard_normal((8, 15)))               # an (n, 2^(L+1)-1) table, n = 8, L = 3
    assert oracle.surethreshold(tab, True) == oracle.surethreshold(tab.reshape(-1, order="F"), False)
    tree = oracle.maketree1d(8, 3, "dwt")
    leaves = oracle.getleaf(tree, "binary")
    assert oracle.relerrorthreshold(tab, True, tree) == oracle.relerrorthreshold(tab[:, leaves].reshape(-1, order="F"), False)


def test_orth2relerror_and_findelbow(oracle):
    r = oracle.orth2relerror(np.array([1.0, -2.0, 2.0]))               # squares sorted down: 4 4 1, total 9
    assert np.allclose(r, [np.sqrt(5.0 / 9), np.sqrt(1.0 / 9), 0.0], rtol=0, atol=1e-16)
    x = np.linspace(0.0, 1.0, 11)
    y = np.where(x < 0.3, 0.0, (x - 0.3) / 0.7)                        # flat, then a straight rise: elbow at x = 0.3
    assert oracle.findelbow(x, y) == 3
