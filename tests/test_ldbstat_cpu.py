"""The oracle's restatements behind the LDB order statistics and density maps: hand-worked cases.  No GPU."""
import numpy as np


def test_emd_pair_hand_worked(oracle):
    # p = {0, 1} and q = {2}: move two half-units over distances 2 and 1 -> (|0.5-0| * 1 + |1-0| * 1) / 2 = 0.75
    assert oracle.emd_pair([0.0, 1.0], [2.0], 0.5, 1.0) == 0.75
    assert oracle.emd_pair([1.0, 2.0], [1.0, 2.0], 0.5, 0.5) == 0.0
    assert oracle.emd_pair_weighted([1.0, 0.0], [2.0], [0.5, 0.5], [1.0]) == 0.75      # unsorted input, weights follow


def test_robust_fishers_hand_worked(oracle):
    # one coefficient, two classes: values {1, 2, 6} and {3, 5}: medians 2 and 4, mads 1 and 1, overall median 3
    X = np.asfortranarray(np.array([[1.0, 3.0, 2.0, 5.0, 6.0]]))
    power, order = oracle.ldb_robust_fishers(X, ["a", "b", "a", "b", "a"])
    p = np.array([0.6, 0.4])
    exp = (((np.array([2.0, 4.0]) - 3.0 * np.array([2.0, 4.0])) ** 2) @ p) / (np.array([1.0, 1.0]) @ p)
    assert power.shape == (1,) and power[0] == exp and order.tolist() == [1]


def test_average_shifted_histogram(oracle):
    # one observation in the middle of a 9-point grid, m = 2: a triangle of half-width 2 points, unit integral
    d = oracle.ash_density([0.0], -4.0, 1.0, 9, 2)
    assert np.allclose(d, [0, 0, 0, 0.25, 0.5, 0.25, 0, 0, 0], atol=1e-16)
    assert abs(d.sum() * 1.0 - 1.0) <= 1e-15
    assert oracle.ash_pdf(d, -4.0, 1.0, -0.5) == 0.375 and oracle.ash_pdf(d, -4.0, 1.0, 10.0) == 0.0
    # observations outside the grid are not counted; the density still integrates to one
    d2 = oracle.ash_density([0.0, 100.0], -4.0, 1.0, 9, 2)
    assert np.array_equal(d, d2)
