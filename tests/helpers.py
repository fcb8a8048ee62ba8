import numpy as np


def random_tree_1d(n, rng, p=0.7):
    """Random valid binary tree (heap order, length n-1) for a dyadic n."""
    tree = np.zeros(n - 1, dtype=bool)
    if n < 2:
        return tree
    tree[0] = rng.random() < 0.95
    for i in range(2, n):
        tree[i - 1] = tree[i // 2 - 1] and (rng.random() < p)
    return tree


def random_tree_2d(m, n, rng, p=0.6):
    from math import log2
    L = 0
    k = min(m, n)
    while k % 2 == 0 and k >= 2:
        k //= 2
        L += 1
    nt = (4 ** L - 1) // 3
    tree = np.zeros(nt, dtype=bool)
    if nt == 0:
        return tree
    tree[0] = True
    for i in range(2, nt + 1):
        parent = (i + 2) // 4
        tree[i - 1] = tree[parent - 1] and (rng.random() < p)
    return tree


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.abs(b).max()
    return float(np.abs(a - b).max() / (den if den > 0 else 1.0))


TOL = {np.dtype(np.float64): 1e-10, np.dtype(np.float32): 1e-5}
