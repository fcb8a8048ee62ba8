"""Tree / index helpers of the hot path (host side, integer code, bit-exact).

Mirrors src/mod/Utils.jl and src/mod/utils/utils_tree.jl of the reference plus the Wavelets.jl
`Util` functions those files extend (maxtransformlevels, isdyadic, ndyadicscales, 1-D maketree /
isvalidtree; semantics in SURVEY.md Appendix C).  Trees are numpy bool vectors in heap order
(`BitVector`), indices in the public API are 1-based exactly like the reference.
`@assert` failures raise AssertionError, `throw(ArgumentError)` raises ArgumentError.
"""
import numpy as np

from .filters import ArgumentError


def _shape(x):
    return tuple(x.shape) if hasattr(x, "shape") else None


def isdyadic(n):
    n = n if isinstance(n, (int, np.integer)) else int(np.prod(_shape(n)))
    return n >= 1 and (n & (n - 1)) == 0


def ndyadicscales(n):
    n = n if isinstance(n, (int, np.integer)) else _shape(n)[0]
    return int(round(np.log2(n)))


def maxtransformlevels(x, dims=None):
    """Wavelets.jl Util.maxtransformlevels + the (x, dims) method of Utils.jl:66-71."""
    if dims is not None:
        shp = _shape(x)
        assert 1 <= dims <= len(shp)
        return maxtransformlevels(int(shp[dims - 1]))
    if not isinstance(x, (int, np.integer)):
        return maxtransformlevels(int(min(_shape(x))))
    n = int(x)
    if n < 2:
        return 0
    tl = 0
    while n % 2 == 0:
        n //= 2
        tl += 1
    return tl


def nodelength(N, L):
    """Utils.jl:242"""
    return N >> L


def getchildindex(idx, child):
    """utils_tree.jl:57-75"""
    assert child in ("left", "right", "topleft", "topright", "bottomleft", "bottomright")
    if child == "left":
        return idx << 1
    if child == "right":
        return (idx << 1) + 1
    return 4 * idx - 2 + ("topleft", "topright", "bottomleft", "bottomright").index(child)


def getparentindex(idx, tree_type):
    """utils_tree.jl:89-99"""
    assert tree_type in ("binary", "quad")
    return idx >> 1 if tree_type == "binary" else (idx + 2) // 4


def getdepth(idx, tree_type):
    """utils_tree.jl:252-263 (integer arithmetic instead of floating log)"""
    assert idx > 0
    assert tree_type in ("binary", "quad")
    idx = int(idx)
    if tree_type == "binary":
        return idx.bit_length() - 1
    t, d = 3 * idx - 2, 0
    while t >= 4:
        t >>= 2
        d += 1
    return d


def gettreelength(*sz):
    """utils_tree.jl:285-293"""
    if len(sz) == 1:
        return (1 << maxtransformlevels(int(sz[0]))) - 1
    L = maxtransformlevels(int(min(sz[0], sz[1])))
    return ((1 << (2 * L)) - 1) // 3


def maketree(*args):
    """maketree(n, L[, s]) / maketree(x::Vector[, s])   (Wavelets.jl, 1-D binary tree)
    maketree(n, m, L[, s]) / maketree(x::Matrix[, s])   (utils_tree.jl:193-222, quad tree)
    s in ("full", "dwt")."""
    args = list(args)
    s = "full"
    if args and isinstance(args[-1], str):
        s = args.pop()
    if len(args) == 1 and not isinstance(args[0], (int, np.integer)):
        shp = _shape(args[0])
        if len(shp) == 1:
            return maketree(int(shp[0]), maxtransformlevels(int(shp[0])), s)
        return maketree(int(shp[0]), int(shp[1]), maxtransformlevels(int(min(shp))), s)
    assert s in ("full", "dwt")
    if len(args) == 2:
        n, L = int(args[0]), int(args[1])
        assert isdyadic(n)
        assert 0 <= L <= maxtransformlevels(n)
        tree = np.zeros(n - 1, dtype=bool)
        if s == "full":
            tree[: (1 << L) - 1] = True
        else:
            for i in range(L):
                tree[(1 << i) - 1] = True
        return tree
    n, m, L = int(args[0]), int(args[1]), int(args[2])
    L0 = maxtransformlevels(min(n, m))
    assert 0 <= L <= L0
    tree = np.zeros(gettreelength(n, m), dtype=bool)
    if s == "full":
        tree[: sum(4 ** i for i in range(L))] = True
    else:
        tree[0] = True
        for i in range(L - 1):
            tree[((1 << (2 * i + 2)) + 2) // 3 - 1] = True
    return tree


def isvalidtree(x, b):
    """Wavelets.jl isvalidtree(x::Vector, b) and utils_tree.jl:13-29 (Matrix)."""
    shp = _shape(x)
    b = np.asarray(b, dtype=bool)
    nb = b.size
    if len(shp) == 1:
        if nb != shp[0] - 1:
            return False
        i = 1
        while 2 * i + 1 <= nb:
            if not b[i - 1] and (b[2 * i - 1] or b[2 * i]):
                return False
            i += 1
        return True
    n, m = shp
    if gettreelength(n, m) != nb:
        return False
    L0 = getdepth(nb, "quad") if nb > 0 else 0
    ns = ((1 << (2 * L0)) - 1) // 3
    for i in range(1, ns + 1):
        haschild = b[4 * i - 3] or b[4 * i - 2] or b[4 * i - 1] or b[4 * i]
        if not b[i - 1] and haschild:
            return False
    return True


def getleaf(tree, tree_type):
    """utils_tree.jl:122-157"""
    assert tree_type in ("binary", "quad")
    tree = np.asarray(tree, dtype=bool)
    nt = tree.size
    L0 = getdepth(nt, tree_type)
    Ent = (1 << (L0 + 1)) - 1 if tree_type == "binary" else ((1 << (2 * L0 + 2)) - 1) // 3
    assert Ent == nt
    n = 1 << (L0 + 1) if tree_type == "binary" else 1 << (2 * L0 + 2)
    ns = 1 << (L0 + 1)
    x = np.empty(ns) if tree_type == "binary" else np.empty((ns, ns))
    assert isvalidtree(x, tree)
    result = np.zeros(n + nt, dtype=bool)
    result[0] = True
    for i in range(1, nt + 1):
        if not tree[i - 1]:
            continue
        result[i - 1] = False
        if tree_type == "binary":
            result[2 * i - 1] = True
            result[2 * i] = True
        else:
            result[4 * i - 3: 4 * i + 1] = True
    return result


def _range(lo, hi):
    return range(lo, hi + 1)


def getrowrange(n, idx):
    """Utils.jl:465-490 -> 1-based inclusive range (python `range(lo, hi+1)`)."""
    L0 = maxtransformlevels(int(n))
    k = ((1 << (2 * L0 + 2)) - 1) // 3
    assert 0 < idx <= k
    if idx == 1:
        return _range(1, n)
    parent = (idx + 2) // 4
    pr = getrowrange(n, parent)
    mid = (pr[0] + pr[-1]) // 2
    return _range(pr[0], mid) if idx < 4 * parent else _range(mid + 1, pr[-1])


def getcolrange(n, idx):
    """Utils.jl:517-542"""
    L0 = maxtransformlevels(int(n))
    k = ((1 << (2 * L0 + 2)) - 1) // 3
    assert 0 < idx <= k
    if idx == 1:
        return _range(1, n)
    parent = (idx + 2) // 4
    pr = getcolrange(n, parent)
    mid = (pr[0] + pr[-1]) // 2
    return _range(pr[0], mid) if idx % 2 == 0 else _range(mid + 1, pr[-1])


def main2depthshift(sm, L):
    """Utils.jl:297-305"""
    assert sm < (1 << L)
    sd, acc = [0], 0
    for d in range(L):
        acc += ((sm >> d) & 1) << d
        sd.append(acc)
    return sd


def coarsestscalingrange(x, tree, redundant=False):
    """Utils.jl:351-371"""
    n = x if isinstance(x, (int, np.integer)) else _shape(x)[0]
    tree = np.asarray(tree, dtype=bool)
    L = getdepth(tree.size, "binary")
    assert L + 1 == maxtransformlevels(int(n))
    i, j = 1, 0
    while i < tree.size and tree[i - 1]:
        i = getchildindex(i, "left")
        j += 1
    return (_range(1, n), i) if redundant else _range(1, n >> j)


def finestdetailrange(x, tree, redundant=False):
    """Utils.jl:416-438"""
    n = x if isinstance(x, (int, np.integer)) else _shape(x)[0]
    tree = np.asarray(tree, dtype=bool)
    L = getdepth(tree.size, "binary")
    assert L + 1 == maxtransformlevels(int(n))
    i, j = 1, 0
    while i <= tree.size and tree[i - 1]:
        i = getchildindex(i, "right")
        j += 1
    return (_range(1, n), i) if redundant else _range(n - nodelength(n, j) + 1, n)


def delete_subtree(bt, i, tree_type):
    """BestBasis.jl:128-140 (in place, returns bt)"""
    assert 1 <= i <= bt.size
    assert tree_type in ("binary", "quad")
    bt[i - 1] = False
    kids = (2 * i, 2 * i + 1) if tree_type == "binary" else tuple(4 * i - 2 + c for c in range(4))
    for c in kids:
        if c <= bt.size and bt[c - 1]:
            delete_subtree(bt, c, tree_type)
    return bt


def leaf_blocks_1d(n, tree):
    """Host helper for the device kernels: for a valid binary tree over a length-n signal return
    (Leff, col) where Leff = depth of the deepest leaf and col[blk] = depth of the leaf that owns
    positions [blk*(n>>Leff), (blk+1)*(n>>Leff)).  Same traversal as getbasiscoef (Utils.jl:117-131)."""
    tree = np.asarray(tree, dtype=bool)
    nt = tree.size
    Leff = 0
    for i in range(1, nt + 1):
        if tree[i - 1]:
            Leff = max(Leff, getdepth(i, "binary") + 1)
    col = np.zeros(1 << Leff, dtype=np.int32)
    stack = [(1, 0, 0)]
    while stack:
        node, d, j = stack.pop()
        if node <= nt and tree[node - 1]:
            stack.append((2 * node, d + 1, 2 * j))
            stack.append((2 * node + 1, d + 1, 2 * j + 1))
        else:
            w = 1 << (Leff - d)
            col[j * w:(j + 1) * w] = d
    return Leff, col
