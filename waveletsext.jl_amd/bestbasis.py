"""Joint best basis (JBB): host-side mirror of the JBB slice of the reference's `BestBasis`
module (src/mod/BestBasis.jl:59-83,128-140,194-201; bestbasis/bestbasis_tree.jl:43-46,150-180;
bestbasis/bestbasis_costs.jl:44-57,127-132) for 1-D signals.  Moments and costs run on the GPU;
the bottom-up tree selection is the library's tiny host routine."""
import ctypes

import numpy as np

from . import _lib
from ._arrays import Arg, qmf_arg, to_numpy
from .dwt import _call
from .filters import ArgumentError
from .util import gettreelength, maxtransformlevels


class LoglpCost:
    """bestbasis_costs.jl:44-46"""

    def __init__(self, p=2):
        self.p = p


class NormCost:
    """bestbasis_costs.jl:55-57"""

    def __init__(self, p=1):
        self.p = p


class JBB:
    """bestbasis_tree.jl:43-46"""

    def __init__(self, cost=None, redundant=False):
        self.cost = LoglpCost(2) if cost is None else cost
        if not isinstance(self.cost, (LoglpCost, NormCost)):
            raise TypeError("JBB cost must be LoglpCost or NormCost")
        self.redundant = bool(redundant)


def _cost_kind(cost):
    return (0, float(cost.p)) if isinstance(cost, LoglpCost) else (1, float(cost.p))


def jbb_moments(X, accumulate_into=None):
    """(sum, sumsq) over the signal (last) axis of a decomposition X (n, k, N) or (n, m, k, N)."""
    X = Arg(X)
    assert 3 <= X.arr.ndim <= 4
    shp, N = X.shape[:-1], X.shape[-1]
    if accumulate_into is None:
        s, q = X.new(shp), X.new(shp)
        acc = 0
    else:
        s, q = Arg(accumulate_into[0]), Arg(accumulate_into[1])
        acc = 1
    _call("wx_jbb_moments", X.suffix, X.ptr, s.ptr, q.ptr, int(np.prod(shp, dtype=np.int64)), N, acc, X.stream())
    return s.arr, q.arr


def costs_from_moments(s, q, Ntot, method=None):
    method = JBB() if method is None else method
    s, q = Arg(s), Arg(q)
    kind, p = _cost_kind(method.cost)
    if s.arr.ndim == 3:                                               # 2-D signals: (n, m, k)
        n, m, k = s.shape
        ncost = k if method.redundant else gettreelength(1 << k, 1 << k)
        costs = s.new((ncost,))
        _call("wx_jbb_costs2d", s.suffix, s.ptr, q.ptr, int(Ntot), n, m, k, int(method.redundant), kind, p, costs.ptr,
              s.stream())
        return costs.arr
    n, k = s.shape
    ncost = k if method.redundant else gettreelength(1 << k)
    costs = s.new((ncost,))
    _call("wx_jbb_costs", s.suffix, s.ptr, q.ptr, int(Ntot), n, k, int(method.redundant), kind, p, costs.ptr,
          s.stream())
    return costs.arr


def tree_costs(X, method=None):
    """tree_costs(X::Array{T,3}, method::JBB) bestbasis_tree.jl:150-180"""
    method = JBB() if method is None else method
    if not isinstance(method, JBB):
        raise _lib.WxError(_lib.WX_EUNSUPPORTED, "only the JBB best-basis type is on the device path")
    Xa = Arg(X)
    assert 3 <= Xa.arr.ndim <= 4
    s, q = jbb_moments(Xa.arr)
    costs = costs_from_moments(s, q, Xa.shape[-1], method)
    c = to_numpy(costs)
    assert not np.isnan(c).any()                                      # @assert all(sigma .>= 0) (:158)
    return costs


def bestbasis_treeselection(costs, n, *args):
    """bestbasis_treeselection(costs, n[, type]) BestBasis.jl:59-83 and (costs, n, m[, type]) :85-110; `costs`
    (host copy) is mutated like the reference, returns the BitVector."""
    args = list(args)
    kind = args.pop() if args and isinstance(args[-1], str) else "min"
    if kind not in ("min", "max"):
        raise ArgumentError("Unsupported type %s." % kind)
    c = np.array(to_numpy(costs), copy=True)
    if c.dtype not in (np.float32, np.float64):
        c = c.astype(np.float64)
    k = c.size
    if args:                                                          # quad tree
        m = int(args[0])
        assert k <= gettreelength(2 * n, 2 * m)                       # BestBasis.jl:90
        tree = np.zeros(gettreelength(n, m), dtype=np.uint8)
        fn = getattr(_lib.lib(), "wx_treeselect2d_f64" if c.dtype == np.float64 else "wx_treeselect2d_f32")
        _lib.check(fn(ctypes.c_void_p(c.ctypes.data), k, n, m, 0 if kind == "min" else 1,
                      ctypes.c_void_p(tree.ctypes.data)))
        return tree.astype(bool)
    assert k <= gettreelength(2 * n)                                  # BestBasis.jl:63
    tree = np.zeros(max(n - 1, 0), dtype=np.uint8)
    fn = getattr(_lib.lib(), "wx_treeselect_f64" if c.dtype == np.float64 else "wx_treeselect_f32")
    _lib.check(fn(ctypes.c_void_p(c.ctypes.data), k, n, 0 if kind == "min" else 1, ctypes.c_void_p(tree.ctypes.data)))
    return tree.astype(bool)


def bestbasistree(X, method=None):
    """bestbasistree(X, JBB(...)) BestBasis.jl:194-201 (X is (n, k, N))"""
    method = JBB() if method is None else method
    Xa = Arg(X)
    assert 3 <= Xa.arr.ndim <= 4
    costs = tree_costs(Xa.arr, method)
    return bestbasis_treeselection(costs, *Xa.shape[:-2])


def acwpd_jbb_moments(x, wt, L=None, accumulate_into=None):
    """acwpdall + JBB moments fused on the device (the (n, 2^(L+1)-1, N) table is never returned):
    BASELINE config 5.  Returns (sum, sumsq), each (n, 2^(L+1)-1)."""
    x = Arg(x)
    assert x.arr.ndim == 2
    n, N = x.shape
    L = maxtransformlevels(n) if L is None else int(L)
    ncols = (1 << (L + 1)) - 1
    if accumulate_into is None:
        s, q = x.new((n, ncols)), x.new((n, ncols))
        acc = 0
    else:
        s, q = Arg(accumulate_into[0]), Arg(accumulate_into[1])
        acc = 1
    qq, qp, F = qmf_arg(wt)
    _call("wx_acwpd_jbb_moments", "_f64", x.ptr, s.ptr, q.ptr, n, L, N, qp, F, acc, x.stream())
    return s.arr, q.arr
