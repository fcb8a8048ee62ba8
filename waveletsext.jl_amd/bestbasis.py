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


class ShannonEntropyCost:
    """bestbasis_costs.jl:76"""


class LogEnergyEntropyCost:
    """bestbasis_costs.jl:86"""


class BB:
    """Standard (per-signal) best basis, bestbasis_tree.jl:60-63"""

    def __init__(self, cost=None, redundant=False):
        self.cost = ShannonEntropyCost() if cost is None else cost
        if not isinstance(self.cost, (ShannonEntropyCost, LogEnergyEntropyCost)):
            raise TypeError("BB cost must be ShannonEntropyCost or LogEnergyEntropyCost")
        self.redundant = bool(redundant)


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


def _bb_costs(Xa, method, batched):
    """(ncost[, N]) costs of one signal (n, k) / (n, m, k) or of a batch (..., N) for BB"""
    nd = Xa.arr.ndim - (1 if batched else 0)
    assert 2 <= nd <= 3
    N = Xa.shape[-1] if batched else 1
    kind = 0 if isinstance(method.cost, ShannonEntropyCost) else 1
    if nd == 2:
        n, k = Xa.shape[:2]
        ncost = k if method.redundant else (1 << k) - 1
        costs = Xa.new((ncost, N) if batched else (ncost,))
        _call("wx_bb_costs", Xa.suffix, Xa.ptr, costs.ptr, n, k, N, int(method.redundant), kind, Xa.stream())
    else:
        n, m, k = Xa.shape[:3]
        ncost = k if method.redundant else gettreelength(1 << k, 1 << k)
        costs = Xa.new((ncost, N) if batched else (ncost,))
        _call("wx_bb_costs2d", Xa.suffix, Xa.ptr, costs.ptr, n, m, k, N, int(method.redundant), kind, Xa.stream())
    return costs


def _bb_trees(costs, sig, N, kind="min"):
    """bestbasis_treeselection for every column of costs (ncost, N) on the device -> uint8 (ntree, N)"""
    ncost = costs.shape[0]
    ntree = sig[0] - 1 if len(sig) == 1 else gettreelength(*sig)
    if costs.kind == "torch":
        import torch
        trees = torch.empty((N, ntree), dtype=torch.uint8, device=costs.device)
        tp, st = ctypes.c_void_p(trees.data_ptr()), costs.stream()
    else:
        trees = np.empty((N, ntree), dtype=np.uint8)
        tp, st = ctypes.c_void_p(trees.ctypes.data), ctypes.c_void_p(0)
    fn = getattr(_lib.lib(), "wx_treeselect_batch" + costs.suffix)
    _lib.check(fn(costs.ptr, ncost, sig[0], sig[1] if len(sig) == 2 else 0, 0 if kind == "min" else 1, N, tp, st))
    return trees


def bestbasistreeall(X, method=None):
    """bestbasistreeall(X, BB(...)) BestBasis.jl:253-262: X (n, k, N) or (n, m, k, N) -> BitMatrix (tree length, N);
    costs and the selection of all N trees run on the device."""
    method = BB() if method is None else method
    if not isinstance(method, BB):
        raise TypeError("bestbasistreeall takes a BB method (BestBasis.jl:253)")
    Xa = Arg(X)
    assert 3 <= Xa.arr.ndim <= 4                                      # BestBasis.jl:254
    sig = Xa.shape[:-2]
    costs = _bb_costs(Xa, method, True)
    trees = _bb_trees(costs, sig, Xa.shape[-1])
    if costs.kind == "torch":
        # device -> page-locked host memory (torch's caching host allocator) -> numpy view of it: .cpu() into pageable memory and a uint8 -> bool
        # copy were most of the call for short signals (19 MB of trees per GiB of table at 64 samples)
        import torch
        host = torch.empty(trees.shape, dtype=torch.uint8, pin_memory=True)
        host.copy_(trees, non_blocking=True)
        torch.cuda.current_stream(trees.device).synchronize()
        t = host.numpy()
    else:
        t = trees
    return t.T.view(np.bool_)                          # (tree length, N), column-major like the reference's BitMatrix; the bytes are 0 / 1


def tree_costs(X, method=None):
    """tree_costs(X::Array{T,3}, method::JBB) bestbasis_tree.jl:150-180; tree_costs(X, ::BB) :210-258 (one signal)"""
    method = JBB() if method is None else method
    if isinstance(method, BB):
        Xa = Arg(X)
        assert 2 <= Xa.arr.ndim <= 3
        return _bb_costs(Xa, method, False).arr
    if not isinstance(method, JBB):
        raise _lib.WxError(_lib.WX_EUNSUPPORTED, "only the JBB best-basis type is on the device path")
    Xa = Arg(X)
    assert 3 <= Xa.arr.ndim <= 4
    s, q = jbb_moments(Xa.arr)
    costs = costs_from_moments(s, q, Xa.shape[-1], method)
    c = to_numpy(costs)
    assert not np.isnan(c).any()                                      # @assert all(sigma .>= 0) (:158)
    return costs


def bestbasis_treeselection(costs, n, *args, return_gap=False):
    """bestbasis_treeselection(costs, n[, type]) BestBasis.jl:59-83 and (costs, n, m[, type]) :85-110; `costs`
    (host copy) is mutated like the reference, returns the BitVector.  return_gap=True (1-D) also returns the margin of the
    closest split decision, min |cc - pc| / |pc| over the decisions taken (`wx_treeselect_gap_*`)."""
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
    if return_gap:
        gap = ctypes.c_double(0.0)
        fn = getattr(_lib.lib(), "wx_treeselect_gap_f64" if c.dtype == np.float64 else "wx_treeselect_gap_f32")
        _lib.check(fn(ctypes.c_void_p(c.ctypes.data), k, n, 0 if kind == "min" else 1, ctypes.c_void_p(tree.ctypes.data),
                      ctypes.byref(gap)))
        return tree.astype(bool), float(gap.value)
    fn = getattr(_lib.lib(), "wx_treeselect_f64" if c.dtype == np.float64 else "wx_treeselect_f32")
    _lib.check(fn(ctypes.c_void_p(c.ctypes.data), k, n, 0 if kind == "min" else 1, ctypes.c_void_p(tree.ctypes.data)))
    return tree.astype(bool)


def bestbasistree(X, method=None):
    """bestbasistree(X, JBB(...)) BestBasis.jl:194-201 (X is (n, k, N)); bestbasistree(X, BB(...)) :203-210 (one
    signal, X is (n, k) or (n, m, k))"""
    method = JBB() if method is None else method
    Xa = Arg(X)
    if isinstance(method, BB):
        assert 2 <= Xa.arr.ndim <= 3                                  # BestBasis.jl:205
        costs = _bb_costs(Xa, method, False)
        trees = _bb_trees(costs, Xa.shape[:-1], 1)
        t = trees.cpu().numpy() if costs.kind == "torch" else trees
        return t[0].astype(bool)
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
        for a in (s, q):
            if a.shape != (n, ncols) or a.kind != x.kind:
                raise TypeError("accumulate_into must be two (n, 2^(L+1)-1) arrays of the same kind as x")
    # ACWT is Float64-only like the reference (acwt_one_level.jl:101-106): the suffix of every array selects the entry
    # point, so Float32 data raises WX_EUNSUPPORTED instead of being reinterpreted
    for a in (s, q):
        if a.suffix != x.suffix:
            raise TypeError("accumulate_into element type differs from x's")
    qq, qp, F = qmf_arg(wt)
    _call("wx_acwpd_jbb_moments", x.suffix, x.ptr, s.ptr, q.ptr, n, L, N, qp, F, acc, x.stream())
    return s.arr, q.arr
