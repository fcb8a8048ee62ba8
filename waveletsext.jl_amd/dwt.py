"""Decimated wavelet-packet transforms: host-side mirror of the reference's `DWT` module
(src/mod/DWT.jl, src/mod/dwt/dwt_all.jl) plus the 1-D `wpt/iwpt` methods it borrows from
Wavelets.jl.  Same names, argument order and error behaviour; the `!` forms are spelled with a
trailing underscore (`wpd_`).  Every function ends in one call into the C ABI (HIP kernels);
there is no CPU path.
"""
import numpy as np

from . import _lib
from ._arrays import Arg, out_arg, qmf_arg, tree_arg
from .util import isdyadic, maxtransformlevels


def _is_tree(a):
    return isinstance(a, np.ndarray) or (isinstance(a, (list, tuple)) and len(a) and isinstance(a[0], (bool, np.bool_)))


def _split_Ltree(arg, default_L):
    """(L | tree | missing) -> (L, tree)"""
    if arg is None:
        return int(default_L), None
    if _is_tree(arg):
        return 0, np.asarray(arg, dtype=bool)
    return int(arg), None


def _call(name, suffix, *args):
    fn = getattr(_lib.lib(), name + suffix, None)
    if fn is None:
        raise _lib.WxError(_lib.WX_EUNSUPPORTED, "%s%s is not exported by libwaveletsext_hip.so" % (name, suffix))
    _lib.check(fn(*args))


# ---------------------------------------------------------------------------------------------
# core batched entry points (signal dims first, batch last)
# ---------------------------------------------------------------------------------------------
def _wpd_batched(x, y, sig_ndim, wt, L):
    q, qp, F = qmf_arg(wt)
    if sig_ndim == 1:
        n, B = x.shape[0], int(np.prod(x.shape[1:], dtype=np.int64))
        _call("wx_wpd1d", x.suffix, x.ptr, y.ptr, n, L, B, qp, F, x.stream())
    else:
        m, n, B = x.shape[0], x.shape[1], int(np.prod(x.shape[2:], dtype=np.int64))
        _call("wx_wpd2d", x.suffix, x.ptr, y.ptr, m, n, L, B, qp, F, x.stream())


def _wpt_batched(name, x, y, sig_ndim, wt, L, tree):
    q, qp, F = qmf_arg(wt)
    tk, tp, nt = tree_arg(tree)
    if sig_ndim == 1:
        n, B = x.shape[0], int(np.prod(x.shape[1:], dtype=np.int64))
        _call(name + "1d", x.suffix, x.ptr, y.ptr, n, L, tp, nt, B, qp, F, x.stream())
    else:
        m, n, B = x.shape[0], x.shape[1], int(np.prod(x.shape[2:], dtype=np.int64))
        _call(name + "2d", x.suffix, x.ptr, y.ptr, m, n, L, tp, nt, B, qp, F, x.stream())


def _iwpd_batched(xw, xh, sig_ndim, wt, L, tree):
    q, qp, F = qmf_arg(wt)
    tk, tp, nt = tree_arg(tree)
    if sig_ndim == 1:
        n, k = xw.shape[0], xw.shape[1]
        B = int(np.prod(xw.shape[2:], dtype=np.int64))
        _call("wx_iwpd1d", xw.suffix, xw.ptr, xh.ptr, n, k, L, tp, nt, B, qp, F, xw.stream())
    else:
        m, n, k = xw.shape[0], xw.shape[1], xw.shape[2]
        B = int(np.prod(xw.shape[3:], dtype=np.int64))
        _call("wx_iwpd2d", xw.suffix, xw.ptr, xh.ptr, m, n, k, L, tp, nt, B, qp, F, xw.stream())


# ---------------------------------------------------------------------------------------------
# wpd / wpd!  (DWT.jl:60-88, 131-209)
# ---------------------------------------------------------------------------------------------
def wpd(x, wt, L=None):
    x = Arg(x)
    assert x.arr.ndim in (1, 2)
    if x.arr.ndim == 1:
        assert isdyadic(x.shape[0])                                   # DWT.jl:64
    L = maxtransformlevels(x.arr) if L is None else int(L)
    assert 0 <= L <= maxtransformlevels(x.arr)                        # DWT.jl:65,79
    y = x.new(x.shape + (L + 1,))
    _wpd_batched(x, y, x.arr.ndim, wt, L)
    return y.arr


def wpd_(y, x, wt, L=None):
    x = Arg(x)
    L = maxtransformlevels(x.arr) if L is None else int(L)
    assert 0 <= L <= maxtransformlevels(x.arr)                        # DWT.jl:137,170
    assert tuple(y.shape) == x.shape + (L + 1,)                       # DWT.jl:138,171
    ya = out_arg(y, x)
    _wpd_batched(x, ya, x.arr.ndim, wt, L)
    return y


def wpdall(x, wt, L=None):
    """dwt/dwt_all.jl:260-282"""
    x = Arg(x)
    assert x.arr.ndim > 1
    sz, N = x.shape[:-1], x.shape[-1]
    Lmax = maxtransformlevels(int(min(sz)))
    L = Lmax if L is None else int(L)
    assert 0 <= L <= Lmax
    y = x.new(sz + (L + 1, N))
    _wpd_batched(x, y, len(sz), wt, L)
    return y.arr


# ---------------------------------------------------------------------------------------------
# iwpd / iwpd!  (DWT.jl:257-401)
# ---------------------------------------------------------------------------------------------
def iwpd(xw, wt, L_or_tree=None):
    xw = Arg(xw)
    assert xw.arr.ndim >= 2
    dim = xw.shape[:-1]
    L, tree = _split_Ltree(L_or_tree, maxtransformlevels(int(min(dim))))
    xh = xw.new(dim)
    _iwpd_batched(xw, xh, len(dim), wt, L, tree)
    return xh.arr


def iwpd_(xh, xw, wt, L_or_tree=None):
    xw = Arg(xw)
    sig = tuple(xh.shape)
    L, tree = _split_Ltree(L_or_tree, maxtransformlevels(int(min(sig))))
    assert sig == xw.shape[:-1]                                       # DWT.jl:344,358-359
    xa = out_arg(xh, xw)
    _iwpd_batched(xw, xa, len(sig), wt, L, tree)
    return xh


def iwpdall(xw, wt, L_or_tree=None):
    """dwt/dwt_all.jl:324-342: iwpd!(x̂ᵢ, xwᵢ, args...) over the last dimension."""
    xw = Arg(xw)
    assert xw.arr.ndim > 2
    sz, N = xw.shape[:-2], xw.shape[-1]
    L, tree = _split_Ltree(L_or_tree, maxtransformlevels(int(min(sz))))
    xh = xw.new(sz + (N,))
    _iwpd_batched(xw, xh, len(sz), wt, L, tree)
    return xh.arr


# ---------------------------------------------------------------------------------------------
# wpt / iwpt  (1-D: Wavelets.jl; 2-D: DWT.jl:440-548, 594-710)
# ---------------------------------------------------------------------------------------------
def _wpt_like(name, x, wt, L_or_tree, y=None):
    x = Arg(x)
    assert x.arr.ndim in (1, 2)
    L, tree = _split_Ltree(L_or_tree, maxtransformlevels(x.arr))
    if y is None:
        ya = x.new(x.shape)
    else:
        assert tuple(y.shape) == x.shape                              # DWT.jl:505,667
        ya = out_arg(y, x)
    _wpt_batched(name, x, ya, x.arr.ndim, wt, L, tree)
    return ya.arr if y is None else y


def wpt(x, wt, L_or_tree=None):
    return _wpt_like("wx_wpt", x, wt, L_or_tree)


def wpt_(y, x, wt, L_or_tree=None):
    return _wpt_like("wx_wpt", x, wt, L_or_tree, y)


def iwpt(xw, wt, L_or_tree=None):
    return _wpt_like("wx_iwpt", xw, wt, L_or_tree)


def iwpt_(xh, xw, wt, L_or_tree=None):
    return _wpt_like("wx_iwpt", xw, wt, L_or_tree, xh)


def _wptall_like(name, x, wt, L_or_tree):
    x = Arg(x)
    assert x.arr.ndim > 1                                             # dwt_all.jl:153,211
    sz = x.shape[:-1]
    L, tree = _split_Ltree(L_or_tree, maxtransformlevels(int(min(sz))))
    y = x.new(x.shape)
    _wpt_batched(name, x, y, len(sz), wt, L, tree)
    return y.arr


def wptall(x, wt, L_or_tree=None):
    """dwt/dwt_all.jl:152-166"""
    return _wptall_like("wx_wpt", x, wt, L_or_tree)


def iwptall(xw, wt, L_or_tree=None):
    """dwt/dwt_all.jl:210-225"""
    return _wptall_like("wx_iwpt", xw, wt, L_or_tree)


# ---------------------------------------------------------------------------------------------
# dwt / idwt / dwtall / idwtall: the non-packet pyramid transform is the packet
# transform along the :dwt tree (test/transforms.jl:42: `dwt(x, wt) ≈ wpt(x, wt, maketree(x,:dwt))`;
# 1-D Wavelets.jl pyramid order [s_L d_L .. d_1] == the leaves of that tree in natural order).
# Batch drivers: dwt/dwt_all.jl:39-54, 95-110.  1-D and 2-D (the reference also admits 3-D).
# ---------------------------------------------------------------------------------------------
def _dwt3d(name, xa, wt, L, batched):
    """3-D signals (dwt_all.jl:8-9 "1-D, 2-D, and 3-D"): Wavelets.jl's separable pyramid on cubes, csrc/wx_dwt3d.hip"""
    import ctypes
    sig = xa.shape[:-1] if batched else xa.shape
    N = xa.shape[-1] if batched else 1
    Lmax = maxtransformlevels(int(min(sig)))
    Lv = Lmax if L is None else int(L)
    assert 0 <= Lv <= Lmax
    out = xa.new(xa.shape)
    q = np.ascontiguousarray(wt.qmf, dtype=np.float64)
    fn = getattr(_lib.lib(), ("wx_dwt3d" if name == "wx_wpt" else "wx_idwt3d") + xa.suffix)
    _lib.check(fn(xa.ptr, out.ptr, sig[0], sig[1], sig[2], Lv, N, ctypes.c_void_p(q.ctypes.data), q.size, xa.stream()))
    return out.arr


def _dwt_like(name, x, wt, L, batched):
    xa = Arg(x)
    sig = xa.shape[:-1] if batched else xa.shape
    assert len(sig) in (1, 2, 3)
    if len(sig) == 3:
        return _dwt3d(name, xa, wt, L, batched)
    if batched:
        assert xa.arr.ndim > 1                                        # dwt_all.jl:40,96
    Lmax = maxtransformlevels(int(min(sig)))
    Lv = Lmax if L is None else int(L)
    assert 0 <= Lv <= Lmax
    if Lv == 0:
        return (_wptall_like if batched else _wpt_like)(name, x, wt, 0)
    from .util import maketree
    tree = maketree(*sig, Lv, "dwt")
    return (_wptall_like if batched else _wpt_like)(name, x, wt, tree)


def dwt(x, wt, L=None):
    return _dwt_like("wx_wpt", x, wt, L, False)


def idwt(xw, wt, L=None):
    return _dwt_like("wx_iwpt", xw, wt, L, False)


def dwtall(x, wt, L=None):
    """dwt/dwt_all.jl:39-54"""
    return _dwt_like("wx_wpt", x, wt, L, True)


def idwtall(xw, wt, L=None):
    """dwt/dwt_all.jl:95-110"""
    return _dwt_like("wx_iwpt", xw, wt, L, True)


# ---------------------------------------------------------------------------------------------
# getbasiscoef / getbasiscoefall (Utils.jl:101-225), 1-D and 2-D tables on device
# ---------------------------------------------------------------------------------------------
def getbasiscoef(Xw, tree):
    Xw = Arg(Xw)
    assert 2 <= Xw.arr.ndim <= 3                                      # Utils.jl:103
    tk, tp, nt = tree_arg(np.asarray(tree, dtype=bool))
    if Xw.arr.ndim == 3:
        m, n, k = Xw.shape
        out = Xw.new((m, n))
        _call("wx_getbasiscoef2d", Xw.suffix, Xw.ptr, out.ptr, m, n, k, tp, nt, 1, Xw.stream())
        return out.arr
    n, k = Xw.shape
    out = Xw.new((n,))
    _call("wx_getbasiscoef1d", Xw.suffix, Xw.ptr, out.ptr, n, k, tp, nt, 1, Xw.stream())
    return out.arr


def getbasiscoefall(Xw, tree):
    Xw = Arg(Xw)
    assert 3 <= Xw.arr.ndim <= 4                                      # Utils.jl:175
    tree = np.asarray(tree, dtype=bool)
    if Xw.arr.ndim == 4:
        mm, nn, k, N = Xw.shape
        import ctypes
        out = Xw.new((mm, nn, N))
        if tree.ndim == 2:                                            # Utils.jl:199-225: one tree per image, one launch
            assert N == tree.shape[1]
            tb = np.asfortranarray(tree.astype(np.uint8))
            _call("wx_getbasiscoef2d_trees", Xw.suffix, Xw.ptr, out.ptr, mm, nn, k, ctypes.c_void_p(tb.ctypes.data), tb.shape[0], N,
                  Xw.stream())
        else:
            tk, tp, nt = tree_arg(tree)
            _call("wx_getbasiscoef2d", Xw.suffix, Xw.ptr, out.ptr, mm, nn, k, tp, nt, N, Xw.stream())
        return out.arr
    n, k, m = Xw.shape
    if tree.ndim == 2:                                                # Utils.jl:199-225: one tree per signal
        nt, mt = tree.shape
        assert m == mt
        out = Xw.new((n, m))
        import ctypes
        tb = np.asfortranarray(tree.astype(np.uint8))                 # (ntree, m): column i = the tree of signal i
        _call("wx_getbasiscoef1d_trees", Xw.suffix, Xw.ptr, out.ptr, n, k, ctypes.c_void_p(tb.ctypes.data), nt, m, Xw.stream())
        return out.arr
    tk, tp, nt = tree_arg(tree)
    out = Xw.new((n, m))
    _call("wx_getbasiscoef1d", Xw.suffix, Xw.ptr, out.ptr, n, k, tp, nt, m, Xw.stream())
    return out.arr
