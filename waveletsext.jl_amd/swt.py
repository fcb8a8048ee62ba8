"""Stationary (undecimated) wavelet transforms: host-side mirror of the reference's `SWT` module
(src/mod/SWT.jl, src/mod/swt/swt_all.jl) for 1-D signals.  `sm=None` selects the average-based
inverse, an integer the shift-based one.  Mutating forms carry a trailing underscore."""
from . import _lib
from ._arrays import Arg, out_arg, qmf_arg, tree_arg
from .dwt import _call, _split_Ltree
from .filters import ArgumentError
from .util import isdyadic, maxtransformlevels, ndyadicscales


def _req_1d(nd):
    if nd not in (1, 2):
        raise _lib.WxError(_lib.WX_EUNSUPPORTED, "redundant transforms are implemented for 1-D and 2-D signals")


def _ncols(kind, L, nd):
    """columns / slices of the coefficient container (SWT.jl:68, 401, 800)"""
    if nd == 1:
        return {"dwt": L + 1, "wpt": 1 << L, "wpd": (1 << (L + 1)) - 1}[kind]
    return {"dwt": 3 * L + 1, "wpt": 1 << (2 * L), "wpd": ((1 << (2 * (L + 1))) - 1) // 3}[kind]


def _check_L(x_shape, L):
    Lmax = maxtransformlevels(int(min(x_shape)))
    if not L <= Lmax:
        raise ArgumentError("Too many transform levels (length(x) < 2^L")      # SWT.jl:64-65
    if not L >= 1:
        raise ArgumentError("L must be >= 1")


def _fwd(name, ncols_of, x, wt, L, batched, y=None):
    x = Arg(x)
    if batched:
        assert 2 <= x.arr.ndim <= 3                                  # swt_all.jl:37,160,283
        sig, N = x.shape[:-1], x.shape[-1]
    else:
        assert 1 <= x.arr.ndim <= 2                                  # SWT.jl:63
        sig, N = x.shape, None
    _req_1d(len(sig))
    L = maxtransformlevels(int(min(sig))) if L is None else int(L)
    _check_L(sig, L)
    shape = sig + (_ncols(ncols_of, L, len(sig)),) + ((N,) if batched else ())
    if y is None:
        ya = x.new(shape)
    else:
        assert tuple(y.shape) == shape
        ya = out_arg(y, x)
    q, qp, F = qmf_arg(wt)
    if len(sig) == 1:
        _call(name, x.suffix, x.ptr, ya.ptr, sig[0], L, 1 if N is None else N, qp, F, x.stream())
    else:
        _call(name.replace("1d", "2d"), x.suffix, x.ptr, ya.ptr, sig[0], sig[1], L, 1 if N is None else N, qp, F,
              x.stream())
    return ya.arr if y is None else y


def sdwt(x, wt, L=None):
    """SWT.jl:60-74"""
    return _fwd("wx_sdwt1d", "dwt", x, wt, L, False)


def sdwt_(xw, x, wt, L=None):
    return _fwd("wx_sdwt1d", "dwt", x, wt, L, False, xw)


def sdwtall(x, wt, L=None):
    """swt_all.jl:33-50"""
    return _fwd("wx_sdwt1d", "dwt", x, wt, L, True)


def swpt(x, wt, L=None):
    """SWT.jl:390-406"""
    return _fwd("wx_swpt1d", "wpt", x, wt, L, False)


def swpt_(xw, x, wt, L=None):
    return _fwd("wx_swpt1d", "wpt", x, wt, L, False, xw)


def swptall(x, wt, L=None):
    """swt_all.jl:156-176"""
    return _fwd("wx_swpt1d", "wpt", x, wt, L, True)


def swpd(x, wt, L=None):
    """SWT.jl:790-806"""
    return _fwd("wx_swpd1d", "wpd", x, wt, L, False)


def swpd_(xw, x, wt, L=None):
    return _fwd("wx_swpd1d", "wpd", x, wt, L, False, xw)


def swpdall(x, wt, L=None):
    """swt_all.jl:279-299"""
    return _fwd("wx_swpd1d", "wpd", x, wt, L, True)


def _inv_common(xw, batched):
    xw = Arg(xw)
    if batched:
        assert 3 <= xw.arr.ndim <= 4                                 # swt_all.jl:90,213,368
        sig, k, N = xw.shape[:-2], xw.shape[-2], xw.shape[-1]
    else:
        assert 2 <= xw.arr.ndim <= 3                                 # SWT.jl:191,564,940
        sig, k, N = xw.shape[:-1], xw.shape[-1], None
    _req_1d(len(sig))
    return xw, sig, k, N


def _smv(sm):
    return -1 if sm is None else int(sm)


def _isdwt(xw, wt, sm, batched, x=None):
    xw, sig, k, N = _inv_common(xw, batched)
    L = k - 1 if len(sig) == 1 else (k - 1) // 3                      # SWT.jl:265, 292
    if sm is not None:
        assert sm >= 1 and sm < (1 << L)                             # SWT.jl:266,293 + Utils.jl:298
    out = xw.new(sig + ((N,) if batched else ())) if x is None else out_arg(x, xw)
    q, qp, F = qmf_arg(wt)
    if len(sig) == 1:
        _call("wx_isdwt1d", xw.suffix, xw.ptr, out.ptr, sig[0], L, _smv(sm), 1 if N is None else N, qp, F, xw.stream())
    else:
        _call("wx_isdwt2d", xw.suffix, xw.ptr, out.ptr, sig[0], sig[1], L, _smv(sm), 1 if N is None else N, qp, F,
              xw.stream())
    return out.arr if x is None else x


def isdwt(xw, wt, sm=None):
    """SWT.jl:190-205, 259-330"""
    return _isdwt(xw, wt, sm, False)


def isdwt_(x, xw, wt, sm=None):
    return _isdwt(xw, wt, sm, False, x)


def isdwtall(xw, wt, sm=None):
    """swt_all.jl:89-122"""
    return _isdwt(xw, wt, sm, True)


def _iswpt(xw, wt, sm, batched, x=None):
    xw, sig, m, N = _inv_common(xw, batched)
    q, qp, F = qmf_arg(wt)
    if len(sig) == 2:
        L = 0
        while (1 << (2 * (L + 1))) <= m:
            L += 1
        if (1 << (2 * L)) != m:
            raise ArgumentError("Size of dimension 3 is not a power of 4.")       # SWT.jl:653
        if x is not None:
            assert tuple(x.shape)[:2] == sig                                       # SWT.jl:654-655
        assert L <= maxtransformlevels(int(min(sig)))                              # SWT.jl:656
        out = xw.new(sig + ((N,) if batched else ())) if x is None else out_arg(x, xw)
        _call("wx_iswpt2d", xw.suffix, xw.ptr, out.ptr, sig[0], sig[1], L, _smv(sm), 1 if N is None else N, qp, F,
              xw.stream())
        return out.arr if x is None else x
    if not isdyadic(m):
        raise ArgumentError("Number of columns of xw is not dyadic.")             # SWT.jl:619
    L = ndyadicscales(m)
    if not L <= maxtransformlevels(sig[0]):
        raise ArgumentError("Number of nodes in `xw` is more than possible number of nodes at any depth "
                            "for signal of length `n`")                            # SWT.jl:620-621
    out = xw.new(sig + ((N,) if batched else ())) if x is None else out_arg(x, xw)
    _call("wx_iswpt1d", xw.suffix, xw.ptr, out.ptr, sig[0], L, _smv(sm), 1 if N is None else N, qp, F, xw.stream())
    return out.arr if x is None else x


def iswpt(xw, wt, sm=None):
    """SWT.jl:563-578, 613-712"""
    return _iswpt(xw, wt, sm, False)


def iswpt_(x, xw, wt, sm=None):
    return _iswpt(xw, wt, sm, False, x)


def iswptall(xw, wt, sm=None):
    """swt_all.jl:212-245"""
    return _iswpt(xw, wt, sm, True)


def _iswpd(xw, wt, L_or_tree, sm, batched, x=None):
    xw, sig, m, N = _inv_common(xw, batched)
    L, tree = _split_Ltree(L_or_tree, maxtransformlevels(int(min(sig))))
    if tree is None:
        if not L <= maxtransformlevels(int(min(sig))):
            raise ArgumentError("Too many transform levels.")                     # SWT.jl:1041-1047
        if not L >= 1:
            raise ArgumentError("L must be >= 1")
    if x is not None:
        assert tuple(x.shape) == sig + ((N,) if batched else ())      # SWT.jl:1039
    out = xw.new(sig + ((N,) if batched else ())) if x is None else out_arg(x, xw)
    q, qp, F = qmf_arg(wt)
    tk, tp, nt = tree_arg(tree)
    if len(sig) == 1:
        _call("wx_iswpd1d", xw.suffix, xw.ptr, out.ptr, sig[0], m, L, tp, nt, _smv(sm), 1 if N is None else N, qp, F,
              xw.stream())
    else:
        _call("wx_iswpd2d", xw.suffix, xw.ptr, out.ptr, sig[0], sig[1], m, L, tp, nt, _smv(sm),
              1 if N is None else N, qp, F, xw.stream())
    return out.arr if x is None else x


def iswpd(xw, wt, L_or_tree=None, sm=None):
    """SWT.jl:939-971, 1035-1160"""
    return _iswpd(xw, wt, L_or_tree, sm, False)


def iswpd_(x, xw, wt, L_or_tree=None, sm=None):
    return _iswpd(xw, wt, L_or_tree, sm, False, x)


def iswpdall(xw, wt, L_or_tree=None, sm=None):
    """swt_all.jl:343-392"""
    return _iswpd(xw, wt, L_or_tree, sm, True)
