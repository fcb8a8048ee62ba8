"""Autocorrelation wavelet transforms: host-side mirror of the reference's `ACWT` module
(src/mod/ACWT.jl, src/mod/acwt/*.jl) for 1-D Float64 signals.  The inverses take no filter
(`iacdwt(xw)` / `iacdwt(xw, wt)` are both accepted, like the reference)."""
import numpy as np

from ._arrays import Arg, out_arg, tree_arg
from .dwt import _call, _split_Ltree
from .filters import ArgumentError, OrthoFilter
from .swt import _fwd, _inv_common
from .util import isdyadic, maxtransformlevels, ndyadicscales


def autocorr(f):
    """acwt_utils.jl:7-18"""
    H = np.asarray(f.qmf, dtype=np.float64)
    l = H.size
    out = np.zeros(l - 1)
    for k in range(1, l):
        acc = 0.0
        for i in range(l - k):
            acc += H[i] * H[i + k]
        out[k - 1] = acc * 2
    return out


def pfilter(f):
    """acwt_utils.jl:27-33"""
    a = autocorr(f)
    c1 = 1 / np.sqrt(2.0)
    b = (c1 / 2) * a
    return np.concatenate([b[::-1], [c1], b])


def qfilter(f):
    """acwt_utils.jl:42-48"""
    a = autocorr(f)
    c1 = 1 / np.sqrt(2.0)
    b = -(c1 / 2) * a
    return np.concatenate([b[::-1], [c1], b])


def make_acqmfpair(f):
    return pfilter(f), qfilter(f)


def make_acreverseqmfpair(f):
    """acwt_utils.jl:69-72"""
    p, q = make_acqmfpair(f)
    return p[::-1].copy(), q[::-1].copy()


def _f64(x):
    a = Arg(x)
    if a.dtype != np.float64:
        raise TypeError("ACWT is Float64-only in the reference (acwt_one_level.jl:101-106)")
    return a


def acdwt(x, wt, L=None):
    """ACWT.jl:60-74"""
    return _fwd("wx_acdwt1d", "dwt", _f64(x).arr, wt, L, False)


def acdwt_(xw, x, wt, L=None):
    return _fwd("wx_acdwt1d", "dwt", _f64(x).arr, wt, L, False, xw)


def acdwtall(x, wt, L=None):
    """acwt_all.jl:33"""
    return _fwd("wx_acdwt1d", "dwt", _f64(x).arr, wt, L, True)


def acwpt(x, wt, L=None):
    """ACWT.jl:379-395"""
    return _fwd("wx_acwpt1d", "wpt", _f64(x).arr, wt, L, False)


def acwpt_(xw, x, wt, L=None):
    return _fwd("wx_acwpt1d", "wpt", _f64(x).arr, wt, L, False, xw)


def acwptall(x, wt, L=None):
    """acwt_all.jl:136"""
    return _fwd("wx_acwpt1d", "wpt", _f64(x).arr, wt, L, True)


def acwpd(x, wt, L=None):
    """ACWT.jl:683-699"""
    return _fwd("wx_acwpd1d", "wpd", _f64(x).arr, wt, L, False)


def acwpd_(xw, x, wt, L=None):
    return _fwd("wx_acwpd1d", "wpd", _f64(x).arr, wt, L, False, xw)


def acwpdall(x, wt, L=None):
    """acwt_all.jl:239"""
    return _fwd("wx_acwpd1d", "wpd", _f64(x).arr, wt, L, True)


def _iacdwt(xw, batched, x=None):
    xw, sig, k, N = _inv_common(_f64(xw).arr, batched)
    out = xw.new(sig + ((N,) if batched else ())) if x is None else out_arg(x, xw)
    if len(sig) == 1:
        _call("wx_iacdwt1d", "_f64", xw.ptr, out.ptr, sig[0], k - 1, 1 if N is None else N, xw.stream())
    else:
        _call("wx_iacdwt2d", "_f64", xw.ptr, out.ptr, sig[0], sig[1], (k - 1) // 3, 1 if N is None else N, xw.stream())
    return out.arr if x is None else x


def iacdwt(xw, wt=None):
    """ACWT.jl:257-265, 287-304"""
    return _iacdwt(xw, False)


def iacdwt_(x, xw, wt=None):
    return _iacdwt(xw, False, x)


def iacdwtall(xw, wt=None):
    return _iacdwt(xw, True)


def _iacwpt(xw, batched, x=None):
    xw, sig, m, N = _inv_common(_f64(xw).arr, batched)
    if len(sig) == 2:
        L = 0
        while (1 << (2 * (L + 1))) <= m:
            L += 1
        if (1 << (2 * L)) != m:
            raise ArgumentError("Size of dimension 3 is not a power of 4.")       # ACWT.jl:617
        assert L <= maxtransformlevels(int(min(sig)))                              # ACWT.jl:620
        out = xw.new(sig + ((N,) if batched else ())) if x is None else out_arg(x, xw)
        _call("wx_iacwpt2d", "_f64", xw.ptr, out.ptr, sig[0], sig[1], L, 1 if N is None else N, xw.stream())
        return out.arr if x is None else x
    if not isdyadic(m):
        raise ArgumentError("Number of columns of xw is not dyadic.")             # ACWT.jl:586
    L = ndyadicscales(m)
    if not L <= maxtransformlevels(sig[0]):
        raise ArgumentError("Number of nodes in `xw` is more than possible number of nodes at any depth "
                            "for signal of length `n`")
    out = xw.new(sig + ((N,) if batched else ())) if x is None else out_arg(x, xw)
    _call("wx_iacwpt1d", "_f64", xw.ptr, out.ptr, sig[0], L, 1 if N is None else N, xw.stream())
    return out.arr if x is None else x


def iacwpt(xw, wt=None):
    """ACWT.jl:551-559, 581-610"""
    return _iacwpt(xw, False)


def iacwpt_(x, xw, wt=None):
    return _iacwpt(xw, False, x)


def iacwptall(xw, wt=None):
    return _iacwpt(xw, True)


def _iacwpd_args(args):
    """(xw[, wt][, L | tree]) like the reference's iacwpd method table (ACWT.jl:844-915)"""
    rest = [a for a in args if not (a is None or isinstance(a, OrthoFilter))]
    return rest[0] if rest else None


def _iacwpd(xw, L_or_tree, batched, x=None):
    xw, sig, m, N = _inv_common(_f64(xw).arr, batched)
    L, tree = _split_Ltree(L_or_tree, maxtransformlevels(int(sig[0])))
    if tree is None:
        if not L <= maxtransformlevels(int(min(sig))):
            raise ArgumentError("Too many transform levels (length(x) < 2^L)")     # ACWT.jl:853-855
        if not L >= 1:
            raise ArgumentError("L must be >= 1")
    if x is not None:
        assert tuple(x.shape)[:1] == sig[:1]                                       # ACWT.jl:946
        assert tuple(x.shape) == sig + ((N,) if batched else ())
    out = xw.new(sig + ((N,) if batched else ())) if x is None else out_arg(x, xw)
    tk, tp, nt = tree_arg(tree)
    if len(sig) == 1:
        _call("wx_iacwpd1d", "_f64", xw.ptr, out.ptr, sig[0], m, L, tp, nt, 1 if N is None else N, xw.stream())
    else:
        _call("wx_iacwpd2d", "_f64", xw.ptr, out.ptr, sig[0], sig[1], m, L, tp, nt, 1 if N is None else N, xw.stream())
    return out.arr if x is None else x


def iacwpd(xw, *args):
    """iacwpd(xw[, wt][, L | tree]) ACWT.jl:844-875"""
    return _iacwpd(xw, _iacwpd_args(args), False)


def iacwpd_(x, xw, *args):
    """iacwpd!(x, xw[, wt][, L | tree]) ACWT.jl:917-968"""
    return _iacwpd(xw, _iacwpd_args(args), False, x)


def iacwpdall(xw, *args):
    """acwt_all.jl:300-333"""
    return _iacwpd(xw, _iacwpd_args(args), True)
