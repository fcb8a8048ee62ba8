"""ctypes front-end of the CPU oracle (oracle/libwx_oracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Arrays use the Julia shapes of the reference (column-major); inputs are converted with
np.asfortranarray, outputs are Fortran-ordered numpy arrays.  `T` is taken from the input dtype
(float64 / float32); filters are always float64, exactly like the reference.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# WX_ORACLE_SAN=1: the AddressSanitizer / UBSan build of the checker (tools/oracle_san.sh; python must run with the sanitizer runtime
# preloaded, which that script arranges)
_SAN = os.environ.get("WX_ORACLE_SAN", "") == "1"
_SO = os.path.join(_HERE, "libwx_oracle_san.so" if _SAN else "libwx_oracle.so")


def build(force=False):
    src_newer = (not os.path.exists(_SO)) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
        for f in ("wx_oracle.c", "wx_oracle_impl.h"))
    if force or src_newer:
        subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(_SO)], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


_P, _I, _L, _D = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double


def _suf(dt):
    return {np.dtype(np.float64): "_f64", np.dtype(np.float32): "_f32"}[np.dtype(dt)]


def _f(a, dtype=None):
    a = np.asarray(a)
    if dtype is None:
        dtype = a.dtype if a.dtype in (np.float32, np.float64) else np.float64
    return np.asfortranarray(a, dtype=dtype)


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def _q(qmf):
    q = np.ascontiguousarray(np.asarray(qmf, dtype=np.float64))
    return q, _p(q), int(q.size)


def _tree(tree):
    t = np.ascontiguousarray(np.asarray(tree).astype(np.uint8))
    return t, _p(t), int(t.size)


def _call(name, dt, *args, restype=_I):
    fn = getattr(lib(), name + _suf(dt))
    fn.restype = restype
    conv = []
    for a in args:
        if isinstance(a, (int, np.integer)):
            conv.append(_L(int(a)))
        elif isinstance(a, float):
            conv.append(_D(a))
        else:
            conv.append(a)
    return fn(*conv)


class OracleAssertion(AssertionError):
    pass


def _chk(rc):
    if rc == -1:
        raise OracleAssertion("reference @assert would fail")
    if rc == -2:
        raise ValueError("reference would throw ArgumentError")
    if rc == -3:
        raise IndexError("reference would throw BoundsError")
    if rc != 0 and rc is not None:
        raise RuntimeError("oracle status %r" % rc)


# ---- filters --------------------------------------------------------------------------------
def makereverseqmfpair(qmf):
    q, qp, F = _q(qmf)
    g, h = np.empty(F), np.empty(F)
    lib().wxo_makereverseqmfpair(qp, _I(F), _p(g), _p(h))
    return g, h


def make_acreverseqmfpair(qmf):
    q, qp, F = _q(qmf)
    P, Q = np.empty(2 * F - 1), np.empty(2 * F - 1)
    lib().wxo_make_acreverseqmfpair(qp, _I(F), _p(P), _p(Q))
    return P, Q


# ---- integer helpers ------------------------------------------------------------------------
def maxtransformlevels(n):
    return lib().wxo_maxtransformlevels(_L(n))


def getdepth(i, kind):
    return lib().wxo_getdepth_binary(_L(i)) if kind == "binary" else lib().wxo_getdepth_quad(_L(i))


def gettreelength(*sz):
    L = lib()
    L.wxo_gettreelength1d.restype = _L
    L.wxo_gettreelength2d.restype = _L
    return L.wxo_gettreelength1d(_L(sz[0])) if len(sz) == 1 else L.wxo_gettreelength2d(_L(sz[0]), _L(sz[1]))


def maketree1d(n, L, s="full"):
    t = np.zeros(max(n - 1, 0), dtype=np.uint8)
    _chk(lib().wxo_maketree1d(_p(t), _L(n), _I(L), _I(0 if s == "full" else 1)))
    return t.astype(bool)


def maketree2d(n, m, L, s="full"):
    t = np.zeros(gettreelength(n, m), dtype=np.uint8)
    _chk(lib().wxo_maketree2d(_p(t), _L(n), _L(m), _I(L), _I(0 if s == "full" else 1)))
    return t.astype(bool)


def isvalidtree1d(n, tree):
    t, tp, nt = _tree(tree)
    return bool(lib().wxo_isvalidtree1d(_L(n), tp, _L(nt)))


def isvalidtree2d(n, m, tree):
    t, tp, nt = _tree(tree)
    return bool(lib().wxo_isvalidtree2d(_L(n), _L(m), tp, _L(nt)))


def getleaf(tree, kind):
    t, tp, nt = _tree(tree)
    if kind == "binary":
        res = np.zeros(2 * nt + 1, dtype=np.uint8)
        _chk(lib().wxo_getleaf_binary(_p(res), tp, _L(nt)))
    else:
        res = np.zeros(4 * nt + 1, dtype=np.uint8)
        _chk(lib().wxo_getleaf_quad(_p(res), tp, _L(nt)))
    return res.astype(bool)


def getrowrange(n, idx):
    lo, hi = _L(), _L()
    _chk(lib().wxo_getrowrange(_L(n), _L(idx), ctypes.byref(lo), ctypes.byref(hi)))
    return lo.value, hi.value


def getcolrange(n, idx):
    lo, hi = _L(), _L()
    _chk(lib().wxo_getcolrange(_L(n), _L(idx), ctypes.byref(lo), ctypes.byref(hi)))
    return lo.value, hi.value


def main2depthshift(sm, L):
    sd = np.zeros(L + 1, dtype=np.int64)
    _chk(lib().wxo_main2depthshift(_L(sm), _I(L), _p(sd)))
    return sd.tolist()


def coarsestscalingrange(n, tree, redundant=False):
    t, tp, nt = _tree(tree)
    fn = lib().wxo_coarsestscalingrange
    fn.restype = _L
    r = fn(_L(n), tp, _L(nt), _I(int(redundant)))
    if r < 0:
        raise OracleAssertion()
    return r


def finestdetailrange(n, tree, redundant=False):
    t, tp, nt = _tree(tree)
    fn = lib().wxo_finestdetailrange
    fn.restype = _L
    r = fn(_L(n), tp, _L(nt), _I(int(redundant)))
    if r < 0:
        raise OracleAssertion()
    return r


# ---- single steps ---------------------------------------------------------------------------
def dwt_step(v, h, g):
    v = _f(v); n = v.shape[0]
    h = np.ascontiguousarray(h, dtype=np.float64); g = np.ascontiguousarray(g, dtype=np.float64)
    if v.ndim == 1:
        w1, w2 = np.empty(n // 2, v.dtype), np.empty(n // 2, v.dtype)
        _call("wxo_dwt_step", v.dtype, _p(w1), _p(w2), _p(v), n, _p(h), _p(g), _I(h.size), restype=None)
        return w1, w2
    n2, m2 = v.shape[0] // 2, v.shape[1] // 2
    ws = [np.empty((n2, m2), v.dtype, order="F") for _ in range(4)]
    _call("wxo_dwt_step2", v.dtype, *[_p(w) for w in ws], _p(v), n2, m2, _p(h), _p(g), _I(h.size), restype=None)
    return tuple(ws)


def idwt_step(*args):
    h = np.ascontiguousarray(args[-2], dtype=np.float64); g = np.ascontiguousarray(args[-1], dtype=np.float64)
    ws = [_f(w) for w in args[:-2]]
    if len(ws) == 2:
        n = 2 * ws[0].shape[0]
        v = np.empty(n, ws[0].dtype)
        _call("wxo_idwt_step", v.dtype, _p(v), _p(ws[0]), _p(ws[1]), n, _p(h), _p(g), _I(h.size), restype=None)
        return v
    n2, m2 = ws[0].shape
    v = np.empty((2 * n2, 2 * m2), ws[0].dtype, order="F")
    _call("wxo_idwt_step2", v.dtype, _p(v), *[_p(w) for w in ws], n2, m2, _p(h), _p(g), _I(h.size), restype=None)
    return v


def sdwt_step(v, d, h, g):
    v = _f(v)
    h = np.ascontiguousarray(h, dtype=np.float64); g = np.ascontiguousarray(g, dtype=np.float64)
    if v.ndim == 1:
        n = v.shape[0]
        w1, w2 = np.zeros(n, v.dtype), np.zeros(n, v.dtype)
        _call("wxo_sdwt_step", v.dtype, _p(w1), _p(w2), _p(v), n, _I(d), _p(h), _p(g), _I(h.size), restype=None)
        return w1, w2
    n, m = v.shape
    ws = [np.empty((n, m), v.dtype, order="F") for _ in range(4)]
    _call("wxo_sdwt_step2", v.dtype, *[_p(w) for w in ws], _p(v), n, m, _I(d), _p(h), _p(g), _I(h.size), restype=None)
    return tuple(ws)


def isdwt_step(*args):
    """isdwt_step(w1, w2, d, h, g) / (w1, w2, d, sv, sw, h, g) and the 4-child 2-D forms."""
    # split by position: leading arrays are children, trailing two are filters
    h = np.ascontiguousarray(args[-2], dtype=np.float64); g = np.ascontiguousarray(args[-1], dtype=np.float64)
    nchild = 2 if np.asarray(args[0]).ndim == 1 else 4
    ws = [_f(w) for w in args[:nchild]]
    rest = args[nchild:-2]
    d = int(rest[0])
    shift = len(rest) == 3
    sv, sw = (int(rest[1]), int(rest[2])) if shift else (0, 0)
    if nchild == 2:
        n = ws[0].shape[0]
        v = np.zeros(n, ws[0].dtype)
        if shift:
            _chk(_call("wxo_isdwt_step_shift", v.dtype, _p(v), _p(ws[0]), _p(ws[1]), n, _I(d), sv, sw, _p(h), _p(g), _I(h.size)))
        else:
            _call("wxo_isdwt_step_avg", v.dtype, _p(v), _p(ws[0]), _p(ws[1]), n, _I(d), _p(h), _p(g), _I(h.size), restype=None)
        return v
    n, m = ws[0].shape
    v = np.zeros((n, m), ws[0].dtype, order="F")
    _chk(_call("wxo_isdwt_step2", v.dtype, _p(v), *[_p(w) for w in ws], n, m, _I(d), _I(int(shift)), sv, sw,
               _p(h), _p(g), _I(h.size)))
    return v


def acdwt_step(v, d, h, g):
    v = _f(v)
    h = np.ascontiguousarray(h, dtype=np.float64); g = np.ascontiguousarray(g, dtype=np.float64)
    if v.ndim == 1:
        n = v.shape[0]
        w1, w2 = np.zeros(n, v.dtype), np.zeros(n, v.dtype)
        _call("wxo_acdwt_step", v.dtype, _p(w1), _p(w2), _p(v), n, _I(d), _p(h), _p(g), _I(h.size), restype=None)
        return w1, w2
    n, m = v.shape
    ws = [np.empty((n, m), v.dtype, order="F") for _ in range(4)]
    _call("wxo_acdwt_step2", v.dtype, *[_p(w) for w in ws], _p(v), n, m, _I(d), _p(h), _p(g), _I(h.size), restype=None)
    return tuple(ws)


def iacdwt_step(*ws):
    ws = [_f(w) for w in ws]
    if len(ws) == 2:
        v = np.empty_like(ws[0])
        _call("wxo_iacdwt_step", v.dtype, _p(v), _p(ws[0]), _p(ws[1]), ws[0].shape[0], restype=None)
        return v
    n, m = ws[0].shape
    v = np.empty((n, m), ws[0].dtype, order="F")
    _call("wxo_iacdwt_step2", v.dtype, _p(v), *[_p(w) for w in ws], n, m, restype=None)
    return v


# ---- decimated packets ----------------------------------------------------------------------
def wpd(x, qmf, L=None):
    x = _f(x); q, qp, F = _q(qmf)
    if x.ndim == 1:
        n = x.shape[0]
        L = maxtransformlevels(n) if L is None else L
        y = np.empty((n, L + 1), x.dtype, order="F")
        _call("wxo_wpd1d", x.dtype, _p(y), _p(x), n, _I(L), qp, _I(F), restype=None)
        return y
    m, n = x.shape
    L = maxtransformlevels(min(m, n)) if L is None else L
    y = np.empty((m, n, L + 1), x.dtype, order="F")
    _call("wxo_wpd2d", x.dtype, _p(y), _p(x), m, n, _I(L), qp, _I(F), restype=None)
    return y


def _tree_for(shape, L_or_tree):
    if isinstance(L_or_tree, np.ndarray):
        return L_or_tree
    if len(shape) == 1:
        L = maxtransformlevels(shape[0]) if L_or_tree is None else L_or_tree
        return maketree1d(shape[0], L, "full")
    L = maxtransformlevels(min(shape)) if L_or_tree is None else L_or_tree
    return maketree2d(shape[0], shape[1], L, "full")


def wpt(x, qmf, L_or_tree=None):
    x = _f(x); q, qp, F = _q(qmf)
    t, tp, nt = _tree(_tree_for(x.shape, L_or_tree))
    y = np.empty(x.shape, x.dtype, order="F")
    if x.ndim == 1:
        _chk(_call("wxo_wpt1d_tree", x.dtype, _p(y), _p(x), x.shape[0], tp, nt, qp, _I(F)))
    else:
        _chk(_call("wxo_wpt2d_tree", x.dtype, _p(y), _p(x), x.shape[0], x.shape[1], tp, nt, qp, _I(F)))
    return y


def iwpt(xw, qmf, L_or_tree=None):
    xw = _f(xw); q, qp, F = _q(qmf)
    t, tp, nt = _tree(_tree_for(xw.shape, L_or_tree))
    y = np.empty(xw.shape, xw.dtype, order="F")
    if xw.ndim == 1:
        _chk(_call("wxo_iwpt1d_tree", xw.dtype, _p(y), _p(xw), xw.shape[0], tp, nt, qp, _I(F)))
    else:
        _chk(_call("wxo_iwpt2d_tree", xw.dtype, _p(y), _p(xw), xw.shape[0], xw.shape[1], tp, nt, qp, _I(F)))
    return y


def iwpd(xw, qmf, L_or_tree=None):
    xw = _f(xw); q, qp, F = _q(qmf)
    sig = xw.shape[:-1]
    k = xw.shape[-1]
    t, tp, nt = _tree(_tree_for(sig, L_or_tree))
    y = np.empty(sig, xw.dtype, order="F")
    if len(sig) == 1:
        _chk(_call("wxo_iwpd1d_tree", xw.dtype, _p(y), _p(xw), sig[0], _I(k), tp, nt, qp, _I(F)))
    else:
        _chk(_call("wxo_iwpd2d_tree", xw.dtype, _p(y), _p(xw), sig[0], sig[1], _I(k), tp, nt, qp, _I(F)))
    return y


def getbasiscoef(Xw, tree):
    Xw = _f(Xw)
    t, tp, nt = _tree(tree)
    n, k = Xw.shape
    out = np.empty(n, Xw.dtype)
    _chk(_call("wxo_getbasiscoef1d", Xw.dtype, _p(out), _p(Xw), n, _I(k), tp, nt))
    return out


def _all(fn, x, sig_ndim, *args):
    """`*all` drivers: loop over the last dimension (dwt_all.jl / swt_all.jl / acwt_all.jl)."""
    x = _f(x)
    outs = [fn(np.asfortranarray(x[..., i]), *args) for i in range(x.shape[-1])]
    return np.asfortranarray(np.stack(outs, axis=-1))


def wpdall(x, qmf, L=None):
    return _all(wpd, x, None, qmf, L)


def iwpdall(xw, qmf, L_or_tree=None):
    return _all(iwpd, xw, None, qmf, L_or_tree)


def wptall(x, qmf, L_or_tree=None):
    return _all(wpt, x, None, qmf, L_or_tree)


def iwptall(x, qmf, L_or_tree=None):
    return _all(iwpt, x, None, qmf, L_or_tree)


# ---- 3-D dwt of a cube (dwt_all.jl:8-9; Wavelets.jl's 3-D transform is not vendored: the separable pyramid out of the
# ---- same one-level step as 1-D, parity unpinned) -------------------------------------------------------------
def dwt3d(x, qmf, L=None):
    x = _f(x)
    n = x.shape[0]
    assert x.ndim == 3 and x.shape == (n, n, n)
    L = maxtransformlevels(n) if L is None else L
    g, h = makereverseqmfpair(qmf)
    y = np.array(x, copy=True, order="F")
    for l in range(L):
        ns = n >> l
        for axis in range(3):
            sub = np.moveaxis(y[:ns, :ns, :ns], axis, 0)
            for i in range(ns):
                for j in range(ns):
                    a, d = dwt_step(np.ascontiguousarray(sub[:, i, j]), h, g)
                    sub[:ns // 2, i, j] = a
                    sub[ns // 2:, i, j] = d
    return y


def idwt3d(xw, qmf, L=None):
    xw = _f(xw)
    n = xw.shape[0]
    assert xw.ndim == 3 and xw.shape == (n, n, n)
    L = maxtransformlevels(n) if L is None else L
    g, h = makereverseqmfpair(qmf)
    y = np.array(xw, copy=True, order="F")
    for l in range(L - 1, -1, -1):
        ns = n >> l
        for axis in (2, 1, 0):
            sub = np.moveaxis(y[:ns, :ns, :ns], axis, 0)
            for i in range(ns):
                for j in range(ns):
                    sub[:, i, j] = idwt_step(np.ascontiguousarray(sub[:ns // 2, i, j]),
                                             np.ascontiguousarray(sub[ns // 2:, i, j]), h, g)
    return y


# ---- stationary -----------------------------------------------------------------------------
def sdwt(x, qmf, L=None):
    x = _f(x); q, qp, F = _q(qmf); n = x.shape[0]
    L = maxtransformlevels(n) if L is None else L
    xw = np.empty((n, L + 1), x.dtype, order="F")
    _chk(_call("wxo_sdwt1d", x.dtype, _p(xw), _p(x), n, _I(L), qp, _I(F)))
    return xw


def isdwt(xw, qmf, sm=None):
    xw = _f(xw); q, qp, F = _q(qmf); n, k = xw.shape
    x = np.empty(n, xw.dtype)
    _chk(_call("wxo_isdwt1d", xw.dtype, _p(x), _p(xw), n, _I(k - 1), -1 if sm is None else sm, qp, _I(F)))
    return x


def swpt(x, qmf, L=None):
    x = _f(x); q, qp, F = _q(qmf); n = x.shape[0]
    L = maxtransformlevels(n) if L is None else L
    xw = np.empty((n, 1 << L), x.dtype, order="F")
    _chk(_call("wxo_swpt1d", x.dtype, _p(xw), _p(x), n, _I(L), qp, _I(F)))
    return xw


def iswpt(xw, qmf, sm=None):
    xw = _f(xw); q, qp, F = _q(qmf); n, m = xw.shape
    x = np.empty(n, xw.dtype)
    _chk(_call("wxo_iswpt1d", xw.dtype, _p(x), _p(xw), n, m, -1 if sm is None else sm, qp, _I(F)))
    return x


def swpd(x, qmf, L=None):
    x = _f(x); q, qp, F = _q(qmf); n = x.shape[0]
    L = maxtransformlevels(n) if L is None else L
    xw = np.empty((n, (1 << (L + 1)) - 1), x.dtype, order="F")
    _chk(_call("wxo_swpd1d", x.dtype, _p(xw), _p(x), n, _I(L), qp, _I(F)))
    return xw


def iswpd(xw, qmf, L_or_tree=None, sm=None):
    xw = _f(xw); q, qp, F = _q(qmf); n, m = xw.shape
    t, tp, nt = _tree(_tree_for((n,), L_or_tree))
    x = np.empty(n, xw.dtype)
    _chk(_call("wxo_iswpd1d", xw.dtype, _p(x), _p(xw), n, m, tp, nt, -1 if sm is None else sm, qp, _I(F)))
    return x


# ---- autocorrelation ------------------------------------------------------------------------
def acdwt(x, qmf, L=None):
    x = _f(x); q, qp, F = _q(qmf); n = x.shape[0]
    L = maxtransformlevels(n) if L is None else L
    xw = np.empty((n, L + 1), x.dtype, order="F")
    _chk(_call("wxo_acdwt1d", x.dtype, _p(xw), _p(x), n, _I(L), qp, _I(F)))
    return xw


def iacdwt(xw):
    xw = _f(xw); n, k = xw.shape
    x = np.empty(n, xw.dtype)
    _call("wxo_iacdwt1d", xw.dtype, _p(x), _p(xw), n, _I(k - 1), restype=None)
    return x


def acwpt(x, qmf, L=None):
    x = _f(x); q, qp, F = _q(qmf); n = x.shape[0]
    L = maxtransformlevels(n) if L is None else L
    xw = np.empty((n, 1 << L), x.dtype, order="F")
    _chk(_call("wxo_acwpt1d", x.dtype, _p(xw), _p(x), n, _I(L), qp, _I(F)))
    return xw


def iacwpt(xw):
    xw = _f(xw); n, m = xw.shape
    x = np.empty(n, xw.dtype)
    _chk(_call("wxo_iacwpt1d", xw.dtype, _p(x), _p(xw), n, m))
    return x


def acwpd(x, qmf, L=None):
    x = _f(x); q, qp, F = _q(qmf); n = x.shape[0]
    L = maxtransformlevels(n) if L is None else L
    xw = np.empty((n, (1 << (L + 1)) - 1), x.dtype, order="F")
    _chk(_call("wxo_acwpd1d", x.dtype, _p(xw), _p(x), n, _I(L), qp, _I(F)))
    return xw


def iacwpd(xw, L_or_tree=None):
    xw = _f(xw); n, m = xw.shape
    t, tp, nt = _tree(_tree_for((n,), L_or_tree))
    x = np.empty(n, xw.dtype)
    _chk(_call("wxo_iacwpd1d", xw.dtype, _p(x), _p(xw), n, m, tp, nt))
    return x


# ---- joint best basis -----------------------------------------------------------------------
def tree_costs_jbb(X, redundant=False, cost="loglp", p=None):
    """tree_costs(X::Array{T,3}, JBB(cost, redundant)); X is (n, L, N)."""
    X = _f(X); n, L, N = X.shape
    p = (2.0 if cost == "loglp" else 1.0) if p is None else float(p)
    ncost = L if redundant else (1 << L) - 1
    costs = np.empty(ncost, X.dtype)
    _chk(_call("wxo_tree_costs_jbb", X.dtype, _p(costs), _p(X), n, L, N, _I(int(redundant)),
               _I(0 if cost == "loglp" else 1), _D(p)))
    return costs


def tree_costs_jbb_sums(EX, EX2, N, redundant=False, cost="loglp", p=None):
    """tree_costs(X, JBB) after its two sums over the signal axis (bestbasis_tree.jl:155-179): EX = sum(X, dims=3),
    EX2 = sum(X.^2, dims=3), both (n, L), of N signals."""
    EX = np.array(_f(EX), copy=True, order="F"); EX2 = np.array(_f(EX2, EX.dtype), copy=True, order="F")
    n, L = EX.shape
    p = (2.0 if cost == "loglp" else 1.0) if p is None else float(p)
    costs = np.empty(L if redundant else (1 << L) - 1, EX.dtype)
    _chk(_call("wxo_tree_costs_jbb_sums", EX.dtype, _p(costs), _p(EX), _p(EX2), n, L, N, _I(int(redundant)),
               _I(0 if cost == "loglp" else 1), _D(p)))
    return costs


def acwpd_jbb_sums(x, qmf, L=None):
    """sum(X, dims=3), sum(X.^2, dims=3) of X = acwpdall(x) (acwt_all.jl:239-259, bestbasis_tree.jl:153-154) without
    holding X: tables of a window of signals at a time, added in signal order with the same roundings as the sequential
    sums (x*x rounded on its own, then the add); threads split the coefficient axis only."""
    x = _f(x, np.float64); n, N = x.shape
    L = maxtransformlevels(n) if L is None else L
    q, qp, F = _q(qmf)
    EX = np.empty((n, (1 << (L + 1)) - 1), np.float64, order="F"); EX2 = np.empty_like(EX)
    _chk(_call("wxo_acwpd_jbb_sums", np.float64, _p(EX), _p(EX2), _p(x), n, _I(L), N, qp, _I(F)))
    return EX, EX2


def bestbasis_treeselection(costs, n, kind="min"):
    costs = np.array(costs, copy=True)
    if costs.dtype not in (np.float32, np.float64):
        costs = costs.astype(np.float64)
    tree = np.zeros(n - 1, dtype=np.uint8)
    _chk(_call("wxo_bestbasis_treeselection", costs.dtype, _p(tree), _p(costs), costs.size, n,
               _I(0 if kind == "min" else 1)))
    return tree.astype(bool)


def bestbasistree_jbb(X, redundant=False, cost="loglp", p=None):
    """bestbasistree(X, JBB(...)) BestBasis.jl:194-201 for 1-D signals."""
    X = _f(X)
    return bestbasis_treeselection(tree_costs_jbb(X, redundant, cost, p), X.shape[0])


# ---- 2-D redundant families, 2-D JBB, 2-D getbasiscoef --------------------------------------------
def _ncols2d(kind, L):
    return {"dwt": 3 * L + 1, "wpt": 1 << (2 * L), "wpd": ((1 << (2 * (L + 1))) - 1) // 3}[kind]


def red2d_fwd(kind, x, qmf, L=None, ac=False):
    """sdwt/swpt/swpd (ac=False) or acdwt/acwpt/acwpd (ac=True) of an (n, m) image"""
    x = _f(x); q, qp, F = _q(qmf); n, m = x.shape
    L = maxtransformlevels(min(n, m)) if L is None else L
    xw = np.empty((n, m, _ncols2d(kind, L)), x.dtype, order="F")
    _chk(_call("wxo_red_%s2d" % kind, x.dtype, _p(xw), _p(x), n, m, _I(L), _I(int(ac)), qp, _I(F)))
    return xw


def red2d_inv(kind, xw, qmf=None, L_or_tree=None, sm=None, ac=False):
    xw = _f(xw); n, m, k = xw.shape
    q, qp, F = _q(qmf if qmf is not None else [1.0, 1.0])
    x = np.empty((n, m), xw.dtype, order="F")
    smv = -1 if sm is None else sm
    if kind == "dwt":
        _chk(_call("wxo_ired_dwt2d", xw.dtype, _p(x), _p(xw), n, m, _I(k), _I(int(ac)), smv, qp, _I(F)))
    elif kind == "wpt":
        _chk(_call("wxo_ired_wpt2d", xw.dtype, _p(x), _p(xw), n, m, k, _I(int(ac)), smv, qp, _I(F)))
    else:
        t, tp, nt = _tree(_tree_for((n, m), L_or_tree))
        _chk(_call("wxo_ired_wpd2d", xw.dtype, _p(x), _p(xw), n, m, k, tp, nt, _I(int(ac)), smv, qp, _I(F)))
    return x


def getbasiscoef2d(Xw, tree):
    Xw = _f(Xw); n, m, k = Xw.shape
    t, tp, nt = _tree(tree)
    out = np.empty((n, m), Xw.dtype, order="F")
    _chk(_call("wxo_getbasiscoef2d", Xw.dtype, _p(out), _p(Xw), n, m, _I(k), tp, nt))
    return out


def tree_costs_jbb2d(X, redundant=False, cost="loglp", p=None):
    """tree_costs(X::Array{T,4}, JBB(cost, redundant)); X is (n, m, L, N)."""
    X = _f(X); n, m, L, N = X.shape
    p = (2.0 if cost == "loglp" else 1.0) if p is None else float(p)
    ncost = L if redundant else ((1 << (2 * L)) - 1) // 3
    costs = np.empty(ncost, X.dtype)
    _chk(_call("wxo_tree_costs_jbb2d", X.dtype, _p(costs), _p(X), n, m, L, N, _I(int(redundant)),
               _I(0 if cost == "loglp" else 1), _D(p)))
    return costs


def bestbasis_treeselection2d(costs, n, m, kind="min"):
    costs = np.array(costs, copy=True)
    if costs.dtype not in (np.float32, np.float64):
        costs = costs.astype(np.float64)
    tree = np.zeros(gettreelength(n, m), dtype=np.uint8)
    _chk(_call("wxo_bestbasis_treeselection2d", costs.dtype, _p(tree), _p(costs), costs.size, n, m,
               _I(0 if kind == "min" else 1)))
    return tree.astype(bool)


def bestbasistree_jbb2d(X, redundant=False, cost="loglp", p=None):
    X = _f(X)
    return bestbasis_treeselection2d(tree_costs_jbb2d(X, redundant, cost, p), X.shape[0], X.shape[1])


# ---- standard (per-signal) best basis, BB -----------------------------------------------------------
def tree_costs_bb(X, redundant=False, cost="shannon"):
    """tree_costs(X, BB(cost, redundant)) bestbasis_tree.jl:209-258; X is (n, L) or (n, m, L)."""
    X = _f(X)
    kind = _I(0 if cost == "shannon" else 1)
    if X.ndim == 2:
        n, L = X.shape
        costs = np.empty(L if redundant else (1 << L) - 1, X.dtype)
        _chk(_call("wxo_tree_costs_bb", X.dtype, _p(costs), _p(X), n, L, _I(int(redundant)), kind))
    else:
        n, m, L = X.shape
        costs = np.empty(L if redundant else ((1 << (2 * L)) - 1) // 3, X.dtype)
        _chk(_call("wxo_tree_costs_bb2d", X.dtype, _p(costs), _p(X), n, m, L, _I(int(redundant)), kind))
    return costs


def bestbasistree_bb(X, redundant=False, cost="shannon"):
    """bestbasistree(X, BB(...)) BestBasis.jl:203-210"""
    X = _f(X)
    c = tree_costs_bb(X, redundant, cost)
    if X.ndim == 2:
        return bestbasis_treeselection(c, X.shape[0])
    return bestbasis_treeselection2d(c, X.shape[0], X.shape[1])


def bestbasistreeall_bb(X, redundant=False, cost="shannon"):
    """bestbasistreeall(X, BB(...)) BestBasis.jl:253-262: (tree length, k) BitMatrix"""
    X = _f(X)
    return np.stack([bestbasistree_bb(np.asfortranarray(X[..., i]), redundant, cost) for i in range(X.shape[-1])], axis=1)


# ---- denoising core (Denoising.jl:214-232, 483-599; Wavelets.jl Threshold) ---------------------------
TH_KINDS = {"hard": 0, "soft": 1, "semisoft": 2, "stein": 3}


def noisest_range(v):
    """mad!(v)/0.6745 of a 1-D array"""
    v = np.ascontiguousarray(_f(v).ravel())
    fn = getattr(lib(), "wxo_noisest_range" + _suf(v.dtype))
    fn.restype = ctypes.c_double if v.dtype == np.float64 else ctypes.c_float
    return float(fn(_p(v), _L(v.size)))


def threshold(x, th, t):
    x = np.array(_f(x), copy=True, order="F")
    fn = getattr(lib(), "wxo_threshold" + _suf(x.dtype))
    fn.restype = None
    fn(_p(x), _L(x.size), _I(TH_KINDS[th]), (ctypes.c_double if x.dtype == np.float64 else ctypes.c_float)(t))
    return x


def noisest(x, redundant, tree=None):
    """Denoising.jl:214-232"""
    x = _f(x)
    n = x.shape[0]
    if not redundant and tree is None:
        return noisest_range(x[n // 2:])
    if not redundant:
        return noisest_range(x[finestdetailrange(n, tree) - 1:])
    if tree is None:
        return noisest_range(x[:, -1])
    return noisest_range(x[:, finestdetailrange(n, tree, True) - 1])


def denoise(x, inputtype, qmf, L=None, tree=None, th="hard", t=None, estnoise=None, smooth="regular"):
    """Denoising.jl:483-599 for dnt = VisuShrink-like (th, t); estnoise None -> noisest, or a number"""
    x = _f(x)
    n = x.shape[0]
    L = maxtransformlevels(n) if L is None else L
    tree = maketree1d(n, L, "dwt") if tree is None else np.asarray(tree, dtype=bool)
    t = np.sqrt(2 * np.log(n)) if t is None else t
    dwt_tree = maketree1d(n, L, "dwt")
    if inputtype == "sig":
        x = wpt(x, qmf, dwt_tree)                                     # dwt(x, wt, L)
        inputtype = "dwt"
    red = inputtype in ("sdwt", "swpd", "acdwt", "acwpd")
    tr = None if inputtype in ("dwt", "sdwt", "acdwt") else tree
    sigma = noisest(x, red, tr) if estnoise is None else float(estnoise)
    xt = np.array(x, copy=True, order="F")
    if inputtype == "dwt":
        lo = (n >> L) if smooth == "undersmooth" else 0
        xt[lo:] = threshold(xt[lo:], th, sigma * t)
        return xt if qmf is None else iwpt(xt, qmf, dwt_tree)         # idwt(x, wt, L)
    if inputtype == "wpt":
        lo = coarsestscalingrange(n, tree) if smooth == "undersmooth" else 0
        xt[lo:] = threshold(xt[lo:], th, sigma * t)
        return xt if qmf is None else iwpt(xt, qmf, tree)
    if inputtype in ("sdwt", "acdwt"):
        c0 = 1 if smooth == "undersmooth" else 0
        xt[:, c0:] = threshold(xt[:, c0:], th, sigma * t)
        if inputtype == "sdwt":
            return xt if qmf is None else isdwt(xt, qmf)
        return iacdwt(xt)
    leaves = np.flatnonzero(getleaf(tree, "binary"))
    if smooth == "undersmooth":
        leaves = leaves[leaves != coarsestscalingrange(n, tree, True) - 1]
    xt[:, leaves] = threshold(xt[:, leaves], th, sigma * t)
    if inputtype == "swpd":
        return xt if qmf is None else iswpd(xt, qmf, tree)
    return iacwpd(xt, tree)


# ---- threshold selection of SureShrink / RelErrorShrink (Denoising.jl:146-166, 285-381) ----------------
# numpy restatement statement by statement; "parity unpinned": the reference's own tests only check the return type
# (test/denoising.jl:92-103) and there is no Julia here.  np.cumsum adds sequentially like Julia's cumsum; np.sum is
# pairwise like Julia's sum (different block size: last-bit differences in `sum(orth)`).
def _shrink_coefs(coef, redundant, tree):
    coef = _f(coef)
    if not redundant:
        return coef.reshape(-1, order="F")
    if tree is None:
        return coef.reshape(-1, order="F")
    leaves = getleaf(np.asarray(tree, dtype=bool), "binary")
    return coef[:, leaves].reshape(-1, order="F")


def surethreshold(coef, redundant, tree=None):
    """Denoising.jl:146-166"""
    y = _shrink_coefs(coef, redundant, tree)
    a = np.sort(np.abs(y)) ** 2
    b = np.cumsum(a)
    n = y.size
    c = np.arange(n - 1, -1, -1).astype(y.dtype)
    s = b + c * a
    risk = ((n - 2 * np.arange(1, n + 1)) + s) / n
    return float(np.sqrt(a[int(np.argmin(risk))]))


def orth2relerror(orth):
    """Denoising.jl:344-349"""
    orth = np.sort(_f(orth) ** 2)[::-1]
    tot = np.sum(orth)
    return np.abs(tot - np.cumsum(orth)) ** 0.5 / tot ** 0.5


def findelbow(x, y):
    """Denoising.jl:367-381; returns the 0-based index of the elbow"""
    v = np.array([x[-1] - x[0], y[-1] - y[0]])
    v = v / np.linalg.norm(v, 2)
    xy = np.stack([x - x[0], y - y[0]], axis=1)
    H = np.sum(xy ** 2, axis=1) ** 0.5
    A = xy @ v
    O = np.abs(H ** 2 - A ** 2) ** 0.5
    return int(np.argmax(O))


def relerrorthreshold(coef, redundant=False, tree=None, elbows=2):
    """Denoising.jl:285-327 (makeplot = false)"""
    assert elbows >= 1
    c = _shrink_coefs(coef, redundant, tree)
    x = np.sort(np.abs(c))[::-1]
    r = orth2relerror(c)
    x = np.append(x, 0)
    r = np.insert(r, 0, r[0])
    xmax, ymax = np.max(x), np.max(r)
    x = x[::-1] / xmax
    y = r[::-1] / ymax
    ix = findelbow(x, y)
    for _ in range(1, elbows):
        ix = findelbow(x[:ix + 1], y[:ix + 1])
    return float(x[ix] * xmax)


# ---- Local Discriminant Basis, TimeFrequency energy map (LDB.jl:186-251, ldb/ldb_energymap.jl:109-141,
# ---- ldb/ldb_measures.jl:139-183, 302-325, 427-479): plain loops in the reference's order ------------
def _unique(y):
    out = []
    for v in list(np.asarray(y).tolist()):
        if v not in out:
            out.append(v)
    return out


def ldb_energy_map(Xw, y):
    Xw = _f(Xw)
    N = Xw.ndim
    classes = _unique(y)
    sz, L = Xw.shape[:N - 2], Xw.shape[N - 2]
    G = np.empty(tuple(sz) + (L, len(classes)), dtype=Xw.dtype, order="F")
    yl = list(np.asarray(y).tolist())
    for ci, c in enumerate(classes):
        idx = [i for i, v in enumerate(yl) if v == c]
        norm_sum = Xw.dtype.type(0)
        en = np.zeros(tuple(sz) + (L,), dtype=Xw.dtype, order="F")
        for i in idx:
            root = Xw[..., 0, i]
            acc = Xw.dtype.type(0)
            for v in root.ravel(order="F"):
                acc = Xw.dtype.type(acc + v * v)
            nrm = Xw.dtype.type(np.sqrt(acc))
            norm_sum = Xw.dtype.type(norm_sum + nrm * nrm)
            en = (en + Xw[..., i] * Xw[..., i]).astype(Xw.dtype)
        G[..., ci] = en / norm_sum
    return G


def _ldb_pair(p, q, dm, lp=2):
    if dm == "are":
        return 0.0 if (p == 0 or q == 0) else p * np.log(p / q)
    if dm == "sre":
        return _ldb_pair(p, q, "are") + _ldb_pair(q, p, "are")
    if dm == "hellinger":
        return (np.sqrt(p) - np.sqrt(q)) ** 2
    return (p - q) ** lp


def ldb_discriminant_measure(G, dm="are", lp=2):
    G = _f(G)
    nc = G.shape[-1]
    D = np.zeros(G.shape[:-1], dtype=G.dtype, order="F")
    flat = D.reshape(-1, order="F")
    for i in range(nc):
        for j in range(i + 1, nc):
            gi, gj = G[..., i].reshape(-1, order="F"), G[..., j].reshape(-1, order="F")
            for e in range(flat.size):
                flat[e] = G.dtype.type(flat[e] + G.dtype.type(_ldb_pair(gi[e], gj[e], dm, lp)))
    return np.asfortranarray(flat.reshape(D.shape, order="F"))


def ldb_fitdec(Xw, y, dm="are", lp=2, top_k=None, dp="basis"):
    """fitdec! -> dict(G, DM, cost, tree, DP, order) with 1-based `order`"""
    Xw = _f(Xw)
    sz, L = Xw.shape[:-2], Xw.shape[-2]
    nelem = int(np.prod(sz))
    top_k = nelem if top_k is None else top_k
    G = ldb_energy_map(Xw, y)
    DM = ldb_discriminant_measure(G, dm, lp)
    one_d = len(sz) == 1
    ncost = (1 << L) - 1 if one_d else ((1 << (2 * L)) - 1) // 3
    cost = np.empty(ncost, dtype=Xw.dtype)
    for i in range(1, ncost + 1):
        d = getdepth(i, "binary" if one_d else "quad")
        if one_d:
            th, nth = i - (1 << d), sz[0] >> d
            v = list(DM[th * nth:(th + 1) * nth, d])
        else:
            r0, r1 = getrowrange(sz[0], i)
            c0, c1 = getcolrange(sz[1], i)
            v = list(DM[r0 - 1:r1, c0 - 1:c1, d].ravel(order="F"))
        if top_k < len(v):
            v = sorted(v, reverse=True)[:top_k]
        s = Xw.dtype.type(0)
        for x in v:
            s = Xw.dtype.type(s + x)
        cost[i - 1] = s
    tree = bestbasis_treeselection(cost, sz[0], "max") if one_d else bestbasis_treeselection2d(cost, sz[0], sz[1], "max")
    if dp == "basis":
        power = getbasiscoef(DM, tree) if one_d else getbasiscoef2d(DM, tree)
    else:
        Xc = np.asfortranarray(np.stack([(getbasiscoef if one_d else getbasiscoef2d)(np.asfortranarray(Xw[..., i]), tree)
                                         for i in range(Xw.shape[-1])], axis=-1))
        power = ldb_fisher_power(Xc, y)
    order = np.argsort(-power.ravel(order="F"), kind="stable") + 1
    return dict(G=G, DM=DM, cost=cost, tree=tree, DP=power, order=order)


def ldb_fisher_power(coefs, y):
    coefs = _f(coefs)
    classes = _unique(y)
    yl = list(np.asarray(y).tolist())
    sz = coefs.shape[:-1]
    E = np.empty(tuple(sz) + (len(classes),), dtype=coefs.dtype)
    V = np.empty_like(E)
    Ni = np.empty(len(classes), dtype=coefs.dtype)
    for ci, c in enumerate(classes):
        idx = [i for i, v in enumerate(yl) if v == c]
        Ni[ci] = len(idx)
        sub = coefs[..., idx]
        E[..., ci] = sub.mean(axis=-1)
        V[..., ci] = sub.var(axis=-1, ddof=1)
    Ea = E.mean(axis=-1, keepdims=True)
    p = Ni / Ni.sum()
    return np.asfortranarray((((E - Ea * E) ** 2) * p).sum(axis=-1) / (V * p).sum(axis=-1))


# ---- LDB order statistics: robust Fisher power (ldb_measures.jl:481-519) and the earth mover's distance between class
# ---- signatures with equal weights (ldb_energymap.jl:186-238, ldb_measures.jl:254-285, 327-360): plain loops ---------
def _median(v):
    """Statistics.median: middle(a, b) = a/2 + b/2 for an even count"""
    s = np.sort(np.asarray(v))
    n = s.size
    return s[n // 2] if n & 1 else s[n // 2 - 1] / 2 + s[n // 2] / 2


def ldb_robust_fishers(coefs, y):
    coefs = _f(coefs)
    classes = _unique(y)
    yl = list(np.asarray(y).tolist())
    sz = coefs.shape[:-1]
    flat = coefs.reshape(-1, coefs.shape[-1], order="F")
    nc = len(classes)
    med = np.empty((flat.shape[0], nc), dtype=coefs.dtype)
    mad = np.empty((flat.shape[0], nc), dtype=coefs.dtype)
    Ni = np.empty(nc, dtype=coefs.dtype)
    for ci, c in enumerate(classes):
        idx = [i for i, v in enumerate(yl) if v == c]
        Ni[ci] = len(idx)
        for e in range(flat.shape[0]):
            v = flat[e, idx]
            med[e, ci] = _median(v)
            mad[e, ci] = _median(np.abs(v - med[e, ci]))                # mad(x, normalize = false)
    meda = np.array([_median(med[e]) for e in range(flat.shape[0])], dtype=coefs.dtype)
    p = Ni / Ni.sum()
    power = (((med - meda[:, None] * med) ** 2) @ p) / (mad @ p)
    return np.asfortranarray(power.reshape(sz, order="F")), np.argsort(-power, kind="stable") + 1


def emd_pair(p, q, wp, wq):
    """pairwise_discriminant_measure(P, Q, EarthMoverDistance()) ldb_measures.jl:327-360 (scalar weights)"""
    p, q = np.sort(np.asarray(p)), np.sort(np.asarray(q))
    w_p, w_q = np.full(p.size, wp), np.full(q.size, wq)
    r = np.sort(np.concatenate([p, q]))
    emd = 0
    for i in range(r.size - 1):
        sp = np.sum(w_p[p <= r[i]])
        sq = np.sum(w_q[q <= r[i]])
        emd += abs(sp - sq) * (r[i + 1] - r[i])
    return emd / (np.sum(w_p) + np.sum(w_q))


def ldb_emd_measure(Xw, y):
    """discriminant_measure(energy_map(Xw, y, Signatures()), EarthMoverDistance()) ldb_measures.jl:185-201"""
    Xw = _f(Xw)
    classes = _unique(y)
    yl = list(np.asarray(y).tolist())
    groups = [[i for i, v in enumerate(yl) if v == c] for c in classes]
    sz = Xw.shape[:-1]
    flat = Xw.reshape(-1, Xw.shape[-1], order="F")
    D = np.zeros(flat.shape[0], dtype=Xw.dtype)
    for a in range(len(classes)):
        for b in range(a + 1, len(classes)):
            for e in range(flat.shape[0]):
                D[e] += emd_pair(flat[e, groups[a]], flat[e, groups[b]], 1 / len(groups[a]), 1 / len(groups[b]))
    return np.asfortranarray(D.reshape(sz, order="F"))


# ---- average shifted histograms (AverageShiftedHistograms.jl 0.8 / 0.9, NOT in the reference tree: its published
# ---- algorithm restated, parity unpinned) and the two energy maps built on them (ldb_energymap.jl:143-184, 216-232) ----
def ash_density(z, a, delta, length, m):
    """ash(z, rng = range(a, step = delta, length = length), m = m, kernel = Kernels.triangular).density"""
    counts = np.zeros(length, dtype=np.int64)
    dinv = 1.0 / delta
    for yi in np.asarray(z, dtype=np.float64):
        ki = int(np.floor((yi - a) * dinv + 1.5))
        if 1 <= ki <= length:
            counts[ki - 1] += 1
    dens = np.zeros(length)
    for k in range(1, length + 1):
        if counts[k - 1] != 0:
            for i in range(max(1, k - m + 1), min(length, k + m - 1) + 1):
                dens[i - 1] += counts[k - 1] * (1.0 - abs((i - k) / m))
    return dens * (1.0 / (dens.sum() * delta))


def ash_pdf(dens, a, delta, x):
    """pdf(o, x): linear interpolation between the points of rng around x, 0 outside"""
    length = dens.size
    rng = a + np.arange(length) * delta
    i = int(np.searchsorted(rng, x, side="right"))                      # searchsortedlast, 1-based
    if 1 <= i < length:
        return dens[i - 1] + (dens[i] - dens[i - 1]) * (x - rng[i - 1]) / (rng[i] - rng[i - 1])
    return 0.0


def _ash_params(Nx):
    nbins = int(np.ceil((30 * Nx) ** (1 / 5)))
    mbins = int(np.ceil(100 / nbins))
    return nbins, mbins, (nbins + 1) * mbins


def ldb_pdf_energy_map(Xw, y):
    """energy_map(Xw, y, ProbabilityDensity()) ldb_energymap.jl:143-184"""
    Xw = _f(Xw)
    classes = _unique(y)
    yl = list(np.asarray(y).tolist())
    Nx = Xw.shape[-1]
    nbins, mbins, plen = _ash_params(Nx)
    flat = Xw.reshape(-1, Nx, order="F").astype(np.float64)
    G = np.empty((flat.shape[0], plen, len(classes)))
    for ci, c in enumerate(classes):
        idx = [i for i, v in enumerate(yl) if v == c]
        for j in range(flat.shape[0]):
            z = flat[j]
            sd = np.std(z, ddof=1)
            delta = (z.max() - z.min() + sd) / (plen - 1)
            G[j, :, ci] = ash_density(z[idx], z.min() - 0.5 * sd, delta, plen, mbins)
    return np.asfortranarray(G.reshape(Xw.shape[:-1] + (plen, len(classes)), order="F"))


def ldb_signature_weights(Xw, y):
    """the :pdf weights of energy_map(Xw, y, Signatures(:pdf)) ldb_energymap.jl:216-232, as an array shaped like Xw"""
    Xw = _f(Xw)
    classes = _unique(y)
    yl = list(np.asarray(y).tolist())
    Nx = Xw.shape[-1]
    nbins, mbins, plen = _ash_params(Nx)
    flat = Xw.reshape(-1, Nx, order="F").astype(np.float64)
    W = np.empty_like(flat)
    for c in classes:
        idx = [i for i, v in enumerate(yl) if v == c]
        for j in range(flat.shape[0]):
            z = flat[j, idx]
            sd = np.std(z, ddof=1)
            delta = (z.max() - z.min() + sd) / (plen - 1)
            a = z.min() - 0.5 * sd
            dens = ash_density(z, a, delta, plen, mbins)
            for k, i in enumerate(idx):
                W[j, i] = ash_pdf(dens, a, delta, z[k])
    return np.asfortranarray(W.reshape(Xw.shape, order="F"))


def emd_pair_weighted(p, q, w_p, w_q):
    """ldb_measures.jl:327-360 with weight vectors"""
    po, qo = np.argsort(p, kind="stable"), np.argsort(q, kind="stable")
    p, q, w_p, w_q = np.asarray(p)[po], np.asarray(q)[qo], np.asarray(w_p)[po], np.asarray(w_q)[qo]
    r = np.sort(np.concatenate([p, q]))
    emd = 0
    for i in range(r.size - 1):
        emd += abs(np.sum(w_p[p <= r[i]]) - np.sum(w_q[q <= r[i]])) * (r[i + 1] - r[i])
    return emd / (np.sum(w_p) + np.sum(w_q))


def ldb_emd_measure_weighted(Xw, W, y):
    Xw, W = _f(Xw), _f(W)
    classes = _unique(y)
    yl = list(np.asarray(y).tolist())
    groups = [[i for i, v in enumerate(yl) if v == c] for c in classes]
    flat, wf = Xw.reshape(-1, Xw.shape[-1], order="F"), W.reshape(-1, Xw.shape[-1], order="F")
    D = np.zeros(flat.shape[0], dtype=Xw.dtype)
    for a in range(len(classes)):
        for b in range(a + 1, len(classes)):
            for e in range(flat.shape[0]):
                D[e] += emd_pair_weighted(flat[e, groups[a]], flat[e, groups[b]], wf[e, groups[a]], wf[e, groups[b]])
    return np.asfortranarray(D.reshape(Xw.shape[:-1], order="F"))


def ldb_pdf_discriminant(G, dm="are", lp=2):
    """discriminant_measure(Gamma, dm) for a density map ldb_measures.jl:139-183, 217-251: pairs of classes, summed over the
    density axis"""
    G = _f(G)
    nc = G.shape[-1]
    D = np.zeros(G.shape[:-2], dtype=G.dtype, order="F")
    for i in range(nc):
        for j in range(i + 1, nc):
            P = np.vectorize(lambda a, b: _ldb_pair(a, b, dm, lp), otypes=[G.dtype])(G[..., i], G[..., j])
            D = D + np.cumsum(P, axis=-1)[..., -1]
    return D


# ---- shift-invariant wavelet packet decomposition (SIWT.jl) ---------------------------
class SIWTObject:
    """ShiftInvariantWaveletTransformObject (siwt/siwt_utls.jl:75-90) with the same field names.  Nodes maps
    (Depth, IndexAtDepth, TransformShift) -> {"Value": vector, "Cost": float}; BestTree is the index list in
    the reference's push / delete order."""

    def __init__(self, signal, qmf, L=0, d=0):
        signal = np.array(signal, dtype=np.asarray(signal).dtype if np.asarray(signal).dtype in (np.float32, np.float64) else np.float64)
        if not 0 <= L <= maxtransformlevels(signal.size):                    # siwt_utls.jl:85
            raise ValueError("Provided MaxTransformLevels is too large.")
        if not 0 <= d < signal.size:                                         # siwt_utls.jl:86
            raise ValueError("Provided MaxShiftedTransformLevels is too large.")
        self.qmf = np.asarray(qmf, dtype=np.float64)
        self.SignalSize = int(signal.size)
        self.MaxTransformLevel = int(L)
        self.MaxShiftedTransformLevels = int(d)
        cost = siwt_nodecost(signal)                                         # siwt_utls.jl:143
        self.Nodes = {(0, 0, 0): {"Value": signal, "Cost": cost}}
        self.MinCost = cost
        self.BestTree = [(0, 0, 0)]


def siwt_nodecost(v, nrm=None):
    """coefcost(v, ShannonEntropyCost(), nrm) with nrm defaulting to norm(v) (siwt_utls.jl:118-126)"""
    v = np.ascontiguousarray(v)
    fn = getattr(lib(), "wxo_siwt_nodecost" + _suf(v.dtype))
    ct = ctypes.c_double if v.dtype == np.float64 else ctypes.c_float
    fn.restype = ct
    return float(fn(_p(v), _L(v.size), ct(-1.0 if nrm is None else nrm)))


def siwt_norm(v):
    v = np.ascontiguousarray(v)
    fn = getattr(lib(), "wxo_siwt_norm" + _suf(v.dtype))
    fn.restype = ctypes.c_double if v.dtype == np.float64 else ctypes.c_float
    return float(fn(_p(v), _L(v.size)))


def sidwt_step(v, h, g, s):
    """sidwt_step!(w1, w2, v, h, g, s) siwt_one_level.jl:71-98"""
    v = np.ascontiguousarray(v); n = v.size
    h = np.ascontiguousarray(h, dtype=np.float64); g = np.ascontiguousarray(g, dtype=np.float64)
    w1, w2 = np.empty(n // 2, v.dtype), np.empty(n // 2, v.dtype)
    _call("wxo_sidwt_step", v.dtype, _p(w1), _p(w2), _p(v), n, _p(h), _p(g), _I(h.size), _I(int(bool(s))), restype=None)
    return w1, w2


def isidwt_step(w1, w2, h, g, s):
    """isidwt_step!(v, w1, w2, h, g, s) siwt_one_level.jl:154-185"""
    w1 = np.ascontiguousarray(w1); w2 = np.ascontiguousarray(w2, dtype=w1.dtype); n = 2 * w1.size
    h = np.ascontiguousarray(h, dtype=np.float64); g = np.ascontiguousarray(g, dtype=np.float64)
    v = np.empty(n, w1.dtype)
    _call("wxo_isidwt_step", v.dtype, _p(v), _p(w1), _p(w2), n, _p(h), _p(g), _I(h.size), _I(int(bool(s))), restype=None)
    return v


def siwpd(x, qmf, L=None, d=None):
    """siwpd(x, wt, L, d) SIWT.jl:57-69 + siwpd_subtree! :92-137"""
    x = np.asarray(x)
    if L is None:
        L = maxtransformlevels(x.size)
    if d is None:
        d = L
    if not 0 <= L <= maxtransformlevels(x.size):
        raise OracleAssertion("0 <= L <= maxtransformlevels(x)")
    if not 1 <= d <= L:
        raise OracleAssertion("1 <= d <= L")
    g, h = makereverseqmfpair(qmf)
    obj = SIWTObject(x, qmf, L, d)
    nrm = siwt_norm(obj.Nodes[(0, 0, 0)]["Value"])

    def step(index, shifted):                                                # siwt_one_level.jl:24-51
        depth, idx, shift = index
        w1, w2 = sidwt_step(obj.Nodes[index]["Value"], h, g, shifted)
        cs = shift + (1 << depth) * int(shifted)
        c1, c2 = (depth + 1, idx << 1, cs), (depth + 1, (idx << 1) + 1, cs)
        obj.Nodes[c1] = {"Value": w1, "Cost": siwt_nodecost(w1, nrm)}
        obj.Nodes[c2] = {"Value": w2, "Cost": siwt_nodecost(w2, nrm)}
        obj.BestTree.append(c1); obj.BestTree.append(c2)
        return c1, c2

    def subtree(index, rem):                                                 # SIWT.jl:92-137
        depth, _, shift = index
        assert 0 <= depth <= L and 0 <= rem <= L - depth
        if depth == L or (rem == 0 and shift > 0):
            return
        c1, c2 = step(index, False)
        crem = rem - 1 if shift > 0 else min(rem, L - (depth + 1))
        subtree(c1, crem); subtree(c2, crem)
        if rem > 0:
            c1, c2 = step(index, True)
            subtree(c1, rem - 1); subtree(c2, rem - 1)

    subtree((0, 0, 0), d)
    return obj


def siwt_delete_node(obj, index):
    """delete_node! siwt_utls.jl:217-236"""
    if index not in obj.Nodes:
        return
    del obj.Nodes[index]
    obj.BestTree = [t for t in obj.BestTree if t != index]
    depth, idx, shift = index
    for ci in ((depth + 1, idx << 1, shift), (depth + 1, (idx << 1) + 1, shift),
               (depth + 1, idx << 1, shift + (1 << depth)), (depth + 1, (idx << 1) + 1, shift + (1 << depth))):
        siwt_delete_node(obj, ci)


def siwt_isvalidtree(obj, literal=False):
    """Wavelets.Util.isvalidtree(siwtObj) siwt_utls.jl:185-207.  literal=True restates :195 as written: the
    parent is looked up with the CHILD's TransformShift, so every node produced by a shifted step counts as an
    orphan, the function returns false for any tree that uses a shift, and bestbasistree!'s closing @assert
    (siwt_bestbasis.jl:34) throws for such signals.  The default also accepts the parent whose shift is the
    child's minus 2^(depth-1) (the shifted step's bookkeeping, siwt_one_level.jl:37-38), which is what the
    docstring describes ("each node has a parent"); the two agree on every tree the reference accepts."""
    assert set(obj.Nodes.keys()) == set(obj.BestTree)
    nodes = set(obj.BestTree)
    for (depth, idx, shift) in nodes:
        is_root = (depth, idx, shift) == (0, 0, 0)
        has_parent = (depth - 1, idx >> 1, shift) in nodes
        if not literal and depth >= 1 and (shift >> (depth - 1)) & 1:
            has_parent = has_parent or (depth - 1, idx >> 1, shift - (1 << (depth - 1))) in nodes
        has_c = (depth + 1, idx << 1, shift) in nodes and (depth + 1, (idx << 1) + 1, shift) in nodes
        sh = shift + (1 << depth)
        has_s = (depth + 1, idx << 1, sh) in nodes and (depth + 1, (idx << 1) + 1, sh) in nodes
        is_leaf = not has_c and not has_s
        if not ((is_root ^ has_parent) and (is_leaf ^ has_c ^ has_s)):
            return False
    return True


def siwt_bestbasistree(obj):
    """bestbasistree!(siwtObj) siwt_bestbasis.jl:28-36 + bestbasis_treeselection! :52-102.  Costs are kept in
    the element type of the signal, like the reference's T2 fields."""
    dt = obj.Nodes[(0, 0, 0)]["Value"].dtype.type

    def select(index):
        if index not in obj.Nodes:
            return None
        depth, idx, shift = index
        c1, c2 = (depth + 1, idx << 1, shift), (depth + 1, (idx << 1) + 1, shift)
        sh = shift + (1 << depth)
        s1, s2 = (depth + 1, idx << 1, sh), (depth + 1, (idx << 1) + 1, sh)
        node = dt(obj.Nodes[index]["Cost"])
        k1, k2, k3, k4 = select(c1), select(c2), select(s1), select(s2)
        ns = None if (k1 is None and k2 is None) else dt(dt(k1) + dt(k2))
        ss = None if (k3 is None and k4 is None) else dt(dt(k3) + dt(k4))
        has_ns, has_ss = ns is not None, ss is not None
        node_lt_ns = has_ns and node < ns
        node_lt_ss = has_ss and node < ss
        ns_lt_ss = (has_ns and not has_ss) or (has_ns and has_ss and ns < ss)
        node_min = (not has_ns and not has_ss) or (node_lt_ns and node_lt_ss)
        if node_min:
            for t in (c1, c2, s1, s2):
                siwt_delete_node(obj, t)
        elif ns_lt_ss:
            siwt_delete_node(obj, s1); siwt_delete_node(obj, s2)
            obj.Nodes[index]["Cost"] = float(ns)
        else:
            siwt_delete_node(obj, c1); siwt_delete_node(obj, c2)
            obj.Nodes[index]["Cost"] = float(ss)
        return obj.Nodes[index]["Cost"]

    select((0, 0, 0))
    obj.MinCost = obj.Nodes[(0, 0, 0)]["Cost"]
    assert siwt_isvalidtree(obj)
    return obj.BestTree


def isiwpd(obj, literal=False):
    """isiwpd(siwtObj) SIWT.jl:166-173 + isiwpd_subtree! :190-229 (children are deleted as they are merged).
    literal=True hands the numeric step the flag exactly as siwt_one_level.jl:126 spells it."""
    g, h = makereverseqmfpair(obj.qmf)
    tree = lambda: set(obj.BestTree)

    def subtree(index):
        depth, idx, shift = index
        has_ns = (depth + 1, idx << 1, shift) in tree()
        has_ss = (depth + 1, idx << 1, shift + (1 << depth)) in tree()
        if not (has_ns or has_ss):
            return
        if not (has_ns ^ has_ss):
            raise OracleAssertion("hasNonShiftedChildren xor hasShiftedChildren")
        cs = shift if has_ns else shift + (1 << depth)
        c1, c2 = (depth + 1, idx << 1, cs), (depth + 1, (idx << 1) + 1, cs)
        subtree(c1); subtree(c2)
        # The flag handed to the numeric step.  siwt_one_level.jl:126 spells it `nodeObj.TransformShift ==
        # child1Obj.TransformShift`, which read literally is true for the NON-shifted children and returns the
        # signal rotated by one sample -- the reference's own known-answer test (test/transforms.jl:261-267,
        # `isiwpd(siwtObj) ≈ signal`) cannot hold under that reading.  The oracle is pinned to the test: `s` is
        # true exactly when the children were produced by the shifted step (sidwt_step!(..., true)), the only
        # choice that inverts siwt_one_level.jl:71-98.  Julia cannot be run here to settle it; noted in DESIGN.md.
        s = (cs == shift) if literal else (cs != shift)
        obj.Nodes[index]["Value"] = isidwt_step(obj.Nodes[c1]["Value"], obj.Nodes[c2]["Value"], h, g, s)
        siwt_delete_node(obj, c1); siwt_delete_node(obj, c2)

    subtree((0, 0, 0))
    return obj.Nodes[(0, 0, 0)]["Value"]
