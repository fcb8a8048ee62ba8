"""Orthogonal wavelet filters: the host-side stand-in for Wavelets.jl's `WT` module.

The reference only ever uses `WT.qmf(wt)` and `WT.makereverseqmfpair(wt, true)` of an
`OrthoFilter` (call sites: src/mod/DWT.jl:141, src/mod/SWT.jl:119, src/mod/acwt/acwt_utils.jl:8).
Wavelets.jl itself is not vendored in /root/reference, so the tables are regenerated from the
published definitions by tools/gen_filters.py (see its docstring for provenance / pinning).
The C ABI takes the QMF vector as an argument, so a Julia caller passes Wavelets.jl's own table.
"""
import numpy as np

from ._filter_tables import QMF as _QMF


class FilterClass:
    """`WT.db4`-style tag (Wavelets.jl: WT.FilterClass singletons)."""

    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return "WT.%s" % self.name


class OrthoFilter:
    """Wavelets.jl `OrthoFilter`: only `.qmf` (Float64 vector) and `.name` are used on the path."""

    def __init__(self, qmf, name="custom"):
        q = np.ascontiguousarray(np.asarray(qmf, dtype=np.float64))
        if q.ndim != 1 or q.size < 2 or q.size % 2:
            raise ArgumentError("qmf must be a vector of even length >= 2")
        self.qmf = q
        self.name = name

    def __len__(self):
        return self.qmf.size

    def __repr__(self):
        return "OrthoFilter(%s, %d taps)" % (self.name, self.qmf.size)


class ArgumentError(ValueError):
    """Julia `ArgumentError` (the reference throws it at e.g. Utils.jl:120, SWT.jl:65)."""


def daubechies(N):
    """Wavelets.jl `daubechies(N)` in double precision (any N >= 1); tabulated N use the
    60-digit tables instead."""
    from math import comb
    if N < 1:
        raise ArgumentError("N must be positive")
    if N == 1:
        return np.array([1.0, 1.0]) / np.sqrt(2.0)
    C = np.array([comb(N - 1 + n, n) for n in range(N)], dtype=np.float64)
    Y = np.roots(C[::-1])
    Z = []
    for y in Y:
        d = 2 * np.sqrt(complex(y * y - y))
        Z += [1 - 2 * y + d, 1 - 2 * y - d]
    roots = [-1.0] * N + [z for z in Z if abs(z) < 1]
    h = np.real(np.poly(roots))
    return h / np.linalg.norm(h)


class _WT:
    """Namespace mirroring `Wavelets.WT` for the names the hot path needs."""

    def __init__(self):
        for k in _QMF:
            setattr(self, k, FilterClass(k))

    @staticmethod
    def qmf(f):
        return f.qmf

    @staticmethod
    def makereverseqmfpair(f, fw=True, T=np.float64):
        """Wavelets.jl `WT.makereverseqmfpair(f, fw)`; the reference always passes fw=true and
        binds the result as `g, h` -> (reverse(qmf), mirror(qmf))."""
        q = np.asarray(f.qmf, dtype=T)
        sgn = np.where(np.arange(q.size) % 2 == 0, 1.0, -1.0).astype(T)
        mq = q * sgn
        if fw:
            return q[::-1].copy(), mq
        return q.copy(), mq[::-1].copy()

    @staticmethod
    def Daubechies(N):
        return FilterClass("db%d" % N)


WT = _WT()


def wavelet(cls):
    """Wavelets.jl `wavelet(WT.db4)` -> OrthoFilter."""
    name = cls.name if isinstance(cls, FilterClass) else str(cls)
    if name in _QMF:
        return OrthoFilter(_QMF[name], name)
    if name.startswith("db") and name[2:].isdigit():
        return OrthoFilter(daubechies(int(name[2:])), name)
    raise ArgumentError("unknown wavelet class %r" % (name,))
