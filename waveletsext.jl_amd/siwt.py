"""Shift-invariant wavelet packet decomposition: host-side mirror of the reference's `SIWT` module
(src/mod/SIWT.jl, siwt/siwt_utls.jl, siwt/siwt_one_level.jl, siwt/siwt_bestbasis.jl).  SURVEY section 8(f) row 4.

The reference stores one Dict of node objects per signal and recurses over it; on the MI355X a batch of signals
shares one flat table (see include/waveletsext_hip.h, "Shift-invariant wavelet packet decomposition") and every
level of the decomposition, the node costs, the three-way best-basis choice and the inverse are level-synchronous
launches over all nodes of all signals.  The classes below present that table through the reference's field names
(`Nodes`, `BestTree`, `MinCost`, ...); `siwpdall` / `bestbasistreeall_` / `isiwpdall` are the batch forms (the
reference has none: its callers loop over signals).

Two places follow the reference's tests rather than the letter of its code (both noted in DESIGN.md): the
inverse step is told `shifted` exactly when the children came from the shifted step (siwt_one_level.jl:126 spells
the flag the other way round, which cannot pass test/transforms.jl:261-267), and `isvalidtree` accepts a shifted
node's parent under the parent's own shift (siwt_utls.jl:195 looks it up under the child's)."""
import ctypes

import numpy as np

from . import _lib
from ._arrays import Arg, is_torch, jl_empty, qmf_arg, to_numpy, torch
from .filters import ArgumentError
from .util import maxtransformlevels


def _ncols(L, d):
    return sum(1 << min(j, d) for j in range(L + 1))


def _coloff(j, d):
    return sum(1 << min(i, d) for i in range(j))


def _nodeoff(j, d):
    return sum((1 << min(i, d)) << i for i in range(j))


def _slot(j, d, shift):
    m = max(0, j - d)
    return shift >> m if shift % (1 << m) == 0 else None


class ShiftInvariantWaveletTransformNode:
    """siwt/siwt_utls.jl:23-52: Depth, IndexAtDepth, TransformShift, Cost, Value (1-D only, like the reference)"""

    def __init__(self, Depth, IndexAtDepth, TransformShift, Cost, Value, N=1):
        nd = Value.ndim if hasattr(Value, "ndim") else np.ndim(Value)
        if N != nd:
            raise TypeError("Value array is not of %d dimension." % N)
        if N == 2:
            raise ArgumentError("2D SIWT not available yet.")
        if N != 1:
            raise ArgumentError("Coefficient array has dimension larger than 2.")
        top = (1 << Depth) - 1
        if IndexAtDepth > top or TransformShift > top:
            raise ArgumentError("Invalid IndexAtDepth or TransformShift for 1D coefficients.")
        self.Depth, self.IndexAtDepth, self.TransformShift = int(Depth), int(IndexAtDepth), int(TransformShift)
        self.Cost, self.Value = Cost, Value

    @classmethod
    def from_data(cls, data, depth, indexAtDepth, transformShift, nrm=None):
        """outer constructor siwt_utls.jl:118-126: Cost = coefcost(data, ShannonEntropyCost(), nrm = norm(data)).
        The cost comes from the same device kernel as the decomposition's (a one-node table)."""
        nd = data.ndim if hasattr(data, "ndim") else np.ndim(data)
        if nd != 1:
            raise ArgumentError("2D SIWT not available yet." if nd == 2 else "Coefficient array has dimension larger than 2.")
        xa = Arg(data)
        n = xa.shape[0]
        c = xa.new((1, 1))
        fn = getattr(_lib.lib(), "wx_bb_costs" + xa.suffix)
        _lib.check(fn(xa.ptr, c.ptr, n, 1, 1, 0, 0, xa.stream()))
        cost = float(to_numpy(c.arr)[0, 0])
        if nrm is not None:
            # sum -(x/N)^2 log (x/N)^2 = r C0 - r log r with r = (norm(x)/N)^2 (the terms (x/norm)^2 sum to one)
            own = float(np.sqrt(float((to_numpy(xa.arr).astype(np.float64) ** 2).sum())))
            if nrm == 0:
                cost = 0.0
            elif own != 0:
                r = (own / float(nrm)) ** 2
                cost = r * cost - r * np.log(r)
        return cls(depth, indexAtDepth, transformShift, cost, xa.arr)


class _Nodes:
    """Dict-like view `(Depth, IndexAtDepth, TransformShift) -> ShiftInvariantWaveletTransformNode` of one signal"""

    def __init__(self, obj):
        self._o = obj

    def _has(self, key):
        o = self._o
        j, i, t = (int(v) for v in key)
        if not (0 <= j <= o._b.L and 0 <= i < (1 << j) and 0 <= t < (1 << j)):
            return False
        s = _slot(j, o._b.d, t)
        if s is None:
            return False
        st = o._status_host()
        return st is None or st[_nodeoff(j, o._b.d) + (s << j) + i] != 0

    def __contains__(self, key):
        return self._has(key)

    def __getitem__(self, key):
        if not self._has(key):
            raise KeyError(key)
        o = self._o
        j, i, t = (int(v) for v in key)
        d = o._b.d
        s = _slot(j, d, t)
        np_ = o.SignalSize >> j
        col = _coloff(j, d) + s
        val = o._table()[i * np_:(i + 1) * np_, col]
        cost = float(o._costs_host()[_nodeoff(j, d) + (s << j) + i])
        return ShiftInvariantWaveletTransformNode(j, i, t, cost, val)

    def keys(self):
        return list(self._o.BestTree)

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def items(self):
        return [(k, self[k]) for k in self.keys()]


class ShiftInvariantWaveletTransformObject:
    """siwt/siwt_utls.jl:75-90.  `ShiftInvariantWaveletTransformObject(signal, wavelet, L=0, d=0)` is the
    undecomposed object (root only, siwt_utls.jl:136-148); `siwpd` returns decomposed ones."""

    def __init__(self, signal, wavelet, maxTransformLevel=0, maxShiftedTransformLevel=0, _batch=None, _index=0):
        if _batch is not None:
            self._b, self._i = _batch, _index
            self.SignalSize = _batch.n
            self.MaxTransformLevel = _batch.L
            self.MaxShiftedTransformLevels = _batch.d
            self.Wavelet = _batch.wt
            return
        xa = Arg(signal)
        if xa.arr.ndim != 1:
            raise ArgumentError("2D SIWT not available yet." if xa.arr.ndim == 2 else "Coefficient array has dimension larger than 2.")
        n = xa.shape[0]
        if not 0 <= maxTransformLevel <= maxtransformlevels(n):
            raise ArgumentError("Provided MaxTransformLevels is too large.")
        if not 0 <= maxShiftedTransformLevel < n:
            raise ArgumentError("Provided MaxShiftedTransformLevels is too large.")
        # root only: a depth-0 table holding the signal and its cost
        self._b = _Batch(xa.arr.reshape(n, 1) if not is_torch(xa.arr) else xa.arr.reshape(n, 1), wavelet, 0, 0)
        self._i = 0
        self.SignalSize = n
        self.MaxTransformLevel = int(maxTransformLevel)
        self.MaxShiftedTransformLevels = int(maxShiftedTransformLevel)
        self.Wavelet = wavelet

    # -- storage of this signal ------------------------------------------------------------------
    def _table(self):
        return self._b.W[:, :, self._i]

    def _costs_host(self):
        return self._b.costs_host()[:, self._i]

    def _status_host(self):
        st = self._b.status_host()
        return None if st is None else st[:, self._i]

    # -- the reference's fields ---------------------------------------------------------------------
    @property
    def Nodes(self):
        return _Nodes(self)

    @property
    def MinCost(self):
        return float(self._costs_host()[0])

    @property
    def BestTree(self):
        """node indices in the reference's order: push order of siwpd_subtree! (SIWT.jl:116-134, children pushed
        by sidwt_step!, siwt_one_level.jl:46-47), minus what bestbasistree! / isiwpd deleted"""
        L, d = self._b.L, self._b.d
        st = self._status_host()
        out = [(0, 0, 0)]

        def alive(j, i, t):
            s = _slot(j, d, t)
            return s is not None and (st is None or st[_nodeoff(j, d) + (s << j) + i] != 0)

        def walk(j, i, t):
            if j == L:
                return
            for sh in ((t,) if st is not None and st[_nodeoff(j, d) + (_slot(j, d, t) << j) + i] == 2 else
                       (t + (1 << j),) if st is not None and st[_nodeoff(j, d) + (_slot(j, d, t) << j) + i] == 3 else
                       (t, t + (1 << j))):
                if _slot(j + 1, d, sh) is None or not alive(j + 1, 2 * i, sh):
                    continue
                out.append((j + 1, 2 * i, sh)); out.append((j + 1, 2 * i + 1, sh))
                walk(j + 1, 2 * i, sh); walk(j + 1, 2 * i + 1, sh)

        if st is None or st[0] > 1:
            walk(0, 0, 0)
        return out


class _Batch:
    """The flat table of a batch: W (n, NS, B), costs (NN, B), status (NN, B) uint8 or None (nothing deleted)."""

    def __init__(self, X, wt, L, d):
        xa = Arg(X)
        assert xa.arr.ndim == 2
        self.n, self.B = xa.shape
        self.wt, self.L, self.d = wt, int(L), int(d)
        self.NS, self.NN = _ncols(self.L, self.d), _nodeoff(self.L + 1, self.d)
        self.kind, self.suffix, self.dtype = xa.kind, xa.suffix, xa.dtype
        w = xa.new((self.n, self.NS, self.B))
        c = xa.new((self.NN, self.B))
        q, qp, F = qmf_arg(wt)
        if self.L == 0:
            # root only (the undecomposed object): the table is the signal, its cost the one-node cost
            if xa.kind == "torch":
                w.arr[:, 0, :] = xa.arr
            else:
                w.arr[:, 0, :] = xa.arr
            fn = getattr(_lib.lib(), "wx_bb_costs" + xa.suffix)
            _lib.check(fn(xa.ptr, c.ptr, self.n, 1, self.B, 0, 0, xa.stream()))
        else:
            fn = getattr(_lib.lib(), "wx_siwpd" + xa.suffix)
            _lib.check(fn(xa.ptr, w.ptr, c.ptr, self.n, self.L, self.d, self.B, qp, F, xa.stream()))
        self.W, self.costs, self.status = w.arr, c.arr, None
        self._wa, self._ca = w, c
        self._ch = self._sh = None

    def costs_host(self):
        if self._ch is None:
            self._ch = to_numpy(self.costs)
        return self._ch

    def status_host(self):
        if self.status is None:
            return None
        if self._sh is None:
            self._sh = self.status.cpu().numpy().T if is_torch(self.status) else self.status
        return self._sh

    def _set_status(self, st_host):
        """replace the status bytes from a host (NN, B) array"""
        st_host = np.asfortranarray(st_host.astype(np.uint8))
        self.status = torch.from_numpy(np.ascontiguousarray(st_host.T)).to(self.W.device) if self.kind == "torch" else st_host
        self._sh = None

    def _status_ptr(self):
        if self.kind == "torch":
            return ctypes.c_void_p(self.status.data_ptr())          # (B, NN) row-major == (NN, B) column-major
        return ctypes.c_void_p(self.status.ctypes.data)

    def bestbasis(self):
        if self.kind == "torch":
            self.status = torch.empty((self.B, self.NN), dtype=torch.uint8, device=self.W.device)
        else:
            self.status = np.empty((self.NN, self.B), dtype=np.uint8, order="F")
        fn = getattr(_lib.lib(), "wx_siwt_bestbasis" + self.suffix)
        _lib.check(fn(self._ca.ptr, self._status_ptr(), self.L, self.d, self.B, self._wa.stream()))
        self._ch = self._sh = None

    def inverse(self, literal=False):
        out = self._wa.new((self.n, self.B))
        q, qp, F = qmf_arg(self.wt)
        fn = getattr(_lib.lib(), "wx_isiwpd" + self.suffix)
        _lib.check(fn(self._wa.ptr, self._status_ptr(), out.ptr, self.n, self.L, self.d, self.B, qp, F, int(bool(literal)),
                      self._wa.stream()))
        # the children were merged into their parents and deleted (SIWT.jl:223-226): only the roots are left
        if self.kind == "torch":
            self.status.zero_()
            self.status[:, 0] = 1
        else:
            self.status[:] = 0
            self.status[0, :] = 1
        self._sh = None
        return out.arr


class ShiftInvariantWaveletTransformBatch:
    """Batch form (no counterpart in the reference, whose callers loop over signals): `batch[i]` is the
    ShiftInvariantWaveletTransformObject of signal i, sharing the batch's device table."""

    def __init__(self, b):
        self._b = b

    def __len__(self):
        return self._b.B

    def __getitem__(self, i):
        if not 0 <= i < self._b.B:
            raise IndexError(i)
        return ShiftInvariantWaveletTransformObject(None, None, _batch=self._b, _index=int(i))

    @property
    def Table(self):
        """W (n, NS, batch)"""
        return self._b.W

    @property
    def Costs(self):
        return self._b.costs

    @property
    def Status(self):
        """(NN, batch) bytes after bestbasistreeall_: 0 deleted, 1 leaf, 2 / 3 kept with (shifted) children"""
        return self._b.status_host()

    @property
    def MinCost(self):
        return self._b.costs_host()[0, :].copy()


def _check_Ld(n, L, d):
    L = maxtransformlevels(n) if L is None else int(L)
    d = L if d is None else int(d)
    assert 0 <= L <= maxtransformlevels(n), "0 <= L <= maxtransformlevels(x) (SIWT.jl:62)"
    assert 1 <= d <= L, "1 <= d <= L (SIWT.jl:63)"
    return L, d


def siwpdall(X, wt, L=None, d=None):
    """siwpd of every column of X (n, batch): one table, one launch per level"""
    xa = Arg(X)
    assert xa.arr.ndim == 2
    L, d = _check_Ld(xa.shape[0], L, d)
    return ShiftInvariantWaveletTransformBatch(_Batch(xa.arr, wt, L, d))


def siwpd(x, wt, L=None, d=None):
    """siwpd(x, wt[, L, d]) SIWT.jl:57-69"""
    xa = Arg(x)
    if xa.arr.ndim != 1:
        raise TypeError("siwpd takes a vector (SIWT.jl:57)")
    L, d = _check_Ld(xa.shape[0], L, d)
    return ShiftInvariantWaveletTransformBatch(_Batch(xa.arr.reshape(xa.shape[0], 1), wt, L, d))[0]


def bestbasistreeall_(batch):
    """bestbasistree! of every signal of the batch; returns the status bytes (NN, batch)"""
    batch._b.bestbasis()
    return batch._b.status_host()


def bestbasistree_(siwtObj):
    """bestbasistree!(siwtObj) siwt/siwt_bestbasis.jl:28-36 (acts on every signal that shares the table)"""
    siwtObj._b.bestbasis()
    assert isvalidtree(siwtObj)                                     # siwt_bestbasis.jl:34
    return siwtObj.BestTree


def isiwpdall(batch, literal=False):
    """isiwpd of every signal; (n, batch).  literal=True passes the inverse step the flag exactly as
    siwt/siwt_one_level.jl:126 spells it (see the module docstring)"""
    b = batch._b
    if b.status is None:
        assert b.L == 0, "hasNonShiftedChildren xor hasShiftedChildren (SIWT.jl:210): run bestbasistreeall_ first"
        return b.W[:, 0, :]
    return b.inverse(literal)


def isiwpd(siwtObj, literal=False):
    """isiwpd(siwtObj) SIWT.jl:166-173: the children are merged bottom-up and deleted; returns the root's Value"""
    b = siwtObj._b
    if b.status is None:
        assert b.L == 0, "hasNonShiftedChildren xor hasShiftedChildren (SIWT.jl:210): run bestbasistree_ first"
    else:
        b.inverse(literal)
    return siwtObj.Nodes[(0, 0, 0)].Value


def delete_node_(siwtObj, index):
    """delete_node!(siwtObj, index) siwt_utls.jl:217-236: the node and every descendant (both kinds)"""
    b, i = siwtObj._b, siwtObj._i
    st = b.status_host()
    st = np.ones((b.NN, b.B), dtype=np.uint8) if st is None else np.array(st, dtype=np.uint8, order="F")
    d, L = b.d, b.L

    def rec(j, k, t):
        s = _slot(j, d, t) if 0 <= j <= L and 0 <= t < (1 << j) else None
        if s is None or st[_nodeoff(j, d) + (s << j) + k, i] == 0:
            return
        st[_nodeoff(j, d) + (s << j) + k, i] = 0
        for sh in (t, t + (1 << j)):
            rec(j + 1, 2 * k, sh); rec(j + 1, 2 * k + 1, sh)

    rec(*(int(v) for v in index))
    b._set_status(st)


def isvalidtree(siwtObj, literal=False):
    """Wavelets.Util.isvalidtree(siwtObj) siwt_utls.jl:185-207: every node but the root has its parent, and every
    node has no children, its non-shifted pair, or its shifted pair -- never both.  literal=True looks the parent
    up under the child's TransformShift, as :195 is written (false for every tree that uses a shift)"""
    nodes = set(siwtObj.BestTree)
    for (j, i, t) in nodes:
        is_root = (j, i, t) == (0, 0, 0)
        has_parent = (j - 1, i >> 1, t) in nodes
        if not literal and j >= 1 and (t >> (j - 1)) & 1:
            has_parent = has_parent or (j - 1, i >> 1, t - (1 << (j - 1))) in nodes
        has_c = (j + 1, 2 * i, t) in nodes and (j + 1, 2 * i + 1, t) in nodes
        has_s = (j + 1, 2 * i, t + (1 << j)) in nodes and (j + 1, 2 * i + 1, t + (1 << j)) in nodes
        leaf = not has_c and not has_s
        if not ((is_root ^ has_parent) and (leaf ^ has_c ^ has_s)):
            return False
    return True
