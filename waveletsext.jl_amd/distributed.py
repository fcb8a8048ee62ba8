"""Multi-GPU layer: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on
the MI355X node, "gloo" in the CPU tests).

Every transform of the hot path is independent per signal (`*all` drivers loop over the last
dimension: dwt/dwt_all.jl:277-279, swt/swt_all.jl:171-173, acwt/acwt_all.jl:254-256), so the batch
shards contiguously across ranks -- in Julia's column-major layout a shard is one contiguous
block -- and the transforms need NO data-path collective.  Two exchange steps exist:

  C1  all-gather of the reconstructed output, only when every rank needs the whole (n, B) array;
  C2  all-reduce(sum) of the JBB moments [sum | sumsq] (bestbasis_tree.jl:153-154 sums over the
      signal axis), after which every rank derives the same costs and the same tree.
"""
import numpy as np

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None


def shard_range(B, world, rank):
    """Contiguous batch shard [lo, hi) of rank `rank`; sizes differ by at most one signal."""
    base, rem = divmod(int(B), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(B, world):
    return [shard_range(B, world, r)[1] - shard_range(B, world, r)[0] for r in range(world)]


def local_shard(x, world, rank):
    """View of this rank's signals (last dimension) of a full-batch array."""
    lo, hi = shard_range(x.shape[-1], world, rank)
    return x[..., lo:hi]


def _as_batch_major(t):
    """(sig..., B) column-major tensor -> contiguous (B, reversed sig...) view of the same memory."""
    return t.permute(*reversed(range(t.dim()))) if t.dim() > 1 else t


def allgather_batch(local, B_total, group=None):
    """C1: gather per-rank shards (sig..., B_r) into the full (sig..., B_total) array on every
    rank.  Shards may be ragged (B_total not divisible by the world size)."""
    world = dist.get_world_size(group)
    sizes = shard_sizes(B_total, world)
    sig = tuple(local.shape[:-1])
    rev = tuple(reversed(sig))
    src = _as_batch_major(local).contiguous()
    if len(set(sizes)) == 1:
        full = torch.empty((B_total,) + rev, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(full, src, group=group)
    else:
        # ragged shards: pad to the largest shard (collectives need equal counts), then trim
        smax = max(sizes)
        padded = torch.zeros((smax,) + rev, dtype=local.dtype, device=local.device)
        padded[: src.shape[0]] = src
        buf = torch.empty((world * smax,) + rev, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(buf, padded, group=group)
        full = torch.cat([buf[r * smax: r * smax + sizes[r]] for r in range(world)], dim=0)
    return _as_batch_major(full) if full.dim() > 1 else full


def chunk_ranges(B_local, nchunks):
    """[c0, c1) pieces of a shard of B_local signals: at most nchunks, sizes differ by at most one, none empty"""
    k = max(1, min(int(nchunks), int(B_local)))
    return [shard_range(B_local, k, c) for c in range(k)] if B_local else []


class OverlappedAllGather:
    """C1 in pieces, overlapped with the compute that produces the pieces.

    `full` is the (sig..., B_total) column-major result on every rank.  A rank's transform writes chunk c of its own
    shard straight into `full[..., lo + c0 : lo + c1]` (no staging copy) and then calls `post(c)`: the chunk goes to the
    other ranks, and their chunk c lands in this rank's `full`, on a side stream, while the caller's stream computes
    chunk c + 1.  The exchange is a grouped point-to-point pattern (send to every peer, receive from every peer): xGMI is
    a full mesh of point-to-point links, so every link carries exactly one pair's traffic, and every piece lands in its
    final place (a ring all-gather wants a rank-major receive buffer, which a chunk of every shard is not).
    `finish()` makes the caller's stream wait for all pieces.  With the gloo backend (CPU validation, or several ranks on
    one GPU) device tensors are staged through the host.
    """

    def __init__(self, full, B_total, nchunks=4, group=None):
        self.full, self.group = full, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.B_total = int(B_total)
        self.lo, self.hi = shard_range(B_total, self.world, self.rank)
        self.nchunks = max(1, int(nchunks))
        self.chunks = chunk_ranges(self.hi - self.lo, self.nchunks)
        # every rank posts the same number of exchanges: the longest chunk list over all shards (a rank whose own list is
        # shorter -- or empty -- still has to receive its peers' later chunks)
        self.nposts = max(len(chunk_ranges(n_r, self.nchunks)) for n_r in shard_sizes(self.B_total, self.world))
        self.posted = 0
        self.cuda = bool(full.is_cuda)
        self.host_staged = self.cuda and dist.get_backend(group) != "nccl"
        self.side = torch.cuda.Stream(full.device) if self.cuda else None
        self.work = []

    def local_chunk(self, c):
        """the slice of `full` this rank's compute must fill for chunk c"""
        c0, c1 = self.chunks[c]
        return self.full[..., self.lo + c0: self.lo + c1]

    def _peer_chunk(self, r, c):
        rlo, rhi = shard_range(self.B_total, self.world, r)
        ch = chunk_ranges(rhi - rlo, self.nchunks)
        if c >= len(ch):
            return None
        return self.full[..., rlo + ch[c][0]: rlo + ch[c][1]]

    def post(self, c):
        """exchange number c (0 .. nposts - 1, in order): this rank's chunk c goes out (if it has one), every peer's chunk c
        comes in"""
        if c != self.posted or c >= self.nposts:
            raise ValueError("OverlappedAllGather.post(%d): exchanges are posted in order, %d of %d done" % (c, self.posted, self.nposts))
        self.posted += 1
        if self.world == 1:
            return
        mine = _as_batch_major(self.local_chunk(c)) if c < len(self.chunks) else None
        ops, recvs = [], []
        for r in range(self.world):
            if r == self.rank:
                continue
            peer = dist.get_global_rank(self.group, r) if self.group is not None else r
            dst = self._peer_chunk(r, c)
            if dst is not None:
                dbm = _as_batch_major(dst)
                if self.host_staged:
                    buf = torch.empty(dbm.shape, dtype=dbm.dtype, device="cpu")
                    recvs.append((dbm, buf))
                    ops.append(dist.P2POp(dist.irecv, buf, peer, self.group))
                else:
                    ops.append(dist.P2POp(dist.irecv, dbm, peer, self.group))
            if mine is not None:
                ops.append(dist.P2POp(dist.isend, mine.cpu() if self.host_staged else mine, peer, self.group))
        if not ops:
            return
        if self.cuda and not self.host_staged:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.full.device))     # chunk c is complete on the caller's stream
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)
                self.work.extend(dist.batch_isend_irecv(ops))
        else:
            if self.cuda:
                torch.cuda.current_stream(self.full.device).synchronize()
            for wk in dist.batch_isend_irecv(ops):
                wk.wait()
            for dbm, buf in recvs:
                dbm.copy_(buf)

    def finish(self):
        """posts the exchanges the caller has not posted (a caller that loops over its own chunks only would leave its
        peers' sends unmatched), then makes the caller's stream wait for every piece"""
        while self.posted < self.nposts:
            self.post(self.posted)
        if self.world == 1:
            self.posted = 0
            return
        if self.cuda and not self.host_staged:
            with torch.cuda.stream(self.side):
                for wk in self.work:
                    wk.wait()
            torch.cuda.current_stream(self.full.device).wait_stream(self.side)
        self.work = []
        self.posted = 0                                    # the object serves the next step


class CollectiveAllGather(OverlappedAllGather):
    """The same exchange schedule as OverlappedAllGather -- chunk c of every shard while chunk c + 1 is computed -- with one
    `all_gather` collective per chunk instead of the grouped point-to-point pattern (`bench.py --gather collective`, and
    what `make_gather(..., mode="auto")` falls back to when the point-to-point group raises).  Pieces of equal size are
    gathered straight into their places in `full`; when the ranks' chunk c differ in size (by at most one signal) they are
    padded to the largest, gathered into a staging buffer and copied to their places."""

    def post(self, c):
        if c != self.posted or c >= self.nposts:
            raise ValueError("CollectiveAllGather.post(%d): exchanges are posted in order, %d of %d done" % (c, self.posted, self.nposts))
        self.posted += 1
        if self.world == 1:
            return
        pieces = [self._peer_chunk(r, c) if r != self.rank else (self.local_chunk(c) if c < len(self.chunks) else None)
                  for r in range(self.world)]
        sizes = [0 if p is None else int(p.shape[-1]) for p in pieces]
        smax = max(sizes)
        if smax == 0:
            return
        dev = self.full.device
        rev = tuple(reversed(tuple(self.full.shape[:-1])))
        stage = "cpu" if self.host_staged else dev

        def run():
            if len(set(sizes)) == 1 and not self.host_staged:
                outs = [_as_batch_major(p) for p in pieces]               # every piece lands in its place
                return [dist.all_gather(outs, outs[self.rank], group=self.group, async_op=True)], None
            src = torch.zeros((smax,) + rev, dtype=self.full.dtype, device=stage)
            if sizes[self.rank]:
                src[: sizes[self.rank]].copy_(_as_batch_major(pieces[self.rank]))
            buf = torch.empty((self.world * smax,) + rev, dtype=self.full.dtype, device=stage)
            wk = dist.all_gather_into_tensor(buf, src, group=self.group, async_op=True)
            return [wk], buf

        def place(buf):
            for r in range(self.world):
                if r != self.rank and sizes[r]:
                    _as_batch_major(pieces[r]).copy_(buf[r * smax: r * smax + sizes[r]])

        if self.cuda and not self.host_staged:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))                       # chunk c is complete on the caller's stream
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)
                works, buf = run()
                if buf is not None:
                    for wk in works:
                        wk.wait()
                    place(buf)
                else:
                    self.work.extend(works)
        else:
            if self.cuda:
                torch.cuda.current_stream(dev).synchronize()
            works, buf = run()
            for wk in works:
                wk.wait()
            if buf is not None:
                place(buf)


def make_gather(full, B_total, nchunks=4, group=None, mode="p2p"):
    """the exchange object of the reconstructed output: mode "p2p" (grouped point-to-point, every piece lands in place) or
    "collective" (one all_gather per chunk)"""
    if mode not in ("p2p", "collective"):
        raise ValueError("gather mode must be p2p or collective")
    return (OverlappedAllGather if mode == "p2p" else CollectiveAllGather)(full, B_total, nchunks=nchunks, group=group)


def allreduce_moments(s, q, group=None):
    """C2: in-place all-reduce(sum) of the JBB moment arrays (one fused buffer, one collective)."""
    fused = torch.stack([_as_batch_major(s).contiguous().reshape(-1), _as_batch_major(q).contiguous().reshape(-1)])
    dist.all_reduce(fused, op=dist.ReduceOp.SUM, group=group)
    shp = tuple(reversed(tuple(s.shape)))
    s_out = _as_batch_major(fused[0].reshape(shp))
    q_out = _as_batch_major(fused[1].reshape(shp))
    return s_out, q_out


def bestbasistree_sharded(X_local, N_total, method=None, group=None):
    """bestbasistree(X, JBB(...)) (BestBasis.jl:194-201) over a batch that is sharded across
    ranks: local moments on the GPU, one all-reduce, costs + tree selection on every rank."""
    from . import bestbasis as bb
    s, q = bb.jbb_moments(X_local)
    s, q = allreduce_moments(s, q, group)
    costs = bb.costs_from_moments(s, q, N_total, method)
    return bb.bestbasis_treeselection(costs, X_local.shape[0])


def acwpd_bestbasistree_sharded(x_local, wt, L, N_total, method=None, group=None):
    """BASELINE config 5: acwpdall + JBB tree over a sharded batch without materialising the
    decomposition; returns the BitVector on every rank."""
    from . import bestbasis as bb
    method = bb.JBB(redundant=True) if method is None else method
    s, q = bb.acwpd_jbb_moments(x_local, wt, L)
    s, q = allreduce_moments(s, q, group)
    costs = bb.costs_from_moments(s, q, N_total, method)
    return bb.bestbasis_treeselection(costs, x_local.shape[0])



def combine_energy_maps(G_local, ns_local, group=None):
    """LDB energy maps of a sharded batch (ldb_energymap.jl:109-141): Gamma = sum_r Gamma_r * ns_r / sum_r ns_r, where
    ns_r are the shard's per-class norm sums (a class absent from a shard has ns_r = 0 and a NaN map, which counts
    as zero).  Two small all-reduces; every rank gets the same maps."""
    is_t = torch is not None and isinstance(G_local, torch.Tensor)
    G = G_local if is_t else torch.from_numpy(np.ascontiguousarray(np.asarray(G_local)))
    ns = torch.as_tensor(np.asarray(ns_local), dtype=G.dtype, device=G.device)
    W = torch.nan_to_num(G, nan=0.0) * ns                              # classes are the last axis: broadcasts
    W = W.contiguous()
    dist.all_reduce(W, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(ns, op=dist.ReduceOp.SUM, group=group)
    out = W / ns
    return out if is_t else np.asfortranarray(out.numpy())


def energy_map_sharded(Xw_local, y_local, classes, group=None):
    """energy_map over a batch sharded across ranks: local class sums on the GPU, then combine_energy_maps"""
    from . import ldb
    G, ns = ldb.energy_map(Xw_local, y_local, classes=classes, return_norm_sum=True)
    return combine_energy_maps(G, ns, group)


# ---- the same two exchange steps through the library's own RCCL entry points ---------------------
# (what a host without torch.distributed -- the Julia shim -- calls; include/waveletsext_hip.h "Multi-GPU
# exchange").  The launcher only has to broadcast the 128-byte id from rank 0.
class NativeComm:
    """RCCL communicator owned by libwaveletsext_hip.so (wx_comm_init / wx_comm_destroy)."""

    def __init__(self, nranks, rank, unique_id):
        import ctypes
        from . import _lib
        self._lib = _lib
        self.nranks, self.rank = int(nranks), int(rank)
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().wx_comm_init(self.nranks, self.rank, ctypes.cast(buf, ctypes.c_void_p), ctypes.byref(h)))
        self.handle = h

    @staticmethod
    def unique_id():
        import ctypes
        from . import _lib
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.lib().wx_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)))
        return buf.raw

    def _fn(self, stem, t):
        return getattr(self._lib.lib(), stem + ("_f64" if t.dtype == torch.float64 else "_f32"))

    def allgather_batch(self, local, B_total):
        """C1 with equal shards (pad ragged shards before calling): (sig..., B_r) -> (sig..., nranks*B_r)"""
        assert local.is_cuda and local.dtype in (torch.float32, torch.float64)
        assert B_total == local.shape[-1] * self.nranks, "equal shards only: pad ragged shards"
        src = _as_batch_major(local).contiguous()
        full = torch.empty((B_total,) + tuple(src.shape[1:]), dtype=local.dtype, device=local.device)
        st = torch.cuda.current_stream(local.device).cuda_stream
        self._lib.check(self._fn("wx_allgather_out", local)(src.data_ptr(), full.data_ptr(), src.numel(), self.handle, st))
        return _as_batch_major(full) if full.dim() > 1 else full

    def allgatherv_batch(self, local, B_total):
        """C1 with ragged shards (wx_allgatherv_out_*): this rank's (sig..., B_r) block of the contiguous sharding
        `shard_range(B_total, nranks, rank)` -> (sig..., B_total) on every rank, every piece landing in place"""
        import ctypes
        import numpy as np
        assert local.is_cuda and local.dtype in (torch.float32, torch.float64)
        sig = int(np.prod(local.shape[:-1], dtype=np.int64)) if local.dim() > 1 else 1
        bounds = [shard_range(B_total, self.nranks, r) for r in range(self.nranks)]
        lo, hi = bounds[self.rank]
        assert hi - lo == local.shape[-1], "local is not this rank's shard of B_total"
        counts = (ctypes.c_int64 * self.nranks)(*[(b - a) * sig for a, b in bounds])
        src = _as_batch_major(local).contiguous()
        full = torch.empty((B_total,) + tuple(src.shape[1:]), dtype=local.dtype, device=local.device)
        st = torch.cuda.current_stream(local.device).cuda_stream
        self._lib.check(self._fn("wx_allgatherv_out", local)(src.data_ptr(), full.data_ptr(), ctypes.cast(counts, ctypes.c_void_p),
                                                             self.nranks, self.handle, st))
        return _as_batch_major(full) if full.dim() > 1 else full

    def allreduce_moments(self, s, q):
        """C2: one in-place sum over the fused [sum | sumsq] buffer"""
        fused = torch.stack([_as_batch_major(s).contiguous().reshape(-1), _as_batch_major(q).contiguous().reshape(-1)])
        st = torch.cuda.current_stream(s.device).cuda_stream
        self._lib.check(self._fn("wx_allreduce_moments", fused)(fused.data_ptr(), fused.numel(), self.handle, st))
        shp = tuple(reversed(tuple(s.shape)))
        return _as_batch_major(fused[0].reshape(shp)), _as_batch_major(fused[1].reshape(shp))

    def close(self):
        if self.handle:
            self._lib.check(self._lib.lib().wx_comm_destroy(self.handle))
            self.handle = None
