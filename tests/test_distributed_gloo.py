"""N > 1 path on CPU: world_size-2 `gloo` processes exercise the batch sharding, the C1 all-gather
of reconstructed output and the C2 all-reduce of JBB moments (the per-rank compute stand-in is the
oracle, since there is no GPU here; on the MI355X node the same functions run over RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import waveletsext_jl_amd as wx
    from waveletsext_jl_amd import distributed as wd   # noqa
    import wx_oracle as wo
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(99)                       # same full batch on every rank
        n = 32
        X = np.asfortranarray(rng.standard_normal((n, B)))
        q = wx.wavelet(wx.WT.db4).qmf
        lo, hi = wd.shard_range(B, world, rank)
        xl = np.asfortranarray(X[:, lo:hi])
        # transforms are per-signal: shard -> local wpdall -> local iwpdall, no collective
        yl = wo.wpdall(xl, q)
        xr = wo.iwpdall(yl, q)
        t_local = torch.from_numpy(np.ascontiguousarray(xr.T)).T       # column-major (n, B_r)
        full = wd.allgather_batch(t_local, B)
        assert tuple(full.shape) == (n, B)
        np.testing.assert_allclose(full.numpy(), X, atol=1e-12)
        # 3-D shards gather too (packet tables)
        t3 = torch.from_numpy(np.ascontiguousarray(yl.transpose(2, 1, 0))).permute(2, 1, 0)
        full3 = wd.allgather_batch(t3, B)
        np.testing.assert_array_equal(full3.numpy(), wo.wpdall(X, q))
        # C2: JBB moments of the local shard, all-reduced == moments of the whole batch
        s = torch.from_numpy(np.ascontiguousarray(yl.sum(axis=2).T)).T
        qq = torch.from_numpy(np.ascontiguousarray((yl ** 2).sum(axis=2).T)).T
        s, qq = wd.allreduce_moments(s, qq)
        Y = wo.wpdall(X, q)
        np.testing.assert_allclose(s.numpy(), Y.sum(axis=2), rtol=0, atol=1e-11)
        np.testing.assert_allclose(qq.numpy(), (Y ** 2).sum(axis=2), rtol=0, atol=1e-10)
        # the tree every rank derives from the reduced moments == the oracle's tree on the full batch
        N = B
        ex, ex2 = s.numpy() / N, qq.numpy() / N
        sig = np.sqrt(ex2 - ex ** 2)
        costs = []
        for lvl in range(Y.shape[1]):
            n0 = n >> lvl
            for node in range(1 << lvl):
                costs.append(2 * np.log(np.abs(sig[node * n0:(node + 1) * n0, lvl])).sum())
        tree = wx.bestbasis_treeselection(np.array(costs), n)
        assert (tree == wo.bestbasistree_jbb(Y)).all()
        # LDB energy maps of a sharded batch: per-shard maps + norm sums, two small all-reduces
        labels = [("u", "v", "w")[i % 3] if i > 0 else "w" for i in range(B)]      # rank-dependent class mix
        classes = wo._unique(labels)
        yl_lab = labels[lo:hi]
        Gr = np.full((n, Y.shape[1], len(classes)), np.nan)
        nsr = np.zeros(len(classes))
        for ci, c in enumerate(classes):
            idx = [i for i, v in enumerate(yl_lab) if v == c]
            if idx:
                nsr[ci] = sum(float(np.sqrt((xl[:, i] ** 2).sum())) ** 2 for i in idx)
                Gr[:, :, ci] = (yl[:, :, idx] ** 2).sum(axis=2) / nsr[ci]
        Gfull = wd.combine_energy_maps(np.asfortranarray(Gr), nsr)
        np.testing.assert_allclose(Gfull, wo.ldb_energy_map(Y, labels), rtol=1e-12, atol=1e-14)
        # C1 in pieces (the bench's overlapped schedule): every rank fills the chunks of its own shard of `full` in
        # place and posts them; afterwards every rank holds the whole array.  Ragged shards, more pieces than signals.
        # ... through both exchange schedules: grouped point-to-point, and one all_gather collective per chunk
        for nchunks, mode in [(k, m) for k in (1, 3, 4, 16) for m in ("p2p", "collective")]:
            fullt = torch.full((B, n), float("nan"), dtype=torch.float64).T          # column-major (n, B)
            g = wd.make_gather(fullt, B, nchunks=nchunks, mode=mode)
            assert type(g) is (wd.OverlappedAllGather if mode == "p2p" else wd.CollectiveAllGather)
            assert (g.lo, g.hi) == (lo, hi)
            assert g.nposts == max(len(wd.chunk_ranges(sz, nchunks)) for sz in wd.shard_sizes(B, world))
            for c in range(g.nposts):
                if c < len(g.chunks):
                    c0, c1 = g.chunks[c]
                    g.local_chunk(c).copy_(torch.from_numpy(np.ascontiguousarray(xr[:, c0:c1].T)).T)
                g.post(c)
            g.finish()
            np.testing.assert_allclose(fullt.numpy(), X, atol=1e-12)
            # a caller that loops over its OWN chunks only (the shards are ragged: the ranks' chunk counts differ when there
            # are more pieces than signals): finish() posts the exchanges it has left out, and the object is reusable
            fullt.fill_(float("nan"))
            for c in range(len(g.chunks)):
                c0, c1 = g.chunks[c]
                g.local_chunk(c).copy_(torch.from_numpy(np.ascontiguousarray(xr[:, c0:c1].T)).T)
                g.post(c)
            g.finish()
            np.testing.assert_allclose(fullt.numpy(), X, atol=1e-12)
            with pytest.raises(ValueError):
                g.post(1)                                   # exchanges are posted in order
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 7])
def test_world2_gloo_shard_gather_reduce(tmp_path, B):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, B, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_chunk_ranges():
    sys.path.insert(0, ROOT)
    from waveletsext_jl_amd import distributed as wd
    for Bl in (0, 1, 3, 4, 5, 8192):
        for k in (1, 4, 7):
            ch = wd.chunk_ranges(Bl, k)
            assert len(ch) == (min(k, Bl) if Bl else 0)
            assert not ch or (ch[0][0] == 0 and ch[-1][1] == Bl and all(a[1] == b[0] for a, b in zip(ch, ch[1:])))
            assert all(c1 > c0 for c0, c1 in ch)


def test_shard_ranges_cover_and_are_contiguous():
    sys.path.insert(0, ROOT)
    import waveletsext_jl_amd  # noqa
    from waveletsext_jl_amd import distributed as wd
    for B in (0, 1, 7, 8, 65536, 262144 + 3):
        for world in (1, 2, 4, 8):
            edges = [wd.shard_range(B, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == B
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = wd.shard_sizes(B, world)
            assert max(sizes) - min(sizes) <= 1
