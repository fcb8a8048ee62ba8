"""One rank of tests/test_gpu_multi.py (run as a fresh child process: `python multi_rank_worker.py RANK WORLD PORT
BACKEND OUTDIR B`).  backend "nccl": one GPU per rank, RCCL over xGMI -- the deployment; backend "gloo": every rank on
GPU 0, exchanges staged through the host -- the same script on a one-GPU box, so that the logic below has run before it
meets a second GPU.

Every rank holds the same seeded batch, transforms its contiguous shard (dwt/dwt_all.jl:277-279: the loop over signals
is what is sharded) with the HIP library, and then checks, bit for bit unless said:
  C1  allgather_batch == OverlappedAllGather (grouped point-to-point) == CollectiveAllGather == this rank's own transform
      of the whole batch; NativeComm.allgather_batch (the library's RCCL entry point, nccl only) for equal shards;
  C2  allreduce_moments of the shards' JBB moments == moments of the whole batch to 1e-12 (the order of the sum over
      signals differs), the same through NativeComm.allreduce_moments, and the tree every rank derives == the tree of
      the whole batch (bestbasis/bestbasis_tree.jl:153-154)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, backend, outdir, B = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], int(sys.argv[6])
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    import waveletsext_jl_amd as wx
    from waveletsext_jl_amd import distributed as wd
    D = sys.modules["waveletsext_jl_amd.dwt"]          # the submodule (the package attribute `dwt` is the function)

    devi = rank if backend == "nccl" else 0
    torch.cuda.set_device(devi)
    dev = torch.device("cuda", devi)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, L = 4096, 6
        wt = wx.wavelet(wx.WT.db4)
        rng = np.random.default_rng(2024)
        X = np.asfortranarray(rng.standard_normal((n, B)))
        Xd = wx.to_device(X, dev)
        lo, hi = wd.shard_range(B, world, rank)
        xl = wx.jl_empty((n, hi - lo), torch.float64, dev)
        xl.copy_(Xd[:, lo:hi])
        # the whole batch on this rank (reference for the gathered results) and the shard
        Y_all = wx.wpdall(Xd, wt, L)
        R_all = wx.iwpdall(Y_all, wt, L)
        yl = wx.wpdall(xl, wt, L)
        rl = wx.iwpdall(yl, wt, L)
        assert torch.equal(rl, R_all[:, lo:hi]), "a shard's transform differs from the same signals inside the whole batch"

        # ---- C1 -------------------------------------------------------------------------------------------------
        full = wd.allgather_batch(rl, B)
        torch.cuda.synchronize(dev)
        assert tuple(full.shape) == (n, B) and torch.equal(full, R_all), "allgather_batch"
        for mode in ("p2p", "collective"):
            for nchunks in (1, 4, 7):
                out = wx.jl_empty((n, B), torch.float64, dev)
                out.fill_(float("nan"))
                g = wd.make_gather(out, B, nchunks=nchunks, mode=mode)
                for rep in range(2):                           # the object serves several steps
                    for c in range(g.nposts):
                        if c < len(g.chunks):
                            c0, c1 = g.chunks[c]
                            # the inverse writes its chunk straight into its place in `out`, like bench.py's step
                            D._iwpd_batched(D.Arg(yl[..., c0:c1]), D.Arg(g.local_chunk(c)), 1, wt, L, None)
                        g.post(c)
                    g.finish()
                    torch.cuda.synchronize(dev)
                    assert torch.equal(out, R_all), "%s gather, %d chunks, step %d" % (mode, nchunks, rep)
        if world > 1:
            dist.barrier()

        # ---- C2 -------------------------------------------------------------------------------------------------
        from waveletsext_jl_amd import bestbasis as bb
        s_all, q_all = bb.jbb_moments(Y_all)
        s, q = bb.jbb_moments(yl)
        s, q = wd.allreduce_moments(s, q)
        torch.cuda.synchronize(dev)
        for a, b in ((s, s_all), (q, q_all)):
            assert float((a - b).abs().max() / b.abs().max()) < 1e-12
        tree_all = wx.bestbasistree(Y_all, wx.JBB())
        costs = bb.costs_from_moments(s, q, B, wx.JBB())
        tree = bb.bestbasis_treeselection(costs, n)
        assert (np.asarray(tree) == np.asarray(tree_all)).all(), "tree from all-reduced moments"
        assert (np.asarray(wd.bestbasistree_sharded(yl, B, wx.JBB())) == np.asarray(tree_all)).all()

        # ---- the library's own RCCL entry points (what the Julia shim calls) ------------------------------------------
        if backend == "nccl":
            holder = [wd.NativeComm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(holder, src=0)
            comm = wd.NativeComm(world, rank, holder[0])
            try:
                Be = (B // world) * world                       # equal shards for the plain collective
                le, he = wd.shard_range(Be, world, rank)
                piece = wx.jl_empty((n, he - le), torch.float64, dev)
                piece.copy_(R_all[:, le:he])
                fulln = comm.allgather_batch(piece, Be)
                torch.cuda.synchronize(dev)
                assert torch.equal(fulln, R_all[:, :Be]), "NativeComm.allgather_batch"
                # ragged shards through the C ABI (wx_allgatherv_out_*): the whole batch B, B mod world != 0 in the callers' runs
                lo_r, hi_r = wd.shard_range(B, world, rank)
                rag = wx.jl_empty((n, hi_r - lo_r), torch.float64, dev)
                rag.copy_(R_all[:, lo_r:hi_r])
                fullv = comm.allgatherv_batch(rag, B)
                torch.cuda.synchronize(dev)
                assert torch.equal(fullv, R_all), "NativeComm.allgatherv_batch (ragged)"
                s2, q2 = bb.jbb_moments(yl)
                s2, q2 = comm.allreduce_moments(s2, q2)
                torch.cuda.synchronize(dev)
                for a, b in ((s2, s_all), (q2, q_all)):
                    assert float((a - b).abs().max() / b.abs().max()) < 1e-12
                p32 = piece.to(torch.float32)
                p32c = wx.jl_empty(tuple(p32.shape), torch.float32, dev)
                p32c.copy_(p32)
                f32 = comm.allgather_batch(p32c, Be)
                torch.cuda.synchronize(dev)
                assert torch.equal(f32, R_all[:, :Be].to(torch.float32))
            finally:
                comm.close()
        if world > 1:
            dist.barrier()
        open(os.path.join(outdir, "ok%d" % rank), "w").write(
            "%s rank %d/%d device %d %s" % (backend, rank, world, devi, torch.cuda.get_device_name(dev)))
    finally:
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
