#!/usr/bin/env python3
"""bench.py -- headline benchmark of the batched wavelet-packet hot path on MI355X.

Metric (BASELINE.json): Msamples/s for forward + inverse wavelet packets, with the dominant
kernel's achieved HBM bandwidth against the 8 TB/s roofline, next to the CPU path.

A step = one forward pass + one inverse pass over one batch of synthetic signals that are already
resident in HBM.  Default workload = BASELINE config 2 (`wpdall` 65536 x 4096 Float64, db8, full
packet tree L=12, then `iwpdall`).  Other workloads (`--workload`): `target` (north-star target
wptall/iwptall db4 L=10), `cfg3` (swptall/iswptall 16384-sample signals, haar, L=12, one resident
chunk of the 8192-signal batch per step), `cfg4` (2-D wptall/iwptall 512x512 Float32 db4 L=6, the
per-GPU shard of 512 images), `cfg5` (acwpd + JBB moments/costs/tree, coif6, L=11, a 2048-signal
slice of the per-GPU shard).  N > 1: one process per GPU (torchrun), each rank owns a fixed-size
shard of the batch (weak scaling); the transforms need no data-path collective.

Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling 6290
FP64_PEAK_TFLOPS = 78.6        # FP64 vector spec; tools/ubench.hip measures 55-61 on this part

WORKLOADS = {
    "cfg2": dict(kind="wpd", n=4096, batch=65536, wavelet="db8", L=12, dtype="f64",
                 kernel="k_fwd1d_fused<double, 16, 512, true, 2>",
                 fwd_kernels=[("k_fwd1d_fused<double, 16, 512, true, 2>", 1)],
                 desc="BASELINE config 2: wpdall+iwpdall 65536x4096 f64 db8 full tree L=12"),
    "target": dict(kind="wpt", n=4096, batch=65536, wavelet="db4", L=10, dtype="f64",
                   kernel="k_fwd1d_inplace<double, 8, 256, false>",
                   fwd_kernels=[("k_fwd1d_inplace<double, 8, 256, false>", 1)],
                   desc="north-star target: wptall+iwptall 65536x4096 f64 db4 L=10"),
    "target_haar": dict(kind="wpt", n=4096, batch=65536, wavelet="haar", L=10, dtype="f64",
                        kernel="k_haar_wpt_f64<256>",
                        fwd_kernels=[("k_haar_wpt_f64<256>", 1)],
                        desc="north-star target with the Haar filter: wptall+iwptall 65536x4096 f64 haar L=10 "
                             "(Walsh-Hadamard kernels, wx_haar.hip)"),
    "cfg3": dict(kind="swpt", n=16384, batch=64, wavelet="haar", L=12, dtype="f64",
                 kernel="k_swt_fwd_multi_rc<double, 8, 8>",
                 fwd_kernels=[("k_swt_fwd_multi<double, 8>", 2), ("k_swt_fwd_multi_rc<double, 8, 8>", 2)],
                 desc="BASELINE config 3: swptall+iswptall (average-based) 16384-sample f64 haar L=12; one resident "
                      "chunk of 64 signals (32 GiB of leaves) of the 8192-signal batch per step"),
    "cfg4": dict(kind="wpt2d", m=512, n=512, batch=512, wavelet="db4", L=6, dtype="f32",
                 kernel="k_rows_fused<float, 8, false, 4, 2>",
                 fwd_kernels=[("k_fwd1d_inplace<float, 8, 64, false>", 1), ("k_rows_fused<float, 8, false, 4, 2>", 1)],
                 desc="BASELINE config 4: 2-D wptall+iwptall 512x512 f32 db4 L=6, 512 images per GPU (4096 / 8)"),
    "cfg5": dict(kind="acwpd_jbb", n=2048, batch=2048, wavelet="coif6", L=11, dtype="f64",
                 kernel="k_acwpd_subtree_moments<5, 4, 9>",
                 fwd_kernels=[("k_swt_fwd_level<double, true>", 6), ("k_jbb_moments<double>", 1),
                              ("k_acwpd_subtree_moments<5, 4, 9>", 1), ("k_jbb_costs<double>", 1)],
                 desc="BASELINE config 5: acwpd + JBB moments/costs/tree 2048-sample f64 coif6 L=11; 2048-signal slice "
                      "of the 32768-signal per-GPU shard per step (no inverse: output is the tree)"),
    "bb": dict(kind="wpd_bb", n=4096, batch=16384, wavelet="db8", L=12, dtype="f64",
               kernel="k_bb_costs1d<double>",
               fwd_kernels=[("k_bb_norms<double>", 1), ("k_bb_costs1d<double>", 1), ("k_bb_treeselect<double, 2>", 1)],
               desc="SURVEY 8(f) row 3: per-signal best basis, bestbasistreeall(wpdall(x), BB()) 16384x4096 f64 db8 L=12; "
                    "timed leg = Shannon costs + tree selection over the resident 6.5 GiB table (wpdall is the other leg)"),
    "ldb": dict(kind="wpd_ldb", n=4096, batch=16384, wavelet="db8", L=12, dtype="f64",
                kernel="k_ldb_class_partial<double>",
                fwd_kernels=[("k_ldb_root_norm2<double>", 1), ("k_ldb_class_sum<double>", 1),
                             ("k_ldb_class_partial<double>", 1), ("k_ldb_class_combine<double>", 1)],
                desc="SURVEY 8(f) row 2: LDB time-frequency energy maps of 4 classes over wpdall(x) 16384x4096 f64 db8 "
                     "L=12; timed leg = energy_map over the resident 6.5 GiB table (wpdall is the other leg)"),
    "siwt": dict(kind="siwt", n=1024, batch=4096, wavelet="db4", L=10, d=3, dtype="f64",
                 kernel="k_siwt_fwd_level<double, true, 8>",
                 fwd_kernels=[("k_siwt_fwd_level<double, false, 8>", 1), ("k_siwt_fwd_level<double, true, 8>", 9), ("k_siwt_norms<double>", 1),
                              ("k_siwt_costs<double>", 1)],
                 desc="SURVEY 8(f) row 4: shift-invariant packet decomposition siwpd(x, wt, 10, 3) + node costs of 4096 "
                      "1024-sample f64 signals (71-column table, 2.4 GB); second leg = bestbasistree! + isiwpd of all signals"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch (debug)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--gather", action="store_true",
                    help="also time an RCCL all-gather of the reconstructed output (reported, not in value)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# CPU baseline: the oracle ("port": scalar restatement of the reference's loops, one thread, like the
# reference) timed on this host on a bounded sample of the same workload
# ------------------------------------------------------------------------------------------------
def cpu_baseline(w, seconds):
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import wx_oracle as wo
    import waveletsext_jl_amd as wx
    lib = wo.lib()
    q = np.ascontiguousarray(wx.wavelet(getattr(wx.WT, w["wavelet"])).qmf)
    rng = np.random.default_rng(1002)
    P = lambda a: ctypes.c_void_p(a.ctypes.data)
    I, L64 = ctypes.c_int, ctypes.c_int64
    kind, L = w["kind"], w["L"]

    def run(B):
        if kind in ("wpd", "wpt"):
            n = w["n"]
            tree = wo.maketree1d(n, L, "full").astype(np.uint8)
            x = np.asfortranarray(rng.standard_normal((n, B)))
            xh = np.empty_like(x)
            t0 = time.perf_counter()
            if kind == "wpd":
                y = np.empty((n, L + 1, B), order="F")
                lib.wxo_wpdall1d_f64(P(y), P(x), L64(n), I(L), L64(B), P(q), I(q.size))
                lib.wxo_iwpdall1d_f64(P(xh), P(y), L64(n), I(L + 1), L64(B), P(tree), L64(tree.size), P(q), I(q.size))
            else:
                y = np.empty_like(x)
                lib.wxo_wptall1d_f64(P(y), P(x), L64(n), L64(B), P(tree), L64(tree.size), P(q), I(q.size))
                lib.wxo_iwptall1d_f64(P(xh), P(y), L64(n), L64(B), P(tree), L64(tree.size), P(q), I(q.size))
            dt = time.perf_counter() - t0
            assert np.abs(xh - x).max() < 1e-9
            return dt, B * n
        if kind == "swpt":
            n = w["n"]
            x = rng.standard_normal((n, B))
            t0 = time.perf_counter()
            for i in range(B):
                xw = wo.swpt(x[:, i], q, L)
                xr = wo.iswpt(xw, q)
            dt = time.perf_counter() - t0
            assert np.abs(xr - x[:, B - 1]).max() < 1e-9
            return dt, B * n
        if kind == "wpt2d":
            m, n = w["m"], w["n"]
            x = rng.standard_normal((m, n, B)).astype(np.float32)
            t0 = time.perf_counter()
            for i in range(B):
                y = wo.wpt(np.asfortranarray(x[:, :, i]), q, L)
                xr = wo.iwpt(y, q, L)
            dt = time.perf_counter() - t0
            assert np.abs(xr - x[:, :, B - 1]).max() < 1e-3
            return dt, B * m * n
        if kind == "wpd_bb":
            n = w["n"]
            x = rng.standard_normal((n, B))
            t0 = time.perf_counter()
            X = np.asfortranarray(np.stack([wo.wpd(x[:, i], q, L) for i in range(B)], axis=-1))
            trees = wo.bestbasistreeall_bb(X)
            dt = time.perf_counter() - t0
            assert trees.shape == (n - 1, B)
            return dt, B * n
        if kind == "wpd_ldb":
            n = w["n"]
            x = rng.standard_normal((n, B))
            y = [i % 4 for i in range(B)]
            t0 = time.perf_counter()
            X = np.asfortranarray(np.stack([wo.wpd(x[:, i], q, L) for i in range(B)], axis=-1))
            G = wo.ldb_energy_map(X, y)
            dt = time.perf_counter() - t0
            assert G.shape[:2] == (n, L + 1)
            return dt, B * n
        if kind == "siwt":
            n = w["n"]
            x = rng.standard_normal((n, B))
            t0 = time.perf_counter()
            for i in range(B):
                obj = wo.siwpd(x[:, i], q, L, w["d"])
                wo.siwt_bestbasistree(obj)
                xr = wo.isiwpd(obj)
            dt = time.perf_counter() - t0
            assert np.abs(xr - x[:, B - 1]).max() < 1e-9
            return dt, B * n
        if kind == "acwpd_jbb":
            n = w["n"]
            x = rng.standard_normal((n, B))
            t0 = time.perf_counter()
            X = np.asfortranarray(np.stack([wo.acwpd(x[:, i], q, L) for i in range(B)], axis=-1))
            tree = wo.bestbasistree_jbb(X, redundant=True)
            dt = time.perf_counter() - t0
            assert tree.size == n - 1
            return dt, B * n
        raise ValueError(kind)

    def run_omp(B):
        """all host cores: OpenMP over the batch (1-D decimated workloads only)"""
        n = w["n"]
        tree = wo.maketree1d(n, L, "full").astype(np.uint8)
        x = np.asfortranarray(rng.standard_normal((n, B)))
        xh = np.empty_like(x)
        y = np.empty((n, L + 1, B) if kind == "wpd" else (n, B), order="F")
        t0 = time.perf_counter()
        if kind == "wpd":
            lib.wxo_wpd_iwpd_roundtrip_omp_f64(P(xh), P(y), P(x), L64(n), I(L), L64(B), P(tree), L64(tree.size), P(q), I(q.size))
        else:
            lib.wxo_wpt_iwpt_roundtrip_omp_f64(P(xh), P(y), P(x), L64(n), L64(B), P(tree), L64(tree.size), P(q), I(q.size))
        dt = time.perf_counter() - t0
        assert np.abs(xh - x).max() < 1e-9
        return dt, B * n

    probe = 16 if kind in ("wpd", "wpt") else 2
    run(probe)                                   # first call: library load, page faults
    t_probe, _ = run(probe)
    cap = {"wpd": 16384, "wpt": 16384, "wpt2d": 320, "wpd_bb": 4096, "wpd_ldb": 2048}.get(kind, 32)
    B = int(max(probe, min(cap, seconds / (t_probe / probe))))
    dt, samples = run(B)
    out = {"value": samples / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": "%d of the %d signals per step, same transform pair, %s, %.1f s"
                     % (B, w["batch"], "oracle: C -O2 steps under the reference's Dict-and-recursion object model in Python, one thread"
                        if kind == "siwt" else "oracle C -O2 single thread", dt)}
    if kind in ("wpd", "wpt"):
        nthr = int(lib.wxo_omp_max_threads())
        Bo = int(min(16384, max(256, B * max(1, nthr // 4))))
        run_omp(min(Bo, 1024))
        dto, so = run_omp(Bo)
        out["all_cores"] = {"value": so / dto / 1e6, "unit": "Msamples/s", "cores": nthr, "kind": "port",
                            "sample": "%d signals, OpenMP over the batch, %.1f s" % (Bo, dto)}
    return out


# ------------------------------------------------------------------------------------------------
# GPU workloads: each returns (fwd, inv, check, info)
# ------------------------------------------------------------------------------------------------
def make_workload(w, wx, torch, dev, rank):
    D = sys.modules["waveletsext_jl_amd.dwt"]        # the submodule (the package attribute `dwt` is the function)
    from waveletsext_jl_amd._arrays import qmf_arg
    kind, L, B = w["kind"], w["L"], w["batch"]
    wt = wx.wavelet(getattr(wx.WT, w["wavelet"]))
    F = len(wt.qmf)
    td = torch.float64 if w["dtype"] == "f64" else torch.float32
    es = 8 if w["dtype"] == "f64" else 4
    gen = torch.Generator(device=dev).manual_seed(1002 + rank)
    A = D.Arg
    if kind in ("wpd", "wpt"):
        n = w["n"]
        x = wx.jl_empty((n, B), td, dev); x.normal_(generator=gen)
        xh = wx.jl_empty((n, B), td, dev)
        if kind == "wpd":
            y = wx.jl_empty((n, L + 1, B), td, dev)
            fwd = lambda: D._wpd_batched(A(x), A(y), 1, wt, L)
            inv = lambda: D._iwpd_batched(A(y), A(xh), 1, wt, L, None)
            fb = es * n * B * (L + 2)                # x read once + (L+1) columns written once
        else:
            y = wx.jl_empty((n, B), td, dev)
            fwd = lambda: D._wpt_batched("wx_wpt", A(x), A(y), 1, wt, L, None)
            inv = lambda: D._wpt_batched("wx_iwpt", A(y), A(xh), 1, wt, L, None)
            fb = es * n * B * 2
        check = lambda: float((xh - x).abs().max() / x.abs().max())
        return fwd, inv, check, dict(fwd_bytes=fb, inv_bytes=es * n * B * 2, fwd_flops=2.0 * F * n * L * B,
                                     samples=n * B, bound="hbm", keep=(x, y, xh))
    if kind == "swpt":
        n = w["n"]
        x = wx.jl_empty((n, B), td, dev); x.normal_(generator=gen)
        xw = wx.jl_empty((n, 1 << L, B), td, dev)
        xh = wx.jl_empty((n, B), td, dev)
        q, qp, Fq = qmf_arg(wt)
        fwd = lambda: D._call("wx_swpt1d", "_f64", A(x).ptr, A(xw).ptr, n, L, B, qp, Fq, A(x).stream())
        inv = lambda: D._call("wx_iswpt1d", "_f64", A(xw).ptr, A(xh).ptr, n, L, -1, B, qp, Fq, A(x).stream())
        fb = es * n * B * (1 + (1 << L))
        check = lambda: float((xh - x).abs().max() / x.abs().max())
        return fwd, inv, check, dict(fwd_bytes=fb, inv_bytes=fb, fwd_flops=((1 << L) - 1) * 4.0 * F * n * B,
                                     samples=n * B, bound="hbm", keep=(x, xw, xh, q))
    if kind == "wpt2d":
        m, n = w["m"], w["n"]
        x = wx.jl_empty((m, n, B), td, dev); x.normal_(generator=gen)
        y = wx.jl_empty((m, n, B), td, dev)
        xh = wx.jl_empty((m, n, B), td, dev)
        fwd = lambda: D._wpt_batched("wx_wpt", A(x), A(y), 2, wt, L, None)
        inv = lambda: D._wpt_batched("wx_iwpt", A(y), A(xh), 2, wt, L, None)
        fb = es * m * n * B * 2
        check = lambda: float((xh - x).abs().max() / x.abs().max())
        return fwd, inv, check, dict(fwd_bytes=fb, inv_bytes=fb, fwd_flops=L * 2.0 * (2 * F * m * n) * B,
                                     samples=m * n * B, bound="hbm", keep=(x, y, xh))
    if kind == "wpd_bb":
        from waveletsext_jl_amd import bestbasis as bbm
        n = w["n"]
        x = wx.jl_empty((n, B), td, dev); x.normal_(generator=gen)
        xw = wx.jl_empty((n, L + 1, B), td, dev)
        qq, qp, Fq = qmf_arg(wt)
        state = {}
        method = wx.BB()

        def inv():      # the transform leg (named inv only because the harness times two legs)
            D._call("wx_wpd1d", "_f64", A(x).ptr, A(xw).ptr, n, L, B, qp, Fq, A(x).stream())

        def fwd():      # the best-basis leg: costs of every node of every signal + all trees, on the device
            costs = bbm._bb_costs(A(xw), method, True)
            state["trees"] = bbm._bb_trees(costs, (n,), B)

        inv()

        def check():
            t = state["trees"][:4].cpu().numpy().astype(bool)
            return 0.0 if all(wx.isvalidtree(torch.empty(n), t[i]) for i in range(4)) else 1.0
        ncost = (1 << (L + 1)) - 1
        return fwd, inv, check, dict(fwd_bytes=es * (n * (L + 1) + ncost) * B + (n - 1) * B, inv_bytes=es * n * (L + 2) * B,
                                     fwd_flops=4.0 * n * (L + 1) * B, samples=n * B, bound="hbm", keep=(x, xw))
    if kind == "wpd_ldb":
        n = w["n"]
        x = wx.jl_empty((n, B), td, dev); x.normal_(generator=gen)
        xw = wx.jl_empty((n, L + 1, B), td, dev)
        qq, qp, Fq = qmf_arg(wt)
        labels = [i % 4 for i in range(B)]
        state = {}

        def inv():      # the transform leg
            D._call("wx_wpd1d", "_f64", A(x).ptr, A(xw).ptr, n, L, B, qp, Fq, A(x).stream())

        def fwd():      # class energy maps of the whole table
            state["G"] = wx.energy_map(xw, labels)

        inv()

        def check():
            g = state["G"]
            s = float(g[:, 0, :].sum().item())                       # root column: energies sum to 1 per class
            return abs(s - 4.0) / 4.0
        return fwd, inv, check, dict(fwd_bytes=es * (n * (L + 1)) * (B + 4), inv_bytes=es * n * (L + 2) * B,
                                     fwd_flops=2.0 * n * (L + 1) * B, samples=n * B, bound="hbm", keep=(x, xw))
    if kind == "siwt":
        n, d = w["n"], w["d"]
        x = wx.jl_empty((n, B), td, dev); x.normal_(generator=gen)
        state = {}

        def fwd():      # decomposition + the cost of every node
            state["b"] = wx.siwpdall(x, wt, L, d)

        def inv():      # best basis of every signal, then the inverse along it
            state["b"]._b.bestbasis()                  # bestbasistreeall_ without the copy of the status bytes to the host
            state["xh"] = wx.isiwpdall(state["b"])

        check = lambda: float((state["xh"] - x).abs().max() / x.abs().max())
        NS = sum(1 << min(j, d) for j in range(L + 1))
        NN = sum((1 << min(j, d)) << j for j in range(L + 1))
        # parents read once, every column written once, one cost per node (the costs of nodes of <= 256 samples
        # come out of the level that creates them; only the top depths are read a second time)
        fb = es * B * (n * (NS - (1 << min(L, d))) + n * NS + NN)
        return fwd, inv, check, dict(fwd_bytes=fb, inv_bytes=es * B * (2 * NN + 3 * n * L) + 2 * NN * B,
                                     fwd_flops=4.0 * F * (n / 2) * (NS - 1) * B, samples=n * B, bound="hbm", keep=(x,))
    if kind == "acwpd_jbb":
        n = w["n"]
        x = wx.jl_empty((n, B), td, dev); x.normal_(generator=gen)
        ncols = (1 << (L + 1)) - 1
        state = {}

        def fwd():
            state["s"], state["q"] = wx.acwpd_jbb_moments(x, wt, L)

        def inv():      # second leg of config 5 = costs + tree selection from the moments
            costs = wx.costs_from_moments(state["s"], state["q"], B, wx.JBB(redundant=True))
            state["tree"] = wx.bestbasis_treeselection(costs, n)

        check = lambda: 0.0 if wx.isvalidtree(torch.empty(n), state["tree"]) else 1.0
        # structure exploited by the kernel: odd lags only, S shared by both children
        flops = (ncols - (1 << L)) * n * (2.0 * (F // 2) * 2 + 4) * B
        return fwd, inv, check, dict(fwd_bytes=es * (n * B + 2 * n * ncols), inv_bytes=es * 2 * n * ncols,
                                     fwd_flops=flops, samples=n * B, bound="fp64", keep=(x,))
    raise ValueError(kind)


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        # started by hand as `python bench.py --gpus N`: run the same command under the launcher as a child
        # process (nothing has touched the GPU yet) and hand back its exit code
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)
    import torch
    import torch.distributed as dist
    import waveletsext_jl_amd as wx

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # WX_BENCH_BACKEND=gloo lets several ranks share one GPU (a validation mode for 1-GPU boxes: same
    # sharding, barriers, max-over-ranks timing and aggregation, no RCCL); the real run is nccl, one GPU per rank
    backend = os.environ.get("WX_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    w = dict(WORKLOADS[a.workload])
    if a.batch:
        w["batch"] = a.batch
    fwd, inv, check, info = make_workload(w, wx, torch, dev, rank)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        fwd()
        inv()
    sync()
    err = check()
    tol = 1e-10 if w["dtype"] == "f64" else 1e-5
    assert err < tol, "round trip broken: %g" % err

    # HIP events on the launch stream (torch's current stream == the stream passed to the C ABI)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(a.steps)]
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        ev[i][0].record()
        fwd()
        ev[i][1].record()
        inv()
        ev[i][2].record()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    fwd_ms = sorted(e[0].elapsed_time(e[1]) for e in ev)
    inv_ms = sorted(e[1].elapsed_time(e[2]) for e in ev)
    fwd_avg = sum(fwd_ms) / len(fwd_ms)
    inv_avg = sum(inv_ms) / len(inv_ms)

    gather_ms = None
    if a.gather and world > 1 and w["kind"] in ("wpd", "wpt"):
        from waveletsext_jl_amd import distributed as wd
        xh = info["keep"][2]
        wd.allgather_batch(xh, world * w["batch"])
        sync()
        t1 = time.perf_counter()
        wd.allgather_batch(xh, world * w["batch"])
        sync()
        gather_ms = (time.perf_counter() - t1) * 1e3

    out = None
    if rank == 0:
        # HBM traffic of one forward pass = sum over its kernels (launches per pass x PMC bytes per launch)
        # from separate rocprofv3 --pmc passes of this same command (profiles/traffic.json, written by
        # tools/collect_evidence.py); null if not profiled.  `kernel` names the dominant one.
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(a.workload, {})
            if not a.batch:
                traffic = sum(cnt * tj[name]["hbm_bytes_per_launch"] for name, cnt in w["fwd_kernels"])
        except Exception:
            traffic = None
        fb = float(info["fwd_bytes"])
        if info["bound"] == "hbm":
            achieved = fb / (fwd_avg * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": w["kernel"], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": fb}
        else:
            achieved = info["fwd_flops"] / (fwd_avg * 1e-3) / 1e12
            roof = {"bound": "fp64-valu (no MFMA on this path)", "kernel": w["kernel"], "achieved": achieved,
                    "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic,
                    "algorithmic_flops_per_step": info["fwd_flops"], "hbm_GBs": fb / (fwd_avg * 1e-3) / 1e9}
        roof.update({"avg_launch_ms": fwd_avg, "median_launch_ms": fwd_ms[len(fwd_ms) // 2],
                     "fwd_TFLOPs": info["fwd_flops"] / (fwd_avg * 1e-3) / 1e12})
        cfg = {"workload": w["desc"], "batch_per_gpu": w["batch"], "wavelet": w["wavelet"], "L": w["L"],
               "sharding": "batch split across ranks, no data-path collective"}
        cfg.update({k: w[k] for k in ("n", "m") if k in w})
        out = {
            "metric": "Msamples/s (fwd+inv wavelet packets)",
            "value": world * info["samples"] * a.steps / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": w["dtype"], "data": "synthetic N(0,1), seed 1002+rank, resident in HBM",
            "config": cfg,
            "roofline": roof,
            "inverse": {"avg_launch_ms": inv_avg, "algorithmic_bytes_per_launch": float(info["inv_bytes"]),
                        "achieved_GBs": info["inv_bytes"] / (inv_avg * 1e-3) / 1e9},
            "roundtrip_rel_err": err,
        }
        if gather_ms is not None:
            out["allgather_reconstructed_ms"] = gather_ms
        if not a.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(w, a.cpu_seconds)
        elif not a.no_cpu:
            out["cpu_baseline"] = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
