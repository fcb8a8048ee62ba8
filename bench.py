#!/usr/bin/env python3
"""bench.py -- headline benchmark of the batched wavelet-packet hot path on MI355X.

Metric (BASELINE.json): Msamples/s for forward + inverse wavelet packets, with the dominant
kernel's achieved HBM bandwidth against the 8 TB/s roofline, next to the CPU path.

A step = one forward pass + one inverse pass over one batch of synthetic signals that are already
resident in HBM.  Default workload = BASELINE config 2 (`wpdall` 65536 x 4096 Float64, db8, full
packet tree L=12, then `iwpdall`).  `--workload target` runs the north-star target configuration
(`wptall`/`iwptall`, db4, L=10).  N > 1: one process per GPU (torchrun), each rank owns a
fixed-size shard of the batch (weak scaling); the transforms need no data-path collective.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling 6290

WORKLOADS = {
    # name: (n, per-GPU batch, wavelet, L, kind)
    "cfg2": dict(n=4096, batch=65536, wavelet="db8", L=12, kind="wpd",
                 desc="BASELINE config 2: wpdall+iwpdall 65536x4096 f64 db8 full tree L=12"),
    "target": dict(n=4096, batch=65536, wavelet="db4", L=10, kind="wpt",
                   desc="north-star target: wptall+iwptall 65536x4096 f64 db4 L=10"),
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


def cpu_baseline(w, seconds):
    """The oracle ("port": scalar restatement of wpdall -> wpd! -> dwt_step!, one thread, like the
    reference) timed on this host on a bounded sample of the same workload."""
    import ctypes
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import wx_oracle as wo
    import waveletsext_jl_amd as wx
    lib = wo.lib()
    n, L = w["n"], w["L"]
    q = np.ascontiguousarray(wx.wavelet(getattr(wx.WT, w["wavelet"])).qmf)
    tree = wo.maketree1d(n, L, "full").astype(np.uint8)
    rng = np.random.default_rng(1002)

    def run(B):
        x = np.asfortranarray(rng.standard_normal((n, B)))
        xh = np.empty_like(x)
        P = lambda a: ctypes.c_void_p(a.ctypes.data)
        I, L64 = ctypes.c_int, ctypes.c_int64
        t0 = time.perf_counter()
        if w["kind"] == "wpd":
            y = np.empty((n, L + 1, B), order="F")
            lib.wxo_wpdall1d_f64(P(y), P(x), L64(n), I(L), L64(B), P(q), I(q.size))
            lib.wxo_iwpdall1d_f64(P(xh), P(y), L64(n), I(L + 1), L64(B), P(tree), L64(tree.size), P(q), I(q.size))
        else:
            y = np.empty_like(x)
            lib.wxo_wptall1d_f64(P(y), P(x), L64(n), L64(B), P(tree), L64(tree.size), P(q), I(q.size))
            lib.wxo_iwptall1d_f64(P(xh), P(y), L64(n), L64(B), P(tree), L64(tree.size), P(q), I(q.size))
        dt = time.perf_counter() - t0
        assert np.abs(xh - x).max() < 1e-9
        return dt
    t_probe = run(16)
    B = int(max(16, min(8192, seconds / (t_probe / 16))))
    dt = run(B)
    return {"value": B * n / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "%d of the %d signals (n=%d), fwd+inv, oracle C -O2 single thread, %.1f s"
                      % (B, w["batch"], n, dt)}


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    import waveletsext_jl_amd as wx

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    w = dict(WORKLOADS[a.workload])
    if a.batch:
        w["batch"] = a.batch
    n, B, L = w["n"], w["batch"], w["L"]
    wt = wx.wavelet(getattr(wx.WT, w["wavelet"]))

    gen = torch.Generator(device=dev).manual_seed(1002 + rank)
    x = wx.jl_empty((n, B), torch.float64, dev)
    x.normal_(generator=gen)

    if w["kind"] == "wpd":
        y = wx.jl_empty((n, L + 1, B), torch.float64, dev)
        xh = wx.jl_empty((n, B), torch.float64, dev)
        fwd = lambda: wx.dwt._wpd_batched(wx.dwt.Arg(x), wx.dwt.Arg(y), 1, wt, L)
        inv = lambda: wx.dwt._iwpd_batched(wx.dwt.Arg(y), wx.dwt.Arg(xh), 1, wt, L, None)
        fwd_bytes = 8.0 * n * B * (1 + L + 1)            # x read once + (L+1) columns written once
        inv_bytes = 8.0 * n * B * 2                      # leaf column read + x written
        kernel = "k_fwd1d_fused<double, 16, 512, true, 2>"
    else:
        y = wx.jl_empty((n, B), torch.float64, dev)
        xh = wx.jl_empty((n, B), torch.float64, dev)
        fwd = lambda: wx.dwt._wpt_batched("wx_wpt", wx.dwt.Arg(x), wx.dwt.Arg(y), 1, wt, L, None)
        inv = lambda: wx.dwt._wpt_batched("wx_iwpt", wx.dwt.Arg(y), wx.dwt.Arg(xh), 1, wt, L, None)
        fwd_bytes = 8.0 * n * B * 2
        inv_bytes = 8.0 * n * B * 2
        kernel = "k_fwd1d_inplace<double, 8, 256, false>"

    def step():
        fwd()
        inv()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    sync()
    err = float((xh - x).abs().max() / x.abs().max())
    assert err < 1e-10, "round trip broken: %g" % err

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
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    fwd_ms = sorted(e[0].elapsed_time(e[1]) for e in ev)
    inv_ms = sorted(e[1].elapsed_time(e[2]) for e in ev)
    fwd_avg = sum(fwd_ms) / len(fwd_ms)
    inv_avg = sum(inv_ms) / len(inv_ms)

    gather_ms = None
    if a.gather and world > 1:
        full = torch.empty((world * B, n), dtype=torch.float64, device=dev)
        src = xh.T if xh.dim() > 1 else xh
        dist.all_gather_into_tensor(full, src.contiguous())
        sync()
        t1 = time.perf_counter()
        dist.all_gather_into_tensor(full, src.contiguous())
        sync()
        gather_ms = (time.perf_counter() - t1) * 1e3

    out = None
    if rank == 0:
        # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes of this same
        # command (profiles/traffic.json, written by tools/summarize_prof.py); null if not profiled.
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if not a.batch:
                traffic = tj.get(a.workload, {}).get(kernel, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
        achieved = fwd_bytes / (fwd_avg * 1e-3) / 1e9
        out = {
            "metric": "Msamples/s (fwd+inv wavelet packets)",
            "value": world * B * n * a.steps / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic N(0,1), seed 1002+rank, resident in HBM",
            "config": {"workload": w["desc"], "n": n, "batch_per_gpu": B, "wavelet": w["wavelet"], "L": L,
                       "sharding": "batch split across ranks, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": fwd_bytes, "avg_launch_ms": fwd_avg,
                         "median_launch_ms": fwd_ms[len(fwd_ms) // 2]},
            "inverse": {"avg_launch_ms": inv_avg, "algorithmic_bytes_per_launch": inv_bytes,
                        "achieved_GBs": inv_bytes / (inv_avg * 1e-3) / 1e9},
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
