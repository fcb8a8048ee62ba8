"""Throughput floor of the 1-D decimated transforms: every dyadic length 64 ... 65536, depths 1 / 4 / full, full trees, pyramids
and a random tree, Float64 and Float32, 1 GiB batches, both directions -- as a fraction of the 8 TB/s HBM peak on the algorithmic
bytes (signal read once + written once).  tests/test_gpu_perf_floor.py asserts the floor and writes the table
(profiles/r04_floor.txt); `python tools/floor_scan.py [wavelet]` prints it.

Timing: HIP events around 5 back-to-back calls, the best of 3 such batches, after 2 warm-up calls (the first call of a shape pays a
61 ms hipMalloc of its 1-2 GiB output inside torch's caching allocator -- that, not a kernel, was the '9.8 ms idwtall' of round 3's
table, which timed one warm-up and averaged)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

HBM_PEAK = 8.0e12


def timed(torch, fn, calls=5, batches=3):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(batches):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(calls):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / calls)
    return best


def scan(wname="db4", lengths=None, out=None, dtypes=("f64", "f32")):
    import numpy as np
    import torch
    import waveletsext_jl_amd as wx
    from helpers import random_tree_1d
    wt = wx.wavelet(getattr(wx.WT, wname))
    rows = []
    for dt, esz, dn in ((torch.float64, 8, "f64"), (torch.float32, 4, "f32")):
        if dn not in dtypes:
            continue
        for n in lengths or [1 << k for k in range(6, 17)]:
            B = (1 << 30) // (n * esz)
            x = wx.jl_empty((n, B), dt, "cuda")
            x.normal_()
            Lmax = wx.maxtransformlevels(n)
            gb = 2.0 * n * B * esz
            cases = []
            for L in sorted({1, min(4, Lmax), Lmax}):
                cases.append(("full tree L=%d" % L, lambda a, L=L: wx.wptall(a, wt, L), lambda a, L=L: wx.iwptall(a, wt, L)))
                cases.append(("pyramid L=%d" % L, lambda a, L=L: wx.dwtall(a, wt, L), lambda a, L=L: wx.idwtall(a, wt, L)))
            rng = np.random.default_rng(n)
            tree = random_tree_1d(n, rng, p=0.7)
            while not (tree[0] and tree[1:3].any() and tree[3:7].any()):          # a tree of at least three levels
                tree = random_tree_1d(n, rng, p=0.7)
            cases.append(("random tree p=0.7", lambda a: wx.wptall(a, wt, tree), lambda a: wx.iwptall(a, wt, tree)))
            for name, fwd, inv in cases:
                tf = timed(torch, lambda: fwd(x))
                y = fwd(x)
                ti = timed(torch, lambda: inv(y))
                err = float((inv(y) - x).abs().max() / x.abs().max())
                rows.append(dict(dtype=dn, n=n, case=name, fwd_ms=tf, inv_ms=ti, fwd_frac=gb / (tf * 1e-3) / HBM_PEAK,
                                 inv_frac=gb / (ti * 1e-3) / HBM_PEAK, roundtrip=err))
                line = "%s n %6d %-18s fwd %6.3f ms (%4.1f %%)  inv %6.3f ms (%4.1f %%)  rt %.0e" % (
                    dn, n, name, tf, 100 * rows[-1]["fwd_frac"], ti, 100 * rows[-1]["inv_frac"], err)
                print(line, flush=True)
                if out is not None:
                    out.write(line + "\n")
                del y
            del x
            torch.cuda.empty_cache()
    return rows


if __name__ == "__main__":
    # python tools/floor_scan.py [wavelet] [f64|f32|both] [n ...]
    dts = ("f64", "f32") if len(sys.argv) < 3 or sys.argv[2] == "both" else (sys.argv[2],)
    scan(sys.argv[1] if len(sys.argv) > 1 else "db4", [int(v) for v in sys.argv[3:]] or None, dtypes=dts)
