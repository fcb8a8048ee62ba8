"""GPU stress: geometry limits rather than values -- batches beyond the 65535 grid limit, the longest signals
of the fused and of the per-level paths, big images, many trees at once.  Checked through size-independent
properties (round trip, energy conservation, batch == repeated single) so that nothing has to run on the CPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    import torch
    return float((a - b).abs().max() / b.abs().max())


def test_batches_beyond_the_grid_limit(wx, torch_mod=None):
    import torch
    wt = wx.wavelet(wx.WT.db2)
    B = 70001                                                         # > 65535 (gridDim.y) and odd
    x = wx.jl_empty((64, B), torch.float64, "cuda"); x.normal_()
    assert _rel(wx.iwpdall(wx.wpdall(x, wt), wt), x) <= 1e-12
    assert _rel(wx.iwptall(wx.wptall(x, wt), wt), x) <= 1e-12
    assert _rel(wx.iswptall(wx.swptall(x, wt, 3), wt), x) <= 1e-12
    assert _rel(wx.isdwtall(wx.sdwtall(x, wt, 3), wt, 5), x) <= 1e-12
    assert _rel(wx.iacwpdall(wx.acwpdall(x, wt, 3), 3), x) <= 1e-12
    xw = wx.wpdall(x, wt)
    trees = wx.bestbasistreeall(xw, wx.BB())
    assert trees.shape == (63, B)
    one = wx.bestbasistree(xw[:, :, B - 1], wx.BB())
    assert (trees[:, B - 1] == one).all()
    y = wx.denoiseall(x, "sig", wt)
    assert tuple(y.shape) == (64, B) and bool(torch.isfinite(y).all())
    img = wx.jl_empty((8, 8, B), torch.float32, "cuda"); img.normal_()
    assert _rel(wx.iwptall(wx.wptall(img, wt, 2), wt, 2), img) <= 1e-5


@pytest.mark.parametrize("n,dtype", [(8192, "f64"), (16384, "f64"), (32768, "f32"), (65536, "f32")])
def test_long_signals(wx, n, dtype):
    import torch
    td = torch.float64 if dtype == "f64" else torch.float32
    tol = 1e-12 if dtype == "f64" else 2e-5
    wt = wx.wavelet(wx.WT.db4)
    x = wx.jl_empty((n, 9), td, "cuda"); x.normal_()
    L = 10
    xw = wx.wpdall(x, wt, L)
    e0 = (x.double() ** 2).sum(0)
    for lvl in (1, L):
        assert float(((xw[:, lvl, :].double() ** 2).sum(0) / e0 - 1).abs().max()) <= 100 * tol     # orthogonality per level
    assert _rel(wx.iwpdall(xw, wt, L), x) <= 10 * tol
    assert _rel(wx.wptall(x, wt, L), xw[:, L, :]) <= tol
    if n <= 16384:
        sp = wx.swptall(x, wt, 4)
        assert _rel(wx.iswptall(sp, wt), x) <= 10 * tol


def test_large_images_and_deep_quadtrees(wx):
    import torch
    wt = wx.wavelet(wx.WT.db2)
    img = wx.jl_empty((1024, 2048, 3), torch.float32, "cuda"); img.normal_()
    y = wx.wptall(img, wt, 7)
    e = float((y.double() ** 2).sum() / (img.double() ** 2).sum())
    assert abs(e - 1) <= 1e-4
    assert _rel(wx.iwptall(y, wt, 7), img) <= 1e-4
    small = wx.jl_empty((256, 256, 5), torch.float64, "cuda"); small.normal_()
    xw = wx.wpdall(small, wt, 5)
    assert _rel(wx.iwpdall(xw, wt, 5), small) <= 1e-11
    trees = wx.bestbasistreeall(xw, wx.BB())
    assert trees.shape[1] == 5 and all(wx.isvalidtree(np.empty((256, 256)), trees[:, i]) for i in range(5))


def test_concurrent_host_threads_on_their_own_streams(wx):
    """the C ABI is re-entrant: four host threads, each on its own HIP stream, run different families at the same
    time (ctypes drops the GIL inside the calls) and get the same results as a serial run"""
    import threading
    import torch
    wt = wx.wavelet(wx.WT.db4)
    rng = torch.Generator(device="cuda"); rng.manual_seed(7)
    xs = []
    for i in range(4):
        x = wx.jl_empty((1024, 257 + i), torch.float64, "cuda"); x.normal_(generator=rng); xs.append(x)

    def work(i, out):
        x = xs[i]
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            res = []
            for _ in range(3):
                res = [wx.iwpdall(wx.wpdall(x, wt), wt), wx.wptall(x, wt, 7), wx.iswptall(wx.swptall(x, wt, 4), wt),
                       wx.bestbasistree(wx.wpdall(x, wt), wx.JBB()), wx.bestbasistreeall(wx.wpdall(x, wt), wx.BB()),
                       wx.denoiseall(x, "sig", wt)]
            s.synchronize()
        out[i] = res

    serial, par = {}, {}
    for i in range(4):
        work(i, serial)
    ts = [threading.Thread(target=work, args=(i, par)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for i in range(4):
        for a, b in zip(serial[i], par[i]):
            if isinstance(a, np.ndarray):
                assert (a == b).all()
            else:
                assert torch.equal(a, b)


def _run_bench(args, env_extra, timeout=900):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    import json
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.parametrize("workload,batch", [("cfg2", 37), ("target", 64), ("cfg4", 6), ("cfg3", 12), ("swpt_db4", 40),
                                            ("target_n2048", 96), ("target_haar", 70)])
def test_bench_two_ranks_gather_equals_one_rank(tmp_path, workload, batch):
    """bench.py's N > 1 path in its validation mode (two fresh child processes share this GPU, gloo instead of RCCL):
    strong scaling of a small batch, the all-gather of the reconstructed output inside the step, and the gathered array
    of two ranks equal to what one rank reconstructs (dwt/dwt_all.jl:277-279 shards over the signal axis)"""
    import numpy as np
    one = str(tmp_path / "one.npy")
    two = str(tmp_path / "two.npy")
    common = ["--workload", workload, "--batch", str(batch), "--steps", "2", "--warmup", "1", "--no-cpu"]
    j1 = _run_bench(common + ["--dump", one], {})
    j2 = _run_bench(common + ["--gpus", "2", "--scaling", "strong", "--chunks", "4", "--dump", two],
                    {"WX_BENCH_BACKEND": "gloo"})
    assert j1["n_gpus"] == 1 and j2["n_gpus"] == 2 and j2["scaling"] == "strong"
    assert j2["ranks"]["nranks"] == 2 and len(j2["ranks"]["per_rank_ms"]) == 2
    ga = j2["with_allgather"]
    assert ga["chunks"] == 4 and ga["ms_per_step"] > 0 and "allgather_alone_ms" in ga and "overlap_ms" in ga
    a, b = np.load(one), np.load(two)
    assert a.shape == b.shape and a.shape[-1] == batch
    assert np.array_equal(a, b)


@pytest.mark.parametrize("workload,batch", [("cfg2", 43), ("cfg4", 11), ("target", 75)])
def test_bench_eight_ranks_ragged_shards(tmp_path, workload, batch):
    """the N = 8 geometry the driver will launch, on one GPU in the gloo validation mode (eight fresh child processes):
    batches that 8 does not divide (shards of 6 and 5, of 2 and 1, of 10 and 9 signals), eight ranks reported, the gathered
    reconstruction of eight ranks bit-identical to one rank's (dwt/dwt_all.jl:277-279)"""
    import numpy as np
    one = str(tmp_path / "one.npy")
    eight = str(tmp_path / "eight.npy")
    common = ["--workload", workload, "--batch", str(batch), "--steps", "2", "--warmup", "1", "--no-cpu"]
    _run_bench(common + ["--dump", one], {})
    j8 = _run_bench(common + ["--gpus", "8", "--scaling", "strong", "--chunks", "4", "--dump", eight],
                    {"WX_BENCH_BACKEND": "gloo"}, timeout=1500)
    assert j8["n_gpus"] == 8 and j8["scaling"] == "strong"
    assert j8["ranks"]["nranks"] == 8 and len(j8["ranks"]["per_rank_ms"]) == 8
    assert j8["with_allgather"]["ms_per_step"] > 0 and j8["with_allgather_value"] == j8["with_allgather"]["value"]
    a, b = np.load(one), np.load(eight)
    assert a.shape == b.shape and a.shape[-1] == batch
    assert np.array_equal(a, b)


def test_bench_config5_eight_ranks_same_tree(tmp_path):
    """config 5 on eight ranks with a batch 8 does not divide: the all-reduced moments give the tree of one rank
    (bestbasis/bestbasis_tree.jl:153-154)"""
    import numpy as np
    one = str(tmp_path / "one.npy")
    eight = str(tmp_path / "eight.npy")
    common = ["--workload", "cfg5", "--batch", "4103", "--steps", "1", "--warmup", "1", "--no-cpu"]
    _run_bench(common + ["--dump", one], {})
    j8 = _run_bench(common + ["--gpus", "8", "--dump", eight], {"WX_BENCH_BACKEND": "gloo"}, timeout=1500)
    assert j8["ranks"]["nranks"] == 8 and "all-reduce" in j8["config"]["collective"]
    assert np.array_equal(np.load(one), np.load(eight))


def test_bench_config5_two_ranks_same_tree(tmp_path):
    """config 5 sharded: moments per shard (accumulated over chunks), the all-reduce inside the step, the same tree on
    two ranks as on one (bestbasis/bestbasis_tree.jl:153-154 sums over the signal axis)"""
    import numpy as np
    one = str(tmp_path / "one.npy")
    two = str(tmp_path / "two.npy")
    common = ["--workload", "cfg5", "--batch", "5000", "--steps", "1", "--warmup", "1", "--no-cpu"]
    _run_bench(common + ["--dump", one], {})
    j2 = _run_bench(common + ["--gpus", "2", "--dump", two], {"WX_BENCH_BACKEND": "gloo"})
    assert j2["roofline"]["launches_per_step"] == 1 and "all-reduce" in j2["config"]["collective"]
    assert np.array_equal(np.load(one), np.load(two))


def test_host_arrays_through_the_pinned_staging_ring(wx):
    """numpy in / numpy out: the D2H of a result larger than the staging threshold runs through the pinned ring and the
    host thread pool (wx_host.hip), in chunks of 32 MiB with a ragged tail -- the same numbers as the device-tensor
    path, bit for bit"""
    import torch
    wt = wx.wavelet(wx.WT.db4)
    rng = np.random.default_rng(21)
    n, B, L = 4096, 333, 9                       # (n, L+1, B) Float64 = 109 MB: three full chunks and a tail
    x = np.asfortranarray(rng.standard_normal((n, B)))
    y_host = wx.wpdall(x, wt, L)
    y_dev = wx.wpdall(wx.to_device(x), wt, L)
    assert isinstance(y_host, np.ndarray) and y_host.flags.f_contiguous
    assert np.array_equal(y_host, y_dev.cpu().numpy())
    xr = wx.iwpdall(y_host, wt, L)
    assert np.abs(xr - x).max() <= 1e-12
    for _ in range(3):                           # the ring and the pool are reused from call to call
        assert np.array_equal(wx.wpdall(x, wt, L), y_host)
    wx.shutdown()                                # releases the ring and the pool; the next call rebuilds them
    assert np.array_equal(wx.wpdall(x, wt, L), y_host)
    del y_dev
    torch.cuda.empty_cache()
