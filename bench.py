#!/usr/bin/env python3
"""bench.py -- headline benchmark of the batched wavelet-packet hot path on MI355X.

Metric (BASELINE.json): Msamples/s for forward + inverse wavelet packets, with the dominant kernel's achieved HBM
bandwidth against the 8 TB/s roofline, next to the CPU path timed on this host.

A step = one forward pass + one inverse pass over the whole batch of the configuration, synthetic signals already
resident in HBM.  Default workload = BASELINE config 2 (`wpdall` 65536 x 4096 Float64, db8, full packet tree L = 12, then
`iwpdall`).  Other workloads (`--workload`): `target` (north-star target: wptall / iwptall db4 L = 10), `cfg3` (swptall /
iswptall 8192 signals of 16384 samples, haar, L = 12: the 4 TiB of leaves exist 64 signals at a time, a step loops over
all chunks of the rank's shard), `cfg3_sdwt` (the same signals through sdwtall / isdwtall), `dwt_long` (dwtall / idwtall of 16384-sample signals), `cfg4` (2-D wptall / iwptall 4096 images 512 x 512 Float32 db4 L = 6), `cfg5` (acwpd + JBB
moments / costs / tree, 262144 signals of 2048 samples, coif6, L = 11: moments accumulate over chunks of 2048 signals, one
all-reduce of the moments when N > 1, costs and tree on every rank), plus the widened rows `bb`, `ldb`, `siwt`.

N > 1: one process per GPU (torchrun).  `--scaling strong` (default): the configuration's batch is split contiguously
over the ranks (distributed.shard_range) -- "the sharded batch" of the north star; `--scaling weak`: every rank gets the
whole configuration batch.  The transforms need no data-path collective; `value` is the sharded compute.  The one
exchange the path has for these workloads -- the all-gather of the reconstructed output -- is timed as a second loop
with the exchange INSIDE the step (the inverse runs in >= 4 chunks, each chunk's all-gather on a side stream overlaps the
next chunk's inverse) and reported next to it (`with_allgather`); for cfg5 the all-reduce of the moments is part of the
algorithm and always inside the step.

Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling 6290
FP64_PEAK_TFLOPS = 78.6        # FP64 vector spec; tools/ubench.hip measures 55-61 on this part

WORKLOADS = {
    "cfg2": dict(inv_bound="fp64 issue: an all-VALU Float64 lattice of 8 rotation stages x 12 levels (63 % of the FP64 vector peak on the lattice's own flop count, profiles/r04_fp64_issue.txt); HBM is the roofline the fraction is quoted against",
                 kind="wpd", n=4096, batch=65536, wavelet="db8", L=12, dtype="f64",
                 kernel="k_lat_wpd_f64<8, 2>", inv_kernel="k_lat_iwpt_f64<8, 2, double>",
                 fwd_kernels=[("k_lat_wpd_f64<8, 2>", 1)],
                 desc="BASELINE config 2: wpdall+iwpdall 65536x4096 f64 db8 full tree L=12"),
    "target": dict(kind="wpt", n=4096, batch=65536, wavelet="db4", L=10, dtype="f64",
                   kernel="k_lat_wpt_f64<4, 3, double>", inv_kernel="k_lat_iwpt_f64<4, 2, double>",
                   fwd_kernels=[("k_lat_wpt_f64<4, 3, double>", 1)],
                   desc="north-star target: wptall+iwptall 65536x4096 f64 db4 L=10"),
    "wpt_db8": dict(inv_bound="fp64 issue (as cfg2's inverse: profiles/r04_fp64_issue.txt)", kind="wpt", n=4096, batch=65536, wavelet="db8", L=12, dtype="f64",
                    kernel="k_lat_wpt_f64<8, 3, double>", inv_kernel="k_lat_iwpt_f64<8, 2, double>",
                    fwd_kernels=[("k_lat_wpt_f64<8, 3, double>", 1)],
                    desc="config 2's signals and filter through wptall+iwptall: 65536x4096 f64 db8 L=12 (the F = 16 lattice kernels)"),
    "tree_random": dict(kind="wpt", n=4096, batch=65536, wavelet="db4", L=12, dtype="f64", tree="random:0.7:3",
                        kernel="k_lat_wpt_treesc_f64<4, 2, 0, double>", inv_kernel="k_lat_iwpt_treesc_f64<4, 2, 0, false, double>",
                        fwd_kernels=[("k_lat_wpt_treesc_f64<4, 2, 0, double>", 1)],
                        desc="the target's batch along a tree, as bestbasistree output is used (DWT.jl:340-351, dwt_all.jl:152-225): "
                             "wptall+iwptall 65536x4096 f64 db4, random tree (every node split with probability 0.7, seed 3, depth 12)"),
    "tree_pyramid": dict(kind="wpt", n=4096, batch=65536, wavelet="db4", L=12, dtype="f64", tree="pyramid",
                         kernel="k_lat_wpt_treesc_f64<4, 2, 0, double>", inv_kernel="k_lat_iwpt_treesc_f64<4, 2, 0, false, double>",
                         fwd_kernels=[("k_lat_wpt_treesc_f64<4, 2, 0, double>", 1)],
                         desc="dwtall+idwtall as wptall+iwptall along maketree(4096, 12, :dwt): 65536x4096 f64 db4 (the levels "
                              "below 64 samples run lane-locally, wx_dwttail.hip)"),
    "dwt_long": dict(kind="wpt", n=16384, batch=16384, wavelet="db4", L=14, dtype="f64", tree="pyramid",
                     kernel="k_top_tile_fwd<double, 8, 2, 4096>", inv_kernel="k_top_tile_inv<double, 8, 2, 2048>",
                     fwd_kernels=[("k_top_tile_fwd<double, 8, 2, 4096>", 1), ("k_lat_wpt_treesc_f64<4, 2, 0, double>", 1)],
                     desc="dwtall+idwtall of long signals as wptall+iwptall along maketree(16384, 14, :dwt): 16384x16384 f64 db4 -- the two "
                          "top levels in one tiled pass (wx_toptile.h), the 4096-sample pyramid on the lattice kernels, the levels "
                          "below 64 samples lane-locally (wx_dev_dwt_long)"),
    "target_n2048": dict(kind="wpt", n=2048, batch=131072, wavelet="db4", L=10, dtype="f64",
                         kernel="k_lat_wpt_sh_f64<4, 2, 1>", inv_kernel="k_lat_iwpt_sh_f64<4, 2, 1>",
                         fwd_kernels=[("k_lat_wpt_sh_f64<4, 2, 1>", 1)],
                         desc="the target's byte count with 2048-sample signals: wptall+iwptall 131072x2048 f64 db4 L=10 "
                              "(two signals interleaved per wavefront, DESIGN 4.20)"),
    "target_n1024": dict(kind="wpt", n=1024, batch=262144, wavelet="db4", L=9, dtype="f64",
                         kernel="k_lat_wpt_sh_f64<4, 2, 2>", inv_kernel="k_lat_iwpt_sh_f64<4, 2, 2>",
                         fwd_kernels=[("k_lat_wpt_sh_f64<4, 2, 2>", 1)],
                         desc="the target's byte count with 1024-sample signals: wptall+iwptall 262144x1024 f64 db4 L=9 "
                              "(four signals interleaved per wavefront, DESIGN 4.20)"),
    "target_f32": dict(kind="wpt", n=4096, batch=131072, wavelet="db4", L=10, dtype="f32",
                       kernel="k_lat_wpt_g_f64<4, 2, 0, float, true>", inv_kernel="k_lat_iwpt_g_f64<4, 2, 0, float, true>",
                       fwd_kernels=[("k_lat_wpt_g_f64<4, 2, 0, float, true>", 1)],
                       desc="the target's byte count in Float32: wptall+iwptall 131072x4096 f32 db4 L=10 -- Float32 arithmetic on pairs "
                            "of signals (lat_f2v, DESIGN 4.8): a wavefront takes two signals, every rotation is one v_pk_fma_f32"),
    "target_haar": dict(kind="wpt", n=4096, batch=65536, wavelet="haar", L=10, dtype="f64",
                        kernel="k_lat_wpt_f64<1, 3, double>", inv_kernel="k_lat_iwpt_f64<1, 2, double>",
                        fwd_kernels=[("k_lat_wpt_f64<1, 3, double>", 1)],
                        desc="north-star target with the Haar filter: wptall+iwptall 65536x4096 f64 haar L=10 "
                             "(the lattice kernels with one rotation; WX_LATTICE=0 selects the Walsh-Hadamard kernels of wx_haar.hip)"),
    "cfg3": dict(kind="swpt", n=16384, batch=8192, chunk=64, wavelet="haar", L=12, dtype="f64",
                 kernel="k_haar_swpt6_fwd<1>", inv_kernel="k_haar_iswpt<5, 1>",
                 fwd_kernels=[("k_swt_fwd_multi<double, 8>", 2), ("k_haar_swpt6_fwd<1>", 1)],
                 desc="BASELINE config 3: swptall+iswptall (average-based) 8192x16384 f64 haar L=12; the leaves exist one "
                      "resident chunk of 64 signals (32 GiB) at a time, a step loops over every chunk of the shard"),
    "cfg3_sdwt": dict(kind="sdwt", n=16384, batch=8192, wavelet="haar", L=12, dtype="f64",
                      kernel="k_sdwt_fused_ip<double, false, 16, 2>", inv_kernel="k_isdwt_avg_fused_ip<double, 16>",
                      fwd_kernels=[("k_sdwt_fused_ip<double, false, 16, 2>", 1)],
                      desc="BASELINE config 3 read as the non-packet transform (SURVEY 8d: report both): sdwtall + isdwtall "
                           "(average-based) 8192x16384 f64 haar L=12, output (16384, 13, 8192) = 13 GiB"),
    "swpt_db4": dict(kind="swpt", n=1024, batch=16384, chunk=2048, wavelet="db4", L=10, dtype="f64",
                     kernel="k_swpt_deep_fwd<8, false, 4>", inv_kernel="k_swpt_deep_inv<8, 4>",
                     fwd_kernels=[("k_swpt_deep_fwd<8, false, 4>", 1)],
                     desc="the redundant packet transform with a general filter at full depth: swptall+iswptall (average-based) "
                          "16384x1024 f64 db4 L=10 in resident chunks of 2048 signals (16 GiB of leaves each); the last four "
                          "levels are the lane-local kernels of DESIGN 4.21 (`traffic` is that kernel's)"),
    "cfg4": dict(kind="wpt2d", m=512, n=512, batch=4096, wavelet="db4", L=6, dtype="f32",
                 kernel="k_lat2d_fused_f32<4, false, 0, false>", inv_kernel="k_lat2d_fused_f32<4, true, 0, false>",
                 fwd_kernels=[("k_lat2d_fused_f32<4, false, 0, false>", 1)],
                 desc="BASELINE config 4: 2-D wptall+iwptall 4096 images 512x512 f32 db4 L=6 -- both passes of a transform in one persistent "
                      "launch, the intermediate image in a 192 MiB ring that stays in the Infinity Cache (round 6)"),
    "cfg4_256": dict(kind="wpt2d", m=256, n=256, batch=16384, wavelet="db4", L=5, dtype="f32",
                     kernel="k_lat2d_fused_f32<4, false, 1, false>", inv_kernel="k_lat2d_fused_f32<4, true, 1, false>",
                     fwd_kernels=[("k_lat2d_fused_f32<4, false, 1, false>", 1)],
                     desc="config 4's bytes as 16384 images 256x256 f32 db4 L=5 (two images per register column)"),
    "cfg4_1024": dict(kind="wpt2d", m=1024, n=1024, batch=1024, wavelet="db4", L=7, dtype="f32",
                      kernel="k_lat2d_fused_f32<4, false, 2, false>", inv_kernel="k_lat2d_fused_f32<4, true, 2, false>",
                      fwd_kernels=[("k_lat2d_fused_f32<4, false, 2, false>", 1)],
                      desc="config 4's bytes as 1024 images 1024x1024 f32 db4 L=7 (eight columns per wavefront)"),
    "cfg5": dict(kind="acwpd_jbb", n=2048, batch=262144, chunk=2048, wavelet="coif6", L=11, dtype="f64",
                 kernel="k_acwpd_subtree_mfma<2>",
                 fwd_kernels=[("k_acwpd_top_two_mom<2>", 3), ("k_acwpd_top_combine", 1),
                              ("k_acwpd_subtree_mfma<2>", 1), ("k_jbb_costs<double>", 1)],
                 desc="BASELINE config 5: acwpd + JBB moments/costs/tree 262144x2048 f64 coif6 L=11; moments accumulate over "
                      "chunks of 2048 signals, all-reduce of the moments when N > 1 (no inverse: the output is the tree)"),
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
    "denoise": dict(kind="denoise", n=4096, batch=65536, wavelet="db4", L=12, dtype="f64",
                    kernel="k_lat_denoise_f64<4, 3, 0>", inv_kernel="k_lat_iwpt_treesc_f64<4, 2, 0, true",
                    fwd_kernels=[("k_lat_denoise_f64<4, 3, 0>", 1)],
                    desc="SURVEY 8(f) row 1: denoiseall(x, :sig, wt) = dwtall -> noisest per signal -> threshold -> idwtall of 65536x4096 f64 db4 "
                         "L=12 (VisuShrink, HardTH); first leg = the one-pass kernel behind wx_denoiseall_sig_* (signal in, denoised signal out), "
                         "second leg = the same result from the separate steps dwtall, then denoiseall(:dwt) (noisest + threshold on the loads of idwtall)"),
    "siwt": dict(kind="siwt", n=1024, batch=4096, wavelet="db4", L=10, d=3, dtype="f64",
                 kernel="k_siwt_fwd_level<double, true, 8>",
                 fwd_kernels=[("k_siwt_fwd_level<double, false, 8>", 1), ("k_siwt_fwd_level<double, true, 8>", 9), ("k_siwt_norms<double>", 1),
                              ("k_siwt_costs<double>", 1)],
                 desc="SURVEY 8(f) row 4: shift-invariant packet decomposition siwpd(x, wt, 10, 3) + node costs of 4096 "
                      "1024-sample f64 signals (71-column table, 2.4 GB); second leg = bestbasistree! + isiwpd of all signals"),
}


def workload_tree(w, n, L, maketree):
    """the tree of a `wpt` workload as a bool array (None: full tree of depth L): "pyramid" = maketree(n, L, :dwt),
    "random:<p>:<seed>" = tests/helpers.random_tree_1d with the root split"""
    import numpy as np
    spec = w.get("tree")
    if not spec:
        return None
    if spec == "pyramid":
        return np.asarray(maketree(n, L, "dwt"), dtype=bool)
    _, pp, seed = spec.split(":")
    rng = np.random.default_rng(int(seed))
    tree = np.zeros(n - 1, dtype=bool)
    tree[0] = rng.random() < 0.95
    for i in range(2, n):
        tree[i - 1] = tree[i // 2 - 1] and (rng.random() < float(pp))
    tree[0] = True
    return tree


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override the configuration's total batch (debug / validation)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the configuration's batch split over the ranks, weak = the whole batch per rank")
    ap.add_argument("--chunks", type=int, default=4, help="pieces of the overlapped all-gather of the reconstructed output")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the second loop with the all-gather in the step")
    ap.add_argument("--dump", default="", help="validation: rank 0 saves the (gathered) reconstructed output as .npy")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--gather", nargs="?", const="auto", default="auto", choices=["p2p", "collective", "auto"],
                    help="N > 1: exchange schedule of the all-gather loop -- grouped point-to-point (every piece lands in place), "
                         "one all_gather collective per chunk, or auto = point-to-point, switching to the collective if it raises")
    ap.add_argument("--watchdog", type=float, default=900.0,
                    help="seconds after which a rank that has not finished dumps where it is and exits 3 (a hung collective must "
                         "not hold the machine); 0 disables")
    ap.add_argument("--gather-timeout", type=float, default=240.0,
                    help="N > 1: seconds the second loop (all-gather inside the step) may take before every rank abandons it; rank 0 "
                         "then prints the line of the main loop with the failure noted (`exchange_abandoned`) and all ranks exit 4")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the `also` block (the other headline workloads, hipEvent-timed in the same run; N = 1 only)")
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
            wt_tree = workload_tree(w, n, L, wo.maketree1d)
            tree = (wo.maketree1d(n, L, "full") if wt_tree is None else wt_tree).astype(np.uint8)
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
        if kind == "sdwt":
            n = w["n"]
            x = rng.standard_normal((n, B))
            t0 = time.perf_counter()
            for i in range(B):
                xw = wo.sdwt(x[:, i], q, L)
                xr = wo.isdwt(xw, q)
            dt = time.perf_counter() - t0
            assert np.abs(xr - x[:, B - 1]).max() < 1e-9
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
        if kind == "denoise":
            n = w["n"]
            x = rng.standard_normal((n, B))
            t0 = time.perf_counter()
            for i in range(B):
                y = wo.denoise(x[:, i], "sig", q, L=L, th="hard")
            dt = time.perf_counter() - t0
            assert y.shape == (n,)
            return dt, B * n
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

    def run_omp(B, min_seconds=0.0):
        """all host cores: OpenMP over the batch (1-D decimated workloads only)"""
        n = w["n"]
        tree = wo.maketree1d(n, L, "full").astype(np.uint8)
        x0 = np.asfortranarray(rng.standard_normal((n, B)))
        # every array is first touched by the thread that works on its signals (the static schedule over the batch of the loops in
        # oracle/wx_oracle.c): the input through a parallel copy, the outputs by an untimed first pass; then passes until `min_seconds`
        x = np.empty_like(x0)
        lib.wxo_copy_omp_f64(P(x), P(x0), L64(n), L64(B))
        xh = np.empty_like(x)
        y = np.empty((n, L + 1, B) if kind == "wpd" else (n, B), order="F")

        def one():
            if kind == "wpd":
                lib.wxo_wpd_iwpd_roundtrip_omp_f64(P(xh), P(y), P(x), L64(n), I(L), L64(B), P(tree), L64(tree.size), P(q), I(q.size))
            else:
                lib.wxo_wpt_iwpt_roundtrip_omp_f64(P(xh), P(y), P(x), L64(n), L64(B), P(tree), L64(tree.size), P(q), I(q.size))
        one()
        assert np.abs(xh - x).max() < 1e-9
        t0, passes = time.perf_counter(), 0
        while passes == 0 or time.perf_counter() - t0 < min_seconds:
            one()
            passes += 1
        dt = time.perf_counter() - t0
        return dt, B * n * passes

    probe = 16 if kind in ("wpd", "wpt") else 2
    run(probe)                                   # first call: library load, page faults
    t_probe, _ = run(probe)
    cap = {"wpd": 16384, "wpt": 16384, "wpt2d": 320, "wpd_bb": 4096, "wpd_ldb": 2048, "denoise": 16384}.get(kind, 32)
    B = int(max(probe, min(cap, seconds / (t_probe / probe))))
    dt, samples = run(B)
    out = {"value": samples / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": "%d of the %d signals per step, same transform pair, %s, %.1f s"
                     % (B, w["batch"], "oracle: C -O2 steps under the reference's Dict-and-recursion object model in Python, one thread"
                        if kind == "siwt" else "oracle C -O2 single thread", dt)}
    if kind in ("wpd", "wpt") and not w.get("tree"):
        nthr = int(lib.wxo_omp_max_threads())
        Bo = int(min(16384, max(256, B * max(1, nthr // 4))))
        dto, so = run_omp(Bo, min_seconds=max(5.0, seconds / 2))
        out["all_cores"] = {"value": so / dto / 1e6, "unit": "Msamples/s", "cores": nthr, "kind": "port",
                            "sample": "%d signals x %d passes, OpenMP over the batch, every array first touched by its thread, %.1f s"
                                      % (Bo, so // (Bo * w["n"]), dto)}
    return out


# ------------------------------------------------------------------------------------------------
# PCIe-inclusive rate: host (numpy) arrays handed to the C ABI, the library stages H2D / D2H itself.  Reported next to
# the headline, never as `value` (the timed region of `value` starts with the inputs resident in HBM).
# ------------------------------------------------------------------------------------------------
def pcie_inclusive(w, wx):
    import numpy as np
    if w["kind"] not in ("wpd", "wpt"):
        return None
    n, L, B = w["n"], w["L"], min(w["batch"], 8192)
    wt = wx.wavelet(getattr(wx.WT, w["wavelet"]))
    rng = np.random.default_rng(7)
    x = np.asfortranarray(rng.standard_normal((n, B)))
    fwd = (lambda a: wx.wpdall(a, wt, L)) if w["kind"] == "wpd" else (lambda a: wx.wptall(a, wt, L))
    inv = (lambda a: wx.iwpdall(a, wt, L)) if w["kind"] == "wpd" else (lambda a: wx.iwptall(a, wt, L))
    y = fwd(x)                                  # first call: pinned ring, host threads, page cache of the allocator
    del y                                       # (freeing 3.5 GB of result pages is the caller's garbage, not the call: until round 5 the
    t0 = time.perf_counter()                    # rebinding below released the previous result INSIDE the timed region, 100-290 ms of munmap)
    y = fwd(x)                                  # the result is a freshly allocated pageable array every time
    t1 = time.perf_counter()
    xr = inv(y)
    t2 = time.perf_counter()
    assert np.abs(xr - x).max() < 1e-9
    return {"signals": B, "forward_ms": (t1 - t0) * 1e3, "inverse_ms": (t2 - t1) * 1e3,
            "forward_host_GBs": (x.nbytes + y.nbytes) / (t1 - t0) / 1e9, "inverse_host_GBs": (x.nbytes + y.nbytes) / (t2 - t1) / 1e9,
            "value": 2.0 * B * n / (t2 - t0) / 1e6, "unit": "Msamples/s",
            "note": "numpy arrays in pageable host memory -> C ABI -> freshly allocated numpy arrays; both directions through the pinned ring + "
                    "host thread pool of wx_host.hip, result pages advised MADV_HUGEPAGE; not the headline"}


# ------------------------------------------------------------------------------------------------
# GPU workloads
# ------------------------------------------------------------------------------------------------
class Legs:
    """HIP events on the launch stream (torch's current stream == the stream passed to the C ABI) around every
    forward / inverse launch of the timed steps"""

    def __init__(self, torch):
        self.torch, self.ev = torch, {"fwd": [], "inv": []}

    def run(self, leg, fn):
        e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        self.ev[leg].append((e0, e1))

    def ms(self, leg):
        return sorted(a.elapsed_time(b) for a, b in self.ev[leg])


class Workload:
    gatherable = False

    def step(self, legs):
        raise NotImplementedError

    def step_gather(self, legs):
        raise NotImplementedError

    def output(self, gathered):
        return None


def make_workload(w, wx, torch, dev, rank, world, a, dist):
    D = sys.modules["waveletsext_jl_amd.dwt"]        # the submodule (the package attribute `dwt` is the function)
    from waveletsext_jl_amd._arrays import qmf_arg
    from waveletsext_jl_amd import distributed as wd
    kind, L, Bt = w["kind"], w["L"], w["batch"]
    wt = wx.wavelet(getattr(wx.WT, w["wavelet"]))
    F = len(wt.qmf)
    td = torch.float64 if w["dtype"] == "f64" else torch.float32
    es = 8 if w["dtype"] == "f64" else 4
    A = D.Arg
    strong = world > 1 and a.scaling == "strong"
    lo, hi = wd.shard_range(Bt, world, rank) if strong else (0, Bt)
    Bl = hi - lo
    B_all = Bt if (strong or world == 1) else world * Bt          # signals the whole job processes per step
    want_gather = world > 1 and not a.no_gather

    def signals(sig):
        """this rank's signals: strong scaling slices one batch generated from one seed (so that N ranks compute what
        one rank computes), weak scaling draws an own batch per rank"""
        gen = torch.Generator(device=dev).manual_seed(1002 if (strong or world == 1) else 1002 + rank)
        if strong:
            full = wx.jl_empty(tuple(sig) + (Bt,), td, dev)
            full.normal_(generator=gen)
            x = wx.jl_empty(tuple(sig) + (Bl,), td, dev)
            x.copy_(full[..., lo:hi])
            del full
            torch.cuda.empty_cache()
            return x
        x = wx.jl_empty(tuple(sig) + (Bl,), td, dev)
        x.normal_(generator=gen)
        return x

    W = Workload()
    W.lo, W.hi, W.B_all = lo, hi, B_all
    G = {"g": None, "mode": "collective" if a.gather == "collective" else "p2p"}     # the exchange object of step_gather
    W.G = G

    if kind in ("wpd", "wpt", "wpt2d"):
        sig = (w["n"],) if kind != "wpt2d" else (w["m"], w["n"])
        npts = 1
        for v in sig:
            npts *= v
        x = signals(sig)
        B_out = (Bt if strong else world * Bt) if want_gather else Bl
        out_lo = (lo if strong else rank * Bt) if want_gather else 0
        full = wx.jl_empty(sig + (B_out,), td, dev)              # reconstructed output of every rank (or just ours)
        xh = full[..., out_lo: out_lo + Bl]
        if kind == "wpd":
            y = wx.jl_empty(sig + (L + 1, Bl), td, dev)
            fwd = lambda: D._wpd_batched(A(x), A(y), 1, wt, L)
            inv_to = lambda c0, c1, dst: D._iwpd_batched(A(y[..., c0:c1]), A(dst), 1, wt, L, None)
            fb = es * npts * Bl * (L + 2)                # x read once + (L+1) columns written once
            flops = 2.0 * F * npts * L * Bl
        else:
            y = wx.jl_empty(sig + (Bl,), td, dev)
            nd = len(sig)
            tree = workload_tree(w, sig[0], L, wx.maketree)
            fwd = lambda: D._wpt_batched("wx_wpt", A(x), A(y), nd, wt, L, tree)
            inv_to = lambda c0, c1, dst: D._wpt_batched("wx_iwpt", A(y[..., c0:c1]), A(dst), nd, wt, L, tree)
            fb = es * npts * Bl * 2
            flops = (2.0 * F * npts * L * Bl) if nd == 1 else L * 2.0 * (2 * F * npts) * Bl
        W.gatherable = want_gather
        g = None
        if want_gather:
            # strong: the shards of one batch; weak: every rank's own batch, concatenated
            G["g"] = g = wd.make_gather(full, B_out, nchunks=a.chunks, mode=G["mode"])
            assert (g.lo, g.hi) == (out_lo, out_lo + Bl), "shards of the gather and of the bench disagree"
            nposts = g.nposts
            W.set_gather_mode = lambda mode: G.update(mode=mode, g=wd.make_gather(full, B_out, nchunks=a.chunks, mode=mode))

        def step(legs):
            legs.run("fwd", fwd)
            legs.run("inv", lambda: inv_to(0, Bl, xh))

        def step_gather(legs):
            g = G["g"]
            legs.run("fwd", fwd)
            for c in range(nposts):
                if c < len(g.chunks):
                    c0, c1 = g.chunks[c]
                    legs.run("inv", lambda: inv_to(c0, c1, g.local_chunk(c)))
                g.post(c)
            g.finish()

        W.step, W.step_gather = step, step_gather
        W.check = lambda: float((xh - x).abs().max() / x.abs().max())
        W.output = lambda gathered: full if gathered else xh
        W.info = dict(fwd_bytes=fb, inv_bytes=es * npts * Bl * 2, fwd_flops=flops, samples=npts * Bl, bound="hbm",
                      gather_bytes=es * npts * (B_out - Bl))
        W.keep = (x, y, full)
        return W

    if kind == "sdwt":
        n = w["n"]
        x = signals((n,))
        xw = wx.jl_empty((n, L + 1, Bl), td, dev)
        xh = wx.jl_empty((n, Bl), td, dev)
        q, qp, Fq = qmf_arg(wt)

        def step(legs):
            legs.run("fwd", lambda: D._call("wx_sdwt1d", "_f64", A(x).ptr, A(xw).ptr, n, L, Bl, qp, Fq, A(x).stream()))
            legs.run("inv", lambda: D._call("wx_isdwt1d", "_f64", A(xw).ptr, A(xh).ptr, n, L, -1, Bl, qp, Fq, A(x).stream()))

        W.step = step
        W.check = lambda: float((xh - x).abs().max() / x.abs().max())
        W.output = lambda gathered: xh
        fb = es * n * Bl * (L + 2)
        W.info = dict(fwd_bytes=fb, inv_bytes=fb, fwd_flops=4.0 * F * n * L * Bl, samples=n * Bl, bound="hbm")
        W.keep = (x, xw, xh, q)
        return W

    if kind == "swpt":
        n, CH = w["n"], min(w["chunk"], max(Bl, 1))
        x = signals((n,))
        B_out = (Bt if strong else world * Bt) if want_gather else Bl
        out_lo = (lo if strong else rank * Bt) if want_gather else 0
        full = wx.jl_empty((n, B_out), td, dev)
        xh = full[:, out_lo: out_lo + Bl]
        xw = wx.jl_empty((n, 1 << L, CH), td, dev)
        q, qp, Fq = qmf_arg(wt)
        chunks = [(c0, min(c0 + CH, Bl)) for c0 in range(0, Bl, CH)]

        def fwd(c0, c1):
            D._call("wx_swpt1d", "_f64", A(x[:, c0:c1]).ptr, A(xw).ptr, n, L, c1 - c0, qp, Fq, A(x).stream())

        def inv(c0, c1, dst):
            D._call("wx_iswpt1d", "_f64", A(xw).ptr, A(dst).ptr, n, L, -1, c1 - c0, qp, Fq, A(x).stream())

        W.gatherable = want_gather
        if want_gather:
            G["g"] = g = wd.make_gather(full, B_out, nchunks=a.chunks, mode=G["mode"])
            nposts = g.nposts
            W.set_gather_mode = lambda mode: G.update(mode=mode, g=wd.make_gather(full, B_out, nchunks=a.chunks, mode=mode))

        def step(legs):
            for c0, c1 in chunks:
                legs.run("fwd", lambda: fwd(c0, c1))
                legs.run("inv", lambda: inv(c0, c1, xh[:, c0:c1]))

        def step_gather(legs):
            # the gather's pieces are groups of whole resident chunks
            g = G["g"]
            for c in range(nposts):
                if c < len(g.chunks):
                    g0, g1 = g.chunks[c]
                    for c0 in range(g0, g1, CH):
                        c1 = min(c0 + CH, g1)
                        legs.run("fwd", lambda: fwd(c0, c1))
                        legs.run("inv", lambda: inv(c0, c1, xh[:, c0:c1]))
                g.post(c)
            g.finish()

        W.step, W.step_gather = step, step_gather
        W.check = lambda: float((xh - x).abs().max() / x.abs().max())
        W.output = lambda gathered: full if gathered else xh
        fb = es * n * CH * (1 + (1 << L))
        W.info = dict(fwd_bytes=fb, inv_bytes=fb, fwd_flops=((1 << L) - 1) * 4.0 * F * n * CH, samples=n * Bl, bound="hbm",
                      launch_unit="one resident chunk of %d signals" % CH, gather_bytes=es * n * (B_out - Bl))
        W.keep = (x, xw, full, q)
        return W

    if kind == "acwpd_jbb":
        n, CH = w["n"], min(w["chunk"], max(Bl, 1))
        x = signals((n,))
        ncols = (1 << (L + 1)) - 1
        chunks = [(c0, min(c0 + CH, Bl)) for c0 in range(0, Bl, CH)]
        state = {}
        N_total = B_all
        method = wx.JBB(redundant=True)
        backend_nccl = world > 1 and dist.get_backend() == "nccl"

        def moments():
            s = q = None
            for c0, c1 in chunks:
                if s is None:
                    s, q = wx.acwpd_jbb_moments(x[:, c0:c1], wt, L)
                else:
                    wx.acwpd_jbb_moments(x[:, c0:c1], wt, L, accumulate_into=(s, q))
            state["s"], state["q"] = s, q

        def tree():
            s, q = state["s"], state["q"]
            if world > 1:                               # C2: the moments of the whole batch on every rank
                if backend_nccl:
                    s, q = wd.allreduce_moments(s, q)
                else:                                   # validation mode: gloo on host copies
                    sc, qc = wd.allreduce_moments(s.cpu(), q.cpu())
                    s, q = wx.to_colmajor(sc.to(dev)), wx.to_colmajor(qc.to(dev))
            costs = wx.costs_from_moments(s, q, N_total, method)
            # the tree and the margin of its closest `cc < pc` decision (BestBasis.jl:72): "bit-exact tree indices" holds across
            # summation orders only while this margin is far above the rounding of the costs (~1e-13)
            state["tree"], state["gap"] = wx.bestbasis_treeselection(costs, n, return_gap=True)

        def step(legs):
            legs.run("fwd", moments)
            legs.run("inv", tree)

        W.step = step
        W.extra = lambda: {"min_rel_cost_gap": state.get("gap"), "tree_split_nodes": int(state["tree"].sum())}
        W.check = lambda: 0.0 if wx.isvalidtree(torch.empty(n), state["tree"]) else 1.0
        W.output = lambda gathered: torch.from_numpy(state["tree"].astype("float64"))
        flops = acwpd_jbb_min_flops(n, L, F) * Bl
        W.info = dict(fwd_bytes=es * (n * Bl + 2 * n * ncols), inv_bytes=es * 2 * n * ncols, fwd_flops=flops, samples=n * Bl,
                      bound="fp64", flops_note="minimal algorithm: per depth the periodised autocorrelation filter (odd lags and the "
                      "centre tap only, symmetric pairs folded), S shared by both children, 3 flops per node sample for sum x and "
                      "sum x^2 (acwpd_jbb_min_flops); what the kernels execute is more (circulant tiles on the matrix pipe)",
                      launch_unit="the rank's whole shard: %d chunks of <= %d signals" % (len(chunks), CH),
                      collective="all-reduce of 2 x %d moments inside the second leg" % (n * ncols) if world > 1 else None)
        W.keep = (x,)
        return W

    # ---- widened rows: sharded like the rest, no exchange step timed -----------------------------------
    n = w["n"]
    x = signals((n,))
    B = Bl
    if kind == "wpd_bb":
        from waveletsext_jl_amd import bestbasis as bbm
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
        W.info = dict(fwd_bytes=es * (n * (L + 1) + ncost) * B + (n - 1) * B, inv_bytes=es * n * (L + 2) * B,
                      fwd_flops=4.0 * n * (L + 1) * B, samples=n * B, bound="hbm")
        W.keep = (x, xw)
    elif kind == "denoise":
        state = {}
        dnt = wx.VisuShrink(n)

        def fwd():      # one pass: wx_denoiseall_sig_*
            state["y"] = wx.denoiseall(x, "sig", wt, L=L, dnt=dnt)

        def inv():      # the separate steps (named inv only because the harness times two legs): dwtall, then noisest + threshold + idwtall
            state["y2"] = wx.denoiseall(wx.dwtall(x, wt, L), "dwt", wt, L=L, dnt=dnt)

        check = lambda: float((state["y"] - state["y2"]).abs().max() / x.abs().max())
        W.info = dict(fwd_bytes=es * 2 * n * B, inv_bytes=es * 4.5 * n * B, fwd_flops=(2 * 8.0 * F + 100.0) * n * B, samples=n * B, bound="hbm")
        W.keep = (x,)
    elif kind == "wpd_ldb":
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
        W.info = dict(fwd_bytes=es * (n * (L + 1)) * (B + 4), inv_bytes=es * n * (L + 2) * B,
                      fwd_flops=2.0 * n * (L + 1) * B, samples=n * B, bound="hbm")
        W.keep = (x, xw)
    elif kind == "siwt":
        d = w["d"]
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
        # come out of the level that creates them; only the top depths are read a second time).  Below depth d only half
        # of a depth's columns have children: a node of shift t has the children of shifts t and t + 2^j, and depth j + 1
        # keeps the multiples of 2^(j+1-d) (SIWT.jl:119-131) -- until round 4 every column was counted as read and the
        # PMC traffic came out BELOW these bytes (0.86 x; profiles/r04_siwt_bytes.txt has the counters)
        RC = sum((1 << j) if j < d else max(1, (1 << d) >> 1) for j in range(L))
        fb = es * B * (n * RC + n * NS + NN)
        W.info = dict(fwd_bytes=fb, inv_bytes=es * B * (2 * NN + 3 * n * L) + 2 * NN * B,
                      fwd_flops=4.0 * F * (n / 2) * (NS - 1) * B, samples=n * B, bound="hbm")
        W.keep = (x,)
    else:
        raise ValueError(kind)

    def step(legs):
        legs.run("fwd", fwd)
        legs.run("inv", inv)

    W.step, W.check = step, check
    return W

def acwpd_jbb_min_flops(n, L, F):
    """Float64 operations per signal of acwpd + the two JBB moments by the cheapest direct algorithm (an FMA = 2).
    acwt/acwt_one_level.jl:101-128: children w1 = c v + S, w2 = c v - S with S = sum over the non-zero taps of the
    autocorrelation filter (2F - 1 taps, only odd lags and the centre are non-zero, symmetric) at dilation 2^d; on a
    signal of n samples the taps of depth d wrap onto n / 2^d positions, taps that land on one position merge.
    bestbasis_tree.jl:150-158: sum x and sum x^2 of every coefficient of every node."""
    total = 3.0 * n                                             # moments of the root column
    for d in range(L):
        t = n >> d                                              # distinct positions k + m 2^d (mod n)
        pos = {l % t for l in range(-(F - 1), F, 2) if l % 2} - {0}
        pairs = len({min(p, t - p) for p in pos if p != t - p and (t - p) in pos})
        singles = len(pos) - 2 * pairs
        per_parent = 3.0 * pairs + 2.0 * singles + 3.0          # S (pair: add + FMA), then c v + S and c v - S
        total += (1 << d) * n * (per_parent + 2 * 3.0)          # + the moments of the two children
    return total


# ------------------------------------------------------------------------------------------------
# `also`: the other headline workloads under the same clock as the default line (N = 1): the north-star target, one
# resident chunk of config 3, config 4, four 2048-signal chunks of config 5 -- per leg the dominant kernel's name, the
# hipEvent-timed average launch (>= 10 launches after 2 warm-up steps), the algorithmic bytes (or flops) of one launch and
# the roofline fraction.  Same code path as `--workload <name>` (make_workload), smaller step counts.
# ------------------------------------------------------------------------------------------------
ALSO = (("target", {}), ("tree_random", {}), ("cfg3", {"batch": 64}), ("cfg3_sdwt", {}), ("cfg4", {}), ("cfg5", {"batch": 8192}))


def also_block(wx, torch, dev, a, dist, steps=10):
    out = {}
    for name, over in ALSO:
        if name not in WORKLOADS:
            continue
        t0 = time.perf_counter()
        try:
            w = dict(WORKLOADS[name])
            w.update(over)
            W = make_workload(w, wx, torch, dev, 0, 1, a, dist)
            warm = Legs(torch)
            for _ in range(2):
                W.step(warm)
            torch.cuda.synchronize(dev)
            err = W.check()
            # the check's temporaries reshuffle the caching allocator's blocks: one more untimed step, or the first timed
            # launch can carry a 20 ms hipMalloc (seen in round 4 as also.target.inv avg 2.9 ms, median 0.80 ms)
            W.step(warm)
            torch.cuda.synchronize(dev)
            legs = Legs(torch)
            n_alloc0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
            gc.collect()
            gc.disable()             # (round 4: also.target.inv showed one 22-97 ms launch in ten -- the collector, not the allocator)
            try:
                for _ in range(steps):
                    W.step(legs)
                torch.cuda.synchronize(dev)
            finally:
                gc.enable()
            n_alloc = torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - n_alloc0
            if os.environ.get("WX_BENCH_DEBUG"):
                sys.stderr.write("also.%s: device allocations inside the timed steps: %d; inv launches (ms, in order): %s\n" % (
                    name, n_alloc, " ".join("%.2f" % a_.elapsed_time(b_) for a_, b_ in legs.ev["inv"])))
            info = W.info
            per_step = {k: len(legs.ev[k]) // steps for k in ("fwd", "inv")}
            fwd = sum(legs.ms("fwd")) / steps          # all forward launches of a step (one, or one per resident chunk)
            inv = sum(legs.ms("inv")) / steps
            fb, ib = float(info["fwd_bytes"]) * per_step["fwd"], float(info["inv_bytes"]) * per_step["inv"]
            rec = {"workload": w["desc"], "batch": w["batch"], "roundtrip_rel_err": err}
            if info["bound"] == "hbm":
                rec["fwd"] = {"kernel": w["kernel"], "avg_launch_ms": fwd, "algorithmic_bytes": fb,
                              "achieved_GBs": fb / fwd / 1e6, "frac": fb / fwd / 1e6 / HBM_PEAK_GBS}
                rec["inv"] = {"kernel": w.get("inv_kernel"), "avg_launch_ms": inv, "algorithmic_bytes": ib,
                              "achieved_GBs": ib / inv / 1e6, "frac": ib / inv / 1e6 / HBM_PEAK_GBS}
                for leg in ("fwd", "inv"):               # one slow launch (an allocation inside the region) shows as avg >> median
                    t = legs.ms(leg)
                    rec[leg]["median_launch_ms"] = t[len(t) // 2]
                    rec[leg]["max_launch_ms"] = t[-1]
                    rec[leg]["launches_per_step"] = per_step[leg]
            else:
                fl = float(info["fwd_flops"])
                rec["fwd"] = {"kernel": w["kernel"], "avg_launch_ms": fwd, "algorithmic_flops": fl,
                              "achieved_TFLOPs": fl / fwd / 1e9, "frac": fl / fwd / 1e9 / FP64_PEAK_TFLOPS,
                              "bound": info["bound"], "hbm_GBs": fb / fwd / 1e6}
                if "chunk" in w:                         # the first chunk of a step allocates the moments, the others accumulate
                    nch = -(-w["batch"] // w["chunk"])
                    rec["fwd"]["chunks_per_launch"] = nch
                    rec["fwd"]["avg_chunk_ms"] = fwd / nch
                rec["inv"] = {"kernel": None, "avg_launch_ms": inv, "note": "costs + tree selection from the moments"}
                if "flops_note" in info:
                    rec["fwd"]["flops_note"] = info["flops_note"]
            rec["Msamples_per_s"] = info["samples"] / ((fwd + inv) * 1e-3) / 1e6
            if getattr(W, "extra", None):
                rec.update(W.extra())
            rec["wall_s"] = time.perf_counter() - t0
            out[name] = rec
            del W, legs, warm
        except Exception as e:  # pragma: no cover - never lose the headline line to a side measurement
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    return out


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
        # the ranks are fresh child processes in their own process group (never a re-exec of this one); if they outlive the
        # watchdog -- each rank has its own, this is the backstop -- the whole group is killed and the exit code says so
        import signal
        child = subprocess.Popen(cmd, start_new_session=True)
        try:
            sys.exit(child.wait(timeout=(a.watchdog + 120.0) if a.watchdog > 0 else None))
        except subprocess.TimeoutExpired:
            sys.stderr.write("bench.py: ranks still running after %.0f s, killing process group %d\n" % (a.watchdog + 120.0, child.pid))
            try:
                os.killpg(child.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            child.wait()
            sys.exit(3)
    import torch
    import torch.distributed as dist
    import waveletsext_jl_amd as wx

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    stage = {"at": "start"}                              # where this rank is, for the watchdog's last words
    wd_timer = None
    if a.watchdog > 0:
        import faulthandler
        import threading

        def _abort():
            sys.stderr.write("bench.py watchdog: rank %d of %d not finished after %.0f s (at: %s); exiting 3\n"
                             % (rank, world, a.watchdog, stage["at"]))
            faulthandler.dump_traceback(file=sys.stderr)
            sys.stderr.flush()
            os._exit(3)

        wd_timer = threading.Timer(a.watchdog, _abort)
        wd_timer.daemon = True
        wd_timer.start()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # WX_BENCH_BACKEND=gloo lets several ranks share one GPU (a validation mode for 1-GPU boxes: same sharding,
    # barriers, exchange schedule, max-over-ranks timing and aggregation, no RCCL); the real run is nccl, one GPU per rank
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
    stage["at"] = "make_workload"
    W = make_workload(w, wx, torch, dev, rank, world, a, dist)
    info = W.info
    red_dev = dev if backend == "nccl" else "cpu"

    def across_ranks(v):
        """one number per rank, in rank order, on every rank"""
        if world == 1:
            return [float(v)]
        t = torch.zeros(world, dtype=torch.float64, device=red_dev)
        t[rank] = float(v)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(x) for x in t.tolist()]

    stage["at"] = "first collective (device indices)"
    rank_devices = [int(v) for v in across_ranks(local)]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(step_fn):
        legs = Legs(torch)
        gc.collect()
        gc.disable()                                     # a generation-2 collection between an event and its launch is a 20-100 ms leg
        try:
            sync()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step_fn(legs)
            sync()
            mine = time.perf_counter() - t0
        finally:
            gc.enable()
        per_rank = [mine]
        if world > 1:
            t = torch.zeros(world, dtype=torch.float64, device=red_dev)
            t[rank] = mine
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            per_rank = [float(v) for v in t.tolist()]
        return max(per_rank), per_rank, legs

    stage["at"] = "warm-up steps"
    warm = Legs(torch)
    for _ in range(a.warmup):
        W.step(warm)
    sync()
    err = W.check()
    tol = 1e-10 if w["dtype"] == "f64" else 1e-5
    assert err < tol, "round trip broken: %g" % err

    stage["at"] = "timed steps"
    elapsed, per_rank, legs = timed(W.step)
    fwd_ms, inv_ms = legs.ms("fwd"), legs.ms("inv")
    fwd_avg = sum(fwd_ms) / len(fwd_ms)
    inv_avg = sum(inv_ms) / len(inv_ms)
    samples_all = W.B_all * (info["samples"] // max(W.hi - W.lo, 1))

    def build_line(gather):
        """the JSON line from the main timed loop (rank 0); `gather` = the second loop's record or None"""
        # HBM traffic of one forward pass = sum over its kernels (launches per pass x PMC bytes per launch)
        # from separate rocprofv3 --pmc passes of this same command (profiles/traffic.json, written by
        # tools/collect_evidence.py); null if not profiled.  `kernel` names the dominant one.
        traffic, traffic_stale = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(a.workload, {})
            if not a.batch and world == 1:
                traffic = sum(cnt * tj[name]["hbm_bytes_per_launch"] for name, cnt in w["fwd_kernels"])
                # the counters belong to the build they were taken with: tools/collect_evidence.py records its source digest
                # (wx_build_info's src=...); another build gets null and says why
                import re as _re
                cur = _re.search(r"src=(\w+)", wx.build_info())
                made = tj.get("_build_src")
                if made != (cur.group(1) if cur else None):
                    traffic_stale = "profiles/traffic.json holds the counters of build src=%s, this library is src=%s" % (made, cur.group(1) if cur else None)
                    traffic = None
        except Exception:
            traffic = None
        fb = float(info["fwd_bytes"])
        if info["bound"] == "hbm":
            achieved = fb / (fwd_avg * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": w["kernel"], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": fb}
        else:
            achieved = info["fwd_flops"] / (fwd_avg * 1e-3) / 1e12
            roof = {"bound": "fp64 (vector + matrix pipe share the FP64 rate; flops counted for the direct form)", "kernel": w["kernel"], "achieved": achieved,
                    "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic,
                    "algorithmic_flops_per_step": info["fwd_flops"], "hbm_GBs": fb / (fwd_avg * 1e-3) / 1e9}
        roof["traffic_source"] = "profiles/traffic.json (offline rocprofv3 --pmc passes of this command with this build, not measured in this run)" if traffic is not None else None
        if traffic_stale:
            roof["traffic_stale"] = traffic_stale
        roof.update({"avg_launch_ms": fwd_avg, "median_launch_ms": fwd_ms[len(fwd_ms) // 2],
                     "launches_per_step": len(fwd_ms) // a.steps,
                     "fwd_TFLOPs_direct_form": info["fwd_flops"] / (fwd_avg * 1e-3) / 1e12})
        if "launch_unit" in info:
            roof["launch_unit"] = info["launch_unit"]
        cfg = {"workload": w["desc"], "batch": w["batch"], "batch_this_rank": W.hi - W.lo, "wavelet": w["wavelet"], "L": w["L"],
               "sharding": ("contiguous batch shards (distributed.shard_range), no data-path collective in `value`"
                            if world > 1 else "one GPU")}
        cfg.update({k: w[k] for k in ("n", "m", "chunk") if k in w})
        if info.get("collective"):
            cfg["collective"] = info["collective"]
        out = {
            "metric": "Msamples/s (fwd+inv wavelet packets)",
            "value": samples_all * a.steps / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": a.scaling if world > 1 else "none", "vs_baseline": None,
            "dtype": w["dtype"],
            "data": "synthetic N(0,1), seed 1002 (one batch, sliced per rank; weak scaling: 1002 + rank), resident in HBM",
            "config": cfg,
            "roofline": roof,
            "inverse": {"kernel": w.get("inv_kernel"), "avg_launch_ms": inv_avg,
                        "algorithmic_bytes_per_launch": float(info["inv_bytes"]),
                        "achieved_GBs": info["inv_bytes"] / (inv_avg * 1e-3) / 1e9,
                        "frac": info["inv_bytes"] / (inv_avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "bound": w.get("inv_bound", "hbm")},
            "roundtrip_rel_err": err,
            "build": wx.build_info(),
            "ranks": {"nranks": dist.get_world_size() if world > 1 else 1, "backend": backend if world > 1 else None,
                      "rccl_version": (".".join(str(v) for v in torch.cuda.nccl.version())
                                       if world > 1 and backend == "nccl" else None),
                      "devices": rank_devices, "device_name": torch.cuda.get_device_name(dev),
                      "visible_devices": torch.cuda.device_count(),
                      "per_rank_ms": [v / a.steps * 1e3 for v in per_rank],
                      "min_ms": min(per_rank) / a.steps * 1e3, "max_ms": max(per_rank) / a.steps * 1e3},
        }
        if getattr(W, "extra", None):
            out["config"].update(W.extra())
        if gather is not None:
            out["with_allgather"] = gather
            # the sharded compute (`value`) and the throughput with the exchange inside the step, side by side
            out["with_allgather_value"] = gather.get("value")
        return out

    # The exchange loop below is reported NEXT to `value`, never part of it -- and it is the one part of this program that has never
    # run on more than one GPU.  If it does not finish in --gather-timeout seconds (a hung grouped send/recv cannot be cancelled from
    # Python), every rank's timer fires at about the same time: rank 0 writes the line of the main loop -- BUILT AND SERIALISED HERE,
    # in the main thread, before the loop starts: the timer thread touches neither torch nor the library while the main thread may sit
    # in RCCL -- with the failure in `with_allgather` and `"exchange_abandoned": true`, and every rank leaves with exit code 4: the
    # headline number survives on stdout, and a launcher that only reads exit codes still sees that the collective hung.
    EXIT_EXCHANGE_ABANDONED = 4
    gather_timer = None
    import threading
    print_lock = threading.Lock()
    printed = {"done": False}

    def emit(text):
        """the ONE JSON line of this process: whoever comes first (main thread or the give-up timer) prints, the other does not"""
        with print_lock:
            if printed["done"]:
                return False
            printed["done"] = True
            sys.stdout.write(text + "\n")
            sys.stdout.flush()
            return True

    if world > 1 and W.gatherable and a.gather_timeout > 0:
        abandoned_line = None
        if rank == 0:
            ab = build_line({"error": "exchange loop not finished after %.0f s; abandoned, `value` is the sharded compute" % a.gather_timeout})
            ab["exchange_abandoned"] = True
            abandoned_line = json.dumps(ab)

        def _give_up_gather():
            if rank == 0:
                emit(abandoned_line)
            sys.stderr.write("bench.py: rank %d gives up the exchange loop after %.0f s (at: %s); exit code %d\n"
                             % (rank, a.gather_timeout, stage["at"], EXIT_EXCHANGE_ABANDONED))
            sys.stderr.flush()
            os._exit(EXIT_EXCHANGE_ABANDONED)

        gather_timer = threading.Timer(a.gather_timeout, _give_up_gather)
        gather_timer.daemon = True
        gather_timer.start()
    gather = None
    if W.gatherable:
      try:
          # second loop: the all-gather of the reconstructed output inside the step, overlapped with the inverse
          stage["at"] = "all-gather loop, warm-up (%s)" % W.G["mode"]
          p2p_error = None
          try:
              for _ in range(max(1, min(a.warmup, 2))):
                  W.step_gather(Legs(torch))
              torch.cuda.synchronize(dev)
              failed = 0.0
          except Exception as e:
              if a.gather != "auto" or W.G["mode"] != "p2p":
                  raise
              p2p_error, failed = "%s: %s" % (type(e).__name__, e), 1.0
          if a.gather == "auto" and W.G["mode"] == "p2p" and max(across_ranks(failed)) > 0:
              # the grouped point-to-point exchange raised on some rank: every rank switches to one collective per chunk
              W.set_gather_mode("collective")
              stage["at"] = "all-gather loop, warm-up (collective, after the point-to-point group raised)"
              for _ in range(max(1, min(a.warmup, 2))):
                  W.step_gather(Legs(torch))
          sync()
          stage["at"] = "all-gather loop, timed (%s)" % W.G["mode"]
          g_elapsed, g_per_rank, _ = timed(W.step_gather)
          # the exchange alone (nothing to overlap with), for the budget: one step's worth of posts
          from waveletsext_jl_amd import distributed as wd
          full = W.output(True)
          sync()
          t1 = time.perf_counter()
          gg = wd.make_gather(full, full.shape[-1], nchunks=a.chunks, mode=W.G["mode"])
          for c in range(gg.nposts):
              gg.post(c)
          gg.finish()
          sync()
          alone_ms = (time.perf_counter() - t1) * 1e3
          gather = {"ms_per_step": g_elapsed / a.steps * 1e3,
                    "value": samples_all * a.steps / g_elapsed / 1e6,
                    "allgather_alone_ms": alone_ms,
                    "exposed_ms": (g_elapsed - elapsed) / a.steps * 1e3,
                    "overlap_ms": max(0.0, alone_ms - (g_elapsed - elapsed) / a.steps * 1e3),
                    "chunks": a.chunks, "mode": W.G["mode"], "p2p_error": p2p_error,
                    "bytes_received_per_rank": float(info.get("gather_bytes", 0)),
                    "per_rank_ms": [v / a.steps * 1e3 for v in g_per_rank],
                    "schedule": "inverse in %d chunks; chunk c's exchange (%s) on a side stream while chunk c+1 is transformed"
                                % (a.chunks, "grouped point-to-point, every piece lands in place" if W.G["mode"] == "p2p"
                                   else "one all_gather collective per chunk")}
          gerr = float((full[..., W.lo:W.hi] - W.keep[0]).abs().max() / W.keep[0].abs().max()) if a.scaling == "strong" else None
          if gerr is not None:
              assert gerr < tol, "gathered output broken: %g" % gerr
      except AssertionError:
          raise
      except Exception as e:  # pragma: no cover - the exchange is reported next to `value`, never part of it
          gather = {"error": "%s: %s" % (type(e).__name__, e)}

    if gather_timer is not None:
        gather_timer.cancel()

    if a.dump:
        import numpy as np
        outp = W.output(W.gatherable)
        if rank == 0 and outp is not None:
            np.save(a.dump, outp.detach().cpu().numpy())

    out = build_line(gather) if rank == 0 else None
    if rank == 0:
        if world == 1 and not a.no_also and not a.batch and a.workload == "cfg2":
            W.keep = None
            del W, legs, warm
            torch.cuda.empty_cache()
            out["also"] = also_block(wx, torch, dev, a, dist)
        if not a.no_cpu and world == 1:
            try:
                pc = pcie_inclusive(w, wx)
            except Exception as e:  # pragma: no cover
                pc = {"error": str(e)}
            if pc is not None:
                out["pcie_inclusive"] = pc
        if not a.no_cpu and world == 1:                 # rank 0 at N = 1 only: the other ranks would wait at the final barrier
            cb = cpu_baseline(w, a.cpu_seconds)
            allc = cb.pop("all_cores", None)
            out["cpu_baseline"] = cb
            if allc is not None:
                out["cpu_baseline_all_cores"] = allc
    stage["at"] = "final barrier"
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if wd_timer is not None:
        wd_timer.cancel()
    if rank == 0:
        emit(json.dumps(out))


if __name__ == "__main__":
    main()
