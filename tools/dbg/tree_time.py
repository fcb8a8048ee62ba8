"""Tree-driven wptall / iwptall (Float64) on the lattice: random trees, the dwt pyramid, against the fused LDS kernels.
usage: python tools/dbg/tree_time.py [n] [wavelet]   (WX_LATTICE_TREE=0 selects the round-2 path)"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import waveletsext_jl_amd as wx
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from helpers import random_tree_1d
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
wname = sys.argv[2] if len(sys.argv) > 2 else "db4"
def t(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
wt = wx.wavelet(getattr(wx.WT, wname))
B = 65536 * 4096 // n
x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
gb = 2e-9 * n * B * 8
Lmax = wx.maxtransformlevels(n)
cases = [("pyramid L=%d" % Lmax, wx.maketree(n, Lmax, "dwt")), ("pyramid L=4", wx.maketree(n, 4, "dwt"))]
for p, seed in ((0.7, 3), (0.7, 4), (0.9, 5), (0.5, 6)):
    tr = random_tree_1d(n, np.random.default_rng(seed), p); tr[0] = True
    cases.append(("random p=%.1f seed %d (depth %d)" % (p, seed, int(np.floor(np.log2(np.flatnonzero(tr).max() + 1))) + 1), tr))
for name, tr in cases:
    f = t(lambda: wx.wptall(x, wt, tr)); y = wx.wptall(x, wt, tr); i = t(lambda: wx.iwptall(y, wt, tr))
    err = float((wx.iwptall(y, wt, tr) - x).abs().max())
    print("n %d %s %-34s wptall %.3f ms (%.1f %%)  iwptall %.3f ms (%.1f %%)  rt %.1e" % (n, wname, name, f, 100 * gb / f / 8, i, 100 * gb / i / 8, err))
f = t(lambda: wx.wptall(x, wt, Lmax)); y = wx.wptall(x, wt, Lmax); i = t(lambda: wx.iwptall(y, wt, Lmax))
print("n %d %s %-34s wptall %.3f ms (%.1f %%)  iwptall %.3f ms (%.1f %%)" % (n, wname, "full tree L=%d" % Lmax, f, 100 * gb / f / 8, i, 100 * gb / i / 8))
