# dwtall / idwtall (pyramid) and tree-driven wptall / iwptall of 4096-sample signals
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import waveletsext_jl_amd as wx
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from helpers import random_tree_1d
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for wname in ("haar", "db4", "db8"):
    wt = wx.wavelet(getattr(wx.WT, wname))
    for dt in (torch.float64, torch.float32):
        n, B = 4096, 65536
        x = wx.jl_empty((n, B), dt, "cuda"); x.normal_()
        gb = 2e-9 * n * B * x.element_size()
        for L in (4, 12):
            f = t(lambda: wx.dwtall(x, wt, L)); y = wx.dwtall(x, wt, L); i = t(lambda: wx.idwtall(y, wt, L))
            print("%-5s %s dwtall L=%2d %.2f ms (%.0f %%)  idwtall %.2f ms (%.0f %%)" % (wname, str(dt)[-7:], L, f, 100 * gb / f / 8, i, 100 * gb / i / 8))
        tr = random_tree_1d(n, np.random.default_rng(3))
        f = t(lambda: wx.wptall(x, wt, tr)); y = wx.wptall(x, wt, tr); i = t(lambda: wx.iwptall(y, wt, tr))
        print("%-5s %s wptall random tree %.2f ms (%.0f %%)  iwptall %.2f ms (%.0f %%)" % (wname, str(dt)[-7:], f, 100 * gb / f / 8, i, 100 * gb / i / 8))
        f = t(lambda: wx.wptall(x, wt, 5)); y = wx.wptall(x, wt, 5); i = t(lambda: wx.iwptall(y, wt, 5))
        print("%-5s %s wptall L=5 %.2f ms (%.0f %%)  iwptall %.2f ms (%.0f %%)" % (wname, str(dt)[-7:], f, 100 * gb / f / 8, i, 100 * gb / i / 8))
        del x, y
