import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch, waveletsext_jl_amd as wx
from helpers import random_tree_2d
wt = wx.wavelet(wx.WT.db4)
def t(f, k=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): r = f(); del r
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
rng = np.random.default_rng(1)
for dt, B in ((torch.float32, 512), (torch.float64, 256)):
    x = wx.jl_empty((512, 512, B), dt, "cuda"); x.normal_()
    dw = wx.maketree(512, 512, 6, "dwt")
    rt = random_tree_2d(512, 512, rng, 0.8); rt[(4 ** 6 - 1) // 3:] = False
    for name, tree in (("dwt tree L=6", dw), ("random tree p=0.8 depth<=6 (%d nodes)" % rt.sum(), rt)):
        y = wx.wptall(x, wt, tree)
        print(dt, name, "wptall %.2f ms" % t(lambda: wx.wptall(x, wt, tree)), "iwptall %.2f ms" % t(lambda: wx.iwptall(y, wt, tree)))
