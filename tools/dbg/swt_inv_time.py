# the inverses that run one level per launch: shift-based isdwt / iswpt, tree-driven iswpd; and swpd forward
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import waveletsext_jl_amd as wx
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
wt = wx.wavelet(wx.WT.db4)
n, B, L = 4096, 16384, 6
x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
sd = wx.sdwtall(x, wt, L)
gb = 8e-9 * n * B * (L + 2)
print("isdwtall avg   %.2f ms (%.0f %%)" % ((lambda v: (v, 100 * gb / v / 8))(t(lambda: wx.isdwtall(sd, wt)))))
print("isdwtall sm=5  %.2f ms (%.0f %%)" % ((lambda v: (v, 100 * gb / v / 8))(t(lambda: wx.isdwtall(sd, wt, 5)))))
del sd
n, B, L = 1024, 2048, 6
x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
sp = wx.swptall(x, wt, L)
gb = 8e-9 * n * B * ((1 << L) + 1)
print("iswptall avg   %.2f ms (%.0f %%)" % ((lambda v: (v, 100 * gb / v / 8))(t(lambda: wx.iswptall(sp, wt)))))
print("iswptall sm=37 %.2f ms (%.0f %%)" % ((lambda v: (v, 100 * gb / v / 8))(t(lambda: wx.iswptall(sp, wt, 37)))))
del sp
pd = wx.swpdall(x, wt, L)
gbp = 8e-9 * n * B * ((2 << L) - 1)
print("swpdall        %.2f ms (%.0f %% of the table)" % ((lambda v: (v, 100 * gbp / v / 8))(t(lambda: wx.swpdall(x, wt, L)))))
print("iswpdall L      %.2f ms (%.0f %% of the leaves)" % ((lambda v: (v, 100 * gb / v / 8))(t(lambda: wx.iswpdall(pd, wt, L)))))
print("iswpdall sm=37  %.2f ms" % t(lambda: wx.iswpdall(pd, wt, L, 37)))
