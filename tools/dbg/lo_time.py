# wptall / iwptall of depth 1 .. 6, 65536 signals of 4096 samples
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import waveletsext_jl_amd as wx
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
n, B = 4096, 65536
x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
gb = 2e-9 * n * B * 8
for wname in ("haar", "db4"):
    wt = wx.wavelet(getattr(wx.WT, wname))
    for L in (1, 2, 3, 4, 5, 6):
        f = t(lambda: wx.wptall(x, wt, L)); y = wx.wptall(x, wt, L); i = t(lambda: wx.iwptall(y, wt, L))
        print("%-5s L=%d wptall %.2f ms (%.0f %%)  iwptall %.2f ms (%.0f %%)  rt %.1e" % (wname, L, f, 100 * gb / f / 8, i, 100 * gb / i / 8,
              float((wx.iwptall(y, wt, L) - x).abs().max())))
