import sys, os
sys.path.insert(0, os.getcwd())
import torch
import waveletsext_jl_amd as wx
wt = wx.wavelet(wx.WT.db4)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for n, B, L in ((4096, 65536, 10), (2048, 131072, 10), (1024, 262144, 9), (2048, 131072, 11), (1024, 262144, 10)):
    x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
    f = t(lambda: wx.wptall(x, wt, L))
    y = wx.wptall(x, wt, L)
    i = t(lambda: wx.iwptall(y, wt, L))
    print("n %5d B %6d L %2d: wptall %.3f ms (%.1f %% HBM)  iwptall %.3f ms (%.1f %%)" % (n, B, L, f, 100 * 2 * 8 * n * B / f / 1e6 / 8e6, i, 100 * 2 * 8 * n * B / i / 1e6 / 8e6))
    del x, y
