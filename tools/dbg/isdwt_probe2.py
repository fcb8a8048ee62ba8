import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import waveletsext_jl_amd as wx
for wname, L in (("db6", 10), ("db4", 10), ("db6", 9), ("db6", 10)):
    wt = wx.wavelet(getattr(wx.WT, wname))
    n, B = 4096, 16384
    x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
    y = wx.sdwtall(x, wt, L)
    ts = []
    for _ in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        xr = wx.isdwtall(y, wt)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(wname, L, " ".join("%.2f" % t for t in ts))
    del x, y, xr
