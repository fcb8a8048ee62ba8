import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import waveletsext_jl_amd as wx
wt = wx.wavelet(wx.WT.db6)
for n, B, L in ((4096, 16384, 6), (4096, 16384, 10), (4096, 16384, 10)):
    x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
    for _ in range(6): y = wx.sdwtall(x, wt, L)
    y = wx.sdwtall(x, wt, L)
    ts = []
    for _ in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter(); xr = wx.isdwtall(y, wt); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(L, "wall", " ".join("%.2f" % t for t in ts), "ptr y %x x %x" % (y.data_ptr(), xr.data_ptr()), torch.cuda.memory_reserved() / 1e9)
    del x, y, xr
