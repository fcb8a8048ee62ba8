# sdwtall / isdwtall / acdwtall timing: algorithmic bytes = (L + 2) n per signal each way
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import waveletsext_jl_amd as wx
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for wname in ("haar", "db4", "db6", "coif6"):
    wt = wx.wavelet(getattr(wx.WT, wname))
    for n, B, L in ((4096, 16384, 6), (4096, 16384, 10), (1024, 65536, 8), (16384, 4096, 8)):
        x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
        gb = 8e-9 * n * B * (L + 2)
        f = t(lambda: wx.sdwtall(x, wt, L))
        y = wx.sdwtall(x, wt, L)
        i = t(lambda: wx.isdwtall(y, wt))
        xr = wx.isdwtall(y, wt)
        a = t(lambda: wx.acdwtall(x, wt, L))
        print("%-5s n %5d B %5d L %2d (%.1f GB): sdwtall %.2f ms (%.0f %% HBM)  isdwtall %.2f ms (%.0f %%)  acdwtall %.2f ms (%.0f %%)  rt %.1e" % (
            wname, n, B, L, gb, f, 100 * gb / f / 8, i, 100 * gb / i / 8, a, 100 * gb / a / 8, float((xr - x).abs().max())))
        del x, y, xr
