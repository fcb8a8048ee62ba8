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
for wname in ("db2", "db4", "coif6"):
    wt = wx.wavelet(getattr(wx.WT, wname))
    for n, B, L in ((1024, 2048, 10), (4096, 128, 12), (2048, 512, 11)):
        x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
        gb = 8e-9 * n * B * (1 << L)
        f = t(lambda: wx.swptall(x, wt, L))
        a = t(lambda: wx.acwptall(x, wt, L))
        y = wx.swptall(x, wt, L)
        i = t(lambda: wx.iswptall(y, wt))
        print("%-5s n %5d B %5d L %2d (%.1f GB leaves): swptall %.2f ms (%.0f %% HBM)  acwptall %.2f ms (%.0f %%)  iswptall %.2f ms (%.0f %%)" % (
            wname, n, B, L, gb, f, 100 * gb / f / 8, a, 100 * gb / a / 8, i, 100 * gb / i / 8))
        del x, y
