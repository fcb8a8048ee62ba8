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
for wname in ("db8", "db4"):
    wt = wx.wavelet(getattr(wx.WT, wname))
    for n, B, L in ((4096, 32768, 12), (2048, 65536, 11), (1024, 131072, 10)):
        x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
        f = t(lambda: wx.wpdall(x, wt, L))
        y = wx.wpdall(x, wt, L)
        i = t(lambda: wx.iwpdall(y, wt, L))
        gb = 8.0 * n * B * (L + 2) / 1e9
        print("%s n %5d B %6d L %2d: wpdall %.3f ms (%.2f TB/s)  iwpdall %.3f ms" % (wname, n, B, L, f, gb / f, i))
        del x, y
