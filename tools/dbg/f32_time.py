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
for n, L in ((4096, 10), (1024, 9), (256, 8)):
    B = (131072 * 4096) // n
    x = wx.jl_empty((n, B), torch.float32, "cuda"); x.normal_()
    f = t(lambda: wx.wptall(x, wt, L))
    y = wx.wptall(x, wt, L)
    i = t(lambda: wx.iwptall(y, wt, L))
    w = t(lambda: wx.wpdall(x[:, :B // 8], wt, L))
    print("f32 n %4d B %8d L %2d: wptall %.3f ms (%.0f %% HBM)  iwptall %.3f ms (%.0f %%)  wpdall(B/8) %.3f ms (%.2f TB/s)" % (
        n, B, L, f, 100 * 4.295 / f / 8.0, i, 100 * 4.295 / i / 8.0, w, 4e-9 * n * (B // 8) * (L + 2) / w))
    del x, y
