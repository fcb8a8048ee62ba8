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
for n in (512, 256, 128, 64):
    B = (65536 * 4096) // n
    L = n.bit_length() - 1
    x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
    f = t(lambda: wx.wptall(x, wt, L))
    y = wx.wptall(x, wt, L)
    i = t(lambda: wx.iwptall(y, wt, L))
    w = t(lambda: wx.wpdall(x[:, :B // 8], wt, L))
    print("n %4d B %8d L %2d: wptall %.3f ms (%.0f %% HBM)  iwptall %.3f ms (%.0f %%)  wpdall(B/8) %.3f ms (%.2f TB/s)" % (
        n, B, L, f, 100 * 4.295 / f / 8.0, i, 100 * 4.295 / i / 8.0, w, 8e-9 * n * (B // 8) * (L + 2) / w))
    del x, y
