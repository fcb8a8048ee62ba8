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
wt = wx.wavelet(wx.WT.db4)
n, B = 4096, 65536
x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
y = wx.jl_empty((n, B), torch.float64, "cuda")
for L in (1, 2, 4, 6, 7, 8, 9, 10, 11, 12):
    f = t(lambda: wx.dwtall(x, wt, L)); yy = wx.dwtall(x, wt, L); i = t(lambda: wx.idwtall(yy, wt, L))
    print("L=%2d dwtall %.2f ms  idwtall %.2f ms" % (L, f, i))
    del yy
