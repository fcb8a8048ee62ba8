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
for n, B, L in ((8192, 32768, 11), (16384, 16384, 12), (32768, 8192, 12)):
    x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
    f = t(lambda: wx.wptall(x, wt, L))
    y = wx.wptall(x, wt, L)
    i = t(lambda: wx.iwptall(y, wt, L))
    d0 = n.bit_length() - 13
    f1 = t(lambda: wx.wptall(x, wt, d0))
    x2 = x.T.contiguous().view(-1, 4096).T  # (4096, B << d0) view of the same bytes (column-major)
    f2 = t(lambda: wx.wptall(x2, wt, L - d0))
    print("n %5d B %6d L %2d: wptall %.3f ms  iwptall %.3f ms | %d top levels %.3f ms + lattice(4096, L-%d) %.3f ms" % (n, B, L, f, i, d0, f1, d0, f2))
    del x, y, x2
