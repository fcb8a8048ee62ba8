import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import waveletsext_jl_amd as wx
wt = wx.wavelet(wx.WT.db4)
n, B = 4096, 16384
x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("denoiseall(:sig) %d x %d: %.3f ms" % (B, n, t(lambda: wx.denoiseall(x, "sig", wt))))
xw = wx.dwtall(x, wt)
print("denoiseall(:dwt): %.3f ms" % t(lambda: wx.denoiseall(xw, "dwt", wt)))
tree = wx.maketree(n, 6, "full")
xp = wx.wptall(x, wt, tree)
print("denoiseall(:wpt, full tree L=6): %.3f ms" % t(lambda: wx.denoiseall(xp, "wpt", wt, tree=tree)))
print("dwtall %.3f  idwtall %.3f  noisest %.3f ms" % (t(lambda: wx.dwtall(x, wt)), t(lambda: wx.idwtall(xw, wt)), t(lambda: wx.noisest(xw[:, 0], False))))
