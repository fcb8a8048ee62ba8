import os, sys, cProfile, pstats, io
sys.path.insert(0, "/root/repo")
import torch
import waveletsext_jl_amd as wx
n = int(sys.argv[1])
wt = wx.wavelet(wx.WT.db4)
L = wx.maxtransformlevels(n)
B = max((1 << 30) // (n * (L + 1) * 8), 1)
x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
labels = [i % 3 for i in range(B)]
f = wx.LocalDiscriminantBasis(wt=wt, n_features=10)
wx.fit_transform(f, x, labels); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    wx.fit_transform(f, x, labels)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
