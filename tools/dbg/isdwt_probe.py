import sys, os
sys.path.insert(0, os.getcwd())
import torch
import waveletsext_jl_amd as wx
wt = wx.wavelet(getattr(wx.WT, sys.argv[1]))
n, B, L = 4096, 16384, int(sys.argv[2])
x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
y = wx.sdwtall(x, wt, L)
for _ in range(4): xr = wx.isdwtall(y, wt)
torch.cuda.synchronize()
