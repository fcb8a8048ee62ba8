import sys, os
sys.path.insert(0, os.getcwd())
import torch
import waveletsext_jl_amd as wx
wt = wx.wavelet(wx.WT.db4)
n, B, L = 4096, 65536, int(sys.argv[1])
x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
for _ in range(4): y = wx.dwtall(x, wt, L)
for _ in range(4): xr = wx.idwtall(y, wt, L)
torch.cuda.synchronize()
