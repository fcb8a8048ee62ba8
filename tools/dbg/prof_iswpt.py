import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import waveletsext_jl_amd as wx
wt = wx.wavelet(wx.WT.db4)
n, L = 16384, int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = (1 << 30) // (n * (1 << L) * 8)
x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
y = wx.swptall(x, wt, L)
for _ in range(3):
    z = wx.iswptall(y, wt)
torch.cuda.synchronize()
