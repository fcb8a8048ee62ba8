import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import waveletsext_jl_amd as wx
wt = wx.wavelet(wx.WT.db4)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = (1 << 30) // (n * 8)
x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
for _ in range(2):
    y = wx.denoiseall(x, "sig", wt)
torch.cuda.synchronize()
