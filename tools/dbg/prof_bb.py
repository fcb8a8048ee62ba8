import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import waveletsext_jl_amd as wx
wt = wx.wavelet(wx.WT.db4)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = wx.maxtransformlevels(n)
Bq = max((1 << 28) // (n * (L + 1) * 8), 1)
xq = wx.jl_empty((n, Bq), torch.float64, "cuda"); xq.normal_()
tab = wx.wpdall(xq, wt, L)
for _ in range(3):
    t = wx.bestbasistreeall(tab, wx.BB())
torch.cuda.synchronize()
print(Bq)
