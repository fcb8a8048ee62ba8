# time of the shallow passes of swptall: the same three passes write depth 6 compactly (L = 6) or at their
# final columns of the depth-10 table (L = 10, every 16th column)
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import waveletsext_jl_amd as wx
L = int(sys.argv[1])
n, B = 1024, 2048
wt = wx.wavelet(wx.WT.db4)
x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
for _ in range(6):
    y = wx.swptall(x, wt, L)
torch.cuda.synchronize()
for _ in range(6):
    xr = wx.iswptall(y, wt)
torch.cuda.synchronize()
print("roundtrip", float((xr - x).abs().max()))
