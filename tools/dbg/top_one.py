# one configuration of the multi-level top pass, a few launches: python top_one.py <n> <f64|f32> <wpt|dwt> [wavelet]
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import waveletsext_jl_amd as wx
n = int(sys.argv[1]); dt = torch.float32 if sys.argv[2] == "f32" else torch.float64
wt = wx.wavelet(getattr(wx.WT, sys.argv[4] if len(sys.argv) > 4 else "db4"))
B = 65536 * 4096 // n
x = wx.jl_empty((n, B), dt, "cuda"); x.normal_()
L = wx.maxtransformlevels(n)
for _ in range(3):
    if sys.argv[3] == "wpt":
        y = wx.wptall(x, wt, L); z = wx.iwptall(y, wt, L)
    else:
        y = wx.dwtall(x, wt); z = wx.idwtall(y, wt)
torch.cuda.synchronize()
