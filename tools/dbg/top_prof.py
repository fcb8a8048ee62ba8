# the multi-level top pass under rocprofv3: full trees and pyramids of long Float64 / Float32 signals, a few launches each
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import waveletsext_jl_amd as wx
wt = wx.wavelet(getattr(wx.WT, sys.argv[1] if len(sys.argv) > 1 else "db4"))
for dt in (torch.float64, torch.float32):
    for n in (8192, 16384, 65536):
        B = 65536 * 4096 // n
        x = wx.jl_empty((n, B), dt, "cuda"); x.normal_()
        L = wx.maxtransformlevels(n)
        for _ in range(3):
            y = wx.wptall(x, wt, L); z = wx.iwptall(y, wt, L)
            y2 = wx.dwtall(x, wt); z2 = wx.idwtall(y2, wt)
        torch.cuda.synchronize()
        del x, y, z, y2, z2
