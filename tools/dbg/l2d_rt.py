import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import waveletsext_jl_amd as wx
wt = wx.wavelet(wx.WT.db4)
for B in (2, 16, 64, 256, 1024, 4096):
    x = wx.jl_empty((512, 512, B), torch.float32, "cuda"); x.normal_()
    y = wx.wptall(x, wt, 6)
    xr = wx.iwptall(y, wt, 6)
    err = (xr - x).abs().amax(dim=(0, 1)) / x.abs().max()
    bad = (err > 1e-4).nonzero().flatten().cpu().numpy()
    print("B", B, "max err", float(err.max()), "bad images", len(bad), bad[:10])
    del x, y, xr
