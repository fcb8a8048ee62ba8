"""diagnostics: the 2-D cells below the bar of profiles/r05_floor2d.txt, a few calls each, for rocprofv3 --kernel-trace --stats"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import waveletsext_jl_amd as wx

wt = wx.wavelet(wx.WT.db4)
which = sys.argv[1] if len(sys.argv) > 1 else "f64full"
for m in (int(v) for v in sys.argv[2:]) or (256, 512, 1024):
    dt, esz = (torch.float64, 8) if which.startswith("f64") else (torch.float32, 4)
    B = (1 << 30) // (m * m * esz)
    x = wx.jl_empty((m, m, B), dt, "cuda")
    x.normal_()
    L = wx.maxtransformlevels(m)
    for _ in range(3):
        if which.endswith("full"):
            y = wx.wptall(x, wt, L)
            z = wx.iwptall(y, wt, L)
        else:
            y = wx.dwtall(x, wt, L)
            z = wx.idwtall(y, wt, L)
        del y, z
    torch.cuda.synchronize()
    del x
    torch.cuda.empty_cache()
