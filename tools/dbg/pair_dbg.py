import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import numpy as np
import waveletsext_jl_amd as wx
import wx_oracle as o
rng = np.random.default_rng(0)
for n in (1024, 2048, 4096):
    for wname in ("db4", "db5", "db6", "db7", "db8", "db10"):
        wt = wx.wavelet(getattr(wx.WT, wname))
        for B in (8, 9):
            x = np.asfortranarray(rng.standard_normal((n, B)).astype(np.float32))
            t = wx.maketree(n, 1, "full")
            e = o.wptall(x, wt.qmf, t)
            y = wx.wptall(x, wt, t)
            err = np.abs(y - e).max(axis=0) / np.abs(e).max()
            back = wx.iwptall(e, wt, t)
            errb = np.abs(back - x).max(axis=0) / np.abs(x).max()
            print(n, wname, B, "fwd err per signal", " ".join("%.0e" % v for v in err), "| inv", " ".join("%.0e" % v for v in errb))
