import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import waveletsext_jl_amd as wx
from oracle import wx_oracle as O
rng = np.random.default_rng(5)
for name in ("db4", "db2"):
    wt = wx.wavelet(getattr(wx.WT, name)); q = np.asarray(wt.qmf)
    x = np.asfortranarray(rng.standard_normal((512, 512, 2)).astype(np.float32))
    ref = O.wptall(x, q, 6)
    got = wx.wptall(x, wt, 6)
    e = np.abs(got - ref).max() / np.abs(ref).max()
    print(name, "fwd err", e)
    if e > 1e-5:
        d = np.abs(got - ref)[:, :, 0]
        print("bad rows", np.where(d.max(axis=1) > 1e-4)[0][:20], "bad cols", np.where(d.max(axis=0) > 1e-4)[0][:20])
        print(got[:4, :4, 0]); print(ref[:4, :4, 0])
    back = wx.iwptall(ref, wt, 6)
    print(name, "inv err", np.abs(back - x).max() / np.abs(x).max())
