import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import waveletsext_jl_amd as wx
from oracle import wx_oracle as O
rng = np.random.default_rng(5)
worst = 0
for name in ("db2", "db4", "db8"):
    wt = wx.wavelet(getattr(wx.WT, name)); q = np.asarray(wt.qmf)
    for L in range(1, 13):
        x = np.asfortranarray(rng.standard_normal((4096, 3)))
        ref = O.wpdall(x, q, L)
        got = wx.wpdall(x, wt, L)
        e = np.abs(got - ref).max() / np.abs(ref).max()
        worst = max(worst, e)
        if e > 1e-12: print("FAIL", name, L, e)
print("worst", worst)
