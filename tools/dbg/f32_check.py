import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle")]
import numpy as np
import waveletsext_jl_amd as wx, wx_oracle as O
rng = np.random.default_rng(6)
for wname in ("db2", "db4", "db7", "coif6"):
    wt = wx.wavelet(getattr(wx.WT, wname))
    for B in (1, 3):
        x = np.asfortranarray(rng.standard_normal((4096, B)).astype(np.float32))
        for L in range(5, 13):
            exp = O.wptall(x.astype(np.float64), wt.qmf, L)
            got = wx.wptall(x, wt, L)
            assert got.dtype == np.float32
            e = np.abs(got - exp).max() / np.abs(exp).max()
            back = wx.iwptall(exp.astype(np.float32), wt, L)
            e2 = np.abs(back - x).max() / np.abs(x).max()
            tab = O.wpdall(x.astype(np.float64), wt.qmf, L)
            e3 = np.abs(wx.iwpdall(tab.astype(np.float32), wt, L) - x).max() / np.abs(x).max()
            if max(e, e2, e3) > 1e-6:
                print("FAIL", wname, B, L, e, e2, e3)
print("done")
