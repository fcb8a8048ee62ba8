import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle")]
import numpy as np
import waveletsext_jl_amd as wx, wx_oracle as O
rng = np.random.default_rng(5)
for n in (512, 256, 128, 64):
    for wname in ("db2", "db3", "db4", "db8"):
        wt = wx.wavelet(getattr(wx.WT, wname))
        for B in (1, 7, 64, 65, 130):
            x = np.asfortranarray(rng.standard_normal((n, B)))
            for L in range(1, int(np.log2(n)) + 1):
                exp = O.wptall(x, wt.qmf, L)
                e = np.abs(wx.wptall(x, wt, L) - exp).max() / np.abs(exp).max()
                e2 = np.abs(wx.iwptall(exp, wt, L) - x).max() / np.abs(x).max()
                tab = O.wpdall(x, wt.qmf, L)
                e3 = np.abs(wx.wpdall(x, wt, L) - tab).max() / np.abs(tab).max()
                e4 = np.abs(wx.iwpdall(tab, wt, L) - x).max() / np.abs(x).max()
                if max(e, e2, e3, e4) > 1e-12:
                    print("FAIL", n, wname, B, L, e, e2, e3, e4)
print("done")
