import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle")]
import numpy as np
import waveletsext_jl_amd as wx, wx_oracle as O
rng = np.random.default_rng(1)
for n in (2048, 1024):
    for wname in ("db2", "db4", "db8"):
        wt = wx.wavelet(getattr(wx.WT, wname))
        for B in (4, 5, 7, 8):
            x = np.asfortranarray(rng.standard_normal((n, B)))
            for L in range(3, int(np.log2(n)) + 1):
                exp = O.wptall(x, wt.qmf, L)
                got = wx.wptall(x, wt, L)
                e = np.abs(got - exp).max() / np.abs(exp).max()
                if e > 1e-12:
                    print("FAIL fwd", n, wname, B, L, e)
                back = wx.iwptall(exp, wt, L)
                e2 = np.abs(back - x).max() / np.abs(x).max()
                if e2 > 1e-12:
                    print("FAIL inv", n, wname, B, L, e2)
print("done")
