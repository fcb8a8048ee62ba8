import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle")]
import numpy as np
import waveletsext_jl_amd as wx, wx_oracle as O
rng = np.random.default_rng(3)
for n in (8192, 16384, 32768, 65536):
    for wname in ("db2", "db4", "coif6"):
        wt = wx.wavelet(getattr(wx.WT, wname))
        for B in (1, 3):
            x = np.asfortranarray(rng.standard_normal((n, B)))
            for L in (5, 7, 8, 10, int(np.log2(n)) - 1, int(np.log2(n))):
                exp = O.wptall(x, wt.qmf, L)
                got = wx.wptall(x, wt, L)
                e = np.abs(got - exp).max() / np.abs(exp).max()
                back = wx.iwptall(exp, wt, L)
                e2 = np.abs(back - x).max() / np.abs(x).max()
                if e > 1e-12 or e2 > 1e-12:
                    print("FAIL", n, wname, B, L, e, e2)
print("done")
