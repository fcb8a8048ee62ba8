import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle")]
import numpy as np
import waveletsext_jl_amd as wx, wx_oracle as O
rng = np.random.default_rng(2)
for n in (2048, 1024):
    for wname in ("db2", "db8"):
        wt = wx.wavelet(getattr(wx.WT, wname))
        for B in (2, 3, 4, 6):
            x = np.asfortranarray(rng.standard_normal((n, B)))
            for L in range(3, int(np.log2(n)) + 1):
                tab = O.wpdall(x, wt.qmf, L)
                got = wx.wpdall(x, wt, L)
                e = np.abs(got - tab).max() / np.abs(tab).max()
                back = wx.iwpdall(tab, wt, L)
                e2 = np.abs(back - x).max() / np.abs(x).max()
                if e > 1e-12 or e2 > 1e-12:
                    print("FAIL", n, wname, B, L, e, e2)
print("done")
