import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle")]
import numpy as np
import waveletsext_jl_amd as wx, wx_oracle as O
rng = np.random.default_rng(8)
for n in (1024, 2048):
    for wname in ("haar", "db2", "db4", "coif6", "db10"):
        wt = wx.wavelet(getattr(wx.WT, wname))
        x = np.asfortranarray(rng.standard_normal((n, 2)))
        for L in (5, 7, 8, int(np.log2(n))):
            exp = np.stack([O.swpt(x[:, b], wt.qmf, L) for b in range(2)], axis=-1)
            got = wx.swptall(x, wt, L)
            e = np.abs(got - exp).max() / np.abs(exp).max()
            expa = np.stack([O.acwpt(x[:, b], wt.qmf, L) for b in range(2)], axis=-1)
            gota = wx.acwptall(x, wt, L)
            e2 = np.abs(gota - expa).max() / np.abs(expa).max()
            back = wx.iswptall(exp, wt)
            e3 = np.abs(back - x).max() / np.abs(x).max()
            if e > 1e-12 or e2 > 1e-12 or e3 > 1e-11:
                print("FAIL", n, wname, L, e, e2, e3)
print("done")
