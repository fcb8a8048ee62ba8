import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle")]
import numpy as np
import waveletsext_jl_amd as wx, wx_oracle as O
rng = np.random.default_rng(4)
# long signals through every path: f64 db7-less filters (lattice), haar, f32 (fused / generic), wpd
for n in (16384, 32768):
    for wname, dt in (("db4", np.float64), ("haar", np.float64), ("db4", np.float32), ("coif6", np.float32)):
        wt = wx.wavelet(getattr(wx.WT, wname))
        tol = 1e-12 if dt == np.float64 else 2e-5
        x = np.asfortranarray(rng.standard_normal((n, 2)).astype(dt))
        for L in (3, 9, int(np.log2(n))):
            exp = O.wptall(x.astype(np.float64), wt.qmf, L)
            got = wx.wptall(x, wt, L).astype(np.float64)
            back = wx.iwptall(exp.astype(dt), wt, L).astype(np.float64)
            e, e2 = np.abs(got - exp).max() / np.abs(exp).max(), np.abs(back - x).max() / np.abs(x).max()
            if e > tol or e2 > tol:
                print("FAIL wpt", n, wname, dt, L, e, e2)
        L = 5
        tab = O.wpdall(x.astype(np.float64), wt.qmf, L)
        got = wx.wpdall(x, wt, L).astype(np.float64)
        e = np.abs(got - tab).max() / np.abs(tab).max()
        back = wx.iwpdall(tab.astype(dt), wt, L).astype(np.float64)
        e2 = np.abs(back - x).max() / np.abs(x).max()
        if e > tol or e2 > tol:
            print("FAIL wpd", n, wname, dt, L, e, e2)
print("done")
