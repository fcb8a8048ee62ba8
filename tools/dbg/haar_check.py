import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle")]
import numpy as np, torch
import waveletsext_jl_amd as wx, wx_oracle as O
rng = np.random.default_rng(7)
wt = wx.wavelet(wx.WT.haar)
for n in (2048, 1024, 512, 256, 128, 64):
    for B in (3, 70):
        x = np.asfortranarray(rng.standard_normal((n, B)))
        for L in range(1, int(np.log2(n)) + 1):
            exp = O.wptall(x, wt.qmf, L)
            e = np.abs(wx.wptall(x, wt, L) - exp).max() / np.abs(exp).max()
            e2 = np.abs(wx.iwptall(exp, wt, L) - x).max() / np.abs(x).max()
            tab = O.wpdall(x, wt.qmf, L)
            e3 = np.abs(wx.wpdall(x, wt, L) - tab).max() / np.abs(tab).max()
            e4 = np.abs(wx.iwpdall(tab, wt, L) - x).max() / np.abs(x).max()
            if max(e, e2, e3, e4) > 1e-12:
                print("FAIL", n, B, L, e, e2, e3, e4)
print("done")
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for n in (2048, 512, 64):
    B = (65536 * 4096) // n
    L = min(10, n.bit_length() - 1)
    x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
    f = t(lambda: wx.wptall(x, wt, L)); y = wx.wptall(x, wt, L); i = t(lambda: wx.iwptall(y, wt, L))
    print("haar n %4d: wptall %.3f ms  iwptall %.3f ms" % (n, f, i))
