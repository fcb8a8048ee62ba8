# scan of common calls at awkward shapes: fraction of the HBM peak on the algorithmic bytes (read + write of the signal once)
import sys, os
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import waveletsext_jl_amd as wx
from helpers import random_tree_1d
def t(fn, reps=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
wt = wx.wavelet(wx.WT.db4)
def line(name, gb, f, i=None):
    print("%-58s %7.2f ms (%4.1f %%)" % (name, f, 100 * gb / f / 8) + ("   inverse %7.2f ms (%4.1f %%)" % (i, 100 * gb / i / 8) if i else ""))
# 2-D pyramids and full trees
for dt, nm, esz in ((torch.float32, "f32", 4), (torch.float64, "f64", 8)):
    for m, B in ((256, 4096), (512, 1024), (1024, 256)):
        x = wx.jl_empty((m, m, B), dt, "cuda"); x.normal_()
        gb = 2e-9 * m * m * B * esz
        y = wx.dwtall(x, wt); line("2-D dwtall %s %dx%dx%d (full depth)" % (nm, m, m, B), gb, t(lambda: wx.dwtall(x, wt)), t(lambda: wx.idwtall(y, wt)))
        y = wx.dwtall(x, wt, 3); line("2-D dwtall %s %dx%dx%d L=3" % (nm, m, m, B), gb, t(lambda: wx.dwtall(x, wt, 3)), t(lambda: wx.idwtall(y, wt, 3)))
        y = wx.wptall(x, wt, 3); line("2-D wptall %s %dx%dx%d L=3 (full tree)" % (nm, m, m, B), gb, t(lambda: wx.wptall(x, wt, 3)), t(lambda: wx.iwptall(y, wt, 3)))
        del x, y
# 1-D random trees on long signals, wpd on long signals
for n in (8192, 16384, 65536):
    B = 32768 * 4096 // n
    x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
    gb = 2e-9 * n * B * 8
    tr = random_tree_1d(n, np.random.default_rng(3), 0.7); tr[0] = True
    y = wx.wptall(x, wt, tr); line("1-D wptall f64 n=%d random tree p=0.7" % n, gb, t(lambda: wx.wptall(x, wt, tr)), t(lambda: wx.iwptall(y, wt, tr)))
    L = 6
    gbw = 1e-9 * n * B * 8 * (L + 2)
    yw = wx.wpdall(x[:, :B // 4], wt, L); line("1-D wpdall f64 n=%d L=6 (B/4)" % n, gbw / 4, t(lambda: wx.wpdall(x[:, :B // 4], wt, L)))
    del x, y, yw
