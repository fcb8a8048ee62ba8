# dwtall / idwtall (pyramid of full depth) and wptall / iwptall (full tree) of long Float64 signals: fraction of the HBM peak
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import waveletsext_jl_amd as wx
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
wname = sys.argv[1] if len(sys.argv) > 1 else "db4"
dt = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else torch.float64
esz = 4 if dt == torch.float32 else 8
wt = wx.wavelet(getattr(wx.WT, wname))
for n in (4096, 8192, 16384, 32768, 65536):
    B = 65536 * 4096 // n
    x = wx.jl_empty((n, B), dt, "cuda"); x.normal_()
    gb = 2e-9 * n * B * esz
    L = wx.maxtransformlevels(n)
    f = t(lambda: wx.dwtall(x, wt)); y = wx.dwtall(x, wt); i = t(lambda: wx.idwtall(y, wt))
    err = float((wx.idwtall(y, wt) - x).abs().max())
    f2 = t(lambda: wx.wptall(x, wt, L)); y2 = wx.wptall(x, wt, L); i2 = t(lambda: wx.iwptall(y2, wt, L))
    f3 = t(lambda: wx.dwtall(x, wt, 4)); y3 = wx.dwtall(x, wt, 4); i3 = t(lambda: wx.idwtall(y3, wt, 4))
    print("n %5d %s: dwtall %.2f ms (%.0f %%) idwtall %.2f ms (%.0f %%) rt %.0e | L=4: %.2f (%.0f %%) / %.2f (%.0f %%) | full tree wptall %.2f (%.0f %%) iwptall %.2f (%.0f %%)" % (
        n, wname, f, 100 * gb / f / 8, i, 100 * gb / i / 8, err, f3, 100 * gb / f3 / 8, i3, 100 * gb / i3 / 8, f2, 100 * gb / f2 / 8, i2, 100 * gb / i2 / 8))
    del x, y, y2, y3
