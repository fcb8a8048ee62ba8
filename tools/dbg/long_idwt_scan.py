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
wt = wx.wavelet(wx.WT.db4)
for n in (16384, 32768, 65536):
    B = 65536 * 4096 // n
    x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
    out = []
    for L in range(1, wx.maxtransformlevels(n) + 1):
        y = wx.dwtall(x, wt, L)
        f = t(lambda: wx.dwtall(x, wt, L)); i = t(lambda: wx.idwtall(y, wt, L))
        err = float((wx.idwtall(y, wt, L) - x).abs().max())
        out.append("L%d %.2f/%.2f%s" % (L, f, i, "" if err < 1e-10 else " ERR %.1e" % err))
        del y
    print("n %d: " % n + "  ".join(out))
