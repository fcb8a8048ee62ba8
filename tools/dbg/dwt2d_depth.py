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
for m, B, dt in ((512, 1024, torch.float32), (256, 2048, torch.float64)):
    x = wx.jl_empty((m, m, B), dt, "cuda"); x.normal_()
    gb = 2e-9 * m * m * B * x.element_size()
    Lmax = m.bit_length() - 1
    for L in (1, 2, 3, 4, 6, Lmax - 1, Lmax):
        f = t(lambda: wx.dwtall(x, wt, L)); y = wx.dwtall(x, wt, L); i = t(lambda: wx.idwtall(y, wt, L))
        print("%dx%d %s L=%d dwtall %.2f ms (%.0f %%)  idwtall %.2f ms (%.0f %%)" % (m, m, str(dt)[-7:], L, f, 100 * gb / f / 8, i, 100 * gb / i / 8))
        del y
