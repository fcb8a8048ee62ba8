# forward swptall with long filters: single levels above the lane-local kernel (default) or two-level composite passes
# (WX_SWTFWD_KF_MAXF=20)
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
for wname in ("coif6", "db10"):
    wt = wx.wavelet(getattr(wx.WT, wname))
    for n, B, L in ((16384, 64, 12), (4096, 2048, 8), (1024, 8192, 8), (1024, 2048, 10)):
        x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
        gb = 8e-9 * n * B * (1 << L)
        f = t(lambda: wx.swptall(x, wt, L))
        print("%-5s n %5d B %5d L %2d: swptall %.2f ms (%.0f %% HBM)" % (wname, n, B, L, f, 100 * gb / f / 8))
        del x
