"""ad-hoc kernel timing helper (diagnostics): python tools/timeit.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import waveletsext_jl_amd as wx
from waveletsext_jl_amd.dwt import Arg, _wpt_batched, _wpd_batched, _iwpd_batched

def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

n, B = 4096, 65536
x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
y1 = wx.jl_empty((n, B), torch.float64, "cuda")
for name, L in (("db8", 12), ("db8", 10), ("db4", 12), ("db4", 10), ("db2", 12), ("db2", 10), ("haar", 12), ("haar", 10)):
    wt = wx.wavelet(getattr(wx.WT, name))
    yt = wx.jl_empty((n, L + 1, B), torch.float64, "cuda")
    f_wpt = t(lambda: _wpt_batched("wx_wpt", Arg(x), Arg(y1), 1, wt, L, None))
    i_wpt = t(lambda: _wpt_batched("wx_iwpt", Arg(y1), Arg(x), 1, wt, L, None))
    f_wpd = t(lambda: _wpd_batched(Arg(x), Arg(yt), 1, wt, L))
    i_wpd = t(lambda: _iwpd_batched(Arg(yt), Arg(y1), 1, wt, L, None))
    flops = 2.0 * len(wt.qmf) * n * L * B
    print("%-5s L=%2d  wpt %.3f ms (%.1f TF)  iwpt %.3f ms (%.1f TF)  wpd %.3f ms (%.0f GB/s)  iwpd %.3f ms" % (
        name, L, f_wpt, flops / f_wpt / 1e9, i_wpt, flops / i_wpt / 1e9, f_wpd, 8.0 * n * B * (L + 2) / f_wpd / 1e6, i_wpd))
    del yt
