import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import waveletsext_jl_amd as wx
from floor_scan import timed, HBM_PEAK
wt = wx.wavelet(wx.WT.db4)
for n in (64, 256, 1024, 4096, 16384):
    B = (1 << 30) // (n * 4)
    x = wx.jl_empty((n, B), torch.float32, "cuda"); x.normal_()
    t = timed(torch, lambda: wx.denoiseall(x, "sig", wt))
    print("f32 n %6d denoiseall %7.3f ms" % (n, t), flush=True)
    del x; torch.cuda.empty_cache()
    L = wx.maxtransformlevels(n)
    Bq = max((1 << 28) // (n * (L + 1) * 4), 1)
    xq = wx.jl_empty((n, Bq), torch.float32, "cuda"); xq.normal_()
    tab = wx.wpdall(xq, wt, L)
    t = timed(torch, lambda: wx.bestbasistreeall(tab, wx.BB()))
    print("f32 n %6d bestbasistreeall(BB) %7.3f ms" % (n, t), flush=True)
    del xq, tab; torch.cuda.empty_cache()
