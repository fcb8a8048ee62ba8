import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import waveletsext_jl_amd as wx
from floor_scan import timed, HBM_PEAK
wt = wx.wavelet(wx.WT.db4)
for side in (32, 64, 128):
    B = (1 << 30) // (side ** 3 * 8)
    x = wx.jl_empty((side, side, side, B), torch.float64, "cuda"); x.normal_()
    for L in (1, 3):
        t = timed(torch, lambda: wx.dwtall(x, wt, L))
        y = wx.dwtall(x, wt, L)
        ti = timed(torch, lambda: wx.idwtall(y, wt, L))
        print("f64 3-D %3d^3 dwt L=%d fwd %7.3f ms (%4.1f %%) inv %7.3f ms (%4.1f %%)" % (side, L, t, 100 * 2 * side ** 3 * B * 8 / (t * 1e-3) / HBM_PEAK, ti, 100 * 2 * side ** 3 * B * 8 / (ti * 1e-3) / HBM_PEAK), flush=True)
    del x, y
    torch.cuda.empty_cache()
