import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import waveletsext_jl_amd as wx
from floor_scan import timed, HBM_PEAK
wt = wx.wavelet(wx.WT.db4)
for m in (64, 256):
    for kind, L in (("sdwt", 2), ("swpd", 2), ("acdwt", 2), ("acwpd", 2), ("swpt", 2)):
        cols = {"sdwt": 3 * L + 1, "acdwt": 3 * L + 1, "swpd": (4 ** (L + 1) - 1) // 3, "acwpd": (4 ** (L + 1) - 1) // 3, "swpt": 4 ** L}[kind]
        B = max((1 << 29) // (m * m * cols * 8), 1)
        x = wx.jl_empty((m, m, B), torch.float64, "cuda"); x.normal_()
        fwd = {"sdwt": wx.sdwtall, "swpd": wx.swpdall, "swpt": wx.swptall, "acdwt": wx.acdwtall, "acwpd": wx.acwpdall}[kind]
        try:
            t = timed(torch, lambda: fwd(x, wt, L))
            gb = m * m * B * 8 * (1 + cols)
            print("f64 2-D %4dx%-4d %-6s L=%d fwd %7.3f ms (%4.1f %%)" % (m, m, kind, L, t, 100 * gb / (t * 1e-3) / HBM_PEAK), flush=True)
        except Exception as e:
            print("f64 2-D", m, kind, "error", str(e)[:80], flush=True)
        del x
        torch.cuda.empty_cache()
