"""cProfile of bestbasistreeall(wpdall(x), BB()) on the host side (GPU box): python3 tools/dbg/bb_cprofile.py <n>"""
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import waveletsext_jl_amd as wx

n = int(sys.argv[1])
wt = wx.wavelet(wx.WT.db4)
L = wx.maxtransformlevels(n)
B = max((1 << 30) // (n * (L + 1) * 8), 1)
x = wx.jl_empty((n, B), torch.float64, "cuda")
x.normal_()
tab = wx.wpdall(x, wt, L)
wx.bestbasistreeall(tab, wx.BB())
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    t = wx.bestbasistreeall(tab, wx.BB())
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(16)
print(s.getvalue()[:3500])
