"""Kernel breakdown of bestbasistreeall(wpdall(x), BB()) at one length: rocprofv3 --kernel-trace --stats -- python3 tools/dbg/prof_bb.py <n>"""
import os
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
for _ in range(6):
    t = wx.bestbasistreeall(tab, wx.BB())
torch.cuda.synchronize()
