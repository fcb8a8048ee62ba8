"""Kernel breakdown of LocalDiscriminantBasis fit_transform at one length (1 GiB packet table): tools/dbg/prof_any.sh tools/dbg/prof_ldb.py <n>"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import waveletsext_jl_amd as wx

n = int(sys.argv[1])
wt = wx.wavelet(wx.WT.db4)
L = wx.maxtransformlevels(n)
B = max((1 << 30) // (n * (L + 1) * 8), 1)
x = wx.jl_empty((n, B), torch.float64, "cuda")
x.normal_()
labels = [i % 3 for i in range(B)]
f = wx.LocalDiscriminantBasis(wt=wt, n_features=10)
wx.fit_transform(f, x, labels)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    wx.fit_transform(f, x, labels)
torch.cuda.synchronize()
print("wall per call %.2f ms" % ((time.perf_counter() - t0) / 3 * 1e3))
