"""Kernel breakdown of denoiseall(x, :sig) on Float32 signals at one length (1 GiB of signals): tools/dbg/prof_any.sh tools/dbg/prof_denoise_f32.py <n>"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import waveletsext_jl_amd as wx

n = int(sys.argv[1])
B = (1 << 30) // (n * 4)
wt = wx.wavelet(wx.WT.db4)
x = wx.jl_empty((n, B), torch.float32, "cuda")
x.normal_()
y = wx.denoiseall(x, "sig", wt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    y = wx.denoiseall(x, "sig", wt)
torch.cuda.synchronize()
print("wall per call %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
