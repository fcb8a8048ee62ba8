"""Kernel breakdown of denoiseall(xw, :dwt) (noisest + threshold on the loads of idwtall) at one length:
rocprofv3 --kernel-trace --stats -- python3 tools/dbg/prof_denoise_dwt.py <n>   (tools/dbg/prof_any.sh drives it)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import waveletsext_jl_amd as wx

n = int(sys.argv[1])
B = (1 << 30) // (n * 8)
wt = wx.wavelet(wx.WT.db4)
x = wx.jl_empty((n, B), torch.float64, "cuda")
x.normal_()
xw = wx.dwtall(x, wt)
for _ in range(6):
    y = wx.denoiseall(xw, "dwt", wt)
torch.cuda.synchronize()
