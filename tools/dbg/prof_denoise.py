"""Kernel breakdown of denoiseall(x, :sig) at a few lengths: run under rocprofv3 --kernel-trace --stats (one length per process so the
stats file is per length):  rocprofv3 --kernel-trace --stats -d gpurun_out/dn<n> -- python3 tools/dbg/prof_denoise.py <n>"""
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
for _ in range(6):
    y = wx.denoiseall(x, "sig", wt)
torch.cuda.synchronize()
