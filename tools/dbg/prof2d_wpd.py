"""diagnostics: wpdall of 16384 Float64 64 x 64 images, depth 3, a few calls (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import waveletsext_jl_amd as wx
wt = wx.wavelet(wx.WT.db4)
x = wx.jl_empty((64, 64, 16384), torch.float64, "cuda"); x.normal_()
for _ in range(3):
    y = wx.wpdall(x, wt, 3)
torch.cuda.synchronize()
print("algorithmic bytes per call:", 64 * 64 * 16384 * 8 * 5)
