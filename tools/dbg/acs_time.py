import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import waveletsext_jl_amd as wx
from waveletsext_jl_amd import bestbasis as bb
wt = wx.wavelet(wx.WT.coif6)
n, L, B = 2048, 11, 2048
x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
def run():
    return bb.acwpd_jbb_moments(x, wt, L)
run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print("acwpd_jbb_moments 2048 signals: %.3f ms" % (e0.elapsed_time(e1) / 10))
