import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import waveletsext_jl_amd as wx
wt = wx.wavelet(wx.WT.db4)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = wx.maxtransformlevels(n)
Bq = max((1 << 30) // (n * (L + 1) * 8), 1)
xq = wx.jl_empty((n, Bq), torch.float64, "cuda"); xq.normal_()
labels = [i % 3 for i in range(Bq)]
f = wx.LocalDiscriminantBasis(wt=wt, n_features=10)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    y = wx.fit_transform(f, xq, labels)
    torch.cuda.synchronize(); print("fit_transform %.2f ms, Bq %d" % ((time.perf_counter() - t0) * 1e3, Bq))
