"""host-pointer path (numpy in / numpy out through the C ABI): python tools/host_path_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import waveletsext_jl_amd as wx
wt = wx.wavelet(wx.WT.db8)
n, B, L = 4096, 8192, 12
x = np.asfortranarray(np.random.default_rng(0).standard_normal((n, B)))
for rep in range(3):
    t0 = time.perf_counter(); y = wx.wpdall(x, wt, L); t1 = time.perf_counter(); xr = wx.iwpdall(y, wt, L); t2 = time.perf_counter()
    print("wpdall %.1f ms (%.1f GB/s out)  iwpdall %.1f ms  err %.1e" % ((t1 - t0) * 1e3, y.nbytes / (t1 - t0) / 1e9, (t2 - t1) * 1e3, np.abs(xr - x).max()))
    del y, xr
