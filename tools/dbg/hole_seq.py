# the sequence of tools/dbg/long_dwt_time.py for one length, every call timed on the host and on the device
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import waveletsext_jl_amd as wx
dt = torch.float64
wt = wx.wavelet(wx.WT.db4)
def call(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-28s host %7.2f ms  total %7.2f ms  torch reserved %.1f GiB" % (name, (t1 - t0) * 1e3, (t2 - t0) * 1e3, torch.cuda.memory_reserved() / 2**30), flush=True)
    return r
for n in (16384, 32768, 65536):
    B = 65536 * 4096 // n
    x = wx.jl_empty((n, B), dt, "cuda"); x.normal_()
    L = wx.maxtransformlevels(n)
    for rep in range(2): y = call("n=%d dwtall" % n, lambda: wx.dwtall(x, wt))
    for rep in range(2): call("idwtall", lambda: wx.idwtall(y, wt))
    for rep in range(2): y2 = call("wptall", lambda: wx.wptall(x, wt, L))
    for rep in range(2): call("iwptall", lambda: wx.iwptall(y2, wt, L))
    for rep in range(2): y3 = call("dwtall L=4", lambda: wx.dwtall(x, wt, 4))
    for rep in range(6): call("idwtall L=4", lambda: wx.idwtall(y3, wt, 4))
    del x, y, y2, y3
