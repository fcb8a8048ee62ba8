import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import waveletsext_jl_amd as wx
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
wt = wx.wavelet(wx.WT.db6)
n, B, L = 4096, 16384, 10
x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
y = wx.sdwtall(x, wt, L)
print("isdwt events", t(lambda: wx.isdwtall(y, wt)))
print("isdwt events again", t(lambda: wx.isdwtall(y, wt)))
f = t(lambda: wx.sdwtall(x, wt, L))
print("after sdwt x6: isdwt events", t(lambda: wx.isdwtall(y, wt)))
ts = []
for _ in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter(); xr = wx.isdwtall(y, wt); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("wall", ts)
print(torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9)
