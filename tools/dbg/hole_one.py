# the kernels behind one suspicious call: python hole_one.py <n> <L> <f64|f32>   (idwtall of a depth-L pyramid)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import waveletsext_jl_amd as wx
n, L = int(sys.argv[1]), int(sys.argv[2]); dt = torch.float32 if sys.argv[3] == "f32" else torch.float64
wt = wx.wavelet(wx.WT.db4)
B = 65536 * 4096 // n
x = wx.jl_empty((n, B), dt, "cuda"); x.normal_()
y = wx.dwtall(x, wt, L)
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    z = wx.idwtall(y, wt, L)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("host %.2f ms, total %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
