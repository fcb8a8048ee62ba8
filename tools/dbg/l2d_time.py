"""Launch times of config 4's two legs (2-D wptall / iwptall, 512 x 512 Float32 db4 L = 6) without any check.
usage: WX_HIP_LIB=... python tools/dbg/l2d_time.py [batch] [wavelet] [side] [L]"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import waveletsext_jl_amd as wx
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
wname = sys.argv[2] if len(sys.argv) > 2 else "db4"
side = int(sys.argv[3]) if len(sys.argv) > 3 else 512
L = int(sys.argv[4]) if len(sys.argv) > 4 else 6
wt = wx.wavelet(getattr(wx.WT, wname))
def t(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
x = wx.jl_empty((side, side, B), torch.float32, "cuda"); x.normal_()
f = t(lambda: wx.wptall(x, wt, L))
y = wx.wptall(x, wt, L)
i = t(lambda: wx.iwptall(y, wt, L))
gb = 2 * 4e-9 * side * side * B
print("%s: %d x %d x %d f32 %s L %d: wptall %.3f ms (%.1f %% of HBM peak)  iwptall %.3f ms (%.1f %%)" % (
    os.path.basename(os.environ.get("WX_HIP_LIB", "") or "base"), side, side, B, wname, L, f, 100 * gb / f / 8.0, i, 100 * gb / i / 8.0))
