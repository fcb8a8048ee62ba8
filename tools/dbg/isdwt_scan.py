# isdwtall (average-based) over filters and depths at n = 4096: looks for shapes that fall off the fused path
import sys, os
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
n, B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 8192
x = wx.jl_empty((n, B), torch.float64, "cuda"); x.normal_()
for wname in ("db4", "db5", "db6", "db7", "db8", "coif4", "coif6"):
    wt = wx.wavelet(getattr(wx.WT, wname))
    out = []
    for L in (6, 7, 8, 9, 10, 11, 12):
        if L > wx.maxtransformlevels(n): break
        y = wx.sdwtall(x, wt, L)
        out.append("L%d %.2f/%.2f" % (L, t(lambda: wx.sdwtall(x, wt, L)), t(lambda: wx.isdwtall(y, wt))))
        del y
    print("%-6s F=%2d " % (wname, len(wt.qmf)) + "  ".join(out))
